# kernel trace of the GAN step with the generator on one / two streams: which hardware queues run what, and how concurrently (dev tool)
R=$PWD; O=$R/gpurun_out/gs; mkdir -p $O; cd /tmp && export TMPDIR=/tmp
for g in 1 2; do
rm -rf /tmp/lt$g
VMASR_GEN_STREAMS=$g timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/lt$g -o t -- python $R/bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-extra-points --no-kernel-timing > /dev/null 2>/tmp/lt$g.err
f=$(find /tmp/lt$g -name "*kernel_trace.csv" | head -1)
python $R/tools/lane_trace.py $f > $O/lane_trace_$g.log 2>&1
tail -40 $O/lane_trace_$g.log
done
