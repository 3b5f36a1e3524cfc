R=$PWD; O=$R/gpurun_out/r06; mkdir -p $O; cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/lt
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/lt -o t -- python $R/bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-extra-points --no-kernel-timing > /dev/null 2>/tmp/lt.err
f=$(find /tmp/lt -name "*kernel_trace.csv" | head -1)
python $R/tools/lane_trace.py $f > $O/lane_trace.log 2>&1
cat $O/lane_trace.log | head -70
