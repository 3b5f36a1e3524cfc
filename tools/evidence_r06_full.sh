# Round-6 evidence run (one gpurun call): PMC HBM traffic first (bench.py quotes it only when measured on the same sources + build flags),
# the full GPU suite (lease c), the default bench line, rocprofv3 kernel stats of the same command (two-stream and one-stream), MFMA-busy,
# the secondary operating points, the wire-dtype run.  Everything lands in gpurun_out/r06/ and is then copied to profiles/r06_*.
export VMASR_BENCH_WATCHDOG=1500
R=$PWD; O=$R/gpurun_out/r06; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export VMASR_TWO_STREAM=0      # PMC passes: one stream — a dispatch's counters must not include another stream's kernels
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pmc_f -o f -- python $R/bench.py --steps 1 --warmup 1 --no-graphs --no-cpu-baseline --no-kernel-timing --no-extra-points > /dev/null 2> /tmp/pmcf.err
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/pmc_w -o w -- python $R/bench.py --steps 1 --warmup 1 --no-graphs --no-cpu-baseline --no-kernel-timing --no-extra-points > /dev/null 2> /tmp/pmcw.err
unset VMASR_TWO_STREAM
cd $R
python tools/pmc_bench_report.py $(find /tmp/pmc_f -name "*counter_collection.csv" | head -1) $(find /tmp/pmc_w -name "*counter_collection.csv" | head -1) $O/pmc_traffic.json 2>&1 | tail -n 8
cp $O/pmc_traffic.json profiles/r06_pmc_traffic.json          # (on the box: the bench below quotes it after checking the digest)
bash tools/evidence_r06.sh ${LEASE:-c}
timeout 1500 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; cat $O/bench.json | cut -c1-600; cp bench_detail.json $O/bench_detail.json
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_e -o e -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra-points --timing-pass shared > $O/bench_prof.json 2> /tmp/prof.err
find /tmp/prof_e -name "*kernel_stats.csv" -exec cp {} $O/trainstep_kernel_stats.csv \;
VMASR_TWO_STREAM=0 timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_1 -o e -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra-points > $O/bench_prof_onestream.json 2> /tmp/prof1.err
find /tmp/prof_1 -name "*kernel_stats.csv" -exec cp {} $O/trainstep_onestream_kernel_stats.csv \;
export VMASR_TWO_STREAM=0
timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/pmc_m -o m -- python $R/bench.py --steps 1 --warmup 1 --no-graphs --no-cpu-baseline --no-kernel-timing --no-extra-points > /dev/null 2> /tmp/pmcm.err
unset VMASR_TWO_STREAM
cd $R
python tools/pmc_mfma_report.py $(find /tmp/pmc_m -name "*counter_collection.csv" | head -1) $O/pmc_mfma.json 2>&1 | tail -n 6
python bench.py --workload vm_asr_48k_16k_MPD_VSSM32_dstate32_nfft2048 --steps 4 --warmup 2 --no-cpu-baseline > $O/bench_dstate32.json 2> /dev/null; cp bench_detail.json $O/bench_dstate32_detail.json
VMASR_TWO_STREAM=0 python bench.py --no-cpu-baseline --no-extra-points > $O/bench_onestream.json 2> /dev/null
VMASR_DIST_BACKEND=gloo python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 \
    bench.py --gpus 2 --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-timing > $O/bench_2proc_gloo.log 2>&1
timeout 600 python tools/wire_dtype_run.py 200 2>&1 | grep -v "amdgpu\|Warning\|warn" > $O/wire_dtype_200steps.log; cat $O/wire_dtype_200steps.log
python tools/kcat.py $O/trainstep_kernel_stats.csv 49 14 2>&1 | tail -n 16
for f in dstate32 onestream; do tail -n 1 $O/bench_$f.json | cut -c1-260; done; tail -n 1 $O/bench_2proc_gloo.log | cut -c1-300
