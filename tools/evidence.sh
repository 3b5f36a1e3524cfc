export VMASR_BENCH_WATCHDOG=500
R=$PWD
python -m pytest tests -m gpu -q 2>&1 | tail -2
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/bench_v5.json 2> gpurun_out/bench_v5.err
timeout 600 python bench.py --amp-scope step --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/bench_v5_step.json 2>> gpurun_out/bench_v5.err
cut -c1-300 gpurun_out/bench_v5.json
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_e -o e -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $R/gpurun_out/bench_v5_prof.json 2> /tmp/prof.err
find /tmp/prof_e -name "*kernel_stats.csv" -exec cp {} $R/gpurun_out/kernel_stats_v5e.csv \;
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pmc_f -o f -- python $R/bench.py --steps 1 --warmup 1 --no-graphs --no-cpu-baseline --no-kernel-timing > /dev/null 2> /tmp/pmcf.err
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/pmc_w -o w -- python $R/bench.py --steps 1 --warmup 1 --no-graphs --no-cpu-baseline --no-kernel-timing > /dev/null 2> /tmp/pmcw.err
cd $R
python tools/pmc_bench_report.py $(find /tmp/pmc_f -name "*counter_collection.csv" | head -1) $(find /tmp/pmc_w -name "*counter_collection.csv" | head -1) gpurun_out/pmc_traffic_v5.json 2>&1 | tail -3
for B in 4 32; do SWEEP=0 B=$B timeout 300 python tools/bench_scan.py 2>&1 | grep -v amdgpu; done > gpurun_out/scan_microbench_v3.log
ls -la gpurun_out/ | grep -E "v5|v3"
