# Round-2 evidence run (one gpurun call): GPU tests, bench line, rocprofv3 kernel stats of the same command, PMC traffic.
export VMASR_BENCH_WATCHDOG=500
R=$PWD; mkdir -p gpurun_out
python -m pytest tests -m gpu -q -s -p no:cacheprovider > gpurun_out/r02_gpu_tests.log 2>&1; tail -n 3 gpurun_out/r02_gpu_tests.log
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/r02_bench.json 2> gpurun_out/r02_bench.err
cut -c1-260 gpurun_out/r02_bench.json
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_e -o e -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $R/gpurun_out/r02_bench_prof.json 2> /tmp/prof.err
find /tmp/prof_e -name "*kernel_stats.csv" -exec cp {} $R/gpurun_out/r02_trainstep_kernel_stats.csv \;
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pmc_f -o f -- python $R/bench.py --steps 1 --warmup 1 --no-graphs --no-cpu-baseline --no-kernel-timing > /dev/null 2> /tmp/pmcf.err
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/pmc_w -o w -- python $R/bench.py --steps 1 --warmup 1 --no-graphs --no-cpu-baseline --no-kernel-timing > /dev/null 2> /tmp/pmcw.err
cd $R
python tools/pmc_bench_report.py $(find /tmp/pmc_f -name "*counter_collection.csv" | head -1) $(find /tmp/pmc_w -name "*counter_collection.csv" | head -1) gpurun_out/r02_pmc_traffic.json 2>&1 | tail -n 30
B=4 python tools/bench_ss2d.py > gpurun_out/r02_ss2d_microbench.log 2>&1
for B in 4 32; do SWEEP=0 B=$B timeout 300 python tools/bench_scan.py 2>&1 | grep -v amdgpu; done > gpurun_out/r02_scan_microbench.log
python tools/kcat.py gpurun_out/r02_trainstep_kernel_stats.csv 49 24
