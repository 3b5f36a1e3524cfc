# Round-3 interim evidence (one gpurun call): new parity tests, headline bench, G-only B=4 bench + rocprofv3 kernel stats.
export VMASR_BENCH_WATCHDOG=500
R=$PWD; mkdir -p gpurun_out/r03a
python -m pytest tests/test_trainer.py tests/test_ss2d_fused.py tests/test_fullsize.py -m gpu -q -s -p no:cacheprovider -k "pinned_to_cpu_oracle or oracle_chain or distribution or adjudicated" > gpurun_out/r03a/tests.log 2>&1; tail -n 5 gpurun_out/r03a/tests.log
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/r03a/bench.json 2> gpurun_out/r03a/bench.err; cut -c1-200 gpurun_out/r03a/bench.json
timeout 600 python bench.py --workload vm_asr_48k --batch 4 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r03a/bench_gonly_b4.json 2> gpurun_out/r03a/bench_g.err; cut -c1-200 gpurun_out/r03a/bench_gonly_b4.json
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_g -o g -- python $R/bench.py --workload vm_asr_48k --batch 4 --steps 20 --warmup 5 --no-cpu-baseline > /dev/null 2> /tmp/profg.err
find /tmp/prof_g -name "*kernel_stats.csv" -exec cp {} $R/gpurun_out/r03a/gonly_b4_kernel_stats.csv \;
cd $R
python tools/kcat.py gpurun_out/r03a/gonly_b4_kernel_stats.csv 49 60
