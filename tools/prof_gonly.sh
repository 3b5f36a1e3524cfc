# rocprofv3 kernel stats of the generator-only train step at B = 4 (bench.py --workload vm_asr_48k --batch 4), categorised
R=$PWD; cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/prof_g
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_g -o g -- python $R/bench.py --workload vm_asr_48k --batch 4 --steps 20 --warmup 5 --no-cpu-baseline > $R/gpurun_out/gonly_prof.json 2> /tmp/profg.err
find /tmp/prof_g -name "*kernel_stats.csv" -exec cp {} $R/gpurun_out/gonly_b4_kernel_stats.csv \;
cd $R; python tools/kcat.py gpurun_out/gonly_b4_kernel_stats.csv 49 70
