# Round-4 evidence run (one gpurun call): PMC HBM traffic FIRST (bench.py quotes it only when it was measured on the same kernel
# sources), GPU tests with their printed numbers, the default bench line (headline + five secondary operating points + the CPU
# baseline), rocprofv3 kernel stats of the same command, PMC MFMA-busy, generator-only step, operator microbenchmarks.
# Everything lands in gpurun_out/r04/ and is then copied to profiles/r04_*.
export VMASR_BENCH_WATCHDOG=1500
R=$PWD; O=$R/gpurun_out/r04; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export VMASR_TWO_STREAM=0      # PMC passes: one stream — a dispatch's counters must not include another stream's kernels
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pmc_f -o f -- python $R/bench.py --steps 1 --warmup 1 --no-graphs --no-cpu-baseline --no-kernel-timing --no-extra-points > /dev/null 2> /tmp/pmcf.err
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/pmc_w -o w -- python $R/bench.py --steps 1 --warmup 1 --no-graphs --no-cpu-baseline --no-kernel-timing --no-extra-points > /dev/null 2> /tmp/pmcw.err
unset VMASR_TWO_STREAM
cd $R
python tools/pmc_bench_report.py $(find /tmp/pmc_f -name "*counter_collection.csv" | head -1) $(find /tmp/pmc_w -name "*counter_collection.csv" | head -1) $O/pmc_traffic.json 2>&1 | tail -n 30
cp $O/pmc_traffic.json profiles/r04_pmc_traffic.json          # (on the box: the bench below quotes it after checking the digest)
python -m pytest tests -m gpu -q -s -p no:cacheprovider > $O/gpu_parity.log 2>&1; tail -n 3 $O/gpu_parity.log
timeout 1500 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; cut -c1-260 $O/bench.json
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_e -o e -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra-points --timing-pass shared > $O/bench_prof.json 2> /tmp/prof.err
find /tmp/prof_e -name "*kernel_stats.csv" -exec cp {} $O/trainstep_kernel_stats.csv \;
VMASR_TWO_STREAM=0 timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_1 -o e -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra-points > $O/bench_prof_onestream.json 2> /tmp/prof1.err
find /tmp/prof_1 -name "*kernel_stats.csv" -exec cp {} $O/trainstep_onestream_kernel_stats.csv \;
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt -o kt -- python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-extra-points --no-kernel-timing > /dev/null 2> /tmp/kt.err
python $R/tools/overlap_report.py $(find /tmp/kt -name "*kernel_trace.csv" | head -1) 6 > $O/overlap_report.log 2>&1
export VMASR_TWO_STREAM=0      # (MFMA-busy PMC passes: one stream, as above)
timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/pmc_m -o m -- python $R/bench.py --steps 1 --warmup 1 --no-graphs --no-cpu-baseline --no-kernel-timing --no-extra-points > /dev/null 2> /tmp/pmcm.err
timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/pmc_m32 -o m -- python $R/bench.py --workload vm_asr_48k_16k_MPD_VSSM32 --steps 1 --warmup 1 --no-graphs --no-cpu-baseline --no-kernel-timing --no-extra-points > /dev/null 2> /tmp/pmcm32.err
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_g -o g -- python $R/bench.py --workload vm_asr_48k --batch 4 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_gonly_b4_prof.json 2> /tmp/profg.err
find /tmp/prof_g -name "*kernel_stats.csv" -exec cp {} $O/gonly_b4_kernel_stats.csv \;
unset VMASR_TWO_STREAM
cd $R
python tools/pmc_mfma_report.py $(find /tmp/pmc_m -name "*counter_collection.csv" | head -1) $O/pmc_mfma.json 2>&1 | tail -n 20
python tools/pmc_mfma_report.py $(find /tmp/pmc_m32 -name "*counter_collection.csv" | head -1) $O/pmc_mfma_vssm32.json 2>&1 | tail -n 12
python bench.py --workload vm_asr_48k --batch 4 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_gonly_b4.json 2> /dev/null
python bench.py --batch 8 --no-cpu-baseline --no-extra-points > $O/bench_b8.json 2> /dev/null
python bench.py --workload vm_asr_48k --no-cpu-baseline > $O/bench_gonly_b35.json 2> /dev/null
python bench.py --amp-scope step --no-cpu-baseline --no-extra-points > $O/bench_amp_step.json 2> /dev/null
VMASR_TWO_STREAM=0 python bench.py --no-cpu-baseline --no-extra-points > $O/bench_onestream.json 2> /dev/null
python tools/phase_probe.py 2>&1 | grep -v amdgpu | tail -n 8 > $O/phase_probe.log; echo '--- VMASR_SIDE_CUS=0 (no CU limit)' >> $O/phase_probe.log; VMASR_SIDE_CUS=0 python tools/phase_probe.py 2>&1 | grep -v amdgpu | tail -n 8 >> $O/phase_probe.log
python tools/overlap_probe.py 4 2>&1 | grep -v amdgpu > $O/overlap_probe.log
python tools/overlap_probe2.py 2>&1 | grep -v amdgpu > $O/overlap_probe2.log
python tools/overlap_probe3.py 2>&1 | grep -v amdgpu > $O/overlap_probe3.log
for c in 0 128 160 192 224; do echo "VMASR_SIDE_CUS=$c: $(VMASR_SIDE_CUS=$c python bench.py --no-cpu-baseline --no-extra-points --no-kernel-timing 2>/dev/null | cut -c70-165)"; done > $O/side_cus_sweep.log
python tools/bench_convgemm.py 4 2>&1 | grep -v amdgpu > $O/convgemm_microbench_b4.log
python tools/bench_convgemm.py 8 2>&1 | grep -v amdgpu > $O/convgemm_microbench_b8.log
B=4 python tools/bench_ss2d.py > $O/ss2d_microbench.log 2>&1
B=4 python tools/bench_ss2d_deep.py 2>&1 | grep -v amdgpu > $O/ss2d_deep_microbench.log
python tools/bench_mlp.py 2>&1 | grep -v amdgpu > $O/mlp_microbench.log
python tools/accuracy_probe.py --ops --core-shapes --families "" --cases "" 2>&1 | grep -v amdgpu > $O/accuracy_ops.log
python tools/linear_accuracy.py 2>&1 | grep -v amdgpu > $O/linear_accuracy.log
VMASR_DIST_BACKEND=gloo python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 \
    bench.py --gpus 2 --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-timing > $O/bench_2proc_gloo.log 2>&1
python tools/rccl_single_rank_probe.py > $O/rccl_single_rank.log 2>&1
python tools/kcat.py $O/trainstep_kernel_stats.csv 49 30
python tools/kcat.py $O/trainstep_onestream_kernel_stats.csv 49 12
python tools/kcat.py $O/gonly_b4_kernel_stats.csv 49 12
for f in b8 gonly_b35 gonly_b4 amp_step onestream; do cut -c1-200 $O/bench_$f.json; done; tail -n 2 $O/bench_2proc_gloo.log | cut -c1-300
