# Round-5 evidence run (one gpurun call): PMC HBM traffic FIRST (bench.py quotes it only when it was measured on the same kernel
# sources), GPU tests with their printed numbers, the default bench line (compact line + bench_detail.json), rocprofv3 kernel stats of
# the same command (two-stream and one-stream), the d_state-32 stress point alone with its kernel stats, the general-N microbenchmarks,
# MFMA-busy, multi-process lines.  Everything lands in gpurun_out/r05/ and is then copied to profiles/r05_*.
export VMASR_BENCH_WATCHDOG=1500
R=$PWD; O=$R/gpurun_out/r05; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export VMASR_TWO_STREAM=0      # PMC passes: one stream — a dispatch's counters must not include another stream's kernels
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pmc_f -o f -- python $R/bench.py --steps 1 --warmup 1 --no-graphs --no-cpu-baseline --no-kernel-timing --no-extra-points > /dev/null 2> /tmp/pmcf.err
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/pmc_w -o w -- python $R/bench.py --steps 1 --warmup 1 --no-graphs --no-cpu-baseline --no-kernel-timing --no-extra-points > /dev/null 2> /tmp/pmcw.err
unset VMASR_TWO_STREAM
cd $R
python tools/pmc_bench_report.py $(find /tmp/pmc_f -name "*counter_collection.csv" | head -1) $(find /tmp/pmc_w -name "*counter_collection.csv" | head -1) $O/pmc_traffic.json 2>&1 | tail -n 12
cp $O/pmc_traffic.json profiles/r05_pmc_traffic.json          # (on the box: the bench below quotes it after checking the digest)
python -m pytest tests -m gpu -q -s -p no:cacheprovider > $O/gpu_parity.log 2>&1; tail -n 3 $O/gpu_parity.log
timeout 1500 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; cat $O/bench.json | cut -c1-400; cp bench_detail.json $O/bench_detail.json
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_e -o e -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra-points --timing-pass shared > $O/bench_prof.json 2> /tmp/prof.err
find /tmp/prof_e -name "*kernel_stats.csv" -exec cp {} $O/trainstep_kernel_stats.csv \;
VMASR_TWO_STREAM=0 timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_1 -o e -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra-points > $O/bench_prof_onestream.json 2> /tmp/prof1.err
find /tmp/prof_1 -name "*kernel_stats.csv" -exec cp {} $O/trainstep_onestream_kernel_stats.csv \;
VMASR_TWO_STREAM=0 timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_s -o s -- python $R/bench.py --workload vm_asr_48k_16k_MPD_VSSM32_dstate32_nfft2048 --steps 4 --warmup 2 --no-cpu-baseline --no-extra-points --no-kernel-timing > $O/bench_dstate32_prof.json 2> /tmp/profs.err
find /tmp/prof_s -name "*kernel_stats.csv" -exec cp {} $O/dstate32_kernel_stats.csv \;
export VMASR_TWO_STREAM=0      # (MFMA-busy PMC pass: one stream, as above)
timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/pmc_m -o m -- python $R/bench.py --steps 1 --warmup 1 --no-graphs --no-cpu-baseline --no-kernel-timing --no-extra-points > /dev/null 2> /tmp/pmcm.err
unset VMASR_TWO_STREAM
cd $R
python tools/pmc_mfma_report.py $(find /tmp/pmc_m -name "*counter_collection.csv" | head -1) $O/pmc_mfma.json 2>&1 | tail -n 8
python bench.py --workload vm_asr_48k_16k_MPD_VSSM32_dstate32_nfft2048 --steps 4 --warmup 2 --no-cpu-baseline > $O/bench_dstate32.json 2> /dev/null; cp bench_detail.json $O/bench_dstate32_detail.json
python bench.py --workload vm_asr_48k --batch 4 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_gonly_b4.json 2> /dev/null
python bench.py --batch 8 --no-cpu-baseline --no-extra-points > $O/bench_b8.json 2> /dev/null
python bench.py --workload vm_asr_48k --no-cpu-baseline > $O/bench_gonly_b35.json 2> /dev/null
VMASR_TWO_STREAM=0 python bench.py --no-cpu-baseline --no-extra-points > $O/bench_onestream.json 2> /dev/null
python tools/bench_scan_n.py 2>&1 | grep -v amdgpu > $O/scan_n_microbench.log
VMASR_SSCAN_N_LEGACY=1 python tools/bench_scan_n.py 2>&1 | grep -v amdgpu > $O/scan_n_microbench_legacy.log
python tools/bench_xproj_n.py 2>&1 | grep -v amdgpu > $O/xproj_n_microbench.log
./tools/scan_prims_probe/probe > $O/scan_prims_probe.log 2>&1
python tools/scan_accuracy.py 1 64 4 32 2048 2>&1 | grep -v amdgpu > $O/scan_accuracy.log
B=4 python tools/bench_ss2d.py > $O/ss2d_microbench.log 2>&1
VMASR_DIST_BACKEND=gloo python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 \
    bench.py --gpus 2 --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-timing > $O/bench_2proc_gloo.log 2>&1
python tools/rccl_single_rank_probe.py 2>&1 | grep -v "amdgpu\|UserWarning\|return func" > $O/rccl_single_rank.log
bash tools/gen_streams_ab.sh > $O/gen_streams_ab.log 2>&1
(python tools/host_bound_probe.py; VMASR_GEN_STREAMS=2 python tools/host_bound_probe.py; python tools/host_bound_probe.py vm_asr_48k 4; VMASR_GEN_STREAMS=1 python tools/host_bound_probe.py vm_asr_48k 4) 2>&1 | grep -v "amdgpu\|Warning" > $O/host_bound_probe.log
python tools/kcat.py $O/trainstep_kernel_stats.csv 49 14
python tools/kcat.py $O/trainstep_onestream_kernel_stats.csv 49 14
python tools/kcat.py $O/dstate32_kernel_stats.csv 9 14
for f in dstate32 gonly_b4 b8 gonly_b35 onestream; do tail -n 1 $O/bench_$f.json | cut -c1-260; done; tail -n 1 $O/bench_2proc_gloo.log | cut -c1-300
