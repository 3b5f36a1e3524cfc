# Round-2 auxiliary evidence (one gpurun call): the bench variants quoted in profiles/r02_README.md, the operator
# microbenchmarks and the torch.profiler view of one eager step.
export VMASR_BENCH_WATCHDOG=500
mkdir -p gpurun_out
python bench.py --mpd-gemm fp32 --no-cpu-baseline > gpurun_out/r02_bench_mpd_fp32.json 2> /dev/null
python bench.py --amp-scope step --no-cpu-baseline > gpurun_out/r02_bench_amp_step.json 2> /dev/null
python bench.py --batch 8 --no-cpu-baseline > gpurun_out/r02_bench_b8.json 2> /dev/null
python bench.py --workload vm_asr_48k --no-cpu-baseline > gpurun_out/r02_bench_gonly_b35.json 2> /dev/null
VMASR_DIST_BACKEND=gloo python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 \
    bench.py --gpus 2 --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-timing > gpurun_out/r02_bench_2proc_gloo.log 2>&1
B=4 python tools/bench_glue.py 2>&1 | grep -v amdgpu > gpurun_out/r02_glue_microbench.log
python tools/bench_xproj.py 2>&1 | grep -v amdgpu > gpurun_out/r02_xproj_microbench.log
python tools/aten_tail.py 90 2>&1 | grep -v "amdgpu.ids\|Warning\|_warn_once" > gpurun_out/r02_aten_tail.log
python tools/bmm_probe_bt.py 2>&1 | grep -v amdgpu > gpurun_out/r02_bmm_probe_bt.log
for f in mpd_fp32 amp_step b8 gonly_b35; do cut -c1-200 gpurun_out/r02_bench_$f.json; done
tail -n 2 gpurun_out/r02_bench_2proc_gloo.log | cut -c1-200
