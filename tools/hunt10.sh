mkdir -p gpurun_out/hunt
for rep in 1 2; do
for v in base nopk mixA; do
  lib=$PWD/vm_asr_amd/libvmasr_hip_$v.so; [ $v = base ] && lib=$PWD/vm_asr_amd/libvmasr_hip.so
  VMASR_LIB=$lib timeout 600 python bench.py --no-cpu-baseline --no-extra-points --no-kernel-timing --steps 60 --warmup 10 --detail gpurun_out/hunt/bench_$v.json 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$v', round(d['value'],2), round(d['ms_per_step'],3))"
done; done
