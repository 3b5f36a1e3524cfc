for v in base pk; do
  lib=$PWD/vm_asr_amd/libvmasr_hip_$v.so; [ $v = base ] && lib=$PWD/vm_asr_amd/libvmasr_hip.so
  VMASR_LIB=$lib timeout 900 python bench.py --workload vm_asr_48k_16k_MPD_VSSM32_dstate32_nfft2048 --steps 4 --warmup 2 --no-cpu-baseline --no-extra-points --no-kernel-timing --detail /tmp/b.json 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$v dstate32', round(d['value'],2), round(d['ms_per_step'],2))"
  VMASR_LIB=$lib timeout 900 python bench.py --workload vm_asr_48k_16k_MPD_VSSM32 --steps 10 --warmup 3 --no-cpu-baseline --no-extra-points --no-kernel-timing --detail /tmp/b.json 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$v vssm32', round(d['value'],2), round(d['ms_per_step'],2))"
  VMASR_LIB=$lib timeout 900 python bench.py --workload vm_asr_48k --steps 10 --warmup 3 --no-cpu-baseline --no-extra-points --no-kernel-timing --detail /tmp/b.json 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$v gonly b35', round(d['value'],2), round(d['ms_per_step'],2))"
done
