mkdir -p gpurun_out/hunt
for v in base nop rl noslp; do
  lib=$PWD/vm_asr_amd/libvmasr_hip_$v.so; [ $v = base ] && lib=$PWD/vm_asr_amd/libvmasr_hip.so
  ( VMASR_LIB=$lib timeout 900 python tools/det_hunt.py --iters 600 ) > gpurun_out/hunt/Q_$v.log 2>&1
  echo "$v: $(tail -n 1 gpurun_out/hunt/Q_$v.log)"
done
