mkdir -p gpurun_out/hunt
for v in mixA mixB; do
  ( VMASR_LIB=$PWD/vm_asr_amd/libvmasr_hip_$v.so timeout 900 python tools/det_hunt.py --iters 700 ) > gpurun_out/hunt/S_$v.log 2>&1
  echo "$v: $(tail -n 1 gpurun_out/hunt/S_$v.log)"
done
