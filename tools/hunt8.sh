mkdir -p gpurun_out/hunt
( timeout 600 tools/pk_hazard/pk_probe 40 ) > gpurun_out/hunt/R_pk_probe.log 2>&1
cat gpurun_out/hunt/R_pk_probe.log
( VMASR_LIB=$PWD/vm_asr_amd/libvmasr_hip_nopk.so timeout 900 python tools/det_hunt.py --iters 800 ) > gpurun_out/hunt/Q_nopk.log 2>&1
echo "nopk: $(tail -n 1 gpurun_out/hunt/Q_nopk.log)"
