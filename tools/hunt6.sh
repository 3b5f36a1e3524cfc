mkdir -p gpurun_out/hunt
( timeout 900 python tools/det_hunt.py --iters 500 --trace ) > gpurun_out/hunt/L_trace_ws.log 2>&1
( VMASR_TWO_STREAM=0 timeout 900 python tools/det_hunt.py --iters 800 ) > gpurun_out/hunt/M_one_800.log 2>&1
( AMD_SERIALIZE_KERNEL=3 timeout 900 python tools/det_hunt.py --iters 500 ) > gpurun_out/hunt/N_serial.log 2>&1
( HIP_FORCE_DEV_KERNARG=0 timeout 900 python tools/det_hunt.py --iters 500 ) > gpurun_out/hunt/O_hostkernarg.log 2>&1
( DEBUG_CLR_KERNARG_HDP_FLUSH_WA=1 timeout 900 python tools/det_hunt.py --iters 500 ) > gpurun_out/hunt/P_hdpwa.log 2>&1
grep -A8 "^   call" gpurun_out/hunt/L_trace_ws.log | head -80
tail -qn 1 gpurun_out/hunt/[L-P]_*.log
