mkdir -p gpurun_out/hunt
( timeout 900 python tools/det_hunt.py --iters 40 ) > gpurun_out/hunt/A_two_det.log 2>&1
( VMASR_TWO_STREAM=0 timeout 600 python tools/det_hunt.py --iters 40 ) > gpurun_out/hunt/B_one_det.log 2>&1
( timeout 900 python tools/det_hunt.py --iters 25 --watch ) > gpurun_out/hunt/C_two_det_watch.log 2>&1
( PYTORCH_NO_CUDA_MEMORY_CACHING=1 timeout 900 python tools/det_hunt.py --iters 25 ) > gpurun_out/hunt/D_two_det_nocache.log 2>&1
tail -n 8 gpurun_out/hunt/*.log
