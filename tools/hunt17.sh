mkdir -p gpurun_out/hunt
export VMASR_TWO_STREAM=force
( timeout 900 python tools/det_hunt.py --iters 150 --batch 4 ) 2>&1 | tail -n 3 > gpurun_out/hunt/U_b4.log; cat gpurun_out/hunt/U_b4.log
( timeout 900 python tools/det_hunt.py --iters 60 --batch 2 --workload vm_asr_48k_16k_MPD_VSSM32 ) 2>&1 | tail -n 3 > gpurun_out/hunt/U_vssm32.log; cat gpurun_out/hunt/U_vssm32.log
( timeout 900 python tools/det_hunt.py --iters 40 --batch 1 --workload vm_asr_48k_16k_nfft2048 ) 2>&1 | tail -n 3 > gpurun_out/hunt/U_nfft2048.log; cat gpurun_out/hunt/U_nfft2048.log
( timeout 1200 python tools/det_hunt.py --iters 12 --batch 1 --workload vm_asr_48k_16k_MPD_VSSM32_dstate32_nfft2048 ) 2>&1 | tail -n 3 > gpurun_out/hunt/U_dstate32.log; cat gpurun_out/hunt/U_dstate32.log
