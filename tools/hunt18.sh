timeout 900 python -m pytest tests/test_determinism.py -m gpu -q -x 2>&1 | grep -v Warn | tail -4
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "xproj" 2>&1 | grep -v Warn | tail -3
( VMASR_TWO_STREAM=force timeout 1200 python tools/det_hunt.py --iters 12 --batch 1 --workload vm_asr_48k_16k_MPD_VSSM32_dstate32_nfft2048 ) 2>&1 | tail -n 2 | cut -c1-600
