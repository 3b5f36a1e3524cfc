python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -p no:cacheprovider -k "im2col2d or gemm_conv2d" 2>&1 | tail -n 3
for r in 1 2; do for k in 1 0; do echo "im2col2d $k: $(VMASR_IM2COL2D=$k python bench.py --no-cpu-baseline --no-extra-points --no-kernel-timing 2>/dev/null | cut -c70-90)"; done; done
for k in 1 0; do echo "G-only b4 im2col2d $k: $(VMASR_IM2COL2D=$k python bench.py --workload vm_asr_48k --batch 4 --no-cpu-baseline --no-extra-points --no-kernel-timing 2>/dev/null | cut -c70-165)"; done
