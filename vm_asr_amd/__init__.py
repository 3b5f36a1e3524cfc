"""vm_asr_amd — MI355X-native (gfx950) implementation of the VM-ASR data-parallel hot path.

The compute path is libvmasr_hip.so (hand-written HIP, C ABI in include/vmasr_hip.h);
this package is the PyTorch-ROCm host side that mirrors the reference's operator /
module interfaces (selective_scan_cuda_core.fwd/bwd, CrossScan/CrossMerge, SS2D,
VSSBlock, wav2spectro/spectro2wav, DualStreamInteractiveMambaUNet, Trainer).

There is NO CPU fallback in this package: ops raise if the HIP library is missing.
"""
__version__ = "0.1.0"
