"""Cross-scan / cross-merge operators (HIP).  Drop-in for the reference's
`CrossScanTriton` / `CrossMergeTriton` (model/csm_triton.py:311-366), which have the
semantics of the PyTorch `CrossScan` / `CrossMerge` (model/vmamba.py:27-73):

    CrossScan.apply(x: (B,C,H,W))        -> (B,4,C,H*W)
    CrossMerge.apply(ys: (B,4,C,H,W))    -> (B,C,H*W)

Each forward is the other's backward; both call `.contiguous()` on their input like the
Triton versions do.  Compute: vm_asr_amd/csrc/csm.hip.  No CPU fallback.
"""
import ctypes

import torch

from . import _lib

__all__ = ["cross_scan", "cross_merge", "CrossScan", "CrossMerge", "CrossScanHIP", "CrossMergeHIP", "CrossScanF32"]


def _p(t):
    return ctypes.c_void_p(t.data_ptr())


def cross_scan(x: torch.Tensor, out_dtype=None) -> torch.Tensor:
    """out_dtype: None = x.dtype; torch.float32 converts 16-bit activations on the fly."""
    if not x.is_cuda:
        raise RuntimeError("cross_scan: expected a CUDA (HIP) tensor; vm_asr_amd has no CPU path")
    B, C, H, W = x.shape
    x = x.contiguous()
    out_dtype = out_dtype or x.dtype
    with torch.cuda.device(x.device):
        xs = torch.empty((B, 4, C, H * W), dtype=out_dtype, device=x.device)
        _lib.check(_lib.lib().vmasr_cross_scan_cvt(_p(x), _p(xs), B, C, H, W, _lib.torch_dtype_code(x.dtype),
                                                   _lib.torch_dtype_code(out_dtype), _lib.current_stream(x.device)),
                   "cross_scan")
    return xs


def cross_merge(ys: torch.Tensor, H: int, W: int, out_dtype=None) -> torch.Tensor:
    """ys (B,4,C,H*W) or (B,4,C,H,W) -> (B,C,H*W); out_dtype 16-bit converts fp32 streams on the fly."""
    if not ys.is_cuda:
        raise RuntimeError("cross_merge: expected a CUDA (HIP) tensor; vm_asr_amd has no CPU path")
    B, K, C = ys.shape[:3]
    if K != 4:
        raise RuntimeError("cross_merge: expected 4 scan directions")
    ys = ys.contiguous()
    out_dtype = out_dtype or ys.dtype
    with torch.cuda.device(ys.device):
        y = torch.empty((B, C, H * W), dtype=out_dtype, device=ys.device)
        _lib.check(_lib.lib().vmasr_cross_merge_cvt(_p(ys), _p(y), B, C, H, W, _lib.torch_dtype_code(ys.dtype),
                                                    _lib.torch_dtype_code(out_dtype), _lib.current_stream(ys.device)),
                   "cross_merge")
    return y


class CrossScan(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x: torch.Tensor):
        B, C, H, W = x.shape
        ctx.shape = (B, C, H, W)
        return cross_scan(x)

    @staticmethod
    def backward(ctx, ys: torch.Tensor):
        B, C, H, W = ctx.shape
        return cross_merge(ys, H, W).view(B, C, H, W)


class CrossMerge(torch.autograd.Function):
    @staticmethod
    def forward(ctx, ys: torch.Tensor):
        B, K, C, H, W = ys.shape
        ctx.shape = (B, C, H, W)
        return cross_merge(ys, H, W)

    @staticmethod
    def backward(ctx, x: torch.Tensor):
        B, C, H, W = ctx.shape
        return cross_scan(x.reshape(B, C, H, W)).view(B, 4, C, H, W)


class CrossScanF32(torch.autograd.Function):
    """CrossScan whose output is fp32 whatever the activation dtype: SS2D feeds the scan with fp32
    (forward type v5 casts xs to float, model/vmamba.py:1487-1491), so the cast is folded into the
    data movement in both directions."""

    @staticmethod
    def forward(ctx, x: torch.Tensor):
        B, C, H, W = x.shape
        ctx.shape, ctx.dt = (B, C, H, W), x.dtype
        return cross_scan(x, torch.float32)

    @staticmethod
    def backward(ctx, ys: torch.Tensor):
        B, C, H, W = ctx.shape
        return cross_merge(ys.float(), H, W, ctx.dt).view(B, C, H, W)


# names used when wiring SS2D (the reference wires CrossScanTriton / CrossMergeTriton)
CrossScanHIP = CrossScan
CrossMergeHIP = CrossMerge
