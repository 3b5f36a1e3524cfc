"""Depthwise 3x3 conv + bias + SiLU (HIP), the `conv2d` + `act` pair of SS2D.forwardv2
(model/vmamba.py:859-868,1543-1545) fused into one operator.

    dwconv3x3_silu(x (B,C,H,W), weight (C,1,3,3), bias (C,)|None) -> (B,C,H,W)

Compute: vm_asr_amd/csrc/dwconv.hip.  No CPU fallback.
"""
import ctypes

import torch

from . import _lib

__all__ = ["dwconv3x3_silu", "DWConv3x3SiLU"]


def _p(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


class DWConv3x3SiLU(torch.autograd.Function):
    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda")
    def forward(ctx, x, weight, bias):
        if not x.is_cuda:
            raise RuntimeError("dwconv3x3_silu: expected a CUDA (HIP) tensor; vm_asr_amd has no CPU path")
        B, C, H, W = x.shape
        if tuple(weight.shape) != (C, 1, 3, 3):
            raise RuntimeError("dwconv3x3_silu: weight must be (C,1,3,3)")
        x = x.contiguous()
        w32 = weight.detach().float().contiguous()
        b32 = None if bias is None else bias.detach().float().contiguous()
        with torch.cuda.device(x.device):
            y = torch.empty_like(x)
            _lib.check(_lib.lib().vmasr_dwconv_silu_fwd(_p(x), _p(w32), _p(b32), _p(y), B, C, H, W,
                                                        _lib.torch_dtype_code(x.dtype),
                                                        _lib.current_stream(x.device)), "dwconv_silu_fwd")
        ctx.save_for_backward(x, w32, b32 if b32 is not None else torch.empty(0, device=x.device))
        ctx.has_bias = bias is not None
        ctx.wdtype = weight.dtype
        ctx.bdtype = None if bias is None else bias.dtype
        return y

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(ctx, gy):
        x, w32, b32 = ctx.saved_tensors
        B, C, H, W = x.shape
        gy = gy.contiguous()
        if gy.dtype != x.dtype:
            gy = gy.to(x.dtype)
        with torch.cuda.device(x.device):
            dx = torch.empty_like(x)
            dw, db = _lib.zeros_f32(x.device, (C, 1, 3, 3), (C,) if ctx.has_bias else None)
            ws = torch.empty((B, C, H, W), dtype=torch.float32, device=x.device)
            _lib.check(_lib.lib().vmasr_dwconv_silu_bwd(_p(x), _p(w32), _p(b32) if ctx.has_bias else None, _p(gy),
                                                        _p(dx), _p(dw), _p(db), _p(ws), B, C, H, W,
                                                        _lib.torch_dtype_code(x.dtype),
                                                        _lib.current_stream(x.device)), "dwconv_silu_bwd")
        return dx, dw.to(ctx.wdtype), (db.to(ctx.bdtype) if ctx.has_bias else None)


def dwconv3x3_silu(x, weight, bias=None):
    return DWConv3x3SiLU.apply(x, weight, bias)
