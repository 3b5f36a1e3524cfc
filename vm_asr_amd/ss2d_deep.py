"""The SS2D core of the deep stages as one operator (vm_asr_amd/csrc/ss2d_deep.hip).

`ss2d_deep(x, x_proj_weight, dt_projs_weight, dt_projs_bias, A_logs, Ds)` = lines 1472-1497 of
SS2D.forward_corev2 (model/vmamba.py: CrossScan -> x_proj / dt_proj einsums -> selective_scan -> CrossMerge) for d_state 1,
dt_rank 2 / 4 / 8, d_inner 64..512, H*W <= 4096: the 64x64, 32x32 and 16x16 stages — 18 of the 28 SS2D calls of a training step,
which vm_asr_amd/ss2d_core.py (d_inner <= 32, dt_rank 1) does not take.  Forward = 2 launches, backward = 3 launches + one small
GEMM (dW_x) and one sum (the per-wave parameter sums) — where the unfused chain ran 4 + 6 launches and their ATen glue.
"""
import ctypes
import os

import torch

from . import _lib
from .linear import _mm_acc

__all__ = ["ss2d_deep", "supported"]


def supported(d_state, dt_rank, d_inner, H, W, dtype=torch.float32):
    if os.environ.get("VMASR_SS2D_DEEP", "1") != "1" or dtype not in (torch.float32, torch.bfloat16):
        return False
    return bool(_lib.lib().vmasr_ss2d_deep_supported(int(d_state), int(dt_rank), int(d_inner), int(H), int(W)))


def _p(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def _f32c(t, shape):
    return t.detach().float().reshape(shape).contiguous()


def _params(x, R, wx, wdt, b32, al, ds, xdbl):
    B, D, H, W = x.shape
    p = _lib.SS2DDeepParams()
    p.B, p.D, p.H, p.W, p.R, p.dtype = B, D, H, W, R, _lib.torch_dtype_code(x.dtype)
    p.x, p.WxT, p.Wdt, p.dtb, p.Alog, p.Ds, p.xdbl = _p(x), _p(wx), _p(wdt), _p(b32), _p(al), _p(ds), _p(xdbl)
    return p


class _SS2DDeepFn(torch.autograd.Function):
    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda")
    def forward(ctx, x, Wx, Wdt, dtb, A_logs, Ds):
        B, D, H, W = x.shape
        R, L = Wdt.shape[-1], H * W
        x = x.contiguous()
        wx = Wx.detach().float().reshape(4, R + 2, D).transpose(1, 2).contiguous()       # (4, D, R + 2)
        wdt, b32 = _f32c(Wdt, (4, D, R)), _f32c(dtb, (4, D))
        al, ds = _f32c(A_logs, (4 * D,)), _f32c(Ds, (4 * D,))
        with torch.cuda.device(x.device):
            xdbl = torch.empty((B, 4, R + 2, L), dtype=torch.float32, device=x.device)
            y = torch.empty((B, D, L), dtype=torch.float32, device=x.device)
            p = _params(x, R, wx, wdt, b32, al, ds, xdbl)
            p.y = _p(y)
            _lib.check(_lib.lib().vmasr_ss2d_deep_fwd(ctypes.byref(p), _lib.current_stream(x.device)), "ss2d_deep_fwd")
        ctx.save_for_backward(x, xdbl, wx, wdt, b32, al, ds)
        ctx.meta = (Wx.dtype, Wx.shape, Wdt.dtype, Wdt.shape, dtb.dtype, dtb.shape, A_logs.dtype, A_logs.shape, Ds.dtype)
        return y

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(ctx, dy):
        x, xdbl, wx, wdt, b32, al, ds = ctx.saved_tensors
        wxdt, wxshape, wdtdt, wdtshape, dtbdt, dtbshape, aldt, alshape, dsdt = ctx.meta
        B, D, H, W = x.shape
        R, L = wdt.shape[-1], H * W
        C = R + 2
        dy = dy.float().contiguous()
        lib = _lib.lib()
        WR = int(lib.vmasr_ss2d_deep_waves_per_row(H, W))
        with torch.cuda.device(x.device):
            du = torch.empty((B, D, L), dtype=torch.float32, device=x.device)
            terms = torch.empty((3, B, 4, D, L), dtype=x.dtype, device=x.device)
            pg = torch.empty((B, 4, D, WR, 20), dtype=torch.float32, device=x.device)
            dx = torch.empty_like(x)
            gpos = torch.empty((B, 4 * C, L), dtype=x.dtype, device=x.device)
            g32 = torch.empty((B, 4 * C, L), dtype=torch.float32, device=x.device)
            p = _params(x, R, wx, wdt, b32, al, ds, xdbl)
            p.dy, p.du, p.tp, p.tb, p.tc, p.pg, p.dx, p.gpos, p.g32 = (_p(dy), _p(du), _p(terms[0]), _p(terms[1]), _p(terms[2]), _p(pg), _p(dx),
                                                                      _p(gpos), _p(g32))
            _lib.check(lib.vmasr_ss2d_deep_bwd(ctypes.byref(p), _lib.current_stream(x.device)), "ss2d_deep_bwd")
            # dW_x[kc][d] = sum_{b,p} gpos[b][kc][p] x[b][d][p]: (B, 4C, L) @ (B, L, D), fp32 accumulation, summed over the batch
            dWx = _mm_acc(gpos, x.view(B, D, L).transpose(1, 2), torch.float32).sum(0)
            s = pg[..., :R + 3].sum((0, 3))                                   # (4, D, R + 3)
        return (dx, dWx.view(wxshape).to(wxdt), s[..., :R].reshape(wdtshape).to(wdtdt), s[..., R].reshape(dtbshape).to(dtbdt),
                s[..., R + 1].reshape(alshape).to(aldt), s[..., R + 2].reshape(-1).to(dsdt))


def ss2d_deep(x, x_proj_weight, dt_projs_weight, dt_projs_bias, A_logs, Ds):
    """-> y (B, d_inner, H*W) fp32, the cross-merged output."""
    if not x.is_cuda:
        raise RuntimeError("ss2d_deep: expected a CUDA (HIP) tensor; vm_asr_amd has no CPU path")
    return _SS2DDeepFn.apply(x, x_proj_weight, dt_projs_weight, dt_projs_bias, A_logs, Ds)
