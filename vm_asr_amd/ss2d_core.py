"""The SS2D core as one HIP operator (vm_asr_amd/csrc/ss2d.hip):

    ss2d_core(x (B,D,H,W), x_proj_weight (4,3,D), dt_projs_weight (4,D,1), dt_projs_bias (4,D), A_logs (4D,1), Ds (4D))
        -> y (B, D, H*W) fp32   ==   CrossMerge(selective_scan(CrossScan(x), dts, -exp(A_logs), Bs, Cs, Ds, bias, softplus))

i.e. lines 1472-1497 of SS2D.forward_corev2 (model/vmamba.py) for d_state 1, dt_rank 1, d_inner <= 32 — the three
high-resolution stages of every shipped config — with cross-scan, x_proj, dt_proj, the four scans and cross-merge
fused around the scan (see the header of ss2d.hip).  Differentiable (one fused backward); no CPU fallback.
"""
import ctypes
import os

import torch

from . import _lib

__all__ = ["ss2d_core", "ss2d_core_pairs", "supported"]


def supported(d_state, dt_rank, d_inner, H, W):
    if os.environ.get("VMASR_SS2D_FUSED", "1") != "1":
        return False
    return bool(_lib.lib().vmasr_ss2d_supported(int(d_state), int(dt_rank), int(d_inner), int(H), int(W)))


def _p(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def _f32c(t, shape):
    return t.detach().float().reshape(shape).contiguous()


class _SS2DCoreFn(torch.autograd.Function):
    """pairs = False: y (B, D, L), the merged output.  pairs = True: (out02, out13) — the two pair outputs in (h,w) / (w,h)
    order, left un-merged for a consumer that adds them itself (ss2d_glue.ln_gate_pairs: no merge launch forward, no
    transpose launch backward; the backward then receives the gradient in both orders)."""

    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda")
    def forward(ctx, x, Wx, Wdt, dtb, A_logs, Ds, pairs=False):
        B, D, H, W = x.shape
        L, nt = H * W, (H * W) // 256
        x = x.contiguous()
        wx, wdt, b32 = _f32c(Wx, (4, 3, D)), _f32c(Wdt, (4, D)), _f32c(dtb, (4, D))
        al, ds = _f32c(A_logs, (4 * D,)), _f32c(Ds, (4 * D,))
        f32 = dict(dtype=torch.float32, device=x.device)
        with torch.cuda.device(x.device):
            xT = torch.empty_like(x)
            state = torch.empty((B, 4 * D, nt, 2), **f32)
            scratch = torch.empty((2, B, D, L), **f32)
            y = None if pairs else torch.empty((B, D, L), **f32)
            p = _lib.SS2DParams()
            p.B, p.D, p.H, p.W, p.dtype, p.flags = B, D, H, W, _lib.torch_dtype_code(x.dtype), (1 if pairs else 0)
            p.x, p.xT, p.Wx, p.Wdt, p.dtb, p.Alog, p.Ds = _p(x), _p(xT), _p(wx), _p(wdt), _p(b32), _p(al), _p(ds)
            p.state, p.out02, p.out13, p.y = _p(state), _p(scratch[0]), _p(scratch[1]), _p(y)
            _lib.check(_lib.lib().vmasr_ss2d_fwd(ctypes.byref(p), _lib.current_stream(x.device)), "ss2d_fwd")
        ctx.save_for_backward(x, xT, state, wx, wdt, b32, al, ds)
        ctx.meta = (Wx.dtype, Wdt.dtype, Wdt.shape, dtb.dtype, dtb.shape, A_logs.dtype, A_logs.shape, Ds.dtype)
        ctx.pairs = pairs
        return (scratch[0], scratch[1]) if pairs else y

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(ctx, dy, dyT=None):
        x, xT, state, wx, wdt, b32, al, ds = ctx.saved_tensors
        wxdt, wdtdt, wdtshape, dtbdt, dtbshape, aldt, alshape, dsdt = ctx.meta
        B, D, H, W = x.shape
        L, nt = H * W, (H * W) // 256
        dy = dy.float().contiguous()
        if ctx.pairs:
            dyT = dyT.float().contiguous()
        f32 = dict(dtype=torch.float32, device=x.device)
        lib = _lib.lib()
        with torch.cuda.device(x.device):
            scratch = torch.empty((3, B, D, L), **f32)            # dyT (unless given), dx02, dx13
            adj = torch.empty((B, 4 * D, nt, 2), **f32)
            part = torch.empty(lib.vmasr_ss2d_part_floats(B, D, H, W), **f32)
            dx = torch.empty_like(x)
            grads = torch.empty(4 * 3 * D + 4 * 4 * D, **f32)
            dWx, dWdt, ddtb, dAl, dDs = torch.split(grads, [12 * D, 4 * D, 4 * D, 4 * D, 4 * D])
            p = _lib.SS2DParams()
            p.B, p.D, p.H, p.W, p.dtype, p.flags = B, D, H, W, _lib.torch_dtype_code(x.dtype), (1 if ctx.pairs else 0)
            p.x, p.xT, p.Wx, p.Wdt, p.dtb, p.Alog, p.Ds = _p(x), _p(xT), _p(wx), _p(wdt), _p(b32), _p(al), _p(ds)
            p.state, p.out02, p.out13 = _p(state), _p(scratch[1]), _p(scratch[2])
            p.dy, p.dyT, p.adj, p.part, p.dx = _p(dy), _p(dyT if ctx.pairs else scratch[0]), _p(adj), _p(part), _p(dx)
            p.dWx, p.dWdt, p.ddtb, p.dAlog, p.dDs = _p(dWx), _p(dWdt), _p(ddtb), _p(dAl), _p(dDs)
            _lib.check(lib.vmasr_ss2d_bwd(ctypes.byref(p), _lib.current_stream(x.device)), "ss2d_bwd")
        return (dx, dWx.view(4, 3, D).to(wxdt), dWdt.view(wdtshape).to(wdtdt), ddtb.view(dtbshape).to(dtbdt),
                dAl.view(alshape).to(aldt), dDs.to(dsdt), None)


def ss2d_core(x, x_proj_weight, dt_projs_weight, dt_projs_bias, A_logs, Ds):
    if not x.is_cuda:
        raise RuntimeError("ss2d_core: expected a CUDA (HIP) tensor; vm_asr_amd has no CPU path")
    return _SS2DCoreFn.apply(x, x_proj_weight, dt_projs_weight, dt_projs_bias, A_logs, Ds, False)


def ss2d_core_pairs(x, x_proj_weight, dt_projs_weight, dt_projs_bias, A_logs, Ds):
    """-> (out02 (B, D, H*W) in (h,w) order, out13 (B, D, W*H) in (w,h) order), both fp32: y = out02 + transpose(out13)."""
    if not x.is_cuda:
        raise RuntimeError("ss2d_core_pairs: expected a CUDA (HIP) tensor; vm_asr_amd has no CPU path")
    return _SS2DCoreFn.apply(x, x_proj_weight, dt_projs_weight, dt_projs_bias, A_logs, Ds, True)
