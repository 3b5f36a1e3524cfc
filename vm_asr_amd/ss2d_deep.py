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
from . import layernorm as _ln
from .linear import _mm_acc
from .wgrad import finish_slabs

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
        wx = _f32c(Wx, (4, R + 2, D))           # as stored (no copy for an fp32 parameter): the kernels index (k, c, d) directly
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
        ps = (Wx, Wdt, dtb, A_logs, Ds)
        if any(ctx.needs_input_grad[1:]):
            _ln.note_use(*ps)
        ctx.params = ps
        ctx.fresh = lambda: all(p.grad is None and p.dtype == torch.float32 for p in ps[1:]) and _ln.used_once(*ps[1:])
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
            pg = torch.empty((B * WR, 4 * D, 20), dtype=torch.float32, device=x.device)      # per-(b, wave) slabs of (4 D, kPG)
            dx = torch.empty_like(x)
            gpos = torch.empty((B, 4 * C, L), dtype=x.dtype, device=x.device)
            g32 = torch.empty((B, 4 * C, L), dtype=torch.float32, device=x.device)
            p = _params(x, R, wx, wdt, b32, al, ds, xdbl)
            p.dy, p.du, p.tp, p.tb, p.tc, p.pg, p.dx, p.gpos, p.g32 = (_p(dy), _p(du), _p(terms[0]), _p(terms[1]), _p(terms[2]), _p(pg), _p(dx),
                                                                      _p(gpos), _p(g32))
            _lib.check(lib.vmasr_ss2d_deep_bwd(ctypes.byref(p), _lib.current_stream(x.device)), "ss2d_deep_bwd")
            # dW_x[kc][d] = sum_{b,p} gpos[b][kc][p] x[b][d][p]: (B, 4C, L) @ (B, L, D), fp32 accumulation, summed over the batch
            px = _mm_acc(gpos, x.view(B, D, L).transpose(1, 2), torch.float32)          # (B, 4C, D): summed over the batch by the finish
            dWx = torch.empty((4 * C, D), dtype=torch.float32, device=x.device)
            Wx_ = ctx.params[0]
            finish_slabs(px, D, (dWx,), (Wx_,), Wx_.grad is None and Wx_.dtype == torch.float32 and _ln.used_once(Wx_))
            # the per-(b, wave) parameter sums -> dW_dt (4, D, R), d dt_bias (4, D), dA_log (4 D), dD (4 D): contiguous tensors,
            # finished together with the pass' other parameter gradients in one launch (wgrad.py)
            dWdt = torch.empty((4 * D, R), dtype=torch.float32, device=x.device)
            ddtb, dal, dds = (torch.empty(4 * D, dtype=torch.float32, device=x.device) for _ in range(3))
            finish_slabs(pg, R, (dWdt, ddtb, dal, dds), ctx.params[1:], ctx.fresh())
        return (dx, dWx.view(wxshape).to(wxdt), dWdt.view(wdtshape).to(wdtdt), ddtb.view(dtbshape).to(dtbdt),
                dal.view(alshape).to(aldt), dds.to(dsdt))


def ss2d_deep(x, x_proj_weight, dt_projs_weight, dt_projs_bias, A_logs, Ds):
    """-> y (B, d_inner, H*W) fp32, the cross-merged output."""
    if not x.is_cuda:
        raise RuntimeError("ss2d_deep: expected a CUDA (HIP) tensor; vm_asr_amd has no CPU path")
    return _SS2DDeepFn.apply(x, x_proj_weight, dt_projs_weight, dt_projs_bias, A_logs, Ds)
