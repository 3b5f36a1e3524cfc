"""SS2D input projections (x_proj + dt_proj) on the HIP kernel (vm_asr_amd/csrc/xproj.hip).

    x_proj_dt(xs (B,K,D,L), x_proj_weight (K,R+2N,D), dt_projs_weight (K,D,R))
        -> dts (B,K*D,L), Bs (B,K,N,L), Cs (B,K,N,L)      all fp32, contiguous, scan-ready

Equivalent to the two einsums + split + contiguous + float casts of SS2D.forward_corev2
(model/vmamba.py:1473-1491), with fp32 accumulation.  No CPU fallback.
"""
import ctypes

import torch

from . import _lib

__all__ = ["x_proj_dt", "supported"]


def _p(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def supported(d_state, dt_rank, d_inner):
    """The memory-bound maps of csrc/xproj.hip (d_state 1 .. 4) or, for a general state dimension, the fp32-MFMA kernels of
    csrc/xproj_n.hip."""
    l = _lib.lib()
    return bool(l.vmasr_xproj_supported(int(d_state), int(dt_rank), int(d_inner))
                or l.vmasr_xproj_n_supported(int(d_state), int(dt_rank), int(d_inner)))


def _general(N, R, D):
    """-> the general-N entry points apply (and the small-C maps do not)."""
    l = _lib.lib()
    return not l.vmasr_xproj_supported(int(N), int(R), int(D)) and bool(l.vmasr_xproj_n_supported(int(N), int(R), int(D)))


class _XProjFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, xs, Wx, Wdt, N):
        B, K, D, L = xs.shape
        R = Wdt.shape[2]
        xs = xs.contiguous()
        wx32, wdt32 = Wx.detach().float().contiguous(), Wdt.detach().float().contiguous()
        f32 = dict(dtype=torch.float32, device=xs.device)
        with torch.cuda.device(xs.device):
            dts = torch.empty((B, K * D, L), **f32)
            Bs = torch.empty((B, K, N, L), **f32)
            Cs = torch.empty((B, K, N, L), **f32)
            dtr = torch.empty((B, K, R, L), **f32)
            fn = _lib.lib().vmasr_xproj_n_fwd if _general(N, R, D) else _lib.lib().vmasr_xproj_fwd
            _lib.check(fn(_p(xs), _p(wx32), _p(wdt32), _p(dts), _p(Bs), _p(Cs), _p(dtr), B, K, D, N, R,
                          L, _lib.torch_dtype_code(xs.dtype), _lib.current_stream(xs.device)), "xproj_fwd")
        ctx.save_for_backward(xs, wx32, wdt32, dtr)
        ctx.meta = (N, Wx.dtype, Wdt.dtype)
        return dts, Bs, Cs

    @staticmethod
    def backward(ctx, ddts, dBs, dCs):
        xs, wx32, wdt32, dtr = ctx.saved_tensors
        N, wxdt, wdtdt = ctx.meta
        B, K, D, L = xs.shape
        R = wdt32.shape[2]
        C = R + 2 * N
        f32 = dict(dtype=torch.float32, device=xs.device)
        z = lambda t, shape: torch.zeros(shape, **f32) if t is None else t.float().contiguous()  # noqa: E731
        ddts, dBs, dCs = z(ddts, (B, K * D, L)), z(dBs, (B, K, N, L)), z(dCs, (B, K, N, L))
        with torch.cuda.device(xs.device):
            dxs = torch.empty_like(xs)
            dWx, dWdt = _lib.zeros_f32(xs.device, (K, C, D), (K, D, R))
            gen = _general(N, R, D)
            ws = torch.empty((B, K, R if gen else C, L), **f32)     # general N: only the low-rank dt rows' gradient is kept
            fn = _lib.lib().vmasr_xproj_n_bwd if gen else _lib.lib().vmasr_xproj_bwd
            _lib.check(fn(_p(xs), _p(wx32), _p(wdt32), _p(dtr), _p(ddts), _p(dBs), _p(dCs), None,
                          _p(dxs), _p(dWx), _p(dWdt), _p(ws), B, K, D, N, R, L,
                          _lib.torch_dtype_code(xs.dtype), _lib.current_stream(xs.device)), "xproj_bwd")
        return dxs, dWx.to(wxdt), dWdt.to(wdtdt), None


def x_proj_dt(xs, x_proj_weight, dt_projs_weight, d_state):
    if not xs.is_cuda:
        raise RuntimeError("x_proj_dt: expected a CUDA (HIP) tensor; vm_asr_amd has no CPU path")
    return _XProjFn.apply(xs, x_proj_weight, dt_projs_weight, int(d_state))
