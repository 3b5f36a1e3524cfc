"""The memory-bound glue of SS2D.forwardv2 around the scan core as two HIP operators (vm_asr_amd/csrc/ss2d_glue.hip):

    ss2d_pre(xz (B,H,W,2D))                      -> xT (B,D,H,W), sz = SiLU(z) (B,H,W,D)
        == x, z = xz.chunk(2, -1); z = SiLU(z); x = x.permute(0,3,1,2).contiguous()        model/vmamba.py:1537-1542
    ln_gate(y (B,D,L) fp32, sz, out_norm.weight, out_norm.bias, eps)  -> (B,H,W,D) in sz.dtype
        == out_norm(y.transpose(1,2).contiguous()).view(B,H,W,D).to(dtype) * z            model/vmamba.py:1528-1531,1550
Both differentiable (one launch per direction); no CPU fallback.
"""
import ctypes
import os

import torch

from . import _lib
from . import layernorm as _ln

__all__ = ["ss2d_pre", "ln_gate", "ln_gate_pairs", "pairs_supported", "supported"]


def _p(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def supported(D, L, dtype):
    if os.environ.get("VMASR_SS2D_GLUE", "1") != "1" or dtype not in (torch.float32, torch.float16, torch.bfloat16):
        return False
    return bool(_lib.lib().vmasr_ss2d_glue_supported(int(D), int(L), _lib.torch_dtype_code(dtype)))


class _PreFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, xz):
        B, H, W, D2 = xz.shape
        D, L = D2 // 2, H * W
        xz = xz.contiguous()
        with torch.cuda.device(xz.device):
            xT = torch.empty((B, D, H, W), dtype=xz.dtype, device=xz.device)
            sz = torch.empty((B, H, W, D), dtype=xz.dtype, device=xz.device)
            _lib.check(_lib.lib().vmasr_ss2d_pre_fwd(_p(xz), _p(xT), _p(sz), B, D, L, _lib.torch_dtype_code(xz.dtype),
                                                     _lib.current_stream(xz.device)), "ss2d_pre_fwd")
        ctx.save_for_backward(xz)
        return xT, sz

    @staticmethod
    def backward(ctx, dxT, dsz):
        (xz,) = ctx.saved_tensors
        B, H, W, D2 = xz.shape
        D, L = D2 // 2, H * W
        dt = xz.dtype
        dxT = (torch.zeros((B, D, H, W), dtype=dt, device=xz.device) if dxT is None else dxT.to(dt)).contiguous()
        dsz = (torch.zeros((B, H, W, D), dtype=dt, device=xz.device) if dsz is None else dsz.to(dt)).contiguous()
        with torch.cuda.device(xz.device):
            dxz = torch.empty_like(xz)
            _lib.check(_lib.lib().vmasr_ss2d_pre_bwd(_p(xz), _p(dxT), _p(dsz), _p(dxz), B, D, L, _lib.torch_dtype_code(dt),
                                                     _lib.current_stream(xz.device)), "ss2d_pre_bwd")
        return dxz


class _LNGateFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, y, sz, gamma, beta, eps):
        B, H, W, D = sz.shape
        L = H * W
        y = y.float().contiguous()
        sz = sz.contiguous()
        g32, b32 = gamma.detach().float().contiguous(), beta.detach().float().contiguous()
        with torch.cuda.device(sz.device):
            out = torch.empty_like(sz)
            stats = torch.empty((2, B, L), dtype=torch.float32, device=sz.device)
            _lib.check(_lib.lib().vmasr_ln_gate_fwd(_p(y), _p(sz), _p(g32), _p(b32), _p(out), _p(stats[0]), _p(stats[1]), B, D, L,
                                                    float(eps), _lib.torch_dtype_code(sz.dtype), _lib.current_stream(sz.device)),
                       "ln_gate_fwd")
        ctx.save_for_backward(y, sz, g32, b32, stats)
        ctx.meta = (gamma.dtype, beta.dtype)
        if any(ctx.needs_input_grad[2:4]):
            _ln.note_use(gamma, beta)
        ctx.fresh = lambda: (all(getattr(t, "grad", None) is None for t in (gamma, beta)) and _ln.used_once(gamma, beta))
        ctx.params = (gamma, beta)
        return out

    @staticmethod
    def backward(ctx, dout):
        y, sz, g32, b32, stats = ctx.saved_tensors
        B, H, W, D = sz.shape
        L = H * W
        dout = dout.to(sz.dtype).contiguous()
        lib, code = _lib.lib(), _lib.torch_dtype_code(sz.dtype)
        with torch.cuda.device(sz.device):
            dy = torch.empty_like(y)
            dsz = torch.empty_like(sz)
            nws = int(lib.vmasr_ln_gate_bwd_workspace(B, D, L, code))
            if nws:
                # d_inner >= 64: per-workgroup partials of dgamma / dbeta, reduced at once or — in the trainer's flat-gradient
                # mode — by the one reduce launch at the end of the backward pass (layernorm.defer_reduction)
                ws = torch.empty(nws, dtype=torch.float32, device=sz.device)
                dg = torch.empty(D, dtype=torch.float32, device=sz.device)
                db = torch.empty(D, dtype=torch.float32, device=sz.device)
                later = (_ln.DEFER_REDUCE and ctx.meta == (torch.float32, torch.float32) and ctx.fresh()
                         and _ln.defer_reduction(ws, dg, db, None, D, *ctx.params, nblk=nws // (2 * D)))
                _lib.check(lib.vmasr_ln_gate_bwd_ws(_p(y), _p(sz), _p(dout), _p(g32), _p(b32), _p(stats[0]), _p(stats[1]), _p(dy), _p(dsz),
                                                    None if later else _p(dg), None if later else _p(db), _p(ws), B, D, L, code,
                                                    _lib.current_stream(sz.device)), "ln_gate_bwd_ws")
                return dy, dsz, dg.to(ctx.meta[0]), db.to(ctx.meta[1]), None
            dgb, = _lib.zeros_f32(sz.device, (2, D))
            _lib.check(lib.vmasr_ln_gate_bwd(_p(y), _p(sz), _p(dout), _p(g32), _p(b32), _p(stats[0]), _p(stats[1]), _p(dy), _p(dsz),
                                             _p(dgb[0]), _p(dgb[1]), B, D, L, code, _lib.current_stream(sz.device)), "ln_gate_bwd")
        return dy, dsz, dgb[0].to(ctx.meta[0]), dgb[1].to(ctx.meta[1]), None


class _LNGatePairsFn(torch.autograd.Function):
    """ln_gate on the un-merged pair outputs of the fused core (csrc/ss2d_glue.hip: ln_gate_pair_*)."""

    @staticmethod
    def forward(ctx, y02, y13, sz, gamma, beta, eps):
        B, H, W, D = sz.shape
        y02, y13 = y02.float().contiguous(), y13.float().contiguous()
        sz = sz.contiguous()
        g32, b32 = gamma.detach().float().contiguous(), beta.detach().float().contiguous()
        with torch.cuda.device(sz.device):
            out = torch.empty_like(sz)
            stats = torch.empty((2, B, H * W), dtype=torch.float32, device=sz.device)
            _lib.check(_lib.lib().vmasr_ln_gate_pair_fwd(_p(y02), _p(y13), _p(sz), _p(g32), _p(b32), _p(out), _p(stats[0]), _p(stats[1]),
                                                         B, D, H, W, float(eps), _lib.torch_dtype_code(sz.dtype),
                                                         _lib.current_stream(sz.device)), "ln_gate_pair_fwd")
        ctx.save_for_backward(y02, y13, sz, g32, b32, stats)
        ctx.meta = (gamma.dtype, beta.dtype)
        return out

    @staticmethod
    def backward(ctx, dout):
        y02, y13, sz, g32, b32, stats = ctx.saved_tensors
        B, H, W, D = sz.shape
        dout = dout.to(sz.dtype).contiguous()
        with torch.cuda.device(sz.device):
            dys = torch.empty((2,) + tuple(y02.shape), dtype=torch.float32, device=sz.device)
            dsz = torch.empty_like(sz)
            dgb, = _lib.zeros_f32(sz.device, (2, D))
            _lib.check(_lib.lib().vmasr_ln_gate_pair_bwd(_p(y02), _p(y13), _p(sz), _p(dout), _p(g32), _p(b32), _p(stats[0]), _p(stats[1]),
                                                         _p(dys[0]), _p(dys[1]), _p(dsz), _p(dgb[0]), _p(dgb[1]), B, D, H, W,
                                                         _lib.torch_dtype_code(sz.dtype), _lib.current_stream(sz.device)), "ln_gate_pair_bwd")
        return dys[0], dys[1], dsz, dgb[0].to(ctx.meta[0]), dgb[1].to(ctx.meta[1]), None


def pairs_supported(D, H, W, dtype):
    if os.environ.get("VMASR_SS2D_PAIRS", "1") != "1" or dtype not in (torch.float32, torch.float16, torch.bfloat16):
        return False
    return bool(_lib.lib().vmasr_ln_gate_pair_supported(int(D), int(H), int(W)))


def ln_gate_pairs(y02, y13, sz, gamma, beta, eps):
    """== ln_gate(y02 + transpose_hw(y13), sz, gamma, beta, eps) without the merged tensor (sz: (B, H, W, D))."""
    if not sz.is_cuda:
        raise RuntimeError("ln_gate_pairs: expected a CUDA (HIP) tensor; vm_asr_amd has no CPU path")
    return _LNGatePairsFn.apply(y02, y13, sz, gamma, beta, eps)


def ss2d_pre(xz):
    if not xz.is_cuda:
        raise RuntimeError("ss2d_pre: expected a CUDA (HIP) tensor; vm_asr_amd has no CPU path")
    return _PreFn.apply(xz)


def ln_gate(y, sz, gamma, beta, eps):
    if not sz.is_cuda:
        raise RuntimeError("ln_gate: expected a CUDA (HIP) tensor; vm_asr_amd has no CPU path")
    return _LNGateFn.apply(y, sz, gamma, beta, eps)
