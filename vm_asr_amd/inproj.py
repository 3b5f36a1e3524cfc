"""The input side of SS2D.forwardv2 as one HIP operator on the matrix cores (vm_asr_amd/csrc/mlp.hip: inproj_kernel):

    fused_in_proj(x, norm, in_proj) -> (xT (B, 2d, H, W), sz (B, H, W, 2d))
        == xz = in_proj(norm(x)); x', z = xz.chunk(2, -1); xT = x'.permute(0, 3, 1, 2).contiguous(); sz = SiLU(z)
           (model/vmamba.py:1826-1827 pre-norm of VSSBlock, :1535-1542 of SS2D.forwardv2)

under bf16 autocast, for the fp32 / bf16 residual stream x (B, H, W, d), d in {8, 16, 32, 64}, in_proj without bias, H*W a
multiple of 32; `norm` a LayerNorm or nn.Identity.  One launch instead of LayerNorm + GEMM + ss2d_pre; backward = one kernel
(recompute) + LayerNorm's backward + one weight-gradient GEMM.  No CPU fallback.
"""
import ctypes
import os

import torch

from . import _lib
from . import layernorm as _ln
from .wgrad import weight_grad_finished
from .linear import LP_ATTR, weight_grad
from .mlp import _bf16_t

__all__ = ["fused_in_proj", "supported"]


def _p(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def supported(x, norm, in_proj):
    if os.environ.get("VMASR_FUSED_INPROJ", "1") != "1" or not x.is_cuda or x.dim() != 4 or x.dtype not in (torch.float32, torch.bfloat16):
        return False
    if not (torch.is_autocast_enabled("cuda") and torch.get_autocast_dtype("cuda") == torch.bfloat16):
        return False
    if not isinstance(in_proj, torch.nn.Linear) or type(in_proj).__name__ == "Linear2d" or in_proj.bias is not None:
        return False
    d = x.shape[-1]
    if isinstance(norm, torch.nn.LayerNorm) and type(norm).__name__ != "LayerNorm2d":
        if tuple(norm.normalized_shape) != (d,) or norm.weight is None or norm.bias is None:
            return False
    elif not isinstance(norm, torch.nn.Identity):
        return False
    if in_proj.in_features != d:
        return False
    return bool(_lib.lib().vmasr_inproj_supported(int(d), int(in_proj.out_features), int(x.shape[1] * x.shape[2])))


def _bf16(w):
    sh = getattr(w, LP_ATTR, None)
    return sh if (sh is not None and sh.dtype == torch.bfloat16) else w.detach().to(torch.bfloat16)


class _InProjFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, weight, eps):
        B, H, W, d = x.shape
        L = H * W
        x2 = x.reshape(-1, d)
        if not x2.is_contiguous():
            x2 = x2.contiguous()
        g32 = None if gamma is None else gamma.detach().float().contiguous()
        b32 = None if beta is None else beta.detach().float().contiguous()
        wb = _bf16(weight).contiguous()
        with torch.cuda.device(x.device):
            xT = torch.empty((B, 2 * d, H, W), dtype=torch.bfloat16, device=x.device)
            sz = torch.empty((B, H, W, 2 * d), dtype=torch.bfloat16, device=x.device)
            _lib.check(_lib.lib().vmasr_inproj_fwd(_p(x2), _p(g32), _p(b32), float(eps), _p(wb), _p(xT), _p(sz), B * L, L, d,
                                                   _lib.torch_dtype_code(x2.dtype), _lib.current_stream(x.device)), "inproj_fwd")
        ctx.save_for_backward(x2, g32 if g32 is not None else torch.empty(0, device=x.device),
                              b32 if b32 is not None else torch.empty(0, device=x.device), wb)
        ctx.meta = (x.shape, eps, gamma is not None, None if gamma is None else gamma.dtype, None if beta is None else beta.dtype, weight.dtype)
        ctx.wt = _bf16_t(weight, wb)
        if gamma is not None and any(ctx.needs_input_grad[1:3]):
            _ln.note_use(gamma, beta)
        ctx.fresh = lambda: gamma is not None and gamma.grad is None and beta.grad is None and _ln.used_once(gamma, beta)
        ctx.params = (gamma, beta)
        if ctx.needs_input_grad[3]:
            _ln.note_use(weight)
        ctx.wparam = weight
        ctx.fresh_w = lambda: weight.grad is None and weight.dtype == torch.float32 and _ln.used_once(weight)
        return xT, sz

    @staticmethod
    def backward(ctx, dxT, dsz):
        x2, g32, b32, wb = ctx.saved_tensors
        shape, eps, has_norm, gdt, bedt, wdt = ctx.meta
        B, H, W, d = shape
        L, rows = H * W, x2.shape[0]
        dev = x2.device
        bf = dict(dtype=torch.bfloat16, device=dev)
        lib = _lib.lib()
        dxT = (torch.zeros((B, 2 * d, H, W), **bf) if dxT is None else dxT.to(torch.bfloat16)).contiguous()
        dsz = (torch.zeros((B, H, W, 2 * d), **bf) if dsz is None else dsz.to(torch.bfloat16)).contiguous()
        with torch.cuda.device(dev):
            wt = ctx.wt if ctx.wt is not None else wb.t().contiguous()
            dxn = torch.empty((rows, d), **bf)
            xn = torch.empty((rows, d), **bf)
            gpre = torch.empty((rows, 4 * d), **bf)
            stats = torch.empty((2, rows), dtype=torch.float32, device=dev) if has_norm else None
            _lib.check(lib.vmasr_inproj_bwd(_p(x2), _p(g32) if has_norm else None, _p(b32) if has_norm else None, float(eps), _p(wb), _p(wt),
                                            _p(dxT), _p(dsz), _p(dxn), _p(xn), _p(gpre), _p(stats[0]) if has_norm else None,
                                            _p(stats[1]) if has_norm else None, rows, L, d, _lib.torch_dtype_code(x2.dtype),
                                            _lib.current_stream(dev)), "inproj_bwd")
            dg = db = None
            if has_norm:
                dx = torch.empty_like(x2)
                dg = torch.empty(d, dtype=torch.float32, device=dev)
                db = torch.empty(d, dtype=torch.float32, device=dev)
                ws = torch.empty(lib.vmasr_layer_norm_bwd_workspace(rows, d) // 4, dtype=torch.float32, device=dev)
                later = (gdt == torch.float32 and bedt == torch.float32 and ctx.fresh() and _ln.defer_reduction(ws, dg, db, rows, d, *ctx.params))
                _lib.check(lib.vmasr_layer_norm_bwd(_p(x2), _p(dxn), _p(g32), _p(stats[0]), _p(stats[1]), _p(dx), None if later else _p(dg),
                                                    None if later else _p(db), _p(ws), rows, d, _lib.torch_dtype_code(x2.dtype), _lib.BF16,
                                                    _lib.current_stream(dev)), "layer_norm_bwd")
                dg, db = dg.to(gdt), db.to(bedt)
            else:
                dx = dxn.to(x2.dtype)
        # (4d, d) fp32, split over the rows when few tiles; finished together with the pass' other weight gradients (wgrad.py)
        dw, _ = weight_grad_finished(gpre, xn, d, ctx.wparam, None, ctx.fresh_w())
        return dx.view(shape), dg, db, dw.to(wdt), None


def fused_in_proj(x, norm, in_proj):
    if not x.is_cuda:
        raise RuntimeError("fused_in_proj: expected a CUDA (HIP) tensor; vm_asr_amd has no CPU path")
    ln = isinstance(norm, torch.nn.LayerNorm)
    return _InProjFn.apply(x, norm.weight if ln else None, norm.bias if ln else None, in_proj.weight, norm.eps if ln else 0.0)
