"""The Mlp branch of a VSSBlock as one HIP operator on the matrix cores (vm_asr_amd/csrc/mlp.hip):

    fused_mlp_residual(x, norm2, mlp, scale=None)  ==  x + scale * mlp(norm2(x))

i.e. `x + self.drop_path(self.mlp(self.norm2(x)))` of VSSBlock._forward (model/vmamba.py:1832-1837; Mlp :483-509 with
exact-erf GELU; timm DropPath as a per-sample scale) under bf16 autocast, for the residual stream x (..., d) in fp32 or bf16,
d in {8, 16, 32, 64}, hidden = 4 d.  Forward: ONE kernel (LayerNorm, fc1, bias, GELU, fc2, bias, residual; bf16
MFMA 32x32x16, fp32 accumulation; the hidden activations never leave the register file).  Backward: one kernel that
recomputes the forward and produces dxn plus the bf16 operands of the weight-gradient GEMMs, then LayerNorm's backward
(with the residual gradient folded in) and two GEMMs whose ones-column carries the bias gradients.  No CPU fallback.
"""
import ctypes
import os

import torch

from . import _lib
from . import layernorm as _ln
from .wgrad import weight_grad_finished
from .linear import LP_ATTR, LPT_ATTR, weight_grad

__all__ = ["fused_mlp_residual", "supported"]


def _p(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def supported(x, norm, mlp):
    """GPU, bf16 autocast, fp32 channel-last stream, plain Mlp (GELU exact, no dropout) of a supported width.
    (The channel-first subclasses Linear2d / LayerNorm2d act on axis 1: never the row kernel, whatever x.shape[-1] is.)"""
    if os.environ.get("VMASR_FUSED_MLP", "1") != "1" or not x.is_cuda or x.dtype not in (torch.float32, torch.bfloat16):
        return False
    if not (torch.is_autocast_enabled("cuda") and torch.get_autocast_dtype("cuda") == torch.bfloat16):
        return False
    fc1, fc2 = getattr(mlp, "fc1", None), getattr(mlp, "fc2", None)
    if not (isinstance(fc1, torch.nn.Linear) and isinstance(fc2, torch.nn.Linear) and isinstance(norm, torch.nn.LayerNorm)
            and all(type(m).__name__ not in ("Linear2d", "LayerNorm2d") for m in (fc1, fc2, norm))):
        return False
    if not isinstance(mlp.act, torch.nn.GELU) or getattr(mlp.act, "approximate", "none") != "none":
        return False
    if (mlp.drop.p != 0.0 and mlp.training) or fc1.bias is None or fc2.bias is None or norm.weight is None or norm.bias is None:
        return False
    d = x.shape[-1]
    if tuple(norm.normalized_shape) != (d,) or fc1.in_features != d or fc2.out_features != d or fc2.in_features != fc1.out_features:
        return False
    return bool(_lib.lib().vmasr_mlp_supported(int(d), int(fc1.out_features)))


def _bf16(w):
    """The trainer's bf16 shadow of a parameter when it has one (no cast kernel), else a cast."""
    sh = getattr(w, LP_ATTR, None)
    return sh if (sh is not None and sh.dtype == torch.bfloat16) else w.detach().to(torch.bfloat16)


def _bf16_t(w, wb):
    """W^T (bf16, contiguous) for the backward kernels when the trainer keeps a transposed shadow of parameter `w` and the
    forward used its shadow `wb` (both are refreshed together by the AdamW kernel); None -> the caller transposes `wb` itself."""
    sh = getattr(w, LPT_ATTR, None)
    if (sh is not None and getattr(w, LP_ATTR, None) is wb and sh.dtype == torch.bfloat16 and wb.dim() == 2
            and tuple(sh.shape) == (wb.shape[1], wb.shape[0])):
        return sh
    return None


class _FusedMlpFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, w1, b1, w2, b2, scale, eps):
        d = x.shape[-1]
        x2 = x.reshape(-1, d)
        if not x2.is_contiguous():
            x2 = x2.contiguous()
        rows = x2.shape[0]
        g32, be32 = gamma.detach().float().contiguous(), beta.detach().float().contiguous()
        b1f, b2f = b1.detach().float().contiguous(), b2.detach().float().contiguous()
        w1b, w2b = _bf16(w1).contiguous(), _bf16(w2).contiguous()
        rps = rows // scale.numel() if scale is not None else 0
        sc = None if scale is None else scale.detach().float().contiguous().view(-1)
        with torch.cuda.device(x.device):
            y = torch.empty_like(x2)
            _lib.check(_lib.lib().vmasr_mlp_fwd(_p(x2), _p(g32), _p(be32), float(eps), _p(w1b), _p(b1f), _p(w2b), _p(b2f), _p(sc), rps,
                                                _p(y), rows, d, _lib.torch_dtype_code(x2.dtype), _lib.current_stream(x.device)), "mlp_fwd")
        ctx.save_for_backward(x2, g32, be32, w1b, b1f, w2b, sc)
        ctx.wts = (_bf16_t(w1, w1b), _bf16_t(w2, w2b))
        ctx.meta = (x.shape, eps, rps, gamma.dtype, beta.dtype, w1.dtype, b1.dtype, w2.dtype, b2.dtype)
        if any(ctx.needs_input_grad[1:3]):
            _ln.note_use(gamma, beta)
        ctx.fresh = lambda: gamma.grad is None and beta.grad is None and _ln.used_once(gamma, beta)
        ctx.params = (gamma, beta)
        if any(ctx.needs_input_grad[3:7]):
            _ln.note_use(w1, b1, w2, b2)
        ctx.wparams = (w1, b1, w2, b2)
        ctx.fresh_w = lambda: (all(p.grad is None and p.dtype == torch.float32 for p in (w1, b1, w2, b2)) and _ln.used_once(w1, b1, w2, b2))
        return y.view(x.shape)

    @staticmethod
    def backward(ctx, gy):
        x2, g32, be32, w1b, b1f, w2b, sc = ctx.saved_tensors
        shape, eps, rps, gdt, bedt, w1dt, b1dt, w2dt, b2dt = ctx.meta
        rows, d = x2.shape
        hd = w1b.shape[0]
        gy2 = gy.reshape(rows, d).to(x2.dtype)
        if not gy2.is_contiguous():
            gy2 = gy2.contiguous()
        lib = _lib.lib()
        dev = x2.device
        bf = dict(dtype=torch.bfloat16, device=dev)
        with torch.cuda.device(dev):
            w1t = ctx.wts[0] if ctx.wts[0] is not None else w1b.t().contiguous()
            w2t = ctx.wts[1] if ctx.wts[1] is not None else w2b.t().contiguous()
            dxn = torch.empty((rows, d), **bf)
            xn_aug = torch.empty((rows, d + 8), **bf)
            gys = torch.empty((rows, d), **bf)
            act_aug = torch.empty((rows, hd + 8), **bf)
            gpre = torch.empty((rows, hd), **bf)
            stats = torch.empty((2, rows), dtype=torch.float32, device=dev)
            _lib.check(lib.vmasr_mlp_bwd(_p(x2), _p(gy2), _p(g32), _p(be32), float(eps), _p(w1b), _p(w1t), _p(b1f), _p(w2t), _p(sc), rps,
                                         _p(dxn), _p(xn_aug), _p(gys), _p(act_aug), _p(gpre), _p(stats[0]), _p(stats[1]), rows, d,
                                         _lib.torch_dtype_code(x2.dtype), _lib.current_stream(dev)), "mlp_bwd")
            # dx = gy + LayerNorm'(dxn); dgamma, dbeta  (one launch: csrc/ln.hip with the residual gradient folded in)
            dx = torch.empty_like(x2)
            dg_ = torch.empty(d, dtype=torch.float32, device=dev)      # separate tensors: autograd adopts each as a .grad
            db_ = torch.empty(d, dtype=torch.float32, device=dev)
            ws = torch.empty(lib.vmasr_layer_norm_bwd_workspace(rows, d) // 4, dtype=torch.float32, device=dev)
            later = (gdt == torch.float32 and bedt == torch.float32 and ctx.fresh() and _ln.defer_reduction(ws, dg_, db_, rows, d, *ctx.params))
            _lib.check(lib.vmasr_layer_norm_bwd_res(_p(x2), _p(dxn), _p(g32), _p(stats[0]), _p(stats[1]), _p(gy2), _p(dx),
                                                    None if later else _p(dg_), None if later else _p(db_), _p(ws), rows, d, _lib.torch_dtype_code(x2.dtype), _lib.BF16, _lib.current_stream(dev)),
                       "layer_norm_bwd_res")
        # [dW1 | db1 | 0] = gpre^T xn_aug,  [dW2 | db2 | 0] = gys^T act_aug  (fp32 accumulation, split over the rows when few tiles);
        # the sum over the slabs and the split into contiguous dW / db: one launch for ALL queued GEMMs of the pass (wgrad.py)
        fresh_w = ctx.fresh_w()
        dw1, db1 = weight_grad_finished(gpre, xn_aug, d, ctx.wparams[0], ctx.wparams[1], fresh_w)
        dw2, db2 = weight_grad_finished(gys, act_aug, hd, ctx.wparams[2], ctx.wparams[3], fresh_w)
        return (dx.view(shape), dg_.to(gdt), db_.to(bedt), dw1.to(w1dt), db1.to(b1dt), dw2.to(w2dt), db2.to(b2dt), None, None)


def fused_mlp_residual(x, norm, mlp, scale=None):
    """x + scale * mlp(norm(x)); `scale`: None or a (B,) / (B,1,1,1) per-sample tensor (DropPath keep mask / keep)."""
    if not x.is_cuda:
        raise RuntimeError("fused_mlp_residual: expected a CUDA (HIP) tensor; vm_asr_amd has no CPU path")
    return _FusedMlpFn.apply(x, norm.weight, norm.bias, mlp.fc1.weight, mlp.fc1.bias, mlp.fc2.weight, mlp.fc2.bias, scale, norm.eps)
