"""The output side of SS2D.forwardv2 as one HIP operator on the matrix cores (vm_asr_amd/csrc/mlp.hip: outproj_kernel):

    fused_out_proj_residual(g, out_proj, x, scale=None)  ==  x + scale * out_proj(g)

i.e. `out = self.out_proj(y)` (model/vmamba.py:1551; no bias, dropout p = 0) followed by the VSSBlock's
`x = input + self.drop_path(...)` (:1826-1827) under bf16 autocast, for g (B, H, W, 2d) bf16 = the gated LayerNorm output of
`ss2d_glue.ln_gate`, the residual stream x (B, H, W, d) in fp32 or bf16 and d in {8, 16, 32, 64}.  Forward: one kernel instead of
GEMM + (DropPath multiply +) add.  Backward: one kernel (dg = scale * gy . W in bf16 for ln_gate's backward, gys = scale * gy as
the operand of the weight gradient) + one split-K GEMM; the stream's own gradient is gy itself.  No CPU fallback.
"""
import ctypes
import os

import torch

from . import _lib
from .wgrad import weight_grad_finished
from . import layernorm as _ln
from .linear import weight_grad
from .mlp import _bf16, _bf16_t

__all__ = ["fused_out_proj_residual", "supported"]


def _p(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def supported(g, out_proj, x, dropout=None):
    """GPU, bf16 autocast, bf16 gate output, fp32 / bf16 stream, bias-free out_proj of a supported width, no active dropout."""
    if os.environ.get("VMASR_FUSED_OUTPROJ", "1") != "1" or not (g.is_cuda and x.is_cuda):
        return False
    if not (torch.is_autocast_enabled("cuda") and torch.get_autocast_dtype("cuda") == torch.bfloat16):
        return False
    if g.dtype != torch.bfloat16 or x.dtype not in (torch.float32, torch.bfloat16):
        return False
    if not isinstance(out_proj, torch.nn.Linear) or type(out_proj).__name__ == "Linear2d" or out_proj.bias is not None:
        return False
    if dropout is not None and not isinstance(dropout, torch.nn.Identity) and getattr(dropout, "p", 0.0) != 0.0 and dropout.training:
        return False
    d, di = x.shape[-1], g.shape[-1]
    if out_proj.in_features != di or out_proj.out_features != d or g.shape[:-1] != x.shape[:-1]:
        return False
    return bool(_lib.lib().vmasr_outproj_supported(int(d), int(di)))


class _OutProjFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, g, w, x, scale):
        d, di = x.shape[-1], g.shape[-1]
        g2, x2 = g.reshape(-1, di), x.reshape(-1, d)
        if not g2.is_contiguous():
            g2 = g2.contiguous()
        if not x2.is_contiguous():
            x2 = x2.contiguous()
        rows = x2.shape[0]
        wb = _bf16(w).contiguous()
        rps = rows // scale.numel() if scale is not None else 0
        sc = None if scale is None else scale.detach().float().contiguous().view(-1)
        with torch.cuda.device(x.device):
            y = torch.empty_like(x2)
            _lib.check(_lib.lib().vmasr_outproj_fwd(_p(g2), _p(wb), _p(x2), _p(sc), rps, _p(y), rows, d, _lib.torch_dtype_code(x2.dtype),
                                                    _lib.current_stream(x.device)), "outproj_fwd")
        ctx.save_for_backward(g2, wb, sc)
        ctx.wt = _bf16_t(w, wb)
        ctx.meta = (g.shape, x.shape, x2.dtype, rps, w.dtype)
        if ctx.needs_input_grad[1]:
            _ln.note_use(w)
        ctx.wparam = w
        ctx.fresh_w = lambda: w.grad is None and w.dtype == torch.float32 and _ln.used_once(w)
        return y.view(x.shape)

    @staticmethod
    def backward(ctx, gy):
        g2, wb, sc = ctx.saved_tensors
        gshape, xshape, xdt, rps, wdt = ctx.meta
        rows, di = g2.shape
        d = di // 2
        gy2 = gy.reshape(rows, d)
        if gy2.dtype not in (torch.float32, torch.bfloat16):
            gy2 = gy2.float()
        if not gy2.is_contiguous():
            gy2 = gy2.contiguous()
        dev = g2.device
        with torch.cuda.device(dev):
            wt = ctx.wt if ctx.wt is not None else wb.t().contiguous()      # (2d, d)
            dg = torch.empty((rows, di), dtype=torch.bfloat16, device=dev)
            gys = torch.empty((rows, d), dtype=torch.bfloat16, device=dev)
            _lib.check(_lib.lib().vmasr_outproj_bwd(_p(gy2), _p(wt), _p(sc), rps, _p(dg), _p(gys), rows, d, _lib.torch_dtype_code(gy2.dtype),
                                                    _lib.current_stream(dev)), "outproj_bwd")
        dw = None
        if ctx.needs_input_grad[1]:                                          # (d, 2d) fp32; finished with the pass' other weight gradients
            dw, _ = weight_grad_finished(gys, g2, di, ctx.wparam, None, ctx.fresh_w())
        return dg.view(gshape), None if dw is None else dw.to(wdt), gy.to(xdt) if ctx.needs_input_grad[2] else None, None


def fused_out_proj_residual(g, out_proj, x, scale=None):
    """x + scale * out_proj(g); `scale`: None or a per-sample tensor (DropPath keep mask / keep)."""
    if not x.is_cuda:
        raise RuntimeError("fused_out_proj_residual: expected a CUDA (HIP) tensor; vm_asr_amd has no CPU path")
    return _OutProjFn.apply(g, out_proj.weight, x, scale)
