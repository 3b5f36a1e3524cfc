"""Channel-last LayerNorm on the HIP kernel (vm_asr_amd/csrc/ln.hip).

`LayerNorm` is an `nn.LayerNorm` (same parameters, same state_dict keys) whose forward runs
`vmasr_layer_norm_fwd/bwd` for GPU tensors normalised over their last dimension (C <= 1024) —
every LayerNorm on the VM-ASR path (model/vmamba.py:767-769,1793,1817; model/model.py:70,105-108,
620,631).  Under autocast the output is fp32, exactly what torch's autocast policy gives
F.layer_norm; the input is consumed in its own dtype (no separate cast pass).
Tensors that are not on the GPU (host-side tests, the cpu_baseline leg) use F.layer_norm.
"""
import ctypes

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _lib

__all__ = ["LayerNorm", "layer_norm"]


def _p(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


class _LayerNormFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, eps, out_dtype):
        C = x.shape[-1]
        x2 = x.reshape(-1, C)
        if not x2.is_contiguous():
            x2 = x2.contiguous()
        rows = x2.shape[0]
        w32 = None if weight is None else weight.detach().float().contiguous()
        b32 = None if bias is None else bias.detach().float().contiguous()
        with torch.cuda.device(x.device):
            y = torch.empty((rows, C), dtype=out_dtype, device=x.device)
            mean = torch.empty(rows, dtype=torch.float32, device=x.device)
            rstd = torch.empty(rows, dtype=torch.float32, device=x.device)
            _lib.check(_lib.lib().vmasr_layer_norm_fwd(_p(x2), _p(w32), _p(b32), _p(y), _p(mean), _p(rstd), rows, C,
                                                       float(eps), _lib.torch_dtype_code(x2.dtype),
                                                       _lib.torch_dtype_code(out_dtype),
                                                       _lib.current_stream(x.device)), "layer_norm_fwd")
        ctx.save_for_backward(x2, w32 if w32 is not None else torch.empty(0, device=x.device), mean, rstd)
        ctx.meta = (x.shape, weight is not None, bias is not None,
                    None if weight is None else weight.dtype, None if bias is None else bias.dtype)
        return y.view(x.shape)

    @staticmethod
    def backward(ctx, gy):
        x2, w32, mean, rstd = ctx.saved_tensors
        shape, has_w, has_b, wdt, bdt = ctx.meta
        rows, C = x2.shape
        gy2 = gy.reshape(rows, C)
        if gy2.dtype != x2.dtype and torch.float32 not in (gy2.dtype, x2.dtype):
            gy2 = gy2.float()  # (fp16, bf16) mixes are not built: go through fp32
        if not gy2.is_contiguous():
            gy2 = gy2.contiguous()
        lib = _lib.lib()
        with torch.cuda.device(x2.device):
            dx = torch.empty_like(x2)
            dg = torch.empty(C, dtype=torch.float32, device=x2.device) if has_w else None
            db = torch.empty(C, dtype=torch.float32, device=x2.device) if has_b else None
            ws = None
            if has_w or has_b:
                ws = torch.empty(lib.vmasr_layer_norm_bwd_workspace(rows, C) // 4, dtype=torch.float32, device=x2.device)
            _lib.check(lib.vmasr_layer_norm_bwd(_p(x2), _p(gy2), _p(w32) if has_w else None, _p(mean), _p(rstd), _p(dx),
                                                _p(dg), _p(db), _p(ws), rows, C, _lib.torch_dtype_code(x2.dtype),
                                                _lib.torch_dtype_code(gy2.dtype), _lib.current_stream(x2.device)),
                       "layer_norm_bwd")
        return (dx.view(shape), dg.to(wdt) if has_w else None, db.to(bdt) if has_b else None, None, None)


def layer_norm(x, weight=None, bias=None, eps=1e-5, feeds_gemm=False):
    """F.layer_norm over the last dimension.  Under autocast the result is fp32 (torch's policy for
    layer_norm) unless `feeds_gemm`: the caller promises that the only consumer is an autocast GEMM,
    which would round this very fp32 result to the autocast dtype first — so it is written in
    that dtype directly (bit-identical GEMM operand, one cast pass fewer in each direction)."""
    C = x.shape[-1]
    if x.is_cuda and C <= 1024 and x.dtype in (torch.float32, torch.float16, torch.bfloat16):
        out_dtype = x.dtype
        if torch.is_autocast_enabled("cuda"):
            out_dtype = torch.get_autocast_dtype("cuda") if feeds_gemm else torch.float32
        return _LayerNormFn.apply(x, weight, bias, eps, out_dtype)
    return F.layer_norm(x, (C,), weight, bias, eps)


class LayerNorm(nn.LayerNorm):
    feeds_gemm = False  # set by owners whose next op is a Linear (VSSBlock.norm / norm2, PatchMerging2D.norm)

    def forward(self, x):
        if len(self.normalized_shape) == 1:
            return layer_norm(x, self.weight, self.bias, self.eps, self.feeds_gemm)
        return super().forward(x)
