"""nn.Linear for the shapes of this path.  Same parameters and state_dict keys as nn.Linear.

* tiny-feature / many-row cases go to the HIP row-map kernels (vm_asr_amd/csrc/linear.hip): the
  d_model = 1 VSS block of the output layer and the 4->1 pointwise conv in front of it
  (model/model.py:862-885);
* many-row cases with a small weight (every projection of the high-resolution stages, the first MPD
  convolutions: rows = B*H*W up to 10^6, weight 32x5 ... 128x160) keep hipBLASLt for y and dx, but
  compute the weight gradient  dW = gy^T x  as a split-K batched GEMM: as one GEMM it is a single
  output tile reduced over all rows on one or two CUs (measured 170-380 us each, 12 ms/step);
* everything else stays F.linear.
"""
import ctypes
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _lib

__all__ = ["Linear", "linear"]

_MIN_ROWS = 16384


def _p(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


class _SmallLinearFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, out_dtype):
        out_f, in_f = weight.shape
        x2 = x.reshape(-1, in_f)
        if not x2.is_contiguous():
            x2 = x2.contiguous()
        rows = x2.shape[0]
        w32 = weight.detach().float().contiguous()
        b32 = None if bias is None else bias.detach().float().contiguous()
        with torch.cuda.device(x.device):
            y = torch.empty((rows, out_f), dtype=out_dtype, device=x.device)
            _lib.check(_lib.lib().vmasr_small_linear_fwd(_p(x2), _p(w32), _p(b32), _p(y), rows, in_f, out_f,
                                                         _lib.torch_dtype_code(x2.dtype), _lib.torch_dtype_code(out_dtype),
                                                         _lib.current_stream(x.device)), "small_linear_fwd")
        ctx.save_for_backward(x2, w32)
        ctx.meta = (x.shape, weight.dtype, None if bias is None else bias.dtype, x.requires_grad)
        return y.view(*x.shape[:-1], out_f)

    @staticmethod
    def backward(ctx, gy):
        x2, w32 = ctx.saved_tensors
        shape, wdt, bdt, need_dx = ctx.meta
        out_f, in_f = w32.shape
        rows = x2.shape[0]
        gy2 = gy.reshape(rows, out_f)
        # supported pairs: (x, gy) equal dtypes, or fp32 x with 16-bit gy
        if gy2.dtype != x2.dtype and not (x2.dtype == torch.float32 and gy2.dtype in (torch.float16, torch.bfloat16)):
            gy2 = gy2.to(x2.dtype)
        if not gy2.is_contiguous():
            gy2 = gy2.contiguous()
        lib = _lib.lib()
        with torch.cuda.device(x2.device):
            dx = torch.empty_like(x2) if need_dx else None
            dw = torch.empty((out_f, in_f), dtype=torch.float32, device=x2.device)
            db = torch.empty(out_f, dtype=torch.float32, device=x2.device) if bdt is not None else None
            ws = torch.empty(lib.vmasr_small_linear_bwd_workspace(rows, in_f, out_f) // 4, dtype=torch.float32,
                             device=x2.device)
            _lib.check(lib.vmasr_small_linear_bwd(_p(x2), _p(w32), _p(gy2), _p(dx), _p(dw), _p(db), _p(ws), rows, in_f,
                                                  out_f, _lib.torch_dtype_code(x2.dtype), _lib.torch_dtype_code(gy2.dtype),
                                                  _lib.current_stream(x2.device)), "small_linear_bwd")
        return (dx.view(shape) if need_dx else None, dw.to(wdt), db.to(bdt) if bdt is not None else None, None)


_SPLITK_CHUNK = int(os.environ.get("VMASR_SPLITK_CHUNK", "2048"))   # fewest rows of one K-split


def splitk_plan(rows, out_f, in_f):
    """Number of K-splits for dW (out_f, in_f) = gy^T (out_f, rows) @ x (rows, in_f); 1 = plain GEMM."""
    tiles = -(-out_f // 64) * -(-in_f // 64)          # output tiles one GEMM would spread over the CUs
    return min(rows // _SPLITK_CHUNK, max(1, 512 // tiles))


_F32_OUT = None   # does this torch build take out_dtype=float32 on 16-bit mm / bmm (hipBLASLt fp32 output)?


def _f32_out_ok(device):
    global _F32_OUT
    if _F32_OUT is None:
        try:
            a = torch.ones(16, 16, dtype=torch.bfloat16, device=device)
            ok = torch.mm(a, a, out_dtype=torch.float32).dtype == torch.float32
            ok = ok and torch.bmm(a[None], a[None], out_dtype=torch.float32).dtype == torch.float32
            _F32_OUT = bool(ok)
        except (RuntimeError, NotImplementedError, TypeError):
            _F32_OUT = False
    return _F32_OUT


def _mm_acc(a, b, acc):
    """a @ b (2-D or batched) returned in the accumulation dtype `acc` without a separate cast kernel
    where the GEMM can write fp32 itself."""
    if a.dtype == acc:
        return torch.bmm(a, b) if a.dim() == 3 else torch.mm(a, b)
    if a.is_cuda and acc == torch.float32 and _f32_out_ok(a.device):
        return torch.bmm(a, b, out_dtype=acc) if a.dim() == 3 else torch.mm(a, b, out_dtype=acc)
    return (torch.bmm(a, b) if a.dim() == 3 else torch.mm(a, b)).to(acc)


def weight_grad(gy2, x2, splits=None):
    """dW (fp32 for 16-bit inputs) = gy2^T @ x2 for row-major (rows, out_f), (rows, in_f); split over the
    rows into a batched GEMM + a sum of the partial products when the output is only a few tiles."""
    rows, out_f = gy2.shape
    in_f = x2.shape[1]
    S = splitk_plan(rows, out_f, in_f) if splits is None else splits
    acc = torch.float32 if gy2.dtype in (torch.float16, torch.bfloat16) else gy2.dtype
    if S < 4:
        return _mm_acc(gy2.t(), x2, acc)
    chunk = rows // S
    main = chunk * S
    # unflatten, not view: x2 may be a column slice of a wider matrix (row stride > in_f)
    part = _mm_acc(gy2[:main].unflatten(0, (S, chunk)).transpose(1, 2), x2[:main].unflatten(0, (S, chunk)), acc)
    dw = part.sum(0)
    if main < rows:
        dw += _mm_acc(gy2[main:].t(), x2[main:], acc)
    return dw


LP_ATTR = "_vmasr_lp"   # parameter attribute: low-precision (autocast dtype) shadow copy kept by the trainer
LPT_ATTR = "_vmasr_lpT"  # the same TRANSPOSED (2-D weights): operand of the fused kernels' backward (mlp.py, inproj.py, outproj.py)


def _shadow(t, like, cdt):
    """The trainer's low-precision shadow of parameter `t` (viewed like `like`, a view of t), or None."""
    sh = getattr(t, LP_ATTR, None) if t is not None else None
    if sh is None or sh.dtype != cdt:
        return None
    return sh if like is t else sh.view(like.shape)


def _skinny_ok(rows, in_f, out_f, x2, weight):
    """Many rows, few features: the streaming kernel of csrc/skinny.hip instead of a GEMM (VMASR_SKINNY=0: off)."""
    # where it beats the GEMM library under graph replay (tools/bench_skinny.py): >= 131 072 rows with <= 16 features each side (6.6-9.2 us
    # against 18.5-19.5 us); at 65 536 rows and 16-72 features hipBLASLt's 4-8 us are out of its reach (VMASR_SKINNY=all: every supported shape)
    mode = os.environ.get("VMASR_SKINNY", "1")
    if mode == "0" or not (x2.is_cuda and x2.dtype in (torch.float32, torch.bfloat16) and weight.dim() == 2):
        return False
    if mode != "all" and not (rows >= 131072 and in_f <= 16 and out_f <= 16):
        return False
    return bool(_lib.lib().vmasr_skinny_linear_supported(int(rows), int(in_f), int(out_f)))


def _skinny(x2, w32, bias32, out_dtype, transposed=False):
    """x2 (rows, in) contiguous, w32 fp32 (out, in) — or, transposed, the layer's (in, out)... weight read as W^T: y = x2 @ w32 (rows, w32.shape[1])."""
    rows, in_f = x2.shape
    if transposed:
        out_f, s_out, s_in = w32.shape[1], w32.stride(1), w32.stride(0)
    else:
        out_f, s_out, s_in = w32.shape[0], w32.stride(0), w32.stride(1)
    with torch.cuda.device(x2.device):
        y = torch.empty((rows, out_f), dtype=out_dtype, device=x2.device)
        _lib.check(_lib.lib().vmasr_skinny_linear(x2.data_ptr(), w32.data_ptr(), None if bias32 is None else bias32.data_ptr(), y.data_ptr(),
                                                  rows, in_f, out_f, s_out, s_in, _lib.torch_dtype_code(x2.dtype), _lib.torch_dtype_code(out_dtype),
                                                  _lib.current_stream(x2.device)), "skinny_linear")
    return y


class _LinearFn(torch.autograd.Function):
    """F.linear with (a) the operands cast to the compute dtype here - or taken from the trainer's
    shadow copies, which saves a cast kernel per weight per step -, (b) dW accumulated in fp32 by the
    GEMM itself and split over the rows where it would otherwise be one tile (weight_grad)."""

    @staticmethod
    def forward(ctx, x, weight, bias, cdt, w_lp, b_lp):
        out_f, in_f = weight.shape
        x2 = x.reshape(-1, in_f).to(cdt)
        if not x2.is_contiguous():
            x2 = x2.contiguous()
        ctx.skinny = _skinny_ok(x2.shape[0], in_f, out_f, x2, weight) and cdt in (torch.float32, torch.bfloat16)
        if ctx.skinny:      # many rows, few features: one streaming pass with the fp32 weight (csrc/skinny.hip)
            wc = weight.detach().float()
            y = _skinny(x2, wc, None if bias is None else bias.detach().float(), cdt)
        else:
            wc = w_lp if w_lp is not None else weight.detach().to(cdt)
            bc = None if bias is None else (b_lp if b_lp is not None else bias.detach().to(cdt))
            y = F.linear(x2, wc, bc)
        ctx.save_for_backward(x2, wc)
        ctx.meta = (x.shape, x.dtype, weight.dtype, None if bias is None else bias.dtype)
        return y.view(*x.shape[:-1], out_f)

    @staticmethod
    def backward(ctx, gy):
        x2, wc = ctx.saved_tensors
        shape, xdt, wdt, bdt = ctx.meta
        gy2 = gy.reshape(x2.shape[0], wc.shape[0]).to(x2.dtype)
        if not gy2.is_contiguous():
            gy2 = gy2.contiguous()
        if not ctx.needs_input_grad[0]:
            dx = None
        elif ctx.skinny:
            dx = _skinny(gy2, wc, None, gy2.dtype, transposed=True).view(shape).to(xdt)
        else:
            dx = (gy2 @ wc).view(shape).to(xdt)
        dw = weight_grad(gy2, x2).to(wdt) if ctx.needs_input_grad[1] else None
        db = None
        if bdt is not None and ctx.needs_input_grad[2]:
            db = gy2.sum(0, dtype=torch.float32 if gy2.dtype != torch.float64 else None).to(bdt)
        return dx, dw, db, None, None, None


_SplitKLinearFn = _LinearFn   # (tests)


def _f64acc(x2, w, b):
    """x2 (rows, in) @ w (out, in)^T + b with float64 accumulation (csrc/linear.hip), fp32 tensors."""
    rows, out_f = x2.shape[0], w.shape[0]
    with torch.cuda.device(x2.device):
        y = torch.empty((rows, out_f), dtype=torch.float32, device=x2.device)
        _lib.check(_lib.lib().vmasr_linear_f64acc(_p(x2), _p(w), _p(b), _p(y), rows, out_f, w.shape[1], _lib.current_stream(x2.device)),
                   "linear_f64acc")
    return y


class _LinearF64AccFn(torch.autograd.Function):
    """fp32 F.linear outside autocast (the parity path): forward and input gradient with float64 accumulation — the result is
    the correctly rounded fp32 value, i.e. this family adds nothing to the distance from the exact answer (hipBLASLt's fp32
    accumulation order at K >= 256 was 1.4-1.7x noisier than the reference's CPU evaluation: tools/linear_accuracy.py); the weight
    gradient keeps the split-K fp32 GEMM (the gradient gates are met with it)."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        out_f, in_f = weight.shape
        x2 = x.reshape(-1, in_f).contiguous()
        w = weight.detach().contiguous()
        y = _f64acc(x2, w, None if bias is None else bias.detach().contiguous())
        ctx.save_for_backward(x2, w)
        ctx.meta = (x.shape, bias is not None)
        return y.view(*x.shape[:-1], out_f)

    @staticmethod
    def backward(ctx, gy):
        x2, w = ctx.saved_tensors
        shape, has_b = ctx.meta
        gy2 = gy.reshape(x2.shape[0], w.shape[0]).float().contiguous()
        dx = _f64acc(gy2, w.t().contiguous(), None).view(shape) if ctx.needs_input_grad[0] else None
        dw = weight_grad(gy2, x2) if ctx.needs_input_grad[1] else None
        db = gy2.sum(0) if has_b and ctx.needs_input_grad[2] else None
        return dx, dw, db


def _use_f64acc(x, weight, bias):
    """The float64-accumulating fp32 Linear (the parity path's GEMM: 64 x 64 tiles, float64 FMAs — an order of magnitude slower
    than hipBLASLt).  VMASR_LINEAR_F64ACC: "1" always, "0" never, default "auto" = only where no gradient is recorded (evaluation,
    the Tester, inference: the fp32 forward + LSD parity claim) — an amp=False TRAINING run keeps the library GEMMs.  The parity
    tests set "1" (tests/conftest.py) so that their fp32 backward is adjudicated at the same accuracy."""
    mode = os.environ.get("VMASR_LINEAR_F64ACC", "auto")
    if mode == "0" or (mode != "1" and torch.is_grad_enabled()):
        return False
    return (x.is_cuda and x.dtype == torch.float32 and weight.dtype == torch.float32
            and (bias is None or bias.dtype == torch.float32) and not torch.is_autocast_enabled("cuda") and weight.dim() == 2
            and weight.shape[1] >= 16)


def linear(x, weight, bias=None, shadow_of=None, bias_shadow_of=None):
    """F.linear for this path.  Tiny in/out features over many GPU rows go to the HIP row-map kernel; under
    autocast, or for many rows with a small weight, the GEMM path goes through _LinearFn.  `shadow_of`:
    the parameter `weight` is a view of (for the shadow lookup)."""
    out_f, in_f = weight.shape
    if (x.is_cuda and x.dtype in (torch.float32, torch.float16, torch.bfloat16) and x.numel() // max(1, in_f) >= _MIN_ROWS
            and _lib.lib().vmasr_small_linear_supported(in_f, out_f)):
        out_dtype = torch.get_autocast_dtype("cuda") if torch.is_autocast_enabled("cuda") else x.dtype
        if out_dtype == x.dtype or x.dtype == torch.float32:
            return _SmallLinearFn.apply(x, weight, bias, out_dtype)
    if _use_f64acc(x, weight, bias):
        return _LinearF64AccFn.apply(x, weight, bias)
    if x.is_cuda and weight.requires_grad and torch.is_grad_enabled() and x.is_floating_point():
        amp = torch.is_autocast_enabled("cuda")
        rows = x.numel() // max(1, in_f)
        if amp or splitk_plan(rows, out_f, in_f) >= 4:
            cdt = torch.get_autocast_dtype("cuda") if amp else x.dtype
            w_lp = _shadow(shadow_of if shadow_of is not None else weight, weight, cdt)
            b_lp = _shadow(bias_shadow_of if bias_shadow_of is not None else bias, bias, cdt) if bias is not None else None
            return _LinearFn.apply(x, weight, bias, cdt, w_lp, b_lp)
    return F.linear(x, weight, bias)


class Linear(nn.Linear):
    def forward(self, x):
        return linear(x, self.weight, self.bias)
