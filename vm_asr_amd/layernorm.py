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


# ---- deferred dgamma / dbeta -----------------------------------------------------------------------------------------
# A LayerNorm backward = one kernel (dx + per-workgroup partials of dgamma / dbeta) + a 9 us reduce launch; a training step has
# 65 of them.  With DEFER_REDUCE the calls of one backward pass only write their partials and queue the reduction; ONE launch at
# the end of the pass (autograd's queue_callback: runs when the engine finishes the current backward) fills all dgamma / dbeta.
# Only safe when nobody reads a parameter gradient before the backward pass ends and gradients are not accumulated in place while
# it runs — the trainer's flat-buffer mode (grads dropped before the pass, packed after it) — so it is OFF unless the trainer
# turns it on; a call whose weight / bias already has a .grad reduces at once.
DEFER_REDUCE = False
_pending = []
_uses = {}          # id(weight) -> forward calls since the last flush / reset: a weight used twice in one graph must not defer
                    # (autograd would add the second call's gradient into the first's still-empty tensor)


def note_use(*params):
    """(called from autograd.Function.forward, where grad mode is off: the caller checks ctx.needs_input_grad)"""
    if DEFER_REDUCE:
        for t in params:
            if t is not None:
                _uses[id(t)] = _uses.get(id(t), 0) + 1


def used_once(*params):
    return all(_uses.get(id(t), 0) == 1 for t in params if t is not None)


_cb_queued = [False]   # the end-of-pass callback of the running backward has been queued
_flush_hooks = []      # other modules' end-of-pass work (vm_asr_amd/wgrad.py): objects with .flush() and .reset()


def ensure_callback():
    """Queue ONE end-of-pass callback per backward pass (autograd runs it when the pass completes)."""
    if not _cb_queued[0]:
        _cb_queued[0] = True
        torch.autograd.Variable._execution_engine.queue_callback(_flush_all)


def _flush_all():
    _cb_queued[0] = False
    _flush_pending()
    for h in _flush_hooks:
        h.flush()


def reset_uses():
    """Start of a step (trainer) / of a test: forget the use counts AND whatever an aborted backward left queued.
    autograd runs the end-of-pass callback only when a backward COMPLETES; after an exception (OOM, kernel error, a failed
    graph capture that the trainer catches) the queue would stay non-empty for good, `defer_reduction` would never queue the
    flush again and every later step would return uninitialised dgamma / dbeta.  Entries of an aborted pass are dropped:
    their gradients belong to a step that did not happen."""
    _uses.clear()
    _pending.clear()
    _cb_queued[0] = False
    for h in _flush_hooks:
        h.reset()


class deferred:
    """`with layernorm.deferred(on):` — DEFER_REDUCE for the forward + backward of one step only (process-global state must not
    leak into other backward passes of the process: a second trainer in ddp / accumulation mode, a tester, user code)."""

    def __init__(self, on=True):
        self.on = bool(on)

    def __enter__(self):
        global DEFER_REDUCE
        self.prev = DEFER_REDUCE
        DEFER_REDUCE = self.on
        reset_uses()
        return self

    def __exit__(self, *exc):
        global DEFER_REDUCE
        DEFER_REDUCE = self.prev
        if exc[0] is not None:
            reset_uses()
        return False


def _flush_pending():
    """Reduce the queued partials (one launch per <= 96 calls) on the current stream."""
    global _pending
    items, _pending = _pending, []
    _uses.clear()
    if not items:
        return
    import numpy as np
    dev = items[0][0].device
    n = len(items)
    parts = np.array([i[0].data_ptr() for i in items], dtype=np.uint64)
    dgs, dbs = np.array([i[1][1] for i in items], dtype=np.uint64), np.array([i[2][1] for i in items], dtype=np.uint64)
    nblk, cs = np.array([i[3] for i in items], dtype=np.int32), np.array([i[4] for i in items], dtype=np.int32)
    with torch.cuda.device(dev):
        _lib.check(_lib.lib().vmasr_layer_norm_bwd_reduce_multi(parts.ctypes.data, dgs.ctypes.data, dbs.ctypes.data, nblk.ctypes.data,
                                                                cs.ctypes.data, n, _lib.current_stream(dev)), "layer_norm_bwd_reduce_multi")
        # autograd normally STEALS the returned gradient tensor as param.grad (fresh .grad, sole owner, contiguous); if it cloned
        # instead, copy the reduced values over (correct either way; the extra copy only in the unusual case)
        for _, (gst, gptr), (bst, bptr), _, C, weight, bias in items:
            for param, st, ptr in ((weight, gst, gptr), (bias, bst, bptr)):
                if param is None or st is None or param.grad is None or param.grad.data_ptr() == ptr:
                    continue
                param.grad.copy_(torch.empty(0, dtype=torch.float32, device=dev).set_(st, 0, (C,)).view_as(param.grad))


def defer_reduction(ws, dg, db, rows, C, weight=None, bias=None, nblk=None):
    """Queue (partials -> dg, db) for the end of the running backward pass; False if deferral does not apply.
    Only the STORAGES of dg / db are kept (a second reference to the tensors themselves would make autograd clone them
    instead of adopting them as .grad).  nblk: rows of 2 C partials in ws (default: LayerNorm's own grid for `rows`)."""
    if not DEFER_REDUCE:
        return False
    ensure_callback()    # (an aborted backward cannot leave entries behind: reset_uses() at the start of every step drops them)
    ref = lambda t: (None, 0) if t is None else (t.untyped_storage(), t.data_ptr())   # noqa: E731
    if nblk is None:
        nblk = int(_lib.lib().vmasr_layer_norm_bwd_blocks(rows, C))
    _pending.append((ws, ref(dg), ref(db), int(nblk), C, weight, bias))
    return True


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
        if any(ctx.needs_input_grad[1:3]):
            note_use(weight, bias)
        ctx.fresh = lambda: (all(getattr(t, "grad", None) is None for t in (weight, bias) if t is not None)
                             and used_once(weight, bias))
        ctx.params = (weight, bias)
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
            # fp32 parameters whose .grad is still empty: their gradients can be filled at the end of the pass (see DEFER_REDUCE)
            later = (ws is not None and DEFER_REDUCE and wdt in (None, torch.float32) and bdt in (None, torch.float32) and ctx.fresh()
                     and defer_reduction(ws, dg, db, rows, C, *ctx.params))
            _lib.check(lib.vmasr_layer_norm_bwd(_p(x2), _p(gy2), _p(w32) if has_w else None, _p(mean), _p(rstd), _p(dx),
                                                None if later else _p(dg), None if later else _p(db), _p(ws), rows, C,
                                                _lib.torch_dtype_code(x2.dtype), _lib.torch_dtype_code(gy2.dtype),
                                                _lib.current_stream(x2.device)), "layer_norm_bwd")
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
