"""Weight gradients of the fused VSS-block operators (vm_asr_amd/mlp.py, inproj.py, outproj.py) finished in ONE launch per
backward pass (csrc/wgrad.hip).

dW = gy^T x over 10^4 .. 10^6 rows runs as a batched GEMM over S row slabs (linear.weight_grad); what follows — the sum over the
slabs, the bias gradient split off the operand's ones column, and both as contiguous tensors (autograd clones a strided gradient
before adopting it as .grad) — was sum + two strided copies per GEMM: ~6 launches of 3-5 us per block backward.  Here the GEMM's
partial products are queued and every queued item of the pass is finished by one `vmasr_wgrad_finish_multi` launch from autograd's
end-of-pass callback (shared with the LayerNorm reductions: layernorm.ensure_callback).  Deferral needs what layernorm.DEFER_REDUCE
needs — nobody reads a parameter gradient before the pass ends, fresh .grad, parameter used once in the graph; otherwise the
finish runs at once (same kernel, one item).
"""
import numpy as np
import torch

from . import _lib
from . import layernorm as _ln
from .linear import _mm_acc, splitk_plan

__all__ = ["weight_grad_finished"]


class _Queue:
    def __init__(self):
        self.items = []

    def reset(self):
        self.items = []

    def flush(self):
        items, self.items = self.items, []
        if items:
            _launch(items)
            # autograd normally adopts the returned tensor as .grad; if it cloned instead, copy the finished values over
            for it in items:
                for param, (st, ptr, shape) in ((it["wparam"], it["dw_ref"]), (it["bparam"], it["db_ref"])) + tuple(it.get("extra", ())):
                    if param is None or st is None or param.grad is None or param.grad.data_ptr() == ptr:
                        continue
                    param.grad.copy_(torch.empty(0, dtype=torch.float32, device=param.grad.device).set_(st, 0, shape).view_as(param.grad))


_Q = _Queue()
_ln._flush_hooks.append(_Q)


def _launch(items):
    n = len(items)
    arr = lambda k, dt: np.array([it[k] for it in items], dtype=dt)   # noqa: E731
    parts, dws, dbs = arr("parts_ptr", np.uint64), arr("dw_ptr", np.uint64), arr("db_ptr", np.uint64)
    e1s = np.array([it.get("e1_ptr", 0) for it in items], dtype=np.uint64)
    e2s = np.array([it.get("e2_ptr", 0) for it in items], dtype=np.uint64)
    Ss, Ns, Ks, lds = arr("S", np.int32), arr("N", np.int32), arr("K", np.int32), arr("ld", np.int32)
    dev = items[0]["parts"].device
    with torch.cuda.device(dev):
        _lib.check(_lib.lib().vmasr_wgrad_finish_multi(parts.ctypes.data, dws.ctypes.data, dbs.ctypes.data, e1s.ctypes.data, e2s.ctypes.data, Ss.ctypes.data, Ns.ctypes.data,
                                                       Ks.ctypes.data, lds.ctypes.data, n, _lib.current_stream(dev)), "wgrad_finish_multi")


def _parts(gy2, x2):
    """(S, out_f, in_f) fp32 partial products of gy2^T x2 over S equal row slabs (S = 1: the whole product)."""
    rows, out_f = gy2.shape
    in_f = x2.shape[1]
    S = splitk_plan(rows, out_f, in_f)
    if S < 4 or rows % S:
        return _mm_acc(gy2.t(), x2, torch.float32).unsqueeze(0)
    chunk = rows // S
    return _mm_acc(gy2.unflatten(0, (S, chunk)).transpose(1, 2), x2.unflatten(0, (S, chunk)), torch.float32)


def weight_grad_finished(gy2, x_aug, K, weight=None, bias=None, fresh=False):
    """dW (out_f, K) and — if x_aug has more than K columns — db (out_f) = column K of gy2^T x_aug, both contiguous fp32 tensors.
    x_aug (rows, >= K) bf16 / fp32: the GEMM operand [x | 1 | 0..] (or plain x).  weight / bias: the parameters the gradients
    belong to; fresh: the caller's check that their .grad is empty and they are used once in the graph — then, under
    layernorm.DEFER_REDUCE, the finish is queued for the end of the pass; the returned tensors are filled by then."""
    parts = _parts(gy2, x_aug).contiguous()
    S, N, ld = parts.shape
    has_b = ld > K
    dev = parts.device
    with torch.cuda.device(dev):
        dw = torch.empty((N, K), dtype=torch.float32, device=dev)
        db = torch.empty(N, dtype=torch.float32, device=dev) if has_b else None
    ref = lambda t: (None, 0, None) if t is None else (t.untyped_storage(), t.data_ptr(), tuple(t.shape))   # noqa: E731
    item = dict(parts=parts, parts_ptr=parts.data_ptr(), dw_ptr=dw.data_ptr(), db_ptr=0 if db is None else db.data_ptr(), S=S, N=N, K=K,
                ld=ld, wparam=weight, bparam=bias, dw_ref=ref(dw), db_ref=ref(db))
    if _ln.DEFER_REDUCE and fresh and dev.type == "cuda":
        _ln.ensure_callback()
        _Q.items.append(item)
    else:
        _launch([item])
    return dw, db


def finish_slabs(parts, K, outs, params, fresh):
    """Generic form: parts (S, N, ld) fp32 slabs; outs = (dW (N, K), col K, col K+1, col K+2) contiguous fp32 tensors (trailing ones
    may be None); params: the parameters they are gradients of (for the adopted-or-cloned check).  Queued like weight_grad_finished."""
    S, N, ld = parts.shape
    dw, db, e1, e2 = (tuple(outs) + (None,) * 4)[:4]
    ref = lambda t: (None, 0, None) if t is None else (t.untyped_storage(), t.data_ptr(), tuple(t.shape))   # noqa: E731
    pw, pb, p1, p2 = (tuple(params) + (None,) * 4)[:4]
    item = dict(parts=parts, parts_ptr=parts.data_ptr(), dw_ptr=dw.data_ptr(), db_ptr=0 if db is None else db.data_ptr(),
                e1_ptr=0 if e1 is None else e1.data_ptr(), e2_ptr=0 if e2 is None else e2.data_ptr(), S=S, N=N, K=K, ld=ld,
                wparam=pw, bparam=pb, dw_ref=ref(dw), db_ref=ref(db), extra=((p1, ref(e1)), (p2, ref(e2))))
    if _ln.DEFER_REDUCE and fresh and parts.device.type == "cuda":
        _ln.ensure_callback()
        _Q.items.append(item)
    else:
        _launch([item])
