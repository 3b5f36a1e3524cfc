"""One-launch AdamW step on the HIP library (csrc/adamw.hip) over the state of an ordinary torch.optim.AdamW.

The optimiser object stays the owner of its state (`exp_avg`, `exp_avg_sq`, `step` per parameter): checkpoints keep
torch's `state_dict` layout, interchangeable with the reference's (utils/optimizer.py:16-50, base/base_trainer.py:130-179).
This class only replaces `optimizer.step()` in the replayed training step: a device table of (param, grad, exp_avg,
exp_avg_sq, bf16 shadow) pointers built once, `step` counters advanced by one multi-tensor add, then one kernel.
"""
import ctypes

import numpy as np
import torch

from . import _lib

__all__ = ["HipAdamWStep"]

_ITEM = np.dtype([("p", "u8"), ("g", "u8"), ("m", "u8"), ("v", "u8"), ("lp", "u8"), ("n", "i8"), ("wd", "f4"), ("vec", "i4"),
                  ("lpt", "u8"), ("rows", "i4"), ("cols", "i4")])


class HipAdamWStep:
    """step() == optimizer.step() for a capturable torch.optim.AdamW (non-amsgrad, no maximize) whose parameters'
    `.grad` are fixed views (the trainer's flat gradient buffers).  `shadows`: {id(param): bf16 tensor} refreshed in
    the same pass; `shadows_t`: {id(2-D param): bf16 tensor of the transposed shape}, likewise.  Raises ValueError if the optimiser does not qualify (the caller keeps optimizer.step())."""

    def __init__(self, optimizer, shadows=None, shadows_t=None):
        if not isinstance(optimizer, torch.optim.AdamW):
            raise ValueError("not an AdamW")
        groups = optimizer.param_groups
        g0 = groups[0]
        if any(g["amsgrad"] or g["maximize"] or not g.get("capturable", False) for g in groups):
            raise ValueError("needs capturable, non-amsgrad, non-maximize AdamW")
        if any(g["betas"] != g0["betas"] or g["eps"] != g0["eps"] for g in groups):
            raise ValueError("param groups differ in betas / eps")
        lr = g0["lr"]
        if not (torch.is_tensor(lr) and lr.is_cuda and all(g["lr"] is lr for g in groups)):
            raise ValueError("needs ONE device learning-rate tensor shared by the param groups (trainer.lr_to_device)")
        self.lr = lr if lr.dtype == torch.float32 else None
        if self.lr is None:
            raise ValueError("learning-rate tensor must be float32")
        self.betas, self.eps = (float(g0["betas"][0]), float(g0["betas"][1])), float(g0["eps"])
        shadows, shadows_t = shadows or {}, shadows_t or {}
        chunk = _lib.lib().vmasr_adamw_chunk()
        items, chunks, steps, keep = [], [], [], []
        for g in groups:
            for p in g["params"]:
                if p.grad is None:
                    continue                                   # never-used parameters: torch skips them too
                st = optimizer.state.get(p)
                if not st or "exp_avg" not in st:
                    raise ValueError("optimizer state not initialised yet (run one optimizer.step() first)")
                m, v, step, lp = st["exp_avg"], st["exp_avg_sq"], st["step"], shadows.get(id(p))
                ok = (p.is_cuda and p.dtype == torch.float32 and p.is_contiguous() and p.grad.is_contiguous() and m.is_contiguous()
                      and v.is_contiguous() and p.grad.dtype == torch.float32 and m.dtype == torch.float32 and v.dtype == torch.float32
                      and torch.is_tensor(step) and step.is_cuda and step.dtype == torch.float32
                      and (lp is None or (lp.is_contiguous() and lp.dtype == torch.bfloat16 and lp.shape == p.shape)))
                if not ok:
                    raise ValueError("a parameter / state tensor does not qualify (dtype, device or layout)")
                lpt = shadows_t.get(id(p))
                if lpt is not None and not (p.dim() == 2 and lpt.is_contiguous() and lpt.dtype == torch.bfloat16
                                            and tuple(lpt.shape) == (p.shape[1], p.shape[0])):
                    raise ValueError("a transposed shadow does not match its parameter")
                ptrs = [p.data_ptr(), p.grad.data_ptr(), m.data_ptr(), v.data_ptr()]
                vec = all(q % 16 == 0 for q in ptrs) and (lp is None or lp.data_ptr() % 8 == 0)
                idx = len(items)
                items.append((*ptrs, lp.data_ptr() if lp is not None else 0, p.numel(), float(g["weight_decay"]), int(vec),
                              lpt.data_ptr() if lpt is not None else 0, p.shape[0] if lpt is not None else 0,
                              p.shape[1] if lpt is not None else 0))
                chunks.extend((idx, c) for c in range(-(-p.numel() // chunk)))
                steps.append(step)
                keep.append((p, p.grad, m, v, (lp, lpt)))
        if not items:
            raise ValueError("no parameter with a gradient")
        # the kernel takes ONE step count for the bias corrections (torch uses each tensor's own): every state must be at
        # the same step (one-off host sync at build time, outside any capture).  A checkpoint whose parameters skipped
        # steps (grad None in some of them) does not qualify: the caller keeps optimizer.step().
        sv = torch.stack([s_.reshape(()) for s_ in steps])
        if not bool((sv == sv[0]).all()):
            raise ValueError("optimizer state holds different step counts per parameter")
        self._skipped = [p for g in groups for p in g["params"] if p.grad is None]
        dev = self.lr.device
        self.items = torch.from_numpy(np.array(items, dtype=_ITEM).view(np.uint8).copy()).to(dev)
        self.chunks = torch.tensor(chunks, dtype=torch.int32, device=dev).contiguous()
        self.nchunks, self.total = len(chunks), sum(it[5] for it in items)
        self.steps, self._keep, self._ptrs = steps, keep, [it[:4] for it in items]
        self.optimizer = optimizer

    def still_valid(self):
        """The table holds raw pointers: it is stale once a parameter's .grad was re-bound (zero_grad(set_to_none), a new
        flat buffer) or the optimiser state was replaced (load_state_dict creates new exp_avg / exp_avg_sq / step tensors)."""
        state = self.optimizer.state
        if any(g["lr"] is not self.lr for g in self.optimizer.param_groups):
            return False                                  # the learning-rate tensor was replaced (scheduler / load_state_dict)
        if any(p.grad is not None for p in self._skipped):
            return False                                  # a parameter that had no gradient at build time has one now
        for (p, _, _, _, _), (pp, gp, mp, vp), step in zip(self._keep, self._ptrs, self.steps):
            st = state.get(p)
            if (p.grad is None or not st or p.data_ptr() != pp or p.grad.data_ptr() != gp or st["exp_avg"].data_ptr() != mp
                    or st["exp_avg_sq"].data_ptr() != vp or st["step"] is not step):
                return False
        return True

    @torch.no_grad()
    def step(self):
        torch._foreach_add_(self.steps, 1)
        dev = self.lr.device
        with torch.cuda.device(dev):
            _lib.check(_lib.lib().vmasr_adamw_step(self.items.data_ptr(), self.chunks.data_ptr(), self.nchunks, self.total,
                                                   self.lr.data_ptr(), self.steps[0].data_ptr(), ctypes.c_float(self.betas[0]),
                                                   ctypes.c_float(self.betas[1]), ctypes.c_float(self.eps), _lib.current_stream(dev)),
                       "adamw_step")
