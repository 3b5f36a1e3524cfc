"""Deterministic, construction-order-independent parameter values for full-size models (shared by the
golden generator and the tests, so a 3 M-parameter state never has to be stored): every tensor of
`state_dict()` is drawn from its own seeded generator, keyed by its position in the SORTED key list, with a
scale chosen by role so that the network keeps realistic dynamics."""
import math

import torch


def synth_state(model, seed=2024):
    sd = model.state_dict()
    with torch.no_grad():
        for i, k in enumerate(sorted(sd)):
            t = sd[k]
            if not t.is_floating_point():
                continue
            g = torch.Generator().manual_seed(seed + i)
            r = torch.randn(t.shape, generator=g, dtype=torch.float32)
            leaf = k.rsplit(".", 1)[-1]
            if leaf == "A_logs":                      # A = -exp(A_logs): around -1
                v = 0.2 * r
            elif leaf == "Ds":
                v = 1.0 + 0.1 * r
            elif leaf == "dt_projs_bias":             # softplus^-1 of 1e-3..1e-1 (model/vmamba.py:886-905)
                v = -4.0 + 0.8 * r
            elif t.dim() <= 1:
                v = (1.0 + 0.1 * r) if leaf == "weight" else 0.05 * r      # norm scales / biases
            else:
                fan_in = t[0].numel()
                v = r / math.sqrt(max(1, fan_in))
            t.copy_(v.to(t.dtype))
    return model


def canonical_phase(phase):
    """The reference's STFT phase with its rounding-noise indeterminacies pinned: frame 0 of a reflect-padded
    clip and the DC / Nyquist bins are exactly real, so angle() is 0 or +-pi by the sign of a zero; take
    0 / +pi (what the oracle and the HIP kernel produce by construction).  phase: (B, 1, F, M)."""
    p = phase.clone()
    pi = math.pi

    def fix(x):
        x = torch.where(x.abs() > pi - 1e-3, torch.full_like(x, pi), x)
        return torch.where(x.abs() < 1e-3, torch.zeros_like(x), x)
    p[..., :, 0] = fix(p[..., :, 0])
    p[..., 0, :] = fix(p[..., 0, :])
    p[..., -1, :] = fix(p[..., -1, :])
    return p
