"""SURVEY.md §5 debug aid: the deterministic-reduction switch (VMASR_DETERMINISTIC=1 / vmasr_set_deterministic).

The reference's backward is non-deterministic (fp32 atomics: cus/selective_scan_bwd_kernel.cuh:218-219,262-271), and so are the
parameter-gradient sums of several kernels here (dwconv, ln_gate, the discriminator's bias / first / last convolutions, the
spectral-norm products, the unfused scan / x_proj operators).  With the switch on, those atomics are taken in workgroup order
(csrc/common.h: det_enter / det_leave) and the hidden-split Mlp kernel adds its partial tiles in wave order: two evaluations
of the same full-size train-step backward from the same state give BIT-IDENTICAL gradients."""
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pytestmark = pytest.mark.gpu


def _grads(tr, batch, snap):
    tr._restore_training_state(snap)
    for m in tr.models.values():
        for p in m.parameters():
            p.grad = None
    torch.manual_seed(77)
    torch.cuda.manual_seed_all(77)
    tr._forward_backward(*batch)
    torch.cuda.synchronize()
    out = {}
    for key, m in tr.models.items():
        for n, p in m.named_parameters():
            if p.grad is not None:
                out[f"{key}.{n}"] = p.grad.detach().clone()
    return out


@pytest.mark.parametrize("workload", ["vm_asr_48k_MPD"])
def test_train_step_gradients_are_bit_reproducible_in_deterministic_mode(workload):
    import bench
    from vm_asr_amd import _lib
    lib = _lib.lib()
    cfg = bench.make_config(workload, 1)                       # the benchmark's model, one clip per step
    dev = torch.device("cuda:0")
    was = lib.vmasr_get_deterministic()
    lib.vmasr_set_deterministic(1)
    try:
        tr = bench.build_trainer(cfg, dev, amp=True, capturable=False)
        for m in tr.models.values():
            m.train()
        batch = bench.synth_batch(cfg, dev, 0)
        tr._forward_backward(*batch)                           # flat buffers / lazily built state exist
        snap = tr._snapshot_training_state()
        a = _grads(tr, batch, snap)
        b = _grads(tr, batch, snap)
        assert len(a) > 400 and a.keys() == b.keys()
        diff = [k for k in a if not torch.equal(a[k], b[k])]
        assert not diff, (len(diff), diff[:8])
        assert all(torch.isfinite(v).all() for v in a.values())
        assert lib.vmasr_det_timeouts() == 0                  # no ordered wait ran out
    finally:
        lib.vmasr_set_deterministic(was)


def test_deterministic_mode_changes_no_value_beyond_rounding():
    """the ordered tails compute the same sums as the atomics (order of the additions only): dwconv + ln_gate gradients with the
    switch on equal the default ones to fp32 rounding."""
    from vm_asr_amd import _lib
    from vm_asr_amd.dwconv import dwconv3x3_silu
    lib = _lib.lib()
    g = torch.Generator().manual_seed(0)
    x = torch.randn(2, 16, 64, 64, generator=g).cuda().requires_grad_(True)
    w = torch.randn(16, 1, 3, 3, generator=g).cuda().requires_grad_(True)
    b = torch.randn(16, generator=g).cuda().requires_grad_(True)
    gy = torch.randn(2, 16, 64, 64, generator=g).cuda()

    def run():
        for t in (x, w, b):
            t.grad = None
        dwconv3x3_silu(x, w, b).backward(gy)
        return [t.grad.clone() for t in (x, w, b)]
    was = lib.vmasr_get_deterministic()
    try:
        lib.vmasr_set_deterministic(0)
        ref = run()
        lib.vmasr_set_deterministic(1)
        d1, d2 = run(), run()
    finally:
        lib.vmasr_set_deterministic(was)
    for r, p, q in zip(ref, d1, d2):
        assert torch.equal(p, q)
        assert torch.allclose(r, p, rtol=1e-5, atol=1e-5 * r.abs().max().item())
