"""SURVEY.md §5 debug aid: the deterministic-reduction switch (VMASR_DETERMINISTIC=1 / vmasr_set_deterministic).

The reference's backward is non-deterministic (fp32 atomics: cus/selective_scan_bwd_kernel.cuh:218-219,262-271), and so are the
parameter-gradient sums of several kernels here (dwconv, ln_gate, the discriminator's bias / first / last convolutions, the
spectral-norm products, the unfused scan / x_proj operators).  With the switch on, those atomics are taken in workgroup order
(csrc/common.h: det_enter / det_leave) and the hidden-split Mlp kernel adds its partial tiles in wave order: two evaluations
of the same full-size train-step backward from the same state give BIT-IDENTICAL gradients.

Round 6: this file is also where the step's STREAM LAYOUTS are pinned against each other.  The round-5 failure of
the first test on the driver's box was not a summation-order effect and not a buffer-lifetime bug: one partial sum of
small_linear_bwd<bf16, bf16, 1, 4> came out different although its inputs were bit-equal — only with kernels of a second stream on
the chip, only in the compiler's packed-fp32 form of the accumulation (profiles/r06_determinism_hunt.md; fixed by building without
packed-fp32 instructions, csrc/Makefile).  test_two_stream_eager_step_is_bit_reproducible is the regression test for exactly that."""
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


def _trainer(workload, batch, amp=True, capturable=False, drop_path=None):
    import bench
    cfg = bench.make_config(workload, batch)
    if drop_path is not None:
        cfg.defrost()
        cfg.MODEL.VSSM.DROP_PATH_RATE = drop_path
        cfg.freeze()
    dev = torch.device("cuda:0")
    tr = bench.build_trainer(cfg, dev, amp=amp, capturable=capturable)
    for m in tr.models.values():
        m.train()
    return tr, bench.synth_batch(cfg, dev, 0)


def test_two_stream_eager_step_is_bit_reproducible(monkeypatch):
    """The EAGER TWO-STREAM step (discriminator on the side stream beside the generator: the layout the headline graph captures) in
    deterministic mode — forced with VMASR_TWO_STREAM=force: generator and discriminator share no ticketed kernel, so the ordered
    tails cannot meet — evaluated 40 times from one snapshot: every evaluation bit-equal to the first, no ordered wait timed out.
    Before the fix this failed in 3-10 % of the evaluations (always a (4,1) weight of the d_model-1 output block: in_proj / fc1)."""
    from vm_asr_amd import _lib
    lib = _lib.lib()
    was = lib.vmasr_get_deterministic()
    lib.vmasr_set_deterministic(1)
    monkeypatch.setenv("VMASR_TWO_STREAM", "force")
    try:
        tr, batch = _trainer("vm_asr_48k_MPD", 1)
        assert tr._two_streams()
        tr._forward_backward(*batch)
        snap = tr._snapshot_training_state()
        ref = _grads(tr, batch, snap)
        assert len(ref) > 400 and all(torch.isfinite(v).all() for v in ref.values())
        bad = {}
        for it in range(40):
            g = _grads(tr, batch, snap)
            for k in ref:
                if not torch.equal(ref[k], g[k]):
                    bad.setdefault(k, []).append(it)
        assert not bad, bad
        assert lib.vmasr_det_timeouts() == 0
        # and the one-stream layout is chosen whenever the mode is on without the test hook — however it was switched on
        monkeypatch.setenv("VMASR_TWO_STREAM", "1")
        assert not tr._two_streams()
    finally:
        lib.vmasr_set_deterministic(was)


def _dist(a, ref):
    worst, name = 0.0, None
    for k, r in ref.items():
        d = float((a[k].double() - r.double()).abs().max()) / max(float(r.double().abs().max()), 1e-30)
        if not d <= worst:
            worst, name = d, k
    return worst, name


LAYOUT_TOL = 2e-3    # of each tensor's max; measured (tools/stream_layout_spread.py --no-amp, profiles/r06_stream_layout_spread.log): the
#                      atomic additions' order alone moves the smallest tensors (|g|max 1e-9 ... 1e-12) by up to 2.5e-4 of their max


def test_stream_layouts_agree_with_the_one_stream_step(monkeypatch):
    """One fp32 train step (amp off: under bf16 autocast a one-ulp flip in the forward moves whole gradient tensors by per cent, which
    hides everything) from one snapshot under every stream layout the trainer can pick, ten times each:
        two streams, eager   |   captured, generator on one stream   |   captured, phase lane (the headline layout)
    every run of every layout equals the ONE-STREAM deterministic step to LAYOUT_TOL on every gradient tensor — no outlier run.
    A missing edge between the streams (a tensor read before it is written, a block re-used while the other stream still reads it)
    shows up as a run that is far off; the order of the atomic additions is the only difference allowed."""
    from vm_asr_amd import _lib
    from vm_asr_amd.trainer import unwrap
    lib = _lib.lib()
    was = lib.vmasr_get_deterministic()
    tr, batch = _trainer("vm_asr_48k_MPD", 1, amp=False, capturable=True)
    tr._forward_backward(*batch)
    snap = tr._snapshot_training_state()
    try:
        lib.vmasr_set_deterministic(1)
        assert not tr._two_streams()
        ref = _grads(tr, batch, snap)
        assert all(torch.equal(v, w) for v, w in zip(ref.values(), _grads(tr, batch, snap).values()))
    finally:
        lib.vmasr_set_deterministic(was)
    assert tr._two_streams()
    worst = {}
    for it in range(10):
        worst["two streams, eager"] = max(worst.get("two streams, eager", (0.0, None)), _dist(_grads(tr, batch, snap), ref))
    tr._restore_training_state(snap)
    state = {k: {n: t.detach().clone() for n, t in unwrap(m).state_dict().items()} for k, m in tr.models.items()}
    for pin in ("one", "lane:0.625"):
        monkeypatch.setenv("VMASR_STEP_VARIANT", pin)
        tr2, _ = _trainer("vm_asr_48k_MPD", 1, amp=False, capturable=True)
        tr2.train_step(*batch)
        assert tr2.enable_graphs(batch, warmup=2)
        assert bool(unwrap(tr2.models["generator"]).phase_lane) == pin.startswith("lane")
        for k, m in tr2.models.items():
            unwrap(m).load_state_dict(state[k])
        tr2._refresh_shadows()
        snap2 = tr2._snapshot_training_state()
        names = {key: [f"{key}.{n}" for n, p in unwrap(tr2.models[key]).named_parameters() if any(p is q for q in tr2._flat_params[key])]
                 for key in tr2._flat_params}
        for it in range(10):
            tr2._restore_training_state(snap2)
            torch.manual_seed(77)
            torch.cuda.manual_seed_all(77)
            tr2._graphed(*batch)
            torch.cuda.synchronize()
            got = {nme: v.detach().clone() for key in tr2._flat_params for nme, v in zip(names[key], tr2._flat_views[key])}
            assert got.keys() == ref.keys()
            worst[f"captured, {pin}"] = max(worst.get(f"captured, {pin}", (0.0, None)), _dist(got, ref))
        del tr2
        torch.cuda.empty_cache()
    print("stream layouts vs the one-stream step (worst tensor of the worst run):", worst)
    assert all(w[0] <= LAYOUT_TOL for w in worst.values()), worst


def test_captured_generator_only_step_in_deterministic_mode():
    """a generator-only workload always put its phase branch on a second stream when captured — also in deterministic mode,
    where both branches launch the same ticketed kernels (one ticket word per kernel id: ONE STREAM ONLY, csrc/common.h).  Now the mode
    keeps the captured step on one stream: replays from one snapshot give bit-equal gradients and no ordered wait runs out."""
    from vm_asr_amd import _lib
    from vm_asr_amd.trainer import unwrap
    lib = _lib.lib()
    was = lib.vmasr_get_deterministic()
    lib.vmasr_set_deterministic(1)
    try:
        tr, batch = _trainer("vm_asr_48k", 1, capturable=True)
        tr.train_step(*batch)
        assert tr.enable_graphs(batch, warmup=2)
        assert not unwrap(tr.models["generator"]).phase_lane
        snap = tr._snapshot_training_state()
        runs = []
        for it in range(6):
            tr._restore_training_state(snap)
            torch.manual_seed(77)
            torch.cuda.manual_seed_all(77)
            tr._graphed(*batch)
            torch.cuda.synchronize()
            runs.append(tr._flat["generator"].detach().clone())
        assert torch.isfinite(runs[0]).all() and runs[0].abs().max() > 0
        assert all(torch.equal(runs[0], r) for r in runs[1:])
        assert lib.vmasr_det_timeouts() == 0
    finally:
        lib.vmasr_set_deterministic(was)


def test_general_state_projection_gradients_are_ordered_in_deterministic_mode():
    """Round 6: the d_state > 1 projections' weight gradients (csrc/xproj_n.hip: 32 x 32 tiles over chunks of positions, float atomics) had no
    ordered form — with the mode on, 54 tensors of the d_state-32 train step still differed between two evaluations (x_proj_weight,
    dt_projs_weight of every block).  Now workgroups add in workgroup order and their four waves in wave order: bit-equal, equal to the
    default mode to rounding, no ordered wait timed out."""
    from vm_asr_amd import _lib
    from vm_asr_amd.xproj import x_proj_dt
    lib = _lib.lib()
    g = torch.Generator().manual_seed(3)
    B, K, D, N, R, L = 2, 4, 64, 32, 4, 4096
    xs = torch.randn(B, K, D, L, generator=g).cuda().requires_grad_(True)
    wx = (0.1 * torch.randn(K, R + 2 * N, D, generator=g)).cuda().requires_grad_(True)
    wdt = (0.1 * torch.randn(K, D, R, generator=g)).cuda().requires_grad_(True)
    gd, gb, gc = (torch.randn(s, generator=g).cuda() for s in ((B, K * D, L), (B, K, N, L), (B, K, N, L)))

    def run():
        for t in (xs, wx, wdt):
            t.grad = None
        dts, Bs, Cs = x_proj_dt(xs, wx, wdt, N)
        (dts.reshape(B, K * D, L) * gd).sum().add((Bs * gb).sum()).add((Cs * gc).sum()).backward()
        torch.cuda.synchronize()
        return [t.grad.clone() for t in (xs, wx, wdt)]
    was = lib.vmasr_get_deterministic()
    try:
        lib.vmasr_set_deterministic(0)
        ref = run()
        lib.vmasr_set_deterministic(1)
        runs = [run() for _ in range(4)]
        assert lib.vmasr_det_timeouts() == 0
    finally:
        lib.vmasr_set_deterministic(was)
    for r in runs[1:]:
        assert all(torch.equal(a, b) for a, b in zip(runs[0], r))
    for a, b in zip(ref, runs[0]):
        assert torch.allclose(a, b, rtol=1e-4, atol=1e-5 * a.abs().max().item())
