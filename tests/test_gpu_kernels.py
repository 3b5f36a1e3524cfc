"""GPU parity tests (pytest -m gpu): every HIP entry point, called through the C ABI
(vm_asr_amd.* -> ctypes -> libvmasr_hip.so), against
  (1) the committed reference-generated goldens (tests/golden/*.npz),
  (2) the CPU oracle (oracle/) on seeded inputs at sizes it finishes in seconds,
  (3) size-independent properties at BASELINE.json's full sizes.
Tolerances: fp32 1e-4 (north_star), bf16/fp16 the reference's own kernel-test tolerances
(kernels/selective_scan/test_selective_scan.py:585-587)."""
import glob
import os

import numpy as np
import pytest
import torch

import oracle

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")
DEV = "cuda:0"


def _t(a, dtype=torch.float32):
    return None if a is None else torch.from_numpy(np.ascontiguousarray(a)).to(DEV).to(dtype)


def _close(got, want, rtol, atol, what=""):
    got = got.detach().float().cpu().numpy().astype(np.float64) if torch.is_tensor(got) else np.asarray(got, np.float64)
    want = np.asarray(want, np.float64)
    assert got.shape == want.shape, (what, got.shape, want.shape)
    assert np.isfinite(got).all(), f"{what}: non-finite values"
    excess = np.abs(got - want) - (atol + rtol * np.abs(want))
    if got.size:
        import errtable
        w = int(np.argmax(np.abs(got - want)))
        errtable.record(what, got, want, atol + rtol * float(np.abs(want).flat[w]))
    assert excess.max() <= 0, f"{what}: max|diff|={np.abs(got - want).max():.3e} (tol rtol={rtol}, atol={atol})"


def _scaled(want, tol=1e-4):
    return tol * max(1.0, float(np.abs(want).max()))


@pytest.fixture(autouse=True)
def _reset_tune():
    from vm_asr_amd import selective_scan as ss
    ss.tune(-1, -1)
    yield
    ss.tune(-1, -1)


# ------------------------------------------------------------------------------------------
# selective scan vs goldens (reference output on the reference's own test distribution)
# ------------------------------------------------------------------------------------------
SCAN_FILES = sorted(glob.glob(os.path.join(GOLDEN, "scan_*.npz")))
TUNES = [(-1, -1), (1, 0), (1, 1), (2, 1), (4, 0), (1, 2), (2, 2)]


@pytest.mark.parametrize("tune", TUNES, ids=[f"r{r}s{s}" for r, s in TUNES])
@pytest.mark.parametrize("path", SCAN_FILES, ids=[os.path.basename(p)[5:-4] for p in SCAN_FILES])
def test_scan_golden(path, tune):
    from vm_asr_amd import selective_scan as ss
    z = np.load(path)
    itype = int(z["meta"][5])
    dt = {0: torch.float32, 1: torch.float16, 2: torch.bfloat16}[itype]
    rtol, atol = {0: (1e-4, 1e-4), 1: (3e-3, 5e-3), 2: (3e-2, 5e-2)}[itype]
    ss.tune(*tune)
    keys = sorted({k.split("_")[0] for k in z.files if k[0] == "D" and "_" in k})
    for key in keys:
        D = _t(z[f"{key}_D"]) if f"{key}_D" in z.files else None
        bias = _t(z[f"{key}_bias"]) if f"{key}_bias" in z.files else None
        sp = key.endswith("s1")
        u, delta, dout = _t(z[f"{key}_u"], dt), _t(z[f"{key}_delta"], dt), _t(z[f"{key}_dout"], dt)
        A, Bm, Cm = _t(z["A"]), _t(z["B"], dt), _t(z["C"], dt)
        out, x = ss.fwd(u, delta, A, Bm, Cm, D, bias, sp, 1)
        _close(out, z[f"{key}_out"], rtol, atol, f"{key} out")
        # last saved state == reference last_state
        N = A.shape[1]
        last = x[:, :, -1].view(x.shape[0], x.shape[1], N, 2)[..., 1]
        _close(last, z[f"{key}_last"], 1e-4, 1e-4, f"{key} last_state")
        du, dd, dA, dB, dC, dD, db = ss.bwd(u, delta, A, Bm, Cm, D, bias, dout, x, sp, 1)
        g = 1 if itype == 0 else 2
        _close(du, z[f"{key}_du"], rtol * g, atol * g if itype else _scaled(z[f"{key}_du"]), f"{key} du")
        _close(dd, z[f"{key}_ddelta"], rtol * 5 if itype else rtol, atol * 10 if itype else _scaled(z[f"{key}_ddelta"]), f"{key} ddelta")
        _close(dA, z[f"{key}_dA"], 1e-3 if itype else 1e-4, _scaled(z[f"{key}_dA"], 5e-3 if itype else 1e-4), f"{key} dA")
        _close(dB, z[f"{key}_dB"], rtol, atol if itype else _scaled(z[f"{key}_dB"]), f"{key} dB")
        _close(dC, z[f"{key}_dC"], rtol, atol if itype else _scaled(z[f"{key}_dC"]), f"{key} dC")
        if D is not None:
            _close(dD, z[f"{key}_dD"], 1e-3 if itype else 1e-4, _scaled(z[f"{key}_dD"], 5e-3 if itype else 1e-4), f"{key} dD")
        if bias is not None:
            _close(db, z[f"{key}_dbias"], 1e-3 if itype else 1e-4, _scaled(z[f"{key}_dbias"], 5e-3 if itype else 1e-4), f"{key} dbias")


# ------------------------------------------------------------------------------------------
# selective scan vs the CPU oracle on the §8 call shapes (scaled down in L where needed)
# ------------------------------------------------------------------------------------------
def _scan_inputs(Bn, KD, G, N, L, seed=0):
    g = torch.Generator().manual_seed(seed)
    A = -0.5 * torch.rand(KD, N, generator=g)
    Bm = torch.randn(Bn, G, N, L, generator=g)
    Cm = torch.randn(Bn, G, N, L, generator=g)
    D = torch.randn(KD, generator=g)
    bias = 0.5 * torch.rand(KD, generator=g)
    u = torch.randn(Bn, KD, L, generator=g)
    delta = 0.5 * torch.rand(Bn, KD, L, generator=g)
    dout = torch.randn(Bn, KD, L, generator=g)
    return u, delta, A, Bm, Cm, D, bias, dout


ORACLE_SHAPES = [  # (B, KD, G, N, L)
    (2, 8, 4, 1, 16384), (1, 64, 4, 1, 4096), (2, 128, 4, 1, 2048), (1, 256, 4, 1, 4096),
    (2, 512, 4, 1, 1024), (2, 1024, 4, 1, 256), (1, 8, 4, 32, 1024), (1, 16, 4, 1, 777),
]


@pytest.mark.parametrize("shape", ORACLE_SHAPES, ids=["x".join(map(str, s)) for s in ORACLE_SHAPES])
def test_scan_vs_oracle(shape):
    from vm_asr_amd import selective_scan as ss
    Bn, KD, G, N, L = shape
    cpu = _scan_inputs(*shape)
    u, delta, A, Bm, Cm, D, bias, dout = [t.to(DEV) for t in cpu]
    want = oracle.sscan_fwd(*[t.numpy() for t in cpu[:7]], True)
    wdu, wdd, wdA, wdB, wdC, wdD, wdb = oracle.sscan_bwd(*[t.numpy() for t in cpu[:7]], cpu[7].numpy(), True)
    for tune in ((-1, -1), (1, 0), (1, 1), (1, 2), (2, 2)):
        ss.tune(*tune)
        out, x = ss.fwd(u, delta, A, Bm, Cm, D, bias, True, 1)
        # 1e-4 of the tensor scale: |out| reaches ~150 on this distribution and the fp32
        # sequential oracle itself sits 1e-4 (absolute) away from a float64 evaluation
        _close(out, want, 1e-4, _scaled(want), f"out tune={tune}")
        du, dd, dA, dB, dC, dD, db = ss.bwd(u, delta, A, Bm, Cm, D, bias, dout, x, True, 1)
        for name, got, w in (("du", du, wdu), ("ddelta", dd, wdd), ("dA", dA, wdA), ("dB", dB, wdB),
                             ("dC", dC, wdC), ("dD", dD, wdD), ("dbias", db, wdb)):
            _close(got, w, 1e-4, _scaled(w), f"{name} tune={tune}")


@pytest.mark.parametrize("shape", [(2, 8, 4, 1, 16384), (2, 128, 4, 1, 2048), (1, 64, 4, 32, 2048), (1, 8, 4, 8, 4113)],
                         ids=["x".join(map(str, s)) for s in [(2, 8, 4, 1, 16384), (2, 128, 4, 1, 2048), (1, 64, 4, 32, 2048), (1, 8, 4, 8, 4113)]])
def test_scan_fp64_adjudicated(shape):
    """The 1e-4-of-scale gates above are loose in absolute terms (|out| reaches ~150): here the HIP kernels' fp32 results are
    adjudicated against the float64 build of the oracle.  Forward: at least as close to float64 (L2) as the oracle's own fp32
    sequential evaluation (x 1.5).  Backward: the oracle's gradient code accumulates in double (its fp32 results sit 4e-8 from
    float64 — not an fp32 algorithm's distance), so the gate is absolute: every gradient within 3e-6 of its float64 norm, i.e. a
    few tens of fp32 epsilons after sums over up to 32 states and 4 096 steps (measured 2e-7 .. 1.8e-6; tools/scan_accuracy.py).
    d_state 1 (sscan.hip) and general d_state (sscan_n.hip: packed pairs, polynomial decay), walk and split plans."""
    from vm_asr_amd import selective_scan as ss
    cpu = _scan_inputs(*shape, seed=11)
    args = [t.numpy() for t in cpu]
    with oracle.float64():
        w64 = (oracle.sscan_fwd(*args[:7], True),) + tuple(oracle.sscan_bwd(*args[:7], args[7], True))
    w32 = (oracle.sscan_fwd(*args[:7], True),) + tuple(oracle.sscan_bwd(*args[:7], args[7], True))
    u, delta, A, Bm, Cm, D, bias, dout = [t.to(DEV) for t in cpu]
    names = ("out", "du", "ddelta", "dA", "dB", "dC", "dD", "dbias")
    for tune in ((-1, -1), (1, 1)):
        ss.tune(*tune)
        out, x = ss.fwd(u, delta, A, Bm, Cm, D, bias, True, 1)
        got = (out,) + tuple(ss.bwd(u, delta, A, Bm, Cm, D, bias, dout, x, True, 1))
        for name, g, r32, r64 in zip(names, got, w32, w64):
            ref = np.asarray(r64, np.float64)
            e_hip = np.linalg.norm(g.double().cpu().numpy().reshape(ref.shape) - ref)
            e_cpu = np.linalg.norm(np.asarray(r32, np.float64) - ref)
            if name == "out":
                assert e_hip <= 1.5 * e_cpu + 2e-7 * np.linalg.norm(ref), (name, tune, e_hip, e_cpu, np.linalg.norm(ref))
            else:
                assert e_hip <= 3e-6 * np.linalg.norm(ref), (name, tune, e_hip / np.linalg.norm(ref), e_cpu / np.linalg.norm(ref))
    ss.tune(-1, -1)


def test_scan_general_n_large_decay_arguments():
    """csrc/sscan_n.hip adds the exponent of 2^(delta A log2 e) to the bits directly while |delta A log2 e| <= 125 for the whole
    workgroup and takes a clamped ldexp form otherwise (a wave-uniform branch per batch of rows): inputs that force the second
    form (delta up to 20, A down to -10: arguments down to -290, decays that underflow to 0) and a mix of both in one call."""
    from vm_asr_amd import selective_scan as ss
    for N, scale in ((8, 40.0), (32, 40.0), (6, 8.0)):
        shape = (2, 16, 4, N, 700)
        u, delta, A, Bm, Cm, D, bias, dout = _scan_inputs(*shape, seed=21)
        delta = delta * scale
        delta[:, ::2] *= 0.02                       # every other row stays in the direct-exponent range
        A = A * 20.0
        cpu = (u, delta, A, Bm, Cm, D, bias, dout)
        want = oracle.sscan_fwd(*[t.numpy() for t in cpu[:7]], True)
        wants = oracle.sscan_bwd(*[t.numpy() for t in cpu[:7]], dout.numpy(), True)
        dev = [t.to(DEV) for t in cpu]
        for tune in ((-1, -1), (1, 1)):
            ss.tune(*tune)
            out, x = ss.fwd(*dev[:7], True, 1)
            assert torch.isfinite(out).all()
            _close(out, want, 1e-4, _scaled(want), f"out N={N} tune={tune}")
            got = ss.bwd(*dev[:7], dev[7], x, True, 1)
            for name, g, w in zip(("du", "ddelta", "dA", "dB", "dC", "dD", "dbias"), got, wants):
                assert torch.isfinite(g).all(), name
                _close(g, w, 1e-4, _scaled(w), f"{name} N={N} tune={tune}")
    ss.tune(-1, -1)


def test_scan_long_sequence_dstate32_stress():
    """BASELINE.json configs[4]: d_state 32 with the n_fft 2048 geometry — the 1024x512 output block, L = 524 288,
    8 rows, 32 states per row (the general-N path, 2 049 saved chunks).  Forward and every gradient vs the
    oracle, walk and split plans agreeing with each other."""
    from vm_asr_amd import selective_scan as ss
    shape = (1, 8, 4, 32, 524288)
    cpu = _scan_inputs(*shape, seed=5)
    u, delta, A, Bm, Cm, D, bias, dout = [t.to(DEV) for t in cpu]
    want = oracle.sscan_fwd(*[t.numpy() for t in cpu[:7]], True)
    wants = oracle.sscan_bwd(*[t.numpy() for t in cpu[:7]], cpu[7].numpy(), True)
    outs = []
    for tune in ((-1, -1), (1, 0), (1, 1)):
        ss.tune(*tune)
        out, x = ss.fwd(u, delta, A, Bm, Cm, D, bias, True, 1)
        assert x.shape == (1, 8, 2048, 64)
        _close(out, want, 1e-4, _scaled(want), f"out tune={tune}")
        got = ss.bwd(u, delta, A, Bm, Cm, D, bias, dout, x, True, 1)
        for name, g, w in zip(("du", "ddelta", "dA", "dB", "dC", "dD", "dbias"), got, wants):
            _close(g, w, 2e-4, 2 * _scaled(w), f"{name} tune={tune}")       # dA/dbias sum 524 288 terms per entry
        outs.append(out)
    ss.tune(-1, -1)
    assert torch.allclose(outs[1], outs[2], rtol=1e-4, atol=1e-4 * outs[1].abs().max().item())


def test_scan_strided_inputs():
    """Non-contiguous batch/dim strides and unaligned bases are part of the operator contract
    (cus/selective_scan.cpp:80-95)."""
    from vm_asr_amd import selective_scan as ss
    Bn, KD, G, N, L = 2, 8, 4, 1, 515
    cpu = _scan_inputs(Bn, KD, G, N, L, seed=3)
    u, delta, A, Bm, Cm, D, bias, dout = [t.to(DEV) for t in cpu]
    big_u = torch.zeros(Bn, KD * 2, L + 5, device=DEV)
    big_u[:, ::2, 3:3 + L] = u
    u_s = big_u[:, ::2, 3:3 + L]
    assert not u_s.is_contiguous() and u_s.stride(-1) == 1
    want = oracle.sscan_fwd(*[t.numpy() for t in cpu[:7]], True)
    out, x = ss.fwd(u_s, delta, A, Bm, Cm, D, bias, True, 1)
    _close(out, want, 1e-4, _scaled(want), "strided out")
    du = ss.bwd(u_s, delta, A, Bm, Cm, D, bias, dout, x, True, 1)[0]
    wdu = oracle.sscan_bwd(*[t.numpy() for t in cpu[:7]], cpu[7].numpy(), True)[0]
    _close(du, wdu, 1e-4, _scaled(wdu), "strided du")


def test_scan_autograd_function_and_full_size_properties():
    """BASELINE full-size call shapes (B=4): modes agree with each other, linearity in u of
    (out - D*u) and gradient consistency (<dout, J v> == <J^T dout, v>)."""
    from vm_asr_amd import selective_scan as ss
    for KD, L in ((8, 262144), (128, 16384), (1024, 256)):
        Bn, G, N = 4, 4, 1
        u, delta, A, Bm, Cm, D, bias, dout = [t.to(DEV) for t in _scan_inputs(Bn, KD, G, N, L, seed=1)]
        outs = []
        for tune in ((1, 0), (1, 1), (2, 1), (1, 2), (2, 2), (-1, -1)):
            ss.tune(*tune)
            outs.append(ss.fwd(u, delta, A, Bm, Cm, D, bias, True, 1)[0])
        sc = 1e-4 * max(1.0, outs[0].abs().max().item())
        for o in outs[1:]:
            assert torch.allclose(o, outs[0], rtol=1e-4, atol=sc)
        ss.tune(-1, -1)
        # linearity in u:  f(2u) - D*2u == 2 (f(u) - D*u)
        o2 = ss.fwd(2 * u, delta, A, Bm, Cm, D, bias, True, 1)[0]
        assert torch.allclose(o2, 2 * outs[-1], rtol=1e-4, atol=2 * sc)
        # adjoint identity on the u-path
        uu = u.clone().requires_grad_()
        out = ss.SelectiveScanCore.apply(uu, delta, A, Bm, Cm, D, bias, True)
        out.backward(dout)
        v = torch.randn_like(u)
        jv = ss.fwd(v, delta, A, Bm, Cm, D, bias, True, 1)[0]  # linear in u
        lhs = (dout.double() * jv.double()).sum()
        rhs = (uu.grad.double() * v.double()).sum()
        # both sides are sums of ~1e7 signed terms: allow 1e-4 relative error per term (random signs)
        tol = 1e-4 * (dout.double() * jv.double()).norm().item() + 1e-4 * abs(lhs.item())
        assert abs(lhs - rhs) <= tol, (KD, L, lhs.item(), rhs.item(), tol)


def test_scan_bf16_io():
    from vm_asr_amd import selective_scan as ss
    Bn, KD, G, N, L = 2, 32, 4, 1, 2048
    cpu = _scan_inputs(Bn, KD, G, N, L, seed=2)
    q = [t.to(torch.bfloat16) if i in (0, 1, 3, 4, 7) else t for i, t in enumerate(cpu)]
    want = oracle.sscan_fwd(*[t.float().numpy() for t in q[:7]], True)
    dev = [t.to(DEV) for t in q]
    out, x = ss.fwd(*dev[:7], True, 1)
    assert out.dtype == torch.bfloat16
    _close(out, want, 3e-2, 5e-2, "bf16 out")
    du, dd, dA, dB, dC, dD, db = ss.bwd(*dev[:7], dev[7], x, True, 1)
    w = oracle.sscan_bwd(*[t.float().numpy() for t in q[:7]], q[7].float().numpy(), True)
    assert du.dtype == torch.bfloat16 and dB.dtype == torch.bfloat16 and dA.dtype == torch.float32
    _close(du, w[0], 6e-2, 1e-1, "bf16 du")
    _close(dA, w[2], 1e-2, _scaled(w[2], 1e-2), "bf16 dA")


# ------------------------------------------------------------------------------------------
# cross scan / merge: bit-exact data movement
# ------------------------------------------------------------------------------------------
def test_cross_scan_merge_golden_and_oracle():
    from vm_asr_amd import csm
    z = np.load(os.path.join(GOLDEN, "csm.npz"))
    for t in "abc":
        x = _t(z[f"{t}_x"]).requires_grad_()
        xs = csm.CrossScan.apply(x)
        assert np.array_equal(xs.detach().cpu().numpy(), z[f"{t}_xs"])
        xs.backward(_t(z[f"{t}_gxs"]))
        _close(x.grad, z[f"{t}_dx"], 1e-6, 1e-6, "scan bwd")
        ys = _t(z[f"{t}_ys"]).requires_grad_()
        y = csm.CrossMerge.apply(ys)
        _close(y, z[f"{t}_y"], 1e-6, 1e-6, "merge")
        y.backward(_t(z[f"{t}_gy"]))
        assert np.array_equal(ys.grad.cpu().numpy(), z[f"{t}_dys"])
    # the six SS2D shapes of the 48 kHz config (B=1) + ragged, vs oracle, bit-exact
    for C, H, W in ((2, 512, 512), (16, 256, 256), (32, 128, 128), (64, 64, 64), (128, 32, 32), (256, 16, 16),
                    (3, 65, 130), (1, 1, 7)):
        g = torch.Generator().manual_seed(C)
        x = torch.randn(1, C, H, W, generator=g)
        assert np.array_equal(csm.cross_scan(x.to(DEV)).cpu().numpy(), oracle.cross_scan(x.numpy()))
        ys = torch.randn(1, 4, C, H, W, generator=g)
        got = csm.cross_merge(ys.to(DEV), H, W).cpu().numpy()
        assert np.array_equal(got, oracle.cross_merge(ys.numpy())), (C, H, W)
    # bf16 payload moves as raw bits
    xb = torch.randn(2, 5, 33, 17).to(torch.bfloat16)
    want = oracle.cross_scan(xb.float().numpy())
    assert np.array_equal(csm.cross_scan(xb.to(DEV)).float().cpu().numpy(), want)


def test_cross_scan_merge_roundtrip_full_size():
    """merge(scan(x)) == 4x at the largest call shape (B=4, 512x512), a size-independent check."""
    from vm_asr_amd import csm
    x = torch.randn(4, 2, 512, 512, device=DEV)
    y = csm.cross_merge(csm.cross_scan(x), 512, 512).view_as(x)
    assert torch.equal(y, 4 * x)


# ------------------------------------------------------------------------------------------
# depthwise conv + SiLU
# ------------------------------------------------------------------------------------------
def test_dwconv_silu_golden_and_oracle():
    from vm_asr_amd import dwconv
    z = np.load(os.path.join(GOLDEN, "dwconv.npz"))
    for t in "abc":
        x = _t(z[f"{t}_x"]).requires_grad_()
        w = _t(z[f"{t}_w"]).requires_grad_()
        b = _t(z[f"{t}_b"]).requires_grad_()
        y = dwconv.dwconv3x3_silu(x, w, b)
        _close(y, z[f"{t}_y"], 1e-4, 1e-5, "y")
        y.backward(_t(z[f"{t}_g"]))
        _close(x.grad, z[f"{t}_dx"], 1e-4, 1e-5, "dx")
        _close(w.grad, z[f"{t}_dw"], 1e-4, _scaled(z[f"{t}_dw"]), "dw")
        _close(b.grad, z[f"{t}_db"], 1e-4, _scaled(z[f"{t}_db"]), "db")
    for C, H, W in ((2, 512, 512), (32, 128, 128), (256, 16, 16), (3, 7, 130)):
        g = torch.Generator().manual_seed(C)
        x = torch.randn(2, C, H, W, generator=g)
        w = 0.3 * torch.randn(C, 1, 3, 3, generator=g)
        b = 0.1 * torch.randn(C, generator=g)
        gy = torch.randn(2, C, H, W, generator=g)
        xd, wd, bd = x.to(DEV).requires_grad_(), w.to(DEV).requires_grad_(), b.to(DEV).requires_grad_()
        y = dwconv.dwconv3x3_silu(xd, wd, bd)
        _close(y, oracle.dwconv_silu_fwd(x.numpy(), w.numpy(), b.numpy()), 1e-4, 1e-5, f"y {C}")
        y.backward(gy.to(DEV))
        dx, dw, db = oracle.dwconv_silu_bwd(x.numpy(), w.numpy(), b.numpy(), gy.numpy())
        _close(xd.grad, dx, 1e-4, 1e-5, f"dx {C}")
        _close(wd.grad, dw.reshape(C, 1, 3, 3), 1e-4, _scaled(dw), f"dw {C}")
        _close(bd.grad, db, 1e-4, _scaled(db), f"db {C}")


# ------------------------------------------------------------------------------------------
# STFT / iSTFT
# ------------------------------------------------------------------------------------------
def _phase_close(a, b, mag, tol, what):
    """See tests/test_oracle.py::_phase_close: frame 0 on the circle, other frames on the same branch."""
    a = a.detach().cpu().numpy().astype(np.float64)
    b = b.astype(np.float64)
    ok = mag > -12
    d = np.abs(a - b)
    d[..., 0] = np.abs(np.angle(np.exp(1j * (a[..., 0] - b[..., 0]))))
    wrapped = ok & (d > tol) & (np.abs(np.abs(a) - np.pi) < 1e-3) & (np.abs(np.abs(b) - np.pi) < 1e-3)
    assert wrapped.sum() <= max(2, ok.sum() // 100000), (what, int(wrapped.sum()))
    bad = ok & (d > tol) & ~wrapped
    assert not bad.any(), (what, int(bad.sum()), float(d[bad].max()))


def test_stft_istft_golden():
    from vm_asr_amd import stft
    z = np.load(os.path.join(GOLDEN, "stft.npz"))
    for t in "stwr":
        n_fft, hop, win = (int(v) for v in z[f"{t}_cfg"])
        mag, ph = stft.wav2spectro(_t(z[f"{t}_wav"]), n_fft, hop, win, "log2")
        assert tuple(mag.shape) == z[f"{t}_mag"].shape
        _close(mag, z[f"{t}_mag"], 1e-4, 2e-4, f"{t} mag")
        _phase_close(ph, z[f"{t}_phase"], z[f"{t}_mag"], 2e-3, f"{t} phase")
        m2, p2 = _t(z[f"{t}_imag"]).requires_grad_(), _t(z[f"{t}_iphase"]).requires_grad_()
        rec = stft.spectro2wav(m2, p2, n_fft, hop, win, "log2")
        _close(rec, z[f"{t}_rec"], 1e-4, 1e-5, f"{t} rec")
        rec.backward(_t(z[f"{t}_grec"]))
        _close(m2.grad, z[f"{t}_dmag"], 1e-4, _scaled(z[f"{t}_dmag"], 1e-5), f"{t} dmag")
        _close(p2.grad, z[f"{t}_dphase"], 1e-4, _scaled(z[f"{t}_dphase"], 1e-5), f"{t} dphase")
    wav = _t(z["big_wav"])
    mag, ph = stft.wav2spectro(wav, 1024, 240, 1024, "log2")
    assert tuple(mag.shape) == tuple(z["big_shape"])
    fr = z["big_frames"]
    _close(mag[..., fr], z["big_mag_frames"], 1e-4, 2e-4, "big frames")
    assert abs(mag.double().sum().item() - float(z["big_mag_sum"])) < 1e-4 * float(z["big_mag_abs"])
    rec = stft.spectro2wav(mag, ph, 1024, 240, 1024, "log2")
    assert (rec - wav).abs().max().item() < 1e-4  # STFT -> iSTFT round trip at the full clip size


def test_stft_vs_oracle_batch4_full_clip():
    from vm_asr_amd import stft
    g = torch.Generator().manual_seed(11)
    wav = 0.1 * torch.randn(4, 1, 122640, generator=g)
    mag, ph = stft.wav2spectro(wav.to(DEV), 1024, 240, 1024, "log2")
    om, op = oracle.stft(wav.numpy(), 1024, 240, 1024)
    # typical |S| is 0.06 here; an fp32 FFT is good to ~1e-7 of that, so the log2 of bins below
    # 2^-10 is dominated by rounding (in torch.stft as well): compare those in the linear domain
    _close(torch.exp2(mag), np.exp2(om.astype(np.float64)), 1e-4, 1e-6, "|S|")
    big = om > -10
    d = np.abs(mag.cpu().numpy().astype(np.float64) - om)[big]
    assert d.max() < 5e-4, f"log2|S| on bins above 2^-10: {d.max():.3e}"
    _phase_close(ph, op, om, 2e-3, "phase")
    re, im = stft.stft_complex(wav.to(DEV)[:, 0], 2048, 512, 2048)  # LSD's STFT (model/metric.py:5-12)
    ore, oim = oracle.stft(wav.numpy()[:, 0], 2048, 512, 2048, normalized=False, logmag=False)
    _close(re, ore, 1e-4, 1e-4, "re")
    _close(im, oim, 1e-4, 1e-4, "im")


# ------------------------------------------------------------------------------------------
# LayerNorm (channel-last, small C): vs torch fp32 F.layer_norm (a floating-point kernel: the
# oracle here is plain PyTorch fp32, as for any fp kernel)
# ------------------------------------------------------------------------------------------
@pytest.mark.parametrize("C", [1, 2, 4, 8, 16, 24, 32, 64, 96, 128, 256, 512, 1024, 3, 130])
def test_layer_norm_vs_torch(C):
    from vm_asr_amd.layernorm import layer_norm
    g = torch.Generator().manual_seed(C)
    for rows_shape in ((2, 16, 16), (3, 5, 7), (1, 512, 512) if C <= 8 else (1, 33, 9)):
        x = torch.randn(*rows_shape, C, generator=g) * 2 + 0.5
        w = torch.randn(C, generator=g)
        b = torch.randn(C, generator=g)
        gy = torch.randn(*rows_shape, C, generator=g)
        xr, wr, br = x.clone().requires_grad_(), w.clone().requires_grad_(), b.clone().requires_grad_()
        yr = torch.nn.functional.layer_norm(xr.double(), (C,), wr.double(), br.double(), 1e-5)
        yr.backward(gy.double())
        xd, wd, bd = (t.to(DEV).requires_grad_() for t in (x, w, b))
        y = layer_norm(xd, wd, bd, 1e-5)
        y.backward(gy.to(DEV))
        _close(y, yr.detach().numpy(), 1e-4, 1e-4, f"y C={C}")
        _close(xd.grad, xr.grad.numpy(), 1e-4, _scaled(xr.grad.numpy()), f"dx C={C}")
        _close(wd.grad, wr.grad.numpy(), 1e-4, _scaled(wr.grad.numpy()), f"dw C={C}")
        _close(bd.grad, br.grad.numpy(), 1e-4, _scaled(br.grad.numpy()), f"db C={C}")
    # bf16 input under autocast: fp32 output like torch's autocast policy
    x = torch.randn(4, 64, 64, C, generator=g).to(torch.bfloat16)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        y = layer_norm(x.to(DEV), None, None, 1e-5)
        yb = layer_norm(x.float().to(DEV), None, None, 1e-5, feeds_gemm=True)
    assert y.dtype == torch.float32 and yb.dtype == torch.bfloat16
    want = torch.nn.functional.layer_norm(x.float(), (C,), None, None, 1e-5)
    _close(y, want.numpy(), 1e-4, 1e-4, f"bf16 in C={C}")
    assert torch.equal(yb.cpu(), want.to(torch.bfloat16)) or (yb.float().cpu() - want).abs().max() < 4e-2  # same rounding


# ------------------------------------------------------------------------------------------
# tiny-width Linear over many rows (the d_model = 1 block): vs torch fp64
# ------------------------------------------------------------------------------------------
@pytest.mark.parametrize("io", [(1, 4), (2, 1), (4, 1), (1, 1), (4, 8), (8, 4), (2, 8), (8, 2)])
def test_small_linear_vs_torch(io):
    from vm_asr_amd.linear import linear
    IN, OUT = io
    g = torch.Generator().manual_seed(IN * 10 + OUT)
    for shape in ((2, 128, 128), (1, 70001)):  # ragged row count too
        x = torch.randn(*shape, IN, generator=g)
        w = torch.randn(OUT, IN, generator=g)
        b = torch.randn(OUT, generator=g)
        gy = torch.randn(*shape, OUT, generator=g)
        xr, wr, br = (t.double().requires_grad_() for t in (x, w, b))
        yr = torch.nn.functional.linear(xr, wr, br)
        yr.backward(gy.double())
        xd, wd, bd = (t.to(DEV).requires_grad_() for t in (x, w, b))
        y = linear(xd, wd, bd)
        y.backward(gy.to(DEV))
        _close(y, yr.detach().numpy(), 1e-4, 1e-4, f"y {io}")
        _close(xd.grad, xr.grad.numpy(), 1e-4, 1e-4, f"dx {io}")
        _close(wd.grad, wr.grad.numpy(), 1e-4, _scaled(wr.grad.numpy()), f"dw {io}")
        _close(bd.grad, br.grad.numpy(), 1e-4, _scaled(br.grad.numpy()), f"db {io}")
    # autocast: fp32 activations in, bf16 out (what F.linear does under autocast)
    x = torch.randn(4, 64, 64, IN, generator=g)
    w = torch.randn(OUT, IN, generator=g)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        y = linear(x.to(DEV), w.to(DEV), None)
    assert y.dtype == torch.bfloat16
    _close(y, torch.nn.functional.linear(x, w).numpy(), 2e-2, 2e-2, f"bf16 out {io}")


# ------------------------------------------------------------------------------------------
# x_proj / dt_proj map vs the reference einsums (model/vmamba.py:1473-1477) in fp64
# ------------------------------------------------------------------------------------------
@pytest.mark.parametrize("cfg", [(2, 2, 1, 1, 4096), (2, 32, 1, 1, 1024), (1, 256, 1, 8, 256), (2, 16, 4, 2, 300),
                                 (1, 128, 1, 4, 1023),
                                 # row-parallel kernels of the deep stages (d_inner >= 64, L % 4 == 0): full / ragged tiles, odd D
                                 (2, 64, 1, 2, 4096), (3, 128, 1, 4, 1000), (1, 100, 2, 3, 200), (2, 64, 1, 2, 16384),
                                 # matrix-core kernels (d_state 1, D % 32 == 0, L % 32 == 0, even dt_rank): the deep-stage call shapes
                                 (4, 128, 1, 4, 1024), (8, 256, 1, 8, 256), (1, 512, 1, 8, 96),
                                 # general d_state (csrc/xproj_n.hip, fp32 MFMA products): configs[4]'s call shapes (d_state 32: C = 66 .. 80
                                 # rows), ragged position tiles, d_inner not a multiple of 32, odd dt_rank, C above 96
                                 (1, 64, 32, 2, 2048), (2, 2, 32, 1, 4096), (1, 32, 32, 1, 1000), (1, 256, 32, 8, 512), (1, 512, 32, 16, 96),
                                 (2, 6, 5, 1, 77), (1, 34, 8, 3, 300), (1, 16, 48, 1, 260)], ids=str)
def test_xproj_vs_einsum(cfg):
    from vm_asr_amd.xproj import x_proj_dt
    Bn, D, N, R, L = cfg
    K, C = 4, R + 2 * N
    g = torch.Generator().manual_seed(D + L)
    xs = torch.randn(Bn, K, D, L, generator=g)
    Wx = torch.randn(K, C, D, generator=g) / D ** 0.5
    Wdt = torch.randn(K, D, R, generator=g)
    gd, gB, gC = torch.randn(Bn, K * D, L, generator=g), torch.randn(Bn, K, N, L, generator=g), torch.randn(Bn, K, N, L, generator=g)
    xr, wxr, wdr = (t.double().requires_grad_() for t in (xs, Wx, Wdt))
    x_dbl = torch.einsum("b k d l, k c d -> b k c l", xr, wxr)
    dtsr, Bsr, Csr = torch.split(x_dbl, [R, N, N], dim=2)
    dtsr = torch.einsum("b k r l, k d r -> b k d l", dtsr, wdr).reshape(Bn, -1, L)
    ((dtsr * gd.double()).sum() + (Bsr * gB.double()).sum() + (Csr * gC.double()).sum()).backward()
    xd, wxd, wdd = (t.to(DEV).requires_grad_() for t in (xs, Wx, Wdt))
    dts, Bs, Cs = x_proj_dt(xd, wxd, wdd, N)
    ((dts * gd.to(DEV)).sum() + (Bs * gB.to(DEV)).sum() + (Cs * gC.to(DEV)).sum()).backward()
    _close(dts, dtsr.detach().numpy(), 1e-4, _scaled(dtsr.detach().numpy()), "dts")
    _close(Bs, Bsr.detach().numpy(), 1e-4, 1e-4, "Bs")
    _close(Cs, Csr.detach().numpy(), 1e-4, 1e-4, "Cs")
    _close(xd.grad, xr.grad.numpy(), 1e-4, _scaled(xr.grad.numpy()), "dxs")
    _close(wxd.grad, wxr.grad.numpy(), 1e-4, _scaled(wxr.grad.numpy()), "dWx")
    _close(wdd.grad, wdr.grad.numpy(), 1e-4, _scaled(wdr.grad.numpy()), "dWdt")


def test_cross_scan_merge_converting():
    from vm_asr_amd import csm
    x = torch.randn(2, 5, 33, 17).to(torch.bfloat16)
    xs = csm.cross_scan(x.to(DEV), torch.float32)
    assert xs.dtype == torch.float32 and np.array_equal(xs.cpu().numpy(), oracle.cross_scan(x.float().numpy()))
    ys = torch.randn(2, 4, 5, 33, 17)
    y = csm.cross_merge(ys.to(DEV), 33, 17, torch.bfloat16)
    want = torch.from_numpy(oracle.cross_merge(ys.numpy())).to(torch.bfloat16)
    assert y.dtype == torch.bfloat16 and torch.equal(y.cpu(), want)


def test_spectral_power_iteration_vs_torch():
    from vm_asr_amd.discriminator import _SpectralNorm
    for shape in ((32, 1, 5, 1), (128, 32, 5, 1), (1024, 512, 5, 1), (1, 1024, 3, 1), (7, 3, 5, 1)):
        torch.manual_seed(sum(shape))
        w = torch.randn(*shape)
        sn_cpu = _SpectralNorm(w)                      # CPU: torch ops
        sn_gpu = _SpectralNorm(w).to(DEV)
        sn_gpu._u.copy_(sn_cpu._u); sn_gpu._v.copy_(sn_cpu._v)
        sn_cpu.train(); sn_gpu.train()
        sn_cpu.n_power_iterations = sn_gpu.n_power_iterations = 3
        out_c = sn_cpu(w)
        out_g = sn_gpu(w.to(DEV))                      # GPU: vmasr_spectral_power_iter
        _close(sn_gpu._u, sn_cpu._u.numpy(), 1e-4, 1e-5, f"u {shape}")
        _close(sn_gpu._v, sn_cpu._v.numpy(), 1e-4, 1e-5, f"v {shape}")
        _close(out_g, out_c.detach().numpy(), 1e-4, 1e-5, f"w/sigma {shape}")


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("H,C,k,stride,pad", [(201, 1, 5, 3, 2), (67, 8, 5, 3, 2), (23, 32, 5, 3, 2), (8, 12, 5, 1, 2),
                                               (8, 5, 3, 1, 1), (1, 4, 3, 1, 1), (40881, 1, 5, 3, 2)])
def test_im2col_col2im_kx1(dtype, H, C, k, stride, pad):
    """HIP im2col / col2im of the period discriminator's (k,1) convolutions vs unfold and its autograd adjoint
    (exact: pure data movement; the adjoint sums at most ceil(k/stride) terms in fp32)."""
    from vm_asr_amd.discriminator import _Im2ColFn
    torch.manual_seed(H + C)
    B, P = 2, 3
    x = torch.randn(B, P, H, C, device="cuda").to(dtype).requires_grad_()
    cols = _Im2ColFn.apply(x, k, stride, pad)
    xr = x.detach().clone().requires_grad_()
    ref = torch.nn.functional.pad(xr, (0, 0, pad, pad)).unfold(2, k, stride).permute(0, 1, 2, 4, 3)   # (B,P,H1,k,C)
    ref = ref.reshape(B, P, -1, k * C)
    assert cols.shape == ref.shape and torch.equal(cols, ref)
    g = torch.randn_like(ref)
    cols.backward(g)
    ref.float().backward(g.float())
    tol = 0 if dtype == torch.float32 and stride >= k else (1e-6 if dtype == torch.float32 else 2e-2)
    assert (x.grad.float() - xr.grad.float()).abs().max() <= tol * max(1.0, xr.grad.float().abs().max().item())


@pytest.mark.gpu
def test_spectral_power_iteration_batched_matches_single():
    """All 30 weights of a MultiPeriodDiscriminator in one launch per phase == one matrix at a time, and both
    == torch's power iteration (CPU)."""
    import copy
    from vm_asr_amd.discriminator import MultiPeriodDiscriminator
    torch.manual_seed(3)
    ref = MultiPeriodDiscriminator(hidden=4)           # CPU: torch ops
    a, b = copy.deepcopy(ref).to(DEV), copy.deepcopy(ref).to(DEV)
    for m in a.spectral_norms():
        m.train()
    assert b.power_iterate_all(3)                      # batched HIP
    assert b._sn_batch.n == 30
    for mod_r, mod_a, mod_b in zip(ref.modules(), a.modules(), b.modules()):
        if isinstance(mod_r, torch.nn.Conv2d):
            w = mod_r.parametrizations.weight.original.detach().flatten(1)
            sr, sa, sb = (m.parametrizations.weight[0] for m in (mod_r, mod_a, mod_b))
            sr._power_method(w, 3)                     # torch (CPU)
            sa._power_method(w.to(DEV), 3)             # single-matrix HIP
            for got, what in ((sa, "single"), (sb, "batched")):
                _close(got._u, sr._u.numpy(), 1e-4, 1e-5, f"u {what} {tuple(w.shape)}")
                _close(got._v, sr._v.numpy(), 1e-4, 1e-5, f"v {what} {tuple(w.shape)}")
    assert b.power_iterate_all(1) and float(b._sn_batch.ws.abs().max()) >= 0   # table reused, scratch sane


def test_layer_norm_deferred_reduction_matches_immediate():
    """layernorm.DEFER_REDUCE: the dgamma / dbeta of every LayerNorm call of a backward pass come from ONE reduce launch at the
    end of the pass (autograd queue_callback); same values as the immediate reduction; a weight used twice in one graph, or
    one that already holds a .grad, is reduced at once."""
    from vm_asr_amd import layernorm as L
    torch.manual_seed(0)
    lns = [L.LayerNorm(c).to(DEV) for c in (8, 16, 64, 128)]
    xs = [torch.randn(4 * 4096 // max(1, c // 8), c, device=DEV, requires_grad=True) for c in (8, 16, 64, 128)]

    def run():
        for m in lns:
            m.zero_grad(set_to_none=True)
        loss = sum((m(x) * (i + 1)).sin().sum() for i, (m, x) in enumerate(zip(lns, xs))) + (lns[1](xs[1] * 2)).sum()   # lns[1] twice
        loss.backward()
        return [p.grad.clone() for m in lns for p in m.parameters()]
    ref = run()
    L.DEFER_REDUCE = True
    deferred = []
    orig = L.defer_reduction
    L.defer_reduction = lambda *a: (deferred.append(a[4]) or True) and orig(*a)
    try:
        L.reset_uses()
        got = run()
        assert not L._pending and sorted(deferred) == [8, 64, 128]          # all but the LayerNorm used twice
        got2 = run()           # (use counts were cleared by the flush)
    finally:
        L.DEFER_REDUCE = False
        L.defer_reduction = orig
        L.reset_uses()
    for a, b, c in zip(ref, got, got2):
        assert torch.allclose(a, b, rtol=1e-5, atol=1e-5 * a.abs().max().item()) and torch.equal(b, c)


def test_layer_norm_deferred_queue_survives_an_aborted_backward():
    """autograd runs the end-of-pass callback only when a backward COMPLETES.  A backward that raises (OOM, kernel
    error, failed graph capture) must not leave the queue non-empty for good: the next step (reset_uses(), as the trainer
    calls it at the start of every step) has to produce correct dgamma / dbeta again."""
    from vm_asr_amd import layernorm as L
    torch.manual_seed(1)
    ln = L.LayerNorm(16).to(DEV)
    x = torch.randn(4096, 16, device=DEV, requires_grad=True)

    class Boom(torch.autograd.Function):
        @staticmethod
        def forward(ctx, t):
            return t.clone()

        @staticmethod
        def backward(ctx, g):
            raise RuntimeError("boom")

    def grads():
        ln.zero_grad(set_to_none=True)
        ln(x).sin().sum().backward()
        return [p.grad.clone() for p in ln.parameters()]
    ref = grads()
    try:
        with pytest.raises(RuntimeError, match="boom"):
            with L.deferred(True):
                ln.zero_grad(set_to_none=True)
                ln(Boom.apply(x)).sin().sum().backward()      # aborts AFTER LayerNorm's backward has queued its reduction
        assert not L._pending and not L.DEFER_REDUCE              # scope left: nothing queued, deferral off again
        L.DEFER_REDUCE = True                                      # the trainer's own protocol: flag + reset_uses() per step
        L._pending.append(("stale",))                              # what an aborted pass used to leave behind
        L.reset_uses()
        got = grads()
        assert not L._pending
    finally:
        L.DEFER_REDUCE = False
        L.reset_uses()
    for a, b in zip(ref, got):
        assert torch.allclose(a, b, rtol=1e-5, atol=1e-5 * a.abs().max().item())


def test_linear_f64acc_is_correctly_rounded():
    """csrc/linear.hip linear_f64acc: fp32 Linear of the parity path with float64 accumulation = the float64 result rounded once
    (the family adds nothing to the forward's distance from the exact answer); backward through autograd against float64."""
    from vm_asr_amd import linear as L
    g = torch.Generator().manual_seed(2)
    for rows, in_f, out_f in ((1000, 512, 300), (4096, 256, 128), (37, 16, 5), (64, 1024, 64)):
        x = torch.randn(rows, in_f, generator=g).to(DEV).requires_grad_(True)
        w = (torch.randn(out_f, in_f, generator=g) / in_f ** 0.5).to(DEV).requires_grad_(True)
        b = torch.randn(out_f, generator=g).to(DEV).requires_grad_(True)
        assert L._use_f64acc(x, w, b)
        y = L.linear(x, w, b)
        want64 = x.detach().double() @ w.detach().double().t() + b.detach().double()
        want = want64.float()
        ulp = torch.finfo(torch.float32).eps * want.abs().clamp(min=1e-30)
        assert ((y.detach() - want).abs() <= ulp).all()
        assert (y.detach() != want).float().mean().item() < 1e-3          # (double rounding can differ on near-ties only)
        # torch's own fp32 GEMM for scale: noisier
        e_hip = (y.detach().double() - want64).abs().max().item()
        e_lt = (torch.nn.functional.linear(x.detach(), w.detach(), b.detach()).double() - want64).abs().max().item()
        assert e_hip <= e_lt
        gy = torch.randn(rows, out_f, generator=g).to(DEV)
        y.backward(gy)
        x64, w64, b64 = (t.detach().double().requires_grad_(True) for t in (x, w, b))
        (x64 @ w64.t() + b64).backward(gy.double())
        for got, ref in ((x.grad, x64.grad), (w.grad, w64.grad), (b.grad, b64.grad)):
            assert (got.double() - ref).abs().max().item() <= 1e-5 * ref.abs().max().item()


@pytest.mark.gpu
@pytest.mark.parametrize("rows,K,N", [(4096 + 37, 9, 8), (65536, 72, 16), (8192, 16, 64), (5000, 96, 96), (4096, 64, 3), (300000, 16, 8)])
@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
def test_skinny_linear_matches_float64(rows, K, N, dt):
    """csrc/skinny.hip: y = x W^T + b and the input gradient g W of the same layer (strides swapped) for many rows and few features,
    against float64 on the values the kernel is given (bf16 inputs are exact in float64).  Tolerance: fp32 accumulation of <= 96 terms
    (1e-6 of scale) + the output's own rounding (bf16: 2^-8)."""
    from vm_asr_amd import linear as L
    dev = torch.device("cuda:0")
    g = torch.Generator(device="cpu").manual_seed(rows % 1000 + K)
    x = torch.randn(rows, K, generator=g).to(dev).to(dt)
    w = (torch.randn(N, K, generator=g) / K ** 0.5).to(dev)
    b = torch.randn(N, generator=g).to(dev)
    assert bool(L._lib.lib().vmasr_skinny_linear_supported(rows, K, N))
    y = L._skinny(x, w, b, dt)
    want = x.double() @ w.double().t() + b.double()
    tol = (2 ** -8 if dt == torch.bfloat16 else 2e-6) * want.abs().max().item()
    assert (y.double() - want).abs().max().item() <= tol
    y0 = L._skinny(x, w, None, torch.float32)
    assert (y0.double() - x.double() @ w.double().t()).abs().max().item() <= 2e-6 * want.abs().max().item()
    gy = torch.randn(rows, N, generator=g).to(dev).to(dt)
    dx = L._skinny(gy, w, None, dt, transposed=True)
    wantx = gy.double() @ w.double()
    tolx = (2 ** -8 if dt == torch.bfloat16 else 2e-6) * wantx.abs().max().item()
    assert dx.shape == (rows, K) and (dx.double() - wantx).abs().max().item() <= tolx


@pytest.mark.gpu
def test_linear_takes_the_skinny_kernel_under_autocast_with_the_same_gradients():
    """vm_asr_amd.linear.linear on (B, H, W, C) activations with many rows: same output and gradients as F.linear in float64, at bf16
    tolerance; VMASR_SKINNY=0 gives the GEMM path."""
    import os
    from vm_asr_amd import linear as L
    dev = torch.device("cuda:0")
    g = torch.Generator(device="cpu").manual_seed(4)
    x = torch.randn(4, 256, 256, 9, generator=g).to(dev).requires_grad_(True)
    w = (torch.randn(12, 9, generator=g) / 3).to(dev).requires_grad_(True)
    b = torch.randn(12, generator=g).to(dev).requires_grad_(True)
    gy = torch.randn(4, 256, 256, 12, generator=g).to(dev)
    res = {}
    for flag in ("1", "0"):
        os.environ["VMASR_SKINNY"] = flag
        try:
            for t in (x, w, b):
                t.grad = None
            with torch.autocast("cuda", dtype=torch.bfloat16):
                y = L.linear(x, w, b)
            y.float().backward(gy)
            res[flag] = (y.detach().float(), x.grad.clone(), w.grad.clone(), b.grad.clone())
        finally:
            os.environ.pop("VMASR_SKINNY", None)
    x64, w64, b64 = x.detach().double().requires_grad_(True), w.detach().double().requires_grad_(True), b.detach().double().requires_grad_(True)
    y64 = torch.nn.functional.linear(x64, w64, b64)
    y64.backward(gy.double())
    want = (y64.detach(), x64.grad, w64.grad, b64.grad)
    for flag in ("1", "0"):
        for got, ref in zip(res[flag], want):
            assert (got.double() - ref).abs().max().item() <= 1e-2 * ref.abs().max().item(), flag


@pytest.mark.gpu
@pytest.mark.parametrize("shape,k,s,p", [((2, 1, 64, 48), (3, 3), (2, 2), (1, 1)), ((3, 8, 33, 40), (3, 3), (2, 2), (1, 1)),
                                         ((2, 4, 17, 19), (3, 2), (1, 2), (0, 1)), ((1, 5, 16, 16), (4, 4), (4, 4), (0, 0))])
@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("channel_last", [False, True])
def test_im2col2d_rows_matches_unfold_and_its_adjoint(shape, k, s, p, dt, channel_last):
    """vmasr_im2col2d_rows == F.unfold(x).transpose(1, 2) row for row (bit-exact: a gather), for dense inputs in either memory layout;
    vmasr_col2im2d_rows == F.fold of the transposed gradient (fp32 sums of <= kh kw terms in a fixed order)."""
    import torch.nn.functional as F
    from vm_asr_amd.model import _Im2ColRowsFn, _rows_ok
    dev = torch.device("cuda:0")
    g = torch.Generator(device="cpu").manual_seed(sum(shape))
    x = torch.randn(shape, generator=g).to(dev).to(dt)
    if channel_last:
        x = x.permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2)
    assert _rows_ok(x)
    x.requires_grad_(True)
    cols = _Im2ColRowsFn.apply(x, k, s, p, dt)
    want = F.unfold(x.detach().float(), k, padding=p, stride=s).transpose(1, 2).reshape(cols.shape)
    assert torch.equal(cols.detach().float(), want)
    gy = torch.randn(cols.shape, generator=g).to(dev).to(dt)
    cols.backward(gy)
    H, W = shape[2], shape[3]
    wantx = F.fold(gy.float().view(shape[0], -1, cols.shape[1]).transpose(1, 2), (H, W), k, padding=p, stride=s)
    tol = (2 ** -8 if dt == torch.bfloat16 else 1e-6) * max(1.0, wantx.abs().max().item())
    assert x.grad.stride() == x.stride() and (x.grad.float() - wantx).abs().max().item() <= tol


@pytest.mark.gpu
def test_gemm_conv2d_rows_path_equals_the_unfold_path(monkeypatch):
    """GemmConv2d through the one-pass rows gather == the F.unfold path (same GEMM, same values) under autocast, forward and gradients."""
    from vm_asr_amd.model import GemmConv2d
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    conv = GemmConv2d(8, 16, kernel_size=3, stride=2, padding=1).to(dev)
    x0 = torch.randn(2, 8, 64, 64, device=dev)
    res = {}
    for flag in ("1", "0"):
        monkeypatch.setenv("VMASR_IM2COL2D", flag)
        x = x0.clone().requires_grad_(True)
        conv.zero_grad()
        with torch.autocast("cuda", dtype=torch.bfloat16):
            y = conv(x)
        y.float().square().sum().backward()
        res[flag] = (y.detach().float(), x.grad.clone(), conv.weight.grad.clone(), conv.bias.grad.clone())
    for a, b in zip(res["1"], res["0"]):
        assert (a - b).abs().max().item() <= 2e-2 * b.abs().max().item()
