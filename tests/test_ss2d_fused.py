"""The fused SS2D core (vm_asr_amd/csrc/ss2d.hip: cross-scan + x_proj + dt_proj + 4 selective scans + cross-merge as
one operator, forward and backward) against the oracle's composition of the SAME reference steps
(model/vmamba.py:1472-1497: CrossScan -> einsum x2 -> selective_scan_ref -> CrossMerge; each oracle piece is pinned
to reference goldens in tests/test_oracle.py), in fp32 and — as adjudicator — in float64."""
import numpy as np
import pytest
import torch

SHAPES = [(2, 32, 16, 16), (1, 16, 32, 32), (2, 8, 16, 32), (1, 4, 32, 32), (2, 2, 64, 64), (1, 32, 64, 128), (1, 16, 128, 64)]
# the three fused call shapes of vm_asr_48k at B = 1 (enc0/out0, out1, out2) — the oracle chain does them in seconds
BENCH_SHAPES = [(1, 32, 128, 128), (1, 16, 256, 256), (1, 2, 512, 512)]


def _params(D, seed, dtype=torch.float32):
    g = torch.Generator().manual_seed(seed)
    r = lambda *s: torch.rand(*s, generator=g, dtype=torch.float64)      # noqa: E731
    Wx = ((2 * r(4, 3, D) - 1) / D ** 0.5)
    Wdt = (2 * r(4, D, 1) - 1)
    dt = torch.exp(r(4, D) * (np.log(0.1) - np.log(1e-3)) + np.log(1e-3))
    dtb = dt + torch.log(-torch.expm1(-dt))                                # softplus^-1 (model/vmamba.py:886-905)
    A_logs = 0.3 * (2 * r(4 * D, 1) - 1)
    Ds = 1 + 0.2 * (2 * r(4 * D) - 1)
    return [t.to(dtype) for t in (Wx, Wdt, dtb, A_logs, Ds)]


def _oracle_core(x, Wx, Wdt, dtb, A_logs, Ds):
    """model/vmamba.py:1472-1497 on the oracle's kernels (dtype follows the active oracle build)."""
    from oracle.torch_backend import OracleCrossMerge, OracleCrossScan, OracleSelectiveScan
    B, D, H, W = x.shape
    L = H * W
    xs = OracleCrossScan.apply(x)
    x_dbl = torch.einsum("b k d l, k c d -> b k c l", xs, Wx)
    dts, Bs, Cs = torch.split(x_dbl, [1, 1, 1], dim=2)
    dts = torch.einsum("b k r l, k d r -> b k d l", dts, Wdt)
    ys = OracleSelectiveScan.apply(xs.reshape(B, -1, L), dts.contiguous().view(B, -1, L), -torch.exp(A_logs), Bs.contiguous(),
                                   Cs.contiguous(), Ds, dtb.reshape(-1), True)
    return OracleCrossMerge.apply(ys.view(B, 4, D, H, W))


def _run(fn, x, params, gy):
    x = x.clone().requires_grad_()
    ps = [p.clone().requires_grad_() for p in params]
    y = fn(x, *ps)
    y.backward(gy.to(y.dtype).to(y.device))
    return [y.detach()] + [t.grad.detach() for t in [x] + ps]


NAMES = ["y", "dx", "dWx", "dWdt", "ddtb", "dA_logs", "dDs"]


@pytest.mark.gpu
@pytest.mark.parametrize("shape", SHAPES + BENCH_SHAPES)
def test_fused_core_matches_oracle_chain_fp32(shape):
    """Fused HIP core vs the oracle's chain of the reference steps, adjudicated by the float64 build of the same chain —
    at small shapes AND at the benchmark call shapes.  Bounds (round 3, after the decay exp stopped being v_exp_f32 —
    csrc/scan_prims.h decay_f): the output and dx are as close to float64 as the sequential fp32 recurrence of
    selective_scan_ref (y: x1.5 in the max norm and in RMS, + 2e-8 of scale; dx: x2.5 / x2); parameter gradients (fp32 sums over up to
    262 144 positions per row) within 6x its distance + 2e-6 of scale.  Measured on MI355X
    (profiles/r03_accuracy_probe.log): y 8e-9 RMS / 8e-8..1.6e-7 max of scale for both; with v_exp_f32 the HIP output
    was 1.7x and d(A_logs), d(dt_bias) 10-40x further from float64 than the oracle chain."""
    import oracle
    from vm_asr_amd.ss2d_core import ss2d_core, supported
    B, D, H, W = shape
    assert supported(1, 1, D, H, W)
    g = torch.Generator().manual_seed(D * 7 + H)
    x = torch.randn(B, D, H, W, generator=g)
    gy = torch.randn(B, D, H * W, generator=g)
    params = _params(D, D + 1)
    got = _run(ss2d_core, x.cuda(), [p.cuda() for p in params], gy)
    ref = _run(_oracle_core, x, params, gy)
    with oracle.float64():
        r64 = _run(_oracle_core, x.double(), _params(D, D + 1, torch.float64), gy.double())
    for n, a, b, c in zip(NAMES, got, ref, r64):
        a, b = a.double().cpu(), b.double()
        scale = max(c.abs().max().item(), 1e-12)
        e_hip, e_cpu = (a - c).abs().max().item() / scale, (b - c).abs().max().item() / scale
        r_hip, r_cpu = (a - c).pow(2).mean().sqrt().item() / scale, (b - c).pow(2).mean().sqrt().item() / scale
        assert a.shape == c.shape, n
        print(f"{shape} {n}: |hip - f64| max {e_hip:.2e} rms {r_hip:.2e}  |oracle fp32 chain - f64| max {e_cpu:.2e} rms {r_cpu:.2e}")
        if n in ("y", "dx"):
            k_max, k_rms = (1.5, 1.5) if n == "y" else (2.5, 2.0)            # dx measured: 1.5-1.6x in RMS, <= 2.3x max
            assert e_hip <= 1e-4, (shape, n, e_hip, e_cpu)                    # north_star (fp32)
            assert e_hip <= k_max * e_cpu + 2e-8, (shape, n, e_hip, e_cpu)
            assert r_hip <= k_rms * r_cpu + 2e-9, (shape, n, r_hip, r_cpu)
        else:
            assert e_hip <= 6 * e_cpu + 2e-6, (shape, n, e_hip, e_cpu)


@pytest.mark.gpu
def test_fused_core_bf16_activations():
    """Under autocast x arrives in bf16 (the depthwise conv's output): values are converted on load, arithmetic stays
    fp32, dx leaves in bf16.  Against the oracle chain on the same bf16-rounded x: y to 1e-4, dx to bf16 rounding."""
    from vm_asr_amd.ss2d_core import ss2d_core
    B, D, H, W = 2, 32, 32, 32
    g = torch.Generator().manual_seed(3)
    x = torch.randn(B, D, H, W, generator=g).to(torch.bfloat16)
    gy = torch.randn(B, D, H * W, generator=g)
    params = _params(D, 9)
    got = _run(ss2d_core, x.cuda(), [p.cuda() for p in params], gy)
    ref = _run(_oracle_core, x.float(), params, gy)
    assert got[0].dtype == torch.float32 and got[1].dtype == torch.bfloat16
    for n, a, b in zip(NAMES, got, ref):
        tol = 1e-2 if n == "dx" else 2e-4
        err, scale = (a.float().cpu() - b).abs().max().item(), b.abs().max().item()
        assert err <= tol * scale, (n, err, scale)


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(4, 32, 128, 128), (4, 16, 256, 256), (4, 2, 512, 512)])
def test_fused_core_equals_unfused_hip_chain_at_benchmark_shapes(shape):
    """BASELINE sizes (B = 4, the three fused stages of vm_asr_48k): the fused operator against the unfused HIP chain
    of round 1 (CrossScanF32 -> xproj -> SelectiveScanCore -> CrossMerge, each oracle-checked in test_gpu_kernels.py):
    output and dx agree to 2e-4 of the tensor scale, parameter gradients (fp32 sums over B*L = 65 k .. 1 M positions
    in different orders) to 1e-3."""
    from vm_asr_amd import xproj
    from vm_asr_amd.csm import CrossMergeHIP, CrossScanF32
    from vm_asr_amd.selective_scan import SelectiveScanCore
    from vm_asr_amd.ss2d_core import ss2d_core
    B, D, H, W = shape
    L = H * W

    def chain(x, Wx, Wdt, dtb, A_logs, Ds):
        xs = CrossScanF32.apply(x)
        dts, Bs, Cs = xproj.x_proj_dt(xs, Wx, Wdt, 1)
        ys = SelectiveScanCore.apply(xs.view(B, -1, L), dts, -torch.exp(A_logs.float()), Bs, Cs, Ds.float(), dtb.view(-1).float(), True)
        return CrossMergeHIP.apply(ys.view(B, 4, D, H, W))
    g = torch.Generator().manual_seed(11)
    x = torch.randn(B, D, H, W, generator=g).cuda()
    gy = torch.randn(B, D, L, generator=g)
    params = [p.cuda() for p in _params(D, 5)]
    got, ref = _run(ss2d_core, x, params, gy), _run(chain, x, params, gy)
    for i, (n, a, b) in enumerate(zip(NAMES, got, ref)):
        err, scale = (a - b).abs().max().item(), max(b.abs().max().item(), 1e-12)
        assert err <= (2e-4 if i < 2 else 1e-3) * scale, (shape, n, err, scale)


@pytest.mark.gpu
def test_fused_core_scan_properties_full_size():
    """Size-independent properties at the largest fused call (d_inner 2, 512 x 512, B = 4): linear in Ds (y(Ds) -
    y(0) = sum_k Ds_k u exactly as cross-merge adds the four D*u terms), and zero input -> zero output."""
    from vm_asr_amd.ss2d_core import ss2d_core
    B, D, H, W = 4, 2, 512, 512
    x = torch.randn(B, D, H, W, device="cuda")
    params = [p.cuda() for p in _params(D, 4)]
    y1 = ss2d_core(x, *params)
    p0 = list(params)
    p0[4] = torch.zeros_like(params[4])
    y0 = ss2d_core(x, *p0)
    want = params[4].view(4, D).sum(0).view(1, D, 1) * x.view(B, D, -1)
    assert torch.allclose(y1 - y0, want, rtol=1e-4, atol=1e-4 * want.abs().max().item())
    assert float(ss2d_core(torch.zeros_like(x), *params).abs().max()) == 0.0
