"""The SS2D core of the deep stages (vm_asr_amd/csrc/ss2d_deep.hip: x_proj kernel + whole-row scan kernel with cross-scan /
cross-merge through an LDS image, forward and backward) against the oracle's composition of the SAME reference steps
(model/vmamba.py:1472-1497: CrossScan -> einsum x2 -> selective_scan_ref -> CrossMerge; each oracle piece is pinned to reference
goldens in tests/test_oracle.py), in fp32 and — as adjudicator — in float64; and against the unfused HIP chain it replaces."""
import numpy as np
import pytest
import torch

# (B, d_inner, H, W, dt_rank): the three deep call shapes of vm_asr_48k (64x64x64 r2, 128x32x32 r4, 256x16x16 r8) at B = 1 and 2,
# non-square images, every waves-per-row count (1, 2, 4, 8, 16), the d_inner 512 / dt_rank 16 stage of the DIMS-32 configs; d_inner in
# {64, 96, 128, 256, 512} x dt_rank in {2, 4, 8, 16} each covered at least twice
SHAPES = [(1, 64, 64, 64, 2), (2, 128, 32, 32, 4), (2, 256, 16, 16, 8), (1, 64, 32, 16, 2), (1, 96, 16, 32, 4), (1, 64, 8, 32, 8),
          (1, 64, 64, 32, 4), (1, 512, 16, 16, 8), (1, 512, 16, 16, 16), (1, 256, 32, 16, 16), (1, 128, 16, 16, 2), (1, 512, 16, 32, 4),
          (1, 128, 16, 32, 16), (1, 256, 64, 64, 2)]
NAMES = ["y", "dx", "dWx", "dWdt", "ddtb", "dA_logs", "dDs"]


def _params(D, R, seed, dtype=torch.float32):
    g = torch.Generator().manual_seed(seed)
    r = lambda *s: torch.rand(*s, generator=g, dtype=torch.float64)      # noqa: E731
    Wx = ((2 * r(4, R + 2, D) - 1) / D ** 0.5)
    Wdt = (2 * r(4, D, R) - 1) * R ** -0.5                                # model/vmamba.py:1204-1225 (dt_init)
    dt = torch.exp(r(4, D) * (np.log(0.1) - np.log(1e-3)) + np.log(1e-3))
    dtb = dt + torch.log(-torch.expm1(-dt))
    A_logs = 0.3 * (2 * r(4 * D, 1) - 1)
    Ds = 1 + 0.2 * (2 * r(4 * D) - 1)
    return [t.to(dtype) for t in (Wx, Wdt, dtb, A_logs, Ds)]


def _oracle_core(x, Wx, Wdt, dtb, A_logs, Ds):
    """model/vmamba.py:1472-1497 on the oracle's kernels (dtype follows the active oracle build)."""
    from oracle.torch_backend import OracleCrossMerge, OracleCrossScan, OracleSelectiveScan
    B, D, H, W = x.shape
    L, R = H * W, Wdt.shape[-1]
    xs = OracleCrossScan.apply(x)
    x_dbl = torch.einsum("b k d l, k c d -> b k c l", xs, Wx)
    dts, Bs, Cs = torch.split(x_dbl, [R, 1, 1], dim=2)
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


@pytest.mark.gpu
@pytest.mark.parametrize("shape", SHAPES)
def test_deep_core_matches_oracle_chain_fp32(shape):
    """Same bounds as the high-resolution fused core (tests/test_ss2d_fused.py): output and dx as close to float64 as the
    sequential fp32 recurrence of selective_scan_ref (x1.5 / x2.5 + a floor), parameter gradients within 6x + 2e-6 of scale."""
    import oracle
    from vm_asr_amd.ss2d_deep import ss2d_deep, supported
    B, D, H, W, R = shape
    assert supported(1, R, D, H, W)
    g = torch.Generator().manual_seed(D * 7 + H)
    x = torch.randn(B, D, H, W, generator=g)
    gy = torch.randn(B, D, H * W, generator=g)
    params = _params(D, R, D + 1)
    got = _run(ss2d_deep, x.cuda(), [p.cuda() for p in params], gy)
    ref = _run(_oracle_core, x, params, gy)
    with oracle.float64():
        r64 = _run(_oracle_core, x.double(), _params(D, R, D + 1, torch.float64), gy.double())
    for n, a, b, c in zip(NAMES, got, ref, r64):
        a, b = a.double().cpu(), b.double()
        scale = max(c.abs().max().item(), 1e-12)
        e_hip, e_cpu = (a - c).abs().max().item() / scale, (b - c).abs().max().item() / scale
        r_hip, r_cpu = (a - c).pow(2).mean().sqrt().item() / scale, (b - c).pow(2).mean().sqrt().item() / scale
        assert a.shape == c.shape, n
        print(f"{shape} {n}: |hip - f64| max {e_hip:.2e} rms {r_hip:.2e}  |oracle fp32 chain - f64| max {e_cpu:.2e} rms {r_cpu:.2e}")
        if n in ("y", "dx"):
            k_max, k_rms = (1.5, 1.5) if n == "y" else (2.5, 2.0)
            assert e_hip <= 1e-4, (shape, n, e_hip, e_cpu)                    # north_star (fp32)
            assert e_hip <= k_max * e_cpu + 5e-8, (shape, n, e_hip, e_cpu)
            assert r_hip <= k_rms * r_cpu + 5e-9, (shape, n, r_hip, r_cpu)
        else:
            assert e_hip <= 6 * e_cpu + 2e-6, (shape, n, e_hip, e_cpu)


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(4, 64, 64, 64, 2), (4, 128, 32, 32, 4), (4, 256, 16, 16, 8)])
def test_deep_core_equals_unfused_hip_chain_at_benchmark_shapes(shape):
    """BASELINE sizes (B = 4, the three deep stages of vm_asr_48k): against the unfused HIP chain this operator replaces
    (CrossScanF32 -> xproj -> SelectiveScanCore -> CrossMerge, each oracle-checked in test_gpu_kernels.py): output and dx to
    2e-4 of the tensor scale, parameter gradients (fp32 sums in different orders) to 1e-3."""
    from vm_asr_amd import xproj
    from vm_asr_amd.csm import CrossMergeHIP, CrossScanF32
    from vm_asr_amd.selective_scan import SelectiveScanCore
    from vm_asr_amd.ss2d_deep import ss2d_deep
    B, D, H, W, R = shape
    L = H * W

    def chain(x, Wx, Wdt, dtb, A_logs, Ds):
        xs = CrossScanF32.apply(x)
        dts, Bs, Cs = xproj.x_proj_dt(xs, Wx, Wdt, 1)
        ys = SelectiveScanCore.apply(xs.view(B, -1, L), dts, -torch.exp(A_logs.float()), Bs, Cs, Ds.float(), dtb.view(-1).float(), True)
        return CrossMergeHIP.apply(ys.view(B, 4, D, H, W))
    g = torch.Generator().manual_seed(11)
    x = torch.randn(B, D, H, W, generator=g).cuda()
    gy = torch.randn(B, D, L, generator=g)
    params = [p.cuda() for p in _params(D, R, 5)]
    got, ref = _run(ss2d_deep, x, params, gy), _run(chain, x, params, gy)
    for i, (n, a, b) in enumerate(zip(NAMES, got, ref)):
        err, scale = (a - b).abs().max().item(), max(b.abs().max().item(), 1e-12)
        print(f"{shape} {n}: {err / scale:.2e}")
        assert err <= (2e-4 if i < 2 else 1e-3) * scale, (shape, n, err, scale)


@pytest.mark.gpu
def test_deep_core_bf16_activations():
    """Under autocast x arrives in bf16: values are converted on load, arithmetic stays fp32, the per-row terms of d(x_dbl)
    and dx leave in bf16 (the reference rounds the same gradients to bf16 for its autocast einsum backward).  Against the oracle
    chain on the same bf16-rounded x: y to 2e-4 of scale, gradients to bf16 rounding."""
    from vm_asr_amd.ss2d_deep import ss2d_deep
    B, D, H, W, R = 2, 128, 32, 32, 4
    g = torch.Generator().manual_seed(3)
    x = torch.randn(B, D, H, W, generator=g).to(torch.bfloat16)
    gy = torch.randn(B, D, H * W, generator=g)
    params = _params(D, R, 9)
    got = _run(ss2d_deep, x.cuda(), [p.cuda() for p in params], gy)
    ref = _run(_oracle_core, x.float(), params, gy)
    assert got[0].dtype == torch.float32 and got[1].dtype == torch.bfloat16
    for n, a, b in zip(NAMES, got, ref):
        tol = 2e-4 if n in ("y", "dA_logs", "dDs", "ddtb", "dWdt") else 1e-2
        err, scale = (a.float().cpu() - b).abs().max().item(), b.abs().max().item()
        print(f"bf16 {n}: {err / scale:.2e}")
        assert err <= tol * scale, (n, err, scale)


@pytest.mark.gpu
def test_ss2d_module_uses_deep_core():
    """SS2D.forward of a deep-stage block (d_model 64 -> d_inner 128, dt_rank 4, 32 x 32) runs the deep core and agrees with
    the unfused path (VMASR_SS2D_DEEP=0) on the same weights: output and every gradient."""
    import os
    from vm_asr_amd import ss2d_deep
    from vm_asr_amd.vmamba import SS2D
    torch.manual_seed(0)
    m = SS2D(d_model=64, d_state=1, ssm_ratio=2.0, dt_rank="auto", forward_type="v5").cuda()
    x = torch.randn(2, 32, 32, 64, device="cuda")
    calls = []
    orig = ss2d_deep.ss2d_deep

    def spy(*a):
        calls.append(a[0].shape)
        return orig(*a)
    out = []
    for flag in ("1", "0"):
        os.environ["VMASR_SS2D_DEEP"] = flag
        try:
            ss2d_deep.ss2d_deep = spy
            xi = x.clone().requires_grad_()
            m.zero_grad()
            y = m(xi)
            y.square().mean().backward()
            out.append([y.detach(), xi.grad] + [p.grad.clone() for p in m.parameters()])
        finally:
            ss2d_deep.ss2d_deep = orig
            os.environ.pop("VMASR_SS2D_DEEP", None)
    assert calls == [torch.Size([2, 128, 32, 32])]
    for a, b in zip(*out):
        err, scale = (a - b).abs().max().item(), max(b.abs().max().item(), 1e-12)
        assert err <= 1e-3 * scale, (err, scale)
