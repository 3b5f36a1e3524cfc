"""ss2d_pre / ln_gate (vm_asr_amd/csrc/ss2d_glue.hip) against the torch expressions of SS2D.forwardv2 they replace
(model/vmamba.py:1537-1542 and :1528-1531,1550), forward and backward, fp32 (vs float64) and bf16 (vs the same
expression on the bf16-rounded inputs)."""
import pytest
import torch
import torch.nn.functional as F

SHAPES = [(2, 8, 8, 2), (1, 16, 16, 16), (2, 16, 8, 32), (1, 8, 8, 64), (2, 8, 8, 128), (1, 8, 4, 256), (1, 4, 4, 512), (4, 128, 128, 32)]


def _ref_pre(xz):
    x, z = xz.chunk(2, dim=-1)
    return x.permute(0, 3, 1, 2).contiguous(), F.silu(z)


def _ref_ln_gate(y, sz, w, b, eps):
    B, H, W, D = sz.shape
    return F.layer_norm(y.transpose(1, 2).contiguous(), (D,), w, b, eps).view(B, H, W, D) * sz


@pytest.mark.gpu
@pytest.mark.parametrize("shape", SHAPES)
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_ss2d_pre(shape, dtype):
    from vm_asr_amd.ss2d_glue import ss2d_pre, supported
    B, H, W, D = shape
    if not supported(D, H * W, dtype):
        pytest.skip("shape not on the fused path")
    g = torch.Generator().manual_seed(D)
    xz = (2 * torch.randn(B, H, W, 2 * D, generator=g)).to(dtype)
    gx, gz = torch.randn(B, D, H, W, generator=g).to(dtype), torch.randn(B, H, W, D, generator=g).to(dtype)
    a = xz.cuda().requires_grad_()
    xT, sz = ss2d_pre(a)
    (xT.float() * gx.cuda().float()).sum().add((sz.float() * gz.cuda().float()).sum()).backward()
    r = xz.double().requires_grad_()
    rx, rz = _ref_pre(r)
    ((rx * gx.double()).sum() + (rz * gz.double()).sum()).backward()
    tol = 1e-6 if dtype == torch.float32 else 1e-2
    assert xT.dtype == sz.dtype == dtype and torch.equal(xT.cpu().double(), rx.detach().to(dtype).double())       # pure data movement
    assert torch.allclose(sz.cpu().double(), rz.detach(), rtol=tol, atol=tol)
    assert torch.allclose(a.grad.cpu().double(), r.grad, rtol=tol, atol=tol * r.grad.abs().max().item())


@pytest.mark.gpu
@pytest.mark.parametrize("shape", SHAPES)
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_ln_gate(shape, dtype):
    from vm_asr_amd.ss2d_glue import ln_gate, supported
    B, H, W, D = shape
    if not supported(D, H * W, dtype):
        pytest.skip("shape not on the fused path")
    g = torch.Generator().manual_seed(D + 1)
    y = 3 * torch.randn(B, D, H * W, generator=g) + 0.5
    sz = torch.randn(B, H, W, D, generator=g).to(dtype)
    w, b = 1 + 0.2 * torch.randn(D, generator=g), 0.1 * torch.randn(D, generator=g)
    go = torch.randn(B, H, W, D, generator=g).to(dtype)
    ins = [t.cuda().requires_grad_() for t in (y, sz, w, b)]
    out = ln_gate(*ins, 1e-5)
    (out.float() * go.cuda().float()).sum().backward()
    ref_in = [t.double().requires_grad_() for t in (y, sz, w, b)]
    ref = _ref_ln_gate(*ref_in, 1e-5)
    (ref * go.double()).sum().backward()
    assert out.dtype == dtype
    tol = 2e-5 if dtype == torch.float32 else 2e-2
    # d_inner = 2: a LayerNorm over two channels amplifies rounding by up to 1/sqrt(eps) where they nearly coincide
    names = ["out", "dy", "dsz", "dgamma", "dbeta"]
    got = [out] + [t.grad for t in ins]
    want = [ref.detach()] + [t.grad for t in ref_in]
    for n, a, r in zip(names, got, want):
        scale = max(r.abs().max().item(), 1e-12)
        lim = tol * (50 if D == 2 and n == "dy" else 1)
        assert (a.cpu().double() - r).abs().max().item() <= lim * scale, (shape, dtype, n, (a.cpu().double() - r).abs().max().item(), scale)


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("shape", [(2, 16, 32, 32), (1, 32, 16, 2), (2, 48, 16, 8), (1, 128, 128, 16)])
def test_ln_gate_pairs_equals_ln_gate_of_the_merged_tensor(shape, dtype):
    """ln_gate_pairs(y02, y13, ...) == ln_gate(y02 + transpose_hw(y13), ...) (what is left of CrossMerge,
    model/vmamba.py:50-73, formed on the fly), forward and every gradient — dy comes back in BOTH orders —, against float64."""
    from vm_asr_amd.ss2d_glue import ln_gate_pairs, pairs_supported
    B, H, W, D = shape
    assert pairs_supported(D, H, W, dtype)
    g = torch.Generator().manual_seed(D + H)
    y02 = 2 * torch.randn(B, D, H * W, generator=g) + 0.3
    y13 = 2 * torch.randn(B, D, W * H, generator=g)
    sz = torch.randn(B, H, W, D, generator=g).to(dtype)
    w, b = 1 + 0.2 * torch.randn(D, generator=g), 0.1 * torch.randn(D, generator=g)
    go = torch.randn(B, H, W, D, generator=g).to(dtype)
    ins = [t.cuda().requires_grad_() for t in (y02, y13, sz, w, b)]
    out = ln_gate_pairs(*ins, 1e-5)
    (out.float() * go.cuda().float()).sum().backward()
    r = [t.double().requires_grad_() for t in (y02, y13, sz, w, b)]
    ymerged = r[0] + r[1].view(B, D, W, H).transpose(2, 3).reshape(B, D, H * W)
    ref = _ref_ln_gate(ymerged, *r[2:], 1e-5)
    (ref * go.double()).sum().backward()
    tol = 2e-5 if dtype == torch.float32 else 2e-2
    for n, a, q in zip(["out", "dy02", "dy13", "dsz", "dgamma", "dbeta"], [out] + [t.grad for t in ins], [ref.detach()] + [t.grad for t in r]):
        scale = max(q.abs().max().item(), 1e-12)
        lim = tol * (50 if D == 2 and n.startswith("dy") else 1)
        assert (a.cpu().double() - q).abs().max().item() <= lim * scale, (shape, dtype, n)


@pytest.mark.gpu
def test_ss2d_forward_with_pairs_equals_merged_path(monkeypatch):
    """SS2D.forward through (fused core -> pair outputs -> ln_gate_pairs) equals the path with the merged tensor
    (VMASR_SS2D_PAIRS=0), output and all gradients, fp32."""
    from vm_asr_amd.vmamba import SS2D
    torch.manual_seed(3)
    m = SS2D(d_model=8, d_state=1, ssm_ratio=2.0, dt_rank="auto", forward_type="v5").cuda()
    x = torch.randn(2, 32, 48, 8, device="cuda")
    gy = torch.randn(2, 32, 48, 8, device="cuda")
    res = {}
    for flag in ("1", "0"):
        monkeypatch.setenv("VMASR_SS2D_PAIRS", flag)
        xi = x.clone().requires_grad_()
        m.zero_grad()
        y = m(xi)
        y.backward(gy)
        res[flag] = [y.detach(), xi.grad] + [p.grad.clone() for p in m.parameters()]
    for a, b in zip(res["1"], res["0"]):
        assert (a - b).abs().max() <= 2e-5 * max(1e-6, b.abs().max().item()) + 1e-7
