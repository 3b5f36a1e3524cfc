"""csrc/convgemm.hip — the period discriminator's (k,1) convolutions as implicit bf16x3 MFMA GEMMs, against a float64
evaluation of the reference's lines (model/discriminator.py:40-104: Conv2d (k,1) stride (s,1) padding (pad,0) -> GELU)
and against torch's own fp32 convolution's distance from float64.

Tolerance: north_star's fp32 bound, 1e-4 of the tensor scale; measured ~3e-6 (the triple carries 16-17 bits per product,
fp32 accumulation)."""
import pytest
import torch
import torch.nn.functional as F

gpu = pytest.mark.gpu


def _ref(xs, W, bias, k, stride, pad, act):
    """float64: per slot x (nseq, H, Cin), W (Cout, k*Cin) in (tap, channel) order -> pre, y (nseq*H1, Cout)"""
    pres, ys = [], []
    for x, w, b in zip(xs, W, bias):
        Cout, Cin = w.shape[0], x.shape[2]
        w4 = w.view(Cout, k, Cin).permute(0, 2, 1).unsqueeze(3)                     # (Cout, Cin, k, 1)
        o = F.conv2d(x.permute(0, 2, 1).unsqueeze(3), w4, b, (stride, 1), (pad, 0))  # (nseq, Cout, H1, 1)
        pre = o.squeeze(3).permute(0, 2, 1).reshape(-1, Cout)
        pres.append(pre)
        ys.append(F.gelu(pre) if act else pre)
    return pres, ys


def _stack(ts, rows):
    out = torch.zeros((len(ts), rows, ts[0].shape[-1]), dtype=ts[0].dtype, device=ts[0].device)
    for i, t in enumerate(ts):
        out[i, :t.shape[0]] = t
    return out


CASES = [
    # (Cin, Cout, k, stride, pad, geom [(nseq, H)])
    (128, 128, 5, 3, 2, [(3, 50), (2, 77)]),
    (128, 256, 5, 3, 2, [(4, 130), (6, 85), (10, 52)]),
    (256, 128, 5, 1, 2, [(2, 64), (3, 41)]),
    (128, 128, 3, 2, 1, [(5, 33)]),
    (128, 128, 5, 3, 2, [(7, 3)]),               # sequences shorter than the kernel
    (32, 128, 5, 3, 2, [(5, 61), (3, 100)]),     # the 32 -> 128 layer: 256 x 128 forward, 256 x 32 dgrad, 128 x 32 wgrad tiles
    (128, 256, 5, 3, 2, [(9, 200)]),             # several 256-row tiles
    (256, 256, 5, 1, 2, [(6, 90), (4, 131)]),    # 256 x 256 tiles everywhere
]


@gpu
@pytest.mark.parametrize("case", CASES)
def test_conv_mfma_fwd_dgrad_wgrad_match_float64(case):
    from vm_asr_amd import convgemm as cg
    from vm_asr_amd.discriminator import split_bf16
    Cin, Cout, k, stride, pad, geom = case
    assert cg.supported(Cin, Cout, k, stride)
    dev = torch.device("cuda:0")
    g = torch.Generator(device="cpu").manual_seed(7)
    n = len(geom)
    xs = [torch.randn(ns, H, Cin, generator=g).to(dev) for ns, H in geom]
    W = (torch.randn(n, Cout, k * Cin, generator=g) / (k * Cin) ** 0.5).to(dev)
    bias = torch.randn(n, Cout, generator=g).to(dev)
    H1 = [cg.out_positions(H, k, stride, pad) for _, H in geom]
    Ms = [ns * h1 for (ns, _), h1 in zip(geom, H1)]
    rows_in = -(-max(ns * H for ns, H in geom) // 256) * 256 + 256        # some zero rows below every slot
    rows_out = -(-max(Ms) // 256) * 256
    x = _stack([t.reshape(-1, Cin) for t in xs], rows_in)
    xh, xl = split_bf16(x)
    wh, wl = split_bf16(W)
    pre, y, yh, yl = cg.conv_fwd(xh, xl, wh, wl, bias, geom, k, stride, pad, rows_out, act=True)
    xs64 = [t.double().requires_grad_(True) for t in xs]
    W64 = W.double().requires_grad_(True)
    pres64, ys64 = _ref(xs64, W64, bias.double(), k, stride, pad, True)
    # torch's fp32 evaluation of the same lines, for scale
    pres32, _ = _ref(xs, W, bias, k, stride, pad, True)
    for i, M in enumerate(Ms):
        sc = pres64[i].abs().max().item()
        e = (pre[i, :M].double() - pres64[i]).abs().max().item()
        e32 = (pres32[i].double() - pres64[i]).abs().max().item()
        assert e <= 1e-4 * sc, (i, e, sc)
        assert e <= 20 * e32 + 1e-6 * sc, (i, e, e32)
        assert (y[i, :M].double() - ys64[i]).abs().max().item() <= 1e-4 * max(1.0, ys64[i].abs().max().item())
        pair = yh[i, :M].float() + yl[i, :M].float()
        assert (pair - y[i, :M]).abs().max().item() <= 2 ** -16 * y[i, :M].abs().max().item()
        for t in (pre, y, yh, yl):
            assert not t[i, M:].any(), "padding rows must be zero"
    # backward: a random output gradient (zero on padding rows), its pair
    gy = [torch.randn(M, Cout, generator=g).to(dev) for M in Ms]
    gst = _stack(gy, rows_out)
    gh, gl = split_bf16(gst)
    loss = sum((p * gg.double()).sum() for p, gg in zip(pres64, gy))
    loss.backward()
    Wt = W.view(n, Cout, k, Cin).permute(0, 3, 2, 1).reshape(n, Cin, k * Cout).contiguous()
    wth, wtl = split_bf16(Wt)
    dx = cg.conv_dgrad(gh, gl, wth, wtl, geom, k, stride, pad, rows_in)
    for i, (ns, H) in enumerate(geom):
        want = xs64[i].grad.reshape(-1, Cin)
        sc = max(want.abs().max().item(), 1e-30)
        assert (dx[i, :ns * H].double() - want).abs().max().item() <= 1e-4 * sc
        assert not dx[i, ns * H:].any(), "rows below the slot's data must be zero"
    for splits in (None, 1, 3):
        dw = cg.conv_wgrad(gh, gl, xh, xl, geom, k, stride, pad, splits=splits)
        sc = W64.grad.abs().max().item()
        assert (dw.double() - W64.grad).abs().max().item() <= 1e-4 * sc, splits


@gpu
def test_conv_mfma_full_size_layer_matches_float64_on_sampled_rows():
    """configs[2]'s last hidden layer (1024 -> 1024 channels, k 5, stride 1) at the size of one period discriminator's pair
    pass: sampled output rows against float64 (the whole convolution in float64 would take minutes)."""
    from vm_asr_amd import convgemm as cg
    from vm_asr_amd.discriminator import split_bf16
    dev = torch.device("cuda:0")
    g = torch.Generator(device="cpu").manual_seed(3)
    Cin = Cout = 1024
    k, stride, pad = 5, 1, 2
    geom = [(16, 758)]
    M = 16 * 758
    rows = -(-M // 256) * 256
    x = torch.zeros(1, rows, Cin, device=dev)
    x[0, :M] = torch.randn(M, Cin, generator=g).to(dev)
    W = (torch.randn(1, Cout, k * Cin, generator=g) / (k * Cin) ** 0.5).to(dev)
    bias = torch.randn(1, Cout, generator=g).to(dev)
    xh, xl = split_bf16(x)
    wh, wl = split_bf16(W)
    pre, y, _, _ = cg.conv_fwd(xh, xl, wh, wl, bias, geom, k, stride, pad, rows, act=True)
    rows_chk = torch.tensor([0, 1, 2, 757, 758, 759, 5000, 9093, M - 3, M - 1], device=dev)
    x64 = x[0, :M].double().view(16, 758, Cin)
    xp = F.pad(x64, (0, 0, pad, pad))                                   # (16, 762, Cin)
    cols = torch.stack([xp[r // 758, (r % 758):(r % 758) + k].reshape(-1) for r in rows_chk.tolist()])
    want = cols @ W[0].double().t() + bias[0].double()
    got = pre[0, rows_chk].double()
    assert (got - want).abs().max().item() <= 1e-4 * want.abs().max().item()
    assert (y[0, rows_chk].double() - F.gelu(want)).abs().max().item() <= 1e-4 * want.abs().max().item()


@gpu
def test_conv_mfma_cu_limit_changes_nothing_but_the_grid():
    """vmasr_conv_set_cu_limit: fewer workgroups than tiles, each looping over several — bit-identical outputs."""
    from vm_asr_amd import convgemm as cg
    from vm_asr_amd.discriminator import split_bf16
    dev = torch.device("cuda:0")
    g = torch.Generator(device="cpu").manual_seed(11)
    Cin, Cout, k, stride, pad = 256, 256, 5, 3, 2
    geom = [(6, 700), (4, 1001), (9, 333)]
    n = len(geom)
    H1 = [cg.out_positions(H, k, stride, pad) for _, H in geom]
    Ms = [ns * h1 for (ns, _), h1 in zip(geom, H1)]
    rows_in = -(-max(ns * H for ns, H in geom) // 256) * 256
    rows_out = -(-max(Ms) // 256) * 256
    x = _stack([torch.randn(ns * H, Cin, generator=g).to(dev) for ns, H in geom], rows_in)
    W = (torch.randn(n, Cout, k * Cin, generator=g) / (k * Cin) ** 0.5).to(dev)
    bias = torch.randn(n, Cout, generator=g).to(dev)
    gy = _stack([torch.randn(M, Cout, generator=g).to(dev) for M in Ms], rows_out)
    xh, xl = split_bf16(x)
    wh, wl = split_bf16(W)
    gh, gl = split_bf16(gy)
    wth, wtl = split_bf16(W.view(n, Cout, k, Cin).permute(0, 3, 2, 1).reshape(n, Cin, k * Cout).contiguous())

    def run():
        f = cg.conv_fwd(xh, xl, wh, wl, bias, geom, k, stride, pad, rows_out, act=True)
        d = cg.conv_dgrad(gh, gl, wth, wtl, geom, k, stride, pad, rows_in)
        w = cg.conv_wgrad(gh, gl, xh, xl, geom, k, stride, pad, splits=2)
        torch.cuda.synchronize()
        return list(f) + [d, w]

    ref = run()
    for cus in (8, 24, 200):
        with cg.cu_limit(cus):
            got = run()
        assert int(cg._lib.lib().vmasr_conv_get_cu_limit()) == 0 and cg._LIMIT["cus"] == 0
        for a, b in zip(ref, got):
            assert torch.equal(a, b), cus


@gpu
@pytest.mark.parametrize("case", [c for c in CASES if c[0] % 128 == 0])
def test_conv_mfma_dgrad_with_the_activation_backward_in_its_epilogue(case):
    """vmasr_conv_mfma_dgrad_gelu == vmasr_conv_mfma_dgrad -> vmasr_masked_l1_bwd_add -> vmasr_gelu_bwd_split / vmasr_gelu_bwd, the three
    passes it replaces (the reference's autograd: conv backward, + the feature-matching gradient, GELU backward): BIT-EXACT — the same fp32
    tile, the same fused multiply-add of the sign term, the same GELU' expression, the same split; with and without the sign term, fp32
    output and pair."""
    import ctypes
    from vm_asr_amd import _lib, convgemm as cg
    from vm_asr_amd.discriminator import split_bf16
    Cin, Cout, k, stride, pad, geom = case
    dev = torch.device("cuda:0")
    g = torch.Generator(device="cpu").manual_seed(11)
    n = len(geom)
    H1 = [cg.out_positions(H, k, stride, pad) for _, H in geom]
    Ms = [ns * h1 for (ns, _), h1 in zip(geom, H1)]
    rows_in = -(-max(ns * H for ns, H in geom) // 256) * 256 + 256
    rows_out = -(-max(Ms) // 256) * 256
    W = (torch.randn(n, Cout, k * Cin, generator=g) / (k * Cin) ** 0.5).to(dev)
    Wt = W.view(n, Cout, k, Cin).permute(0, 3, 2, 1).reshape(n, Cin, k * Cout).contiguous()
    wth, wtl = split_bf16(Wt)
    gh, gl = split_bf16(_stack([torch.randn(M, Cout, generator=g).to(dev) for M in Ms], rows_out))
    pre = (2.0 * torch.randn(n, rows_in, Cin, generator=g)).to(dev)
    sgn = torch.randint(-1, 2, (n, rows_in, Cin), generator=g, dtype=torch.int8).to(dev)
    gtok = torch.tensor([0.37], device=dev)
    valid = [max(ns * H - 5 * i, 0) for i, (ns, H) in enumerate(geom)]
    scale = [0.5 + 0.25 * i for i in range(n)]
    lib = _lib.lib()
    dx = cg.conv_dgrad(gh, gl, wth, wtl, geom, k, stride, pad, rows_in)
    for with_sgn in (False, True):
        t = dx
        if with_sgn:
            t = torch.empty_like(dx)
            v = (ctypes.c_int64 * n)(*valid)
            sc = (ctypes.c_float * n)(*scale)
            _lib.check(lib.vmasr_masked_l1_bwd_add(sgn.data_ptr(), gtok.data_ptr(), dx.data_ptr(), t.data_ptr(), v, sc, n, rows_in, Cin,
                                                   _lib.current_stream(dev)), "masked_l1_bwd_add")
        rh, rl, r32 = torch.empty_like(gh[:, :0]).new_empty((n, rows_in, Cin)), None, torch.empty_like(t)
        rl = torch.empty_like(rh)
        _lib.check(lib.vmasr_gelu_bwd_split(pre.data_ptr(), t.data_ptr(), rh.data_ptr(), rl.data_ptr(), None, None, n, rows_in, Cin,
                                            _lib.current_stream(dev)), "gelu_bwd_split")
        _lib.check(lib.vmasr_gelu_bwd(pre.data_ptr(), t.data_ptr(), r32.data_ptr(), None, n, rows_in, Cin, _lib.current_stream(dev)), "gelu_bwd")
        kw = dict(sgn=sgn, gtok=gtok, scale=scale, valid=valid) if with_sgn else {}
        g32, pair = cg.conv_dgrad_gelu(gh, gl, wth, wtl, geom, k, stride, pad, rows_in, pre, want_f32=True, want_pair=True, **kw)
        assert torch.equal(g32, r32), (with_sgn, (g32 - r32).abs().max().item())
        assert torch.equal(pair[0], rh) and torch.equal(pair[1], rl), with_sgn
        g32b, pairb = cg.conv_dgrad_gelu(gh, gl, wth, wtl, geom, k, stride, pad, rows_in, pre, want_f32=False, want_pair=True, **kw)
        assert g32b is None and torch.equal(pairb[0], rh) and torch.equal(pairb[1], rl)
        g32c, pairc = cg.conv_dgrad_gelu(gh, gl, wth, wtl, geom, k, stride, pad, rows_in, pre, want_f32=True, want_pair=False, **kw)
        assert pairc is None and torch.equal(g32c, r32)
        for i, (ns, H) in enumerate(geom):
            assert not g32[i, ns * H:].any() and not pair[0][i, ns * H:].any(), "rows below the slot's data must be zero"
        # the bias gradient's column sums from the same epilogue == gelu_bwd_split's (fp32 atomics in another order: 1e-5 of the scale)
        db_ref = torch.zeros(n, Cin, device=dev)
        _lib.check(lib.vmasr_gelu_bwd_split(pre.data_ptr(), t.data_ptr(), rh.data_ptr(), rl.data_ptr(), None, db_ref.data_ptr(), n, rows_in, Cin,
                                            _lib.current_stream(dev)), "gelu_bwd_split")
        db = torch.zeros(n, Cin, device=dev)
        _, paird = cg.conv_dgrad_gelu(gh, gl, wth, wtl, geom, k, stride, pad, rows_in, pre, want_f32=False, want_pair=True, db=db, **kw)
        assert torch.equal(paird[0], rh) and torch.equal(paird[1], rl)
        want = r32.double().sum(1)
        sc = want.abs().max().item()
        assert (db.double() - want).abs().max().item() <= 1e-5 * sc and (db_ref.double() - want).abs().max().item() <= 1e-5 * sc


F32_CASES = [
    # (Cin, Cout, k, stride, pad, geom): the 32 -> 128 layer's geometry (ragged sequences, several tiles, sequences shorter than the kernel)
    (32, 128, 5, 3, 2, [(5, 61), (3, 100)]),
    (32, 128, 5, 3, 2, [(9, 333), (4, 517), (2, 1031)]),
    (32, 128, 5, 3, 2, [(7, 3)]),
    (32, 256, 5, 3, 2, [(6, 200)]),
]


@gpu
@pytest.mark.parametrize("case", F32_CASES)
def test_conv_f32_fwd_dgrad_match_float64_at_fp32_accuracy(case):
    """The exact-f32 form (csrc/convgemm.hip OPS 1: fp32 operands, v_mfma_f32_32x32x2_f32) of the 32 -> 128 layer's forward and input
    gradient against float64 — held to what an fp32 evaluation of the same lines achieves (torch's own fp32 convolution x 4), two orders
    tighter than the bf16x3 form's 16-17 bits; epilogue (bias, GELU, bf16 pair, zero padding rows) as the pair form."""
    from vm_asr_amd import convgemm as cg
    Cin, Cout, k, stride, pad, geom = case
    dev = torch.device("cuda:0")
    g = torch.Generator(device="cpu").manual_seed(11)
    n = len(geom)
    xs = [torch.randn(ns, H, Cin, generator=g).to(dev) for ns, H in geom]
    W = (torch.randn(n, Cout, k * Cin, generator=g) / (k * Cin) ** 0.5).to(dev)
    bias = torch.randn(n, Cout, generator=g).to(dev)
    H1 = [cg.out_positions(H, k, stride, pad) for _, H in geom]
    Ms = [ns * h1 for (ns, _), h1 in zip(geom, H1)]
    rows_in = -(-max(ns * H for ns, H in geom) // 256) * 256 + 256
    rows_out = -(-max(Ms) // 256) * 256
    x = _stack([t.reshape(-1, Cin) for t in xs], rows_in)
    pre, y, yh, yl = cg.conv_fwd_f32(x, W, bias, geom, k, stride, pad, rows_out, act=True)
    xs64 = [t.double().requires_grad_(True) for t in xs]
    W64 = W.double().requires_grad_(True)
    pres64, ys64 = _ref(xs64, W64, bias.double(), k, stride, pad, True)
    pres32, _ = _ref(xs, W, bias, k, stride, pad, True)
    for i, M in enumerate(Ms):
        sc = pres64[i].abs().max().item()
        e = (pre[i, :M].double() - pres64[i]).abs().max().item()
        e32 = (pres32[i].double() - pres64[i]).abs().max().item()
        assert e <= 4 * e32 + 1e-7 * sc, (i, e, e32, sc)          # fp32 accuracy (measured: at or below torch's fp32 convolution)
        assert (y[i, :M].double() - ys64[i]).abs().max().item() <= 4 * e32 + 1e-6 * max(1.0, ys64[i].abs().max().item())
        pair = yh[i, :M].float() + yl[i, :M].float()
        assert (pair - y[i, :M]).abs().max().item() <= 2 ** -16 * y[i, :M].abs().max().item()
        for t in (pre, y, yh, yl):
            assert not t[i, M:].any(), "padding rows must be zero"
    gy = [torch.randn(M, Cout, generator=g).to(dev) for M in Ms]
    gst = _stack(gy, rows_out)
    sum((p * gg.double()).sum() for p, gg in zip(pres64, gy)).backward()
    xs32 = [t.clone().requires_grad_(True) for t in xs]
    p32, _ = _ref(xs32, W, bias, k, stride, pad, True)
    sum((p * gg).sum() for p, gg in zip(p32, gy)).backward()
    Wt = W.view(n, Cout, k, Cin).permute(0, 3, 2, 1).reshape(n, Cin, k * Cout).contiguous()
    dx = cg.conv_dgrad_f32(gst, Wt, geom, k, stride, pad, rows_in)
    for i, (ns, H) in enumerate(geom):
        want = xs64[i].grad.reshape(-1, Cin)
        sc = max(want.abs().max().item(), 1e-30)
        e32 = (xs32[i].grad.reshape(-1, Cin).double() - want).abs().max().item()
        assert (dx[i, :ns * H].double() - want).abs().max().item() <= 4 * e32 + 1e-7 * sc, (i, e32, sc)
        assert not dx[i, ns * H:].any(), "rows below the slot's data must be zero"
