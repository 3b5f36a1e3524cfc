"""Module-level parity against reference-generated goldens (tests/golden/ss2d.npz,
model_tiny.npz): SS2D, VSSBlock and the tiny DualStreamInteractiveMambaUNet are built by
vm_asr_amd, loaded with the reference's state_dict (strict=True => identical key names and
shapes) and must reproduce the reference's outputs and every gradient.

CPU variant: host logic with the oracle plugged into the operator hooks.
GPU variant (-m gpu): the shipped HIP path.  Tolerance 1e-4 fp32 (north_star)."""
import os

import numpy as np
import pytest
import torch

from oracle.torch_backend import oracle_stft_patch, use_oracle

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def _sd(z, prefix):
    return {k[len(prefix):]: torch.from_numpy(z[k]) for k in z.files if k.startswith(prefix)}


def _close(got, want, tol=1e-4, what=""):
    got = got.detach().float().cpu().numpy().astype(np.float64)
    want = np.asarray(want, np.float64)
    assert got.shape == want.shape, (what, got.shape, want.shape)
    lim = tol * max(1.0, float(np.abs(want).max()))
    err = np.abs(got - want) - tol * np.abs(want)
    if got.size:
        import errtable
        w = int(np.argmax(np.abs(got - want)))
        errtable.record(what, got, want, lim + tol * float(np.abs(want).flat[w]))
    assert err.max() <= lim, f"{what}: max|diff| {np.abs(got - want).max():.3e} > {lim:.3e}"


def _build(tag):
    from vm_asr_amd.vmamba import SS2D, VSSBlock
    if tag == "vssblock16":
        return VSSBlock(hidden_dim=16, drop_path=0.0, norm_layer=torch.nn.LayerNorm, channel_first=False,
                        ssm_d_state=1, ssm_ratio=2.0, ssm_dt_rank="auto", ssm_act_layer=torch.nn.SiLU,
                        ssm_conv=3, ssm_conv_bias=True, ssm_drop_rate=0.0, ssm_init="v0", forward_type="v5",
                        mlp_ratio=4.0, mlp_act_layer=torch.nn.GELU, mlp_drop_rate=0.0, gmlp=False)
    d_model, d_state = {"ss2d16": (16, 1), "ss2d8n4": (8, 4), "ss2d1": (1, 1)}[tag]
    return SS2D(d_model=d_model, d_state=d_state, ssm_ratio=2.0, dt_rank="auto", act_layer=torch.nn.SiLU,
                d_conv=3, conv_bias=True, dropout=0.0, initialize="v0", forward_type="v5", channel_first=False)


def _run_block(tag, device):
    z = np.load(os.path.join(GOLDEN, "ss2d.npz"))
    m = _build(tag)
    m.load_state_dict(_sd(z, f"{tag}_sd::"), strict=True)
    if device == "cpu":
        use_oracle(m)
    m = m.to(device)
    x = torch.from_numpy(z[f"{tag}_x"]).to(device).requires_grad_()
    y = m(x)
    _close(y, z[f"{tag}_y"], what=f"{tag} y")
    y.backward(torch.from_numpy(z[f"{tag}_g"]).to(device))
    _close(x.grad, z[f"{tag}_dx"], what=f"{tag} dx")
    grads = _sd(z, f"{tag}_grad::")
    assert grads
    for k, p in m.named_parameters():
        assert p.grad is not None, k
        _close(p.grad, grads[k].numpy(), what=f"{tag} grad {k}")


TAGS = ["ss2d16", "ss2d8n4", "ss2d1", "vssblock16"]


@pytest.mark.parametrize("tag", TAGS)
def test_ss2d_vssblock_cpu_oracle_backend(tag):
    _run_block(tag, "cpu")


@pytest.mark.gpu
@pytest.mark.parametrize("tag", TAGS)
def test_ss2d_vssblock_hip(tag):
    _run_block(tag, "cuda:0")


def _tiny_model(interact="dual"):
    from vm_asr_amd.model import DualStreamInteractiveMambaUNet
    return DualStreamInteractiveMambaUNet(
        in_chans=1, patch_size=4, depths=[2, 2, 2, 2], dims=8, ssm_d_state=1, ssm_ratio=2.0, ssm_dt_rank="auto",
        ssm_act_layer="silu", ssm_conv=3, ssm_conv_bias=True, ssm_drop_rate=0.0, ssm_init="v0", forward_type="v5",
        mlp_ratio=4.0, mlp_act_layer="gelu", mlp_drop_rate=0.0, gmlp=False, drop_path_rate=0.1, patch_norm=True,
        norm_layer="LN", patchembed_version="v2", downsample_version="v1", upsample_version="v1",
        output_version="v3", concat_skip=True, interact=interact, n_fft=128, hop_length=32, win_length=128,
        spectro_scale="log2", low_freq_replacement=True)


def _run_variant(device, variant):
    """Ablation forwards m2p / p2m / single (model/model.py:1229-1552) on model_tiny's weights and inputs:
    output, LSD and every parameter-gradient norm vs the reference (tests/golden/model_variants.npz)."""
    import oracle
    import vm_asr_amd.model as M
    z, v = np.load(os.path.join(GOLDEN, "model_tiny.npz")), np.load(os.path.join(GOLDEN, "model_variants.npz"))
    m = _tiny_model(variant)
    missing, unexpected = m.load_state_dict(_sd(z, "sd::"), strict=False)
    assert not missing and len(unexpected) == int(v[f"{variant}_n_unexpected"])   # same parameter set as the reference variant
    m.eval()
    if device == "cpu":
        use_oracle(m)
    m = m.to(device)
    wave, hf = torch.from_numpy(z["wave"]).to(device), torch.from_numpy(z["hf"]).to(device)
    mag_in, phase_in = torch.from_numpy(z["mag_in"]).to(device), torch.from_numpy(z["phase_in"]).to(device)
    saved = M.wav2spectro
    M.wav2spectro = lambda *a, **k: (mag_in.clone(), phase_in.clone())
    try:
        y = m(wave, hf)
    finally:
        M.wav2spectro = saved
    _close(y, v[f"{variant}_y"], what=f"{variant} wave out")
    y.backward(torch.from_numpy(z["gy"]).to(device))
    n_none = 0
    for k, p in m.named_parameters():
        key = f"{variant}_gradnorm::{k}"
        if p.grad is None:
            n_none += 1
            assert key not in v.files, k
        else:
            want = float(v[key])
            assert abs(p.grad.double().norm().item() - want) <= 2e-3 * max(want, 1e-3), (k, want)
    assert n_none == int(v[f"{variant}_n_unused"])
    lsd = oracle.lsd(y.detach().cpu().numpy()[:, 0], z["target"][:, 0])
    assert abs(lsd - float(v[f"{variant}_lsd"])) < 1e-4


@pytest.mark.parametrize("variant", ["m2p", "p2m", "single"])
def test_ablation_variants_cpu_oracle_backend(variant):
    with oracle_stft_patch():
        _run_variant("cpu", variant)


@pytest.mark.gpu
@pytest.mark.parametrize("variant", ["m2p", "p2m", "single"])
def test_ablation_variants_hip(variant):
    _run_variant("cuda:0", variant)


def _run_tiny(device):
    import oracle
    z = np.load(os.path.join(GOLDEN, "model_tiny.npz"))
    m = _tiny_model()
    m.load_state_dict(_sd(z, "sd::"), strict=True)  # same keys/shapes as the reference generator
    m.eval()
    if device == "cpu":
        use_oracle(m)
    m = m.to(device)
    wave, hf = torch.from_numpy(z["wave"]).to(device), torch.from_numpy(z["hf"]).to(device)
    # inject the reference's own spectrogram (frame-0 phase is +-pi by FFT rounding noise in any
    # implementation, the reference included; the STFT itself is pinned in the STFT tests)
    import vm_asr_amd.model as M
    mag_in, phase_in = torch.from_numpy(z["mag_in"]).to(device), torch.from_numpy(z["phase_in"]).to(device)
    saved = M.wav2spectro
    M.wav2spectro = lambda *a, **k: (mag_in, phase_in)
    try:
        y = m(wave, hf)
    finally:
        M.wav2spectro = saved
    _close(y, z["y"], what="wave out")
    y.backward(torch.from_numpy(z["gy"]).to(device))
    grads, samples, norms = _sd(z, "grad::"), _sd(z, "gradsample::"), _sd(z, "gradnorm::")
    n_none = nz = 0
    for k, p in m.named_parameters():
        if p.grad is None:
            n_none += 1
            assert k not in grads and k not in samples, k
        elif k in grads:
            # gradients cross 34 SS2D blocks + the iSTFT adjoint: fp32 re-association noise grows to
            # a few 1e-4 of the tensor scale (the reference's own kernel tests allow 1e-3..1e-2 on grads)
            _close(p.grad, grads[k].numpy(), tol=1e-3, what=f"grad {k}")
            nz += int(grads[k].abs().max() > 0)
        else:  # large tensors: 256 evenly spaced elements + L2 norm
            flat = p.grad.detach().flatten().cpu()
            idx = torch.linspace(0, flat.numel() - 1, 256).long()
            _close(flat[idx], samples[k].numpy(), tol=1e-3, what=f"grad sample {k}")
            assert abs(flat.double().norm().item() - float(norms[k])) <= 1e-3 * max(1.0, float(norms[k])), k
            nz += int(float(norms[k]) > 0)
    assert nz > 500  # gradients really reach the whole network (a dims=4 model would cut them)
    assert n_none == int(z["n_unused"]) == 129  # layers_decoder_phase.{1,2,3} never used (quirk 0.2-1)
    lsd = oracle.lsd(y.detach().cpu().numpy()[:, 0], z["target"][:, 0])
    assert abs(lsd - float(z["lsd"])) < 1e-4, (lsd, float(z["lsd"]))


def test_tiny_generator_cpu_oracle_backend():
    with oracle_stft_patch():
        _run_tiny("cpu")


@pytest.mark.gpu
def test_tiny_generator_hip():
    _run_tiny("cuda:0")


@pytest.mark.gpu
def test_tiny_generator_hip_default_training_gemms(monkeypatch):
    """tests/conftest.py forces the float64-accumulating Linear for the whole suite, so the fp32 parity tests never ran the
    product's DEFAULT training GEMMs (VMASR_LINEAR_F64ACC=auto: library GEMMs wherever a gradient is recorded).  Same golden, same
    gates (forward 1e-4, gradients 1e-3 of each tensor's scale), default mode."""
    monkeypatch.setenv("VMASR_LINEAR_F64ACC", "auto")
    _run_tiny("cuda:0")


def test_reference_parameter_count_and_keys():
    """dims=16 generator: 3,010,352 parameters in 714 tensors (README.md:8)."""
    from vm_asr_amd.model import DualStreamInteractiveMambaUNet
    m = DualStreamInteractiveMambaUNet(in_chans=1, patch_size=4, depths=[2, 2, 2, 2], dims=16, ssm_d_state=1,
                                       forward_type="v5", output_version="v3", concat_skip=True, n_fft=1024,
                                       hop_length=240, win_length=1024, low_freq_replacement=True)
    assert sum(p.numel() for p in m.parameters()) == 3010352
    assert len(m.state_dict()) == 714
    z = np.load(os.path.join(GOLDEN, "model_tiny.npz"))
    tiny = _tiny_model()
    assert sorted(tiny.state_dict().keys()) == sorted(k[4:] for k in z.files if k.startswith("sd::"))


def test_splitk_linear_matches_f_linear():
    """The split-K weight-gradient linear (vm_asr_amd/linear.py) is F.linear with a re-associated dW sum."""
    from vm_asr_amd.linear import _SplitKLinearFn, splitk_plan, weight_grad
    torch.manual_seed(0)
    assert splitk_plan(327040, 32, 5) >= 4 and splitk_plan(1024, 32, 5) < 4 and splitk_plan(65536, 1024, 5120) < 4
    gy, xx = torch.randn(10007, 8), torch.randn(10007, 3)
    assert torch.allclose(weight_grad(gy, xx, splits=7), gy.t() @ xx, rtol=1e-4, atol=1e-3)
    x = torch.randn(3, 5000, 6, requires_grad=True, dtype=torch.double)
    w = torch.randn(4, 6, requires_grad=True, dtype=torch.double)
    b = torch.randn(4, requires_grad=True, dtype=torch.double)
    _SplitKLinearFn.apply(x, w, b, torch.double, None, None).square().sum().backward()
    got = [t.grad.clone() for t in (x, w, b)]
    for t in (x, w, b):
        t.grad = None
    torch.nn.functional.linear(x, w, b).square().sum().backward()
    for g, t in zip(got, (x, w, b)):
        assert torch.allclose(g, t.grad, rtol=1e-10, atol=1e-8)


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_splitk_linear_gpu(dtype):
    """linear() on many rows x small weight takes the split-K path on the GPU: same y, dx, dW, db as F.linear."""
    from vm_asr_amd.linear import linear, splitk_plan
    torch.manual_seed(1)
    rows, in_f, out_f = 65536 + 40, 16, 64
    assert splitk_plan(rows, out_f, in_f) >= 4
    x = torch.randn(4, rows // 4, in_f, device="cuda", requires_grad=True)
    w = (0.2 * torch.randn(out_f, in_f, device="cuda")).requires_grad_()
    b = torch.randn(out_f, device="cuda", requires_grad=True)
    gy = torch.randn(4, rows // 4, out_f, device="cuda")
    with torch.autocast("cuda", dtype=torch.bfloat16, enabled=dtype == torch.bfloat16):
        y = linear(x, w, b)
        assert y.dtype == dtype
        y.backward(gy.to(dtype))
    got = [t.grad.clone() for t in (x, w, b)]
    for t in (x, w, b):
        t.grad = None
    ref = torch.nn.functional.linear(x.double(), w.double(), b.double())
    ref.backward(gy.double())
    tol = 2e-2 if dtype == torch.bfloat16 else 1e-4
    assert (y.double() - ref).abs().max() <= tol * ref.abs().max()
    for g, t in zip(got, (x, w, b)):
        assert g.dtype == torch.float32
        assert (g.double() - t.grad.double()).abs().max() <= tol * t.grad.abs().max(), (g - t.grad).abs().max()


@pytest.mark.gpu
def test_vssm_backbone_constructs_and_runs_on_hip():
    """north_star: "keeps the VSSM/SS2D nn.Module ... API surface" — the backbone class of model/vmamba.py:1847-1877 with
    the reference's constructor keywords builds, exposes the reference's parameter names per block, runs forward +
    backward on the HIP operators and equals the same module on the CPU with the oracle kernels plugged in."""
    from oracle.torch_backend import use_oracle
    from vm_asr_amd.vmamba import VSSM
    kw = dict(patch_size=4, in_chans=3, num_classes=10, depths=[1, 1, 1, 1], dims=[16, 32, 64, 128], ssm_d_state=1, ssm_ratio=2.0,
              ssm_dt_rank="auto", ssm_act_layer="silu", ssm_conv=3, ssm_conv_bias=True, ssm_drop_rate=0.0, ssm_init="v0",
              forward_type="v5", mlp_ratio=4.0, mlp_act_layer="gelu", mlp_drop_rate=0.0, gmlp=False, drop_path_rate=0.0,
              patch_norm=True, norm_layer="LN", downsample_version="v1", patchembed_version="v1", use_checkpoint=False)
    torch.manual_seed(5)
    ref = VSSM(**kw)
    keys = set(ref.state_dict())
    for leaf in ("x_proj_weight", "dt_projs_weight", "dt_projs_bias", "A_logs", "Ds", "in_proj.weight", "conv2d.weight", "conv2d.bias",
                 "out_norm.weight", "out_norm.bias", "out_proj.weight"):
        assert f"layers.0.0.0.op.{leaf}" in keys, leaf
    gpu = VSSM(**kw)
    gpu.load_state_dict(ref.state_dict(), strict=True)
    gpu = gpu.cuda()
    use_oracle(ref)
    x = torch.randn(2, 3, 128, 128)
    gy = torch.randn(2, 10)
    y_ref = ref(x)
    y_ref.backward(gy)
    y = gpu(x.cuda())
    y.backward(gy.cuda())
    assert y.shape == (2, 10)
    assert (y.cpu() - y_ref).abs().max() <= 1e-4 * max(1.0, y_ref.abs().max().item())
    for (n, p), (_, q) in zip(gpu.named_parameters(), ref.named_parameters()):
        assert p.grad is not None and (p.grad.cpu() - q.grad).abs().max() <= 5e-4 * max(1e-3, q.grad.abs().max().item()), n
