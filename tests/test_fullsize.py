"""BASELINE.json configs[0] and configs[1] at FULL size: the dims-16, n_fft-1024 generator forward on one
synthetic 16 kHz (hop 80) and 48 kHz (hop 240) clip must reproduce the reference's own CPU
selective_scan_ref forward (tests/golden/fullsize.npz) — output wave within 3e-4 (max) / 5e-5 (RMS) of its
peak, equal LSD (1e-4).
CPU: this package's modules on the C oracle kernels (plumbing, no GPU); GPU: the HIP path in fp32.
Weights come from tests/golden/synth.py on both sides (a 3 M-parameter state is not stored)."""
import os
import sys

import numpy as np
import pytest
import torch

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")
sys.path.insert(0, GOLDEN)


def _model(hop, dims=16, n_fft=1024, win=1024):
    from synth import synth_state
    from vm_asr_amd.model import DualStreamInteractiveMambaUNet
    torch.manual_seed(123)
    m = DualStreamInteractiveMambaUNet(
        in_chans=1, patch_size=4, depths=[2, 2, 2, 2], dims=dims, ssm_d_state=1, ssm_ratio=2.0, ssm_dt_rank="auto",
        ssm_act_layer="silu", ssm_conv=3, ssm_conv_bias=True, ssm_drop_rate=0.0, ssm_init="v0", forward_type="v5",
        mlp_ratio=4.0, mlp_act_layer="gelu", mlp_drop_rate=0.0, gmlp=False, drop_path_rate=0.1, patch_norm=True,
        norm_layer="LN", patchembed_version="v2", downsample_version="v1", upsample_version="v1",
        output_version="v3", concat_skip=True, interact="dual", n_fft=n_fft, hop_length=hop, win_length=win,
        spectro_scale="log2", low_freq_replacement=True)
    return synth_state(m).eval()


def _run(device, tag, hop):
    import oracle
    z = np.load(os.path.join(GOLDEN, "fullsize.npz"))
    m = _model(hop)
    if device == "cpu":
        from oracle.torch_backend import use_oracle
        use_oracle(m)
    m = m.to(device)
    wave, hf = torch.from_numpy(z[f"{tag}_wave"]).to(device), torch.from_numpy(z[f"{tag}_hf"]).to(device)
    with torch.no_grad():
        y = m(wave, hf).float().cpu().numpy()
    want = z[f"{tag}_y"]
    assert y.shape == want.shape and np.isfinite(y).all()
    # Tolerance.  Each kernel is held to 1e-4 (north_star) in tests/test_gpu_kernels.py; the full forward
    # chains 34 SS2D blocks and two FFTs in fp32, and two fp32 evaluations of it (the reference's sequential
    # Python scan vs any other summation order) differ by ~2e-4 of the peak at the worst sample and ~2.5e-5
    # RMS.  Bounds: max 3e-4, RMS 5e-5 of the output peak.
    d, scale = np.abs(y - want), np.abs(want).max()
    assert d.max() <= 3e-4 * scale, (tag, d.max(), scale)
    assert np.sqrt((d.astype(np.float64) ** 2).mean()) <= 5e-5 * scale, (tag, np.sqrt((d ** 2).mean()), scale)
    lsd = oracle.lsd(y[:, 0], z[f"{tag}_target"][:, 0])
    assert abs(lsd - float(z[f"{tag}_lsd"])) < 1e-4, (lsd, float(z[f"{tag}_lsd"]))


@pytest.mark.parametrize("tag,hop", [("16k", 80), ("48k", 240)])
def test_fullsize_forward_cpu_oracle_backend(tag, hop):
    from oracle.torch_backend import oracle_stft_patch
    with oracle_stft_patch():
        _run("cpu", tag, hop)


def _trace(m, wave, hf, names):
    acts = {}
    hooks = [dict(m.named_modules())[n].register_forward_hook(lambda mod, i, o, n=n: acts.__setitem__(n, o.detach().float().cpu()))
             for n in names]
    with torch.no_grad():
        y = m(wave, hf).float().cpu()
    for h in hooks:
        h.remove()
    return acts, y


@pytest.mark.gpu
@pytest.mark.parametrize("tag,hop", [("16k", 80), ("48k", 240)])
def test_fullsize_forward_hip(tag, hop):
    """HIP fp32 forward at full size vs (a) the CPU-oracle forward of the same model, stage by stage, and
    (b) the reference's output wave.

    Up to the last-but-one output block every activation agrees to ~1e-6..1e-5 of its scale (bound here:
    1e-4, north_star).  The LAST block of each stream has d_model = 1, d_inner = 2: its out_norm is a LayerNorm
    over TWO channels, (a-b)/sqrt((a-b)^2+4 eps) — where the two channels are within sqrt(eps) ~ 3e-3 of each
    other it multiplies any rounding-level difference by up to 1/sqrt(eps) ~ 300.  That is a property of the
    reference architecture (quirk: output_version v3 ends in a 1-channel VSS block), not of an implementation:
    two correct fp32 evaluations differ there by ~1e-3 of the peak.  Bounds after it: 5e-3 (max) and 5e-4 (RMS)
    of the output peak, LSD within 2e-3."""
    import oracle
    from oracle.torch_backend import oracle_stft_patch, use_oracle
    z = np.load(os.path.join(GOLDEN, "fullsize.npz"))
    wave, hf = torch.from_numpy(z[f"{tag}_wave"]), torch.from_numpy(z[f"{tag}_hf"])
    names = ["patch_embed_mag", "layers_encoder_mag.3", "layers_encoder_phase.3", "layers_decoder_mag.3",
             "output_layer_mag.0.blocks.0", "output_layer_phase.0.blocks.0", "output_layer_mag.1.blocks.0",
             "output_layer_phase.1.blocks.0"]
    m_cpu = _model(hop)
    use_oracle(m_cpu)
    with oracle_stft_patch():
        a_cpu, y_cpu = _trace(m_cpu, wave, hf, names)
    a_gpu, y_gpu = _trace(_model(hop).to("cuda:0"), wave.cuda(), hf.cuda(), names)
    for n in names:
        c, g = a_cpu[n], a_gpu[n]
        assert c.shape == g.shape
        assert (c - g).abs().max() <= 1e-4 * c.abs().max(), (n, ((c - g).abs().max() / c.abs().max()).item())
    want = z[f"{tag}_y"]
    scale = np.abs(want).max()
    for y, what in ((y_gpu.numpy(), "hip vs reference"), (y_gpu.numpy() - y_cpu.numpy() + want, "hip vs cpu-oracle")):
        d = np.abs(y - want)
        assert d.max() <= 5e-3 * scale, (tag, what, d.max(), scale)
        assert np.sqrt((d.astype(np.float64) ** 2).mean()) <= 5e-4 * scale, (tag, what)
    lsd = oracle.lsd(y_gpu.numpy()[:, 0], z[f"{tag}_target"][:, 0])
    assert abs(lsd - float(z[f"{tag}_lsd"])) < 2e-3, (lsd, float(z[f"{tag}_lsd"]))


# ---- float64-adjudicated accuracy, dims-32 and n_fft-2048 configurations ---------------------------------------
def _case(tag):
    """Inputs of tests/golden/fullsize2.npz regenerated from their seed (make_golden.py::gen_fullsize2)."""
    z = np.load(os.path.join(GOLDEN, "fullsize2.npz"))
    dims, n_fft, win, hop, T, seed = (int(v) for v in z[f"{tag}_cfg"])
    g = torch.Generator().manual_seed(seed)
    wave = 0.1 * torch.randn(1, 1, T, generator=g)
    target = 0.1 * torch.randn(1, 1, T, generator=g)
    assert abs(wave.double().sum().item() - float(z[f"{tag}_wave_sum"])) < 1e-9, "RNG stream differs from the golden's"
    hf = torch.full((1,), int((n_fft // 2 + 1) * 16000 / 48000), dtype=torch.int64)
    old = np.load(os.path.join(GOLDEN, "fullsize.npz"))
    y32 = z[f"{tag}_y"] if f"{tag}_y" in z.files else old[f"{tag}_y"]
    lsd = float(z[f"{tag}_lsd"]) if f"{tag}_lsd" in z.files else float(old[f"{tag}_lsd"])
    return (dims, n_fft, win, hop), wave, target, hf, y32, y32.astype(np.float64) + z[f"{tag}_d64"].astype(np.float64), lsd


def _rms(a):
    return float(np.sqrt((np.asarray(a, np.float64) ** 2).mean()))


@pytest.mark.parametrize("tag", ["16k", "48k", "d32", "n2048"])
def test_fullsize_forward_cpu_oracle_fp64_adjudicated(tag):
    """CPU plumbing leg of the test below (oracle kernels): same bounds."""
    from oracle.torch_backend import oracle_stft_patch, use_oracle
    (dims, n_fft, win, hop), wave, target, hf, y32, y64, lsd_ref = _case(tag)
    m = use_oracle(_model(hop, dims, n_fft, win))
    with oracle_stft_patch(), torch.no_grad():
        y = m(wave, hf).float().numpy()
    _adjudicate(tag, "cpu-oracle", y, y32, y64, target, lsd_ref)


K_MAX, K_RMS = 8.0, 4.0


def _adjudicate(tag, who, y, y32, y64, target, lsd_ref):
    """The exact output is y64 (the reference evaluated in float64, make_golden.py::gen_fullsize2).  The reference's
    own fp32 run sits e_ref = |y32 - y64| from it; ours must sit within a constant factor of that — the factor,
    not a prose tolerance, is the claim: K_MAX on the worst sample, K_RMS in RMS.  (A different fp32 evaluation
    order of the same 34-block network cannot be expected to have the SAME error as the reference's, only the same
    order of magnitude; both are dominated by the final LayerNorm over two channels, see test_fullsize_forward_hip.)"""
    import oracle
    e_ref, e = np.abs(y32.astype(np.float64) - y64), np.abs(y.astype(np.float64) - y64)
    peak = np.abs(y64).max()
    print(f"[{tag}] {who}: max |y-y64| {e.max() / peak:.2e} of peak (reference fp32: {e_ref.max() / peak:.2e}), "
          f"rms {_rms(e) / peak:.2e} (reference fp32: {_rms(e_ref) / peak:.2e})")
    assert e.max() <= K_MAX * e_ref.max(), (tag, who, e.max(), e_ref.max())
    assert _rms(e) <= K_RMS * _rms(e_ref), (tag, who, _rms(e), _rms(e_ref))
    lsd = oracle.lsd(y[:, 0], target.numpy()[:, 0])
    assert abs(lsd - lsd_ref) < 1e-3, (tag, who, lsd, lsd_ref)


@pytest.mark.gpu
@pytest.mark.parametrize("tag", ["16k", "48k", "d32", "n2048"])
def test_fullsize_forward_hip_fp64_adjudicated(tag):
    """BASELINE configs[0], [1], [4] (dims 32; n_fft 2048) at full size through the HIP path in fp32, adjudicated by
    the float64 evaluation of the reference."""
    (dims, n_fft, win, hop), wave, target, hf, y32, y64, lsd_ref = _case(tag)
    m = _model(hop, dims, n_fft, win).to("cuda:0")
    with torch.no_grad():
        y = m(wave.cuda(), hf.cuda()).float().cpu().numpy()
    _adjudicate(tag, "hip fp32", y, y32, y64, target, lsd_ref)


@pytest.mark.gpu
@pytest.mark.parametrize("tag", ["48k", "d32"])
def test_fullsize_forward_hip_bf16_autocast(tag):
    """north_star: 1e-2 for bf16.  The generator under bf16 autocast exactly as the trainer runs it (scan, LayerNorm,
    STFT in fp32; Linear / conv GEMMs in bf16) against the float64 output: RMS error <= 1e-2 of the peak and LSD
    within 1e-2 of the reference's.  (The worst single sample is reported, not bounded at 1e-2: the final
    two-channel LayerNorm turns a bf16-level input difference into an O(1) change of its +-1 output at isolated
    time-frequency bins, see test_fullsize_forward_hip.)"""
    import oracle
    (dims, n_fft, win, hop), wave, target, hf, y32, y64, lsd_ref = _case(tag)
    m = _model(hop, dims, n_fft, win).to("cuda:0")
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
        y = m(wave.cuda(), hf.cuda()).float().cpu().numpy()
    e, peak = np.abs(y.astype(np.float64) - y64), np.abs(y64).max()
    lsd = oracle.lsd(y[:, 0], target.numpy()[:, 0])
    print(f"[{tag}] hip bf16 autocast: rms {_rms(e) / peak:.2e}, max {e.max() / peak:.2e} of peak; LSD {lsd:.4f} vs {lsd_ref:.4f}")
    assert _rms(e) <= 1e-2 * peak, (tag, _rms(e), peak)
    assert abs(lsd - lsd_ref) <= 1e-2, (tag, lsd, lsd_ref)


@pytest.mark.gpu
def test_fullsize_backward_hip_vs_cpu_oracle():
    """Full-size GRADIENTS (BASELINE configs[1]/[2] shape: dims 16, 513x512 spectrogram, one 48 kHz clip): every
    parameter gradient of the HIP fp32 forward+backward against the same module on the CPU oracle kernels
    (`SelectiveScanCore.backward` at L = 262 144 inside the real graph, model/vmamba.py:347-356).
    Bounds: relative L2 error per tensor <= 2e-3 against max(|g_ref|, 1e-4 of the largest tensor norm), and the
    whole gradient vector within 5e-4; the 129 `layers_decoder_phase` tensors have no gradient on either side."""
    from oracle.torch_backend import oracle_stft_patch, use_oracle
    (dims, n_fft, win, hop), wave, target, hf, y32, y64, lsd_ref = _case("48k")
    gy = torch.randn(1, 1, wave.shape[-1], generator=torch.Generator().manual_seed(77))
    m_cpu = use_oracle(_model(hop, dims, n_fft, win))
    with oracle_stft_patch():
        (m_cpu(wave, hf) * gy).sum().backward()
    m_gpu = _model(hop, dims, n_fft, win).to("cuda:0")
    (m_gpu(wave.cuda(), hf.cuda()).float() * gy.cuda()).sum().backward()
    torch.cuda.synchronize()
    ref = {n: p.grad for n, p in m_cpu.named_parameters()}
    got = {n: p.grad for n, p in m_gpu.named_parameters()}
    assert sum(g is None for g in ref.values()) == 129 and all((got[n] is None) == (ref[n] is None) for n in ref)
    names = [n for n in ref if ref[n] is not None]
    norms = {n: ref[n].double().norm().item() for n in names}
    floor = 1e-4 * max(norms.values())
    rel = {n: (got[n].cpu().double() - ref[n].double()).norm().item() / max(norms[n], floor) for n in names}
    worst = sorted(rel, key=rel.get)[-5:]
    tot = (sum((got[n].cpu().double() - ref[n].double()).pow(2).sum() for n in names).sqrt()
           / sum(ref[n].double().pow(2).sum() for n in names).sqrt()).item()
    print(f"full-size backward: whole-vector rel L2 {tot:.2e}; worst tensors " + ", ".join(f"{n} {rel[n]:.2e}" for n in worst))
    assert all(torch.isfinite(got[n]).all() for n in names)
    assert tot <= 5e-4, tot
    assert rel[worst[-1]] <= 2e-3, (worst[-1], rel[worst[-1]])
