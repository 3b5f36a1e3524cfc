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


def _model(hop, dims=16, n_fft=1024, win=1024, d_state=1):
    from synth import synth_state
    from vm_asr_amd.model import DualStreamInteractiveMambaUNet
    torch.manual_seed(123)
    m = DualStreamInteractiveMambaUNet(
        in_chans=1, patch_size=4, depths=[2, 2, 2, 2], dims=dims, ssm_d_state=d_state, ssm_ratio=2.0, ssm_dt_rank="auto",
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
    two correct fp32 evaluations differ there by a few 1e-4 of the peak (each sits 1.5e-4 .. 2.6e-4 from the float64
    answer, test_fullsize_forward_hip_fp64_adjudicated).  Bounds after it: 2e-3 (max) and 1e-4 (RMS) of the output
    peak, LSD within 1e-4."""
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
        assert d.max() <= 2e-3 * scale, (tag, what, d.max(), scale)
        assert np.sqrt((d.astype(np.float64) ** 2).mean()) <= 1e-4 * scale, (tag, what)
    lsd = oracle.lsd(y_gpu.numpy()[:, 0], z[f"{tag}_target"][:, 0])
    print(f"[{tag}] hip fp32 LSD {lsd:.6f} vs the reference's {float(z[f'{tag}_lsd']):.6f}: delta {abs(lsd - float(z[f'{tag}_lsd'])):.1e}")
    assert abs(lsd - float(z[f"{tag}_lsd"])) < 1e-4, (lsd, float(z[f"{tag}_lsd"]))      # LSD parity 1e-4


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
    # the CPU oracle is the OTHER sequential-fp32 evaluation (the checker, not the product): its own distance from float64 is a draw
    # of the same heavy-tailed statistic as the reference's (2.06x on the 16 kHz clip, 1.52x on the n_fft-2048 clip) — round-3 factors
    _adjudicate(tag, "cpu-oracle", y, y32, y64, target, lsd_ref, k_max=3.0, k_rms=2.0, lsd_tol=1e-4)


# The statistic |y - y64| is ONE DRAW of a heavy-tailed quantity: the network ends in a LayerNorm over two channels
# (amplification up to 1/(2 sqrt(eps)) = 158 where the channels nearly coincide), so a handful of spectrogram bins carry
# the whole error and ANY change of rounding anywhere upstream re-draws it.  Measured on MI355X in round 3
# (tools/accuracy_probe.py, profiles/r03_accuracy_probe.log, 6 clips x 4 configs): the ratio (HIP fp32) / (CPU-oracle
# fp32, i.e. the sequential recurrence of selective_scan_ref) ranges 0.47 .. 1.96 in RMS and 0.27 .. 3.9 at the worst
# sample from clip to clip, with a geometric mean of 0.84 .. 1.21 (RMS) — and on clip 0 of the 16 kHz config the CPU
# oracle's fp32 run itself is 2.06x further from float64 than the reference's fp32 run (4.89e-5 vs 2.37e-5), although
# both are "the sequential fp32 recurrence".  Computing ONE operator family in float64 moves the final error of the
# n_fft-2048 clip anywhere between 0.6x and 2.2x.  So the per-clip bounds below are relative to the NOISIER of the two
# sequential fp32 evaluations and cover that spread; the claim "as accurate as the reference's fp32 arithmetic" is the
# DISTRIBUTION test (test_fullsize_forward_hip_error_distribution: geometric mean of the ratio over 8 clips), and
# per operator tests/test_ss2d_fused.py (x1.5 of the sequential recurrence's distance from float64, benchmark shapes).
K_MAX, K_RMS = 2.0, 1.5     # round 4: the fp32 Linear layers accumulate in float64 now (csrc/linear.hip) — measured on the golden
                            # clips: HIP 0.33 .. 0.86x the noisier sequential-fp32 evaluation (worst sample), 0.38 .. 0.66x (RMS)
LSD_TOL = 2e-5              # BASELINE.md §3 asks 1e-4; measured 1e-7 .. 5.5e-6 (the reference's own fp32 run sits 0 .. 3.7e-6 from the
                            # float64 LSD): gated at 4x the worst measured value, five times tighter than the contract


def _adjudicate(tag, who, y, y32, y64, target, lsd_ref, y_cpu=None, k_max=None, k_rms=None, lsd_tol=None):
    """The exact output is y64 (the reference evaluated in float64, make_golden.py::gen_fullsize2).  The reference's
    own fp32 run sits e_ref = |y32 - y64| from it (1.5e-4 .. 6.7e-4 of the peak at the worst sample — above the
    1e-4 of north_star, because the last VSS block's LayerNorm over two channels amplifies rounding); ours must be
    AS CLOSE to the exact answer as the reference's fp32 run is, within the factors K_MAX (worst sample) and K_RMS
    (RMS): the factor, not a prose tolerance, is the claim."""
    import oracle
    e_ref, e = np.abs(y32.astype(np.float64) - y64), np.abs(y.astype(np.float64) - y64)
    peak = np.abs(y64).max()
    print(f"[{tag}] {who}: max |y-y64| {e.max() / peak:.2e} of peak (reference fp32: {e_ref.max() / peak:.2e}), "
          f"rms {_rms(e) / peak:.2e} (reference fp32: {_rms(e_ref) / peak:.2e})")
    ref_max, ref_rms = e_ref.max(), _rms(e_ref)
    if y_cpu is not None:            # the other sequential-fp32 evaluation of the same clip (CPU oracle)
        e_cpu = np.abs(y_cpu.astype(np.float64) - y64)
        print(f"[{tag}] cpu-oracle fp32: max {e_cpu.max() / peak:.2e} rms {_rms(e_cpu) / peak:.2e}")
        ref_max, ref_rms = max(ref_max, e_cpu.max()), max(ref_rms, _rms(e_cpu))
    k_max, k_rms = K_MAX if k_max is None else k_max, K_RMS if k_rms is None else k_rms
    assert e.max() <= k_max * ref_max, (tag, who, e.max(), ref_max)
    assert _rms(e) <= k_rms * ref_rms, (tag, who, _rms(e), ref_rms)
    lsd = oracle.lsd(y[:, 0], target.numpy()[:, 0])
    lsd64 = oracle.lsd(y64.astype(np.float32)[:, 0], target.numpy()[:, 0])
    print(f"[{tag}] {who}: LSD {lsd:.6f}  reference fp32 {lsd_ref:.6f} (delta {abs(lsd - lsd_ref):.1e})  float64 {lsd64:.6f} "
          f"(delta {abs(lsd - lsd64):.1e}; reference fp32 vs float64 {abs(lsd_ref - lsd64):.1e})")
    assert abs(lsd - lsd_ref) < (LSD_TOL if lsd_tol is None else lsd_tol), (tag, who, lsd, lsd_ref)


@pytest.mark.gpu
@pytest.mark.parametrize("tag", ["16k", "48k", "d32", "n2048"])
def test_fullsize_forward_hip_fp64_adjudicated(tag):
    """BASELINE configs[0], [1], [4] (dims 32; n_fft 2048) at full size through the HIP path in fp32, adjudicated by
    the float64 evaluation of the reference."""
    from oracle.torch_backend import oracle_stft_patch, use_oracle
    (dims, n_fft, win, hop), wave, target, hf, y32, y64, lsd_ref = _case(tag)
    m = _model(hop, dims, n_fft, win).to("cuda:0")
    with torch.no_grad():
        y = m(wave.cuda(), hf.cuda()).float().cpu().numpy()
    with oracle_stft_patch(), torch.no_grad():
        y_cpu = use_oracle(_model(hop, dims, n_fft, win))(wave, hf).float().numpy()
    _adjudicate(tag, "hip fp32", y, y32, y64, target, lsd_ref, y_cpu)


@pytest.mark.parametrize("tag", ["16k", "n2048"])
def test_f64ref_equals_float64_reference(tag):
    """Pins the SECOND adjudicator, tests/f64ref.py (this package's modules in float64 with every HIP-backed family
    through a plain-torch float64 restatement; runs on the GPU in seconds): it reproduces the reference's own float64
    evaluation to 1e-9 of the peak.  Used where no golden exists (fresh clips of the distribution test below)."""
    import f64ref
    (dims, n_fft, win, hop), wave, target, hf, y32, y64, lsd_ref = _case(tag)
    y = f64ref.forward64(_model(hop, dims, n_fft, win), wave, hf)
    assert y.dtype == np.float64 and np.abs(y - y64).max() <= 1e-9 * np.abs(y64).max()


# round 4: measured 0.50 / 0.58 / 0.55 / 0.67 (RMS) and 0.47 / 0.53 / 0.56 / 0.68 (worst sample) over 16 clips (profiles/r04_gpu_parity.log):
# the HIP fp32 forward is CLOSER to float64 than the sequential fp32 evaluation since its Linear layers accumulate in float64
N_CLIPS, GM_RMS, GM_MAX = int(os.environ.get("VMASR_TEST_CLIPS", "16")), 1.0, 1.0


@pytest.mark.gpu
@pytest.mark.parametrize("tag", ["16k", "48k", "d32", "n2048"])
def test_fullsize_forward_hip_error_distribution(tag):
    """"As accurate as the reference's fp32 arithmetic" as a statement about the DISTRIBUTION of the error, which is
    what can be measured (see the note above K_MAX): over N_CLIPS fresh full-size clips the geometric mean of
        |y_hip - y64| / |y_cpu-oracle-fp32 - y64|      (RMS and worst sample; y64 = tests/f64ref.py on the GPU)
    is <= GM_RMS / GM_MAX.  Measured in round 3 (profiles/r03_accuracy_probe.log, r03_gpu_parity.log): 0.84 / 1.05 /
    1.21 / 1.18 (16k / 48k / d32 / n2048) over one set of 6 clips, 1.46 / 1.38 / 1.31 / 1.06 over the 8 clips used here;
    pooled over the 14 clips 1.15 / 1.23 / 1.27 / 1.11 with a standard error of ~0.12 (a single clip's log-ratio has a
    spread of ~0.4).  EVERY HIP operator family alone is at 0.6 .. 1.2x torch's CPU fp32 evaluation of the same lines
    (tools/accuracy_probe.py --ops); what remains at network level (~1.2x) is within two standard errors of 1 and has
    one known contributor: hipBLASLt's fp32 GEMMs are 1.4 .. 1.7x further from float64 than MKL's for K >= 256
    (tools/linear_accuracy.py; accumulation order), and the Linear layers are the largest single-family term of the
    final error in three of the four configs (isolation runs in the same log).
    16 clips since the deep-stage core (csrc/ss2d_deep.hip) went in: with 8 the statistic's own scatter (+-15 %) reached the
    bound on one config; over 16 clips 1.41 / 1.17 / 1.51 / 1.19 (RMS; worst sample 1.62 / 1.16 / 1.63 / 1.22) with the deep
    core and 1.35 / 1.21 / 1.41 / 1.13 (1.56 / 1.21 / 1.44 / 1.07) with the unfused chain on the same clips — no difference
    beyond the standard error of ~0.1; at operator level the deep core's output is as far from float64 as the oracle chain's
    (ratio 1.00, tests/test_ss2d_deep.py)."""
    import f64ref
    from oracle.torch_backend import oracle_stft_patch, use_oracle
    (dims, n_fft, win, hop), wave0, target, hf, y32, y64, lsd_ref = _case(tag)
    m = _model(hop, dims, n_fft, win).to("cuda:0")
    m_cpu = use_oracle(_model(hop, dims, n_fft, win))
    assert np.abs(f64ref.forward64(m, wave0, hf) - y64).max() <= 1e-9 * np.abs(y64).max()   # the adjudicator, on this device
    lr, lm = [], []
    for s in range(N_CLIPS):
        wave = 0.1 * torch.randn(wave0.shape, generator=torch.Generator().manual_seed(9100 + s))
        y64s = f64ref.forward64(m, wave, hf)
        with torch.no_grad():
            y = m(wave.cuda(), hf.cuda()).float().cpu().numpy()
        with oracle_stft_patch(), torch.no_grad():
            yc = m_cpu(wave, hf).float().numpy()
        e, ec = np.abs(y - y64s), np.abs(yc - y64s)
        assert e.max() <= 3e-3 * np.abs(y64s).max()                                           # sanity, absolute
        lr.append(np.log(_rms(e) / _rms(ec)))
        lm.append(np.log(e.max() / ec.max()))
    gm_rms, gm_max = float(np.exp(np.mean(lr))), float(np.exp(np.mean(lm)))
    print(f"[{tag}] hip / cpu-oracle-fp32 error ratio over {N_CLIPS} clips: geometric mean rms {gm_rms:.2f} max {gm_max:.2f}; "
          f"single clips rms {np.exp(min(lr)):.2f} .. {np.exp(max(lr)):.2f}")
    assert gm_rms <= GM_RMS and gm_max <= GM_MAX, (tag, gm_rms, gm_max)


def _oracle64_forward(tag):
    import oracle
    from oracle.torch_backend import oracle_stft_patch, use_oracle
    (dims, n_fft, win, hop), wave, target, hf, y32, y64, lsd_ref = _case(tag)
    m = use_oracle(_model(hop, dims, n_fft, win), f64=True)
    with oracle.float64(), oracle_stft_patch(), torch.no_grad():
        y = m(wave.double(), hf)
    return y.numpy(), y64


@pytest.mark.parametrize("tag", ["16k", "48k", "d32", "n2048"])
def test_float64_oracle_equals_float64_reference(tag):
    """Pins the ADJUDICATOR: this package's modules in float64 on the float64 build of the oracle's C source
    (oracle.float64) reproduce the reference's own float64 evaluation (module.double() + selective_scan_ref in
    double, tests/golden/fullsize2.npz) to 1e-9 of the peak — two independent float64 evaluations of the same
    network agree, so either can stand for "the exact answer" (used for gradients, where no reference golden of
    a full-size backward exists: its Python-loop scan cannot be back-propagated at L = 262 144)."""
    y, y64 = _oracle64_forward(tag)
    assert y.dtype == np.float64 and np.abs(y - y64).max() <= 1e-9 * np.abs(y64).max()


@pytest.mark.gpu
@pytest.mark.parametrize("tag", ["48k", "d32"])
def test_fullsize_forward_hip_bf16_autocast(tag):
    """north_star: 1e-2 for bf16.  The generator under bf16 autocast exactly as the trainer runs it (scan, LayerNorm,
    STFT in fp32; Linear / conv GEMMs in bf16) against the float64 output.  What holds at 1e-2: the LSD (the parity
    metric of BASELINE.json).  The WAVEFORM error is ~3e-2 of the peak in RMS — a property of bf16 autocast on this
    architecture, not of the HIP path: torch's own CPU autocast of the same module on the fp32 oracle kernels
    (bf16 Linear/conv, everything else fp32 — the reference's autocast policy) lands at the same distance, and that
    is the bound: RMS error <= 1.5x the CPU-autocast run's, and <= 5e-2 of the peak absolutely."""
    import oracle
    from oracle.torch_backend import oracle_stft_patch, use_oracle
    (dims, n_fft, win, hop), wave, target, hf, y32, y64, lsd_ref = _case(tag)
    m = _model(hop, dims, n_fft, win).to("cuda:0")
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
        y = m(wave.cuda(), hf.cuda()).float().cpu().numpy()
    m_cpu = use_oracle(_model(hop, dims, n_fft, win))
    with oracle_stft_patch(), torch.no_grad(), torch.autocast("cpu", dtype=torch.bfloat16):
        y_cpu = m_cpu(wave, hf).float().numpy()
    peak = np.abs(y64).max()
    e, e_cpu = np.abs(y.astype(np.float64) - y64), np.abs(y_cpu.astype(np.float64) - y64)
    lsd = oracle.lsd(y[:, 0], target.numpy()[:, 0])
    print(f"[{tag}] bf16 autocast: hip rms {_rms(e) / peak:.2e} max {e.max() / peak:.2e} | torch-CPU autocast rms "
          f"{_rms(e_cpu) / peak:.2e} max {e_cpu.max() / peak:.2e} (of peak); LSD {lsd:.4f} vs {lsd_ref:.4f}")
    assert abs(lsd - lsd_ref) <= 1e-2, (tag, lsd, lsd_ref)
    assert _rms(e) <= 1.5 * _rms(e_cpu) and _rms(e) <= 5e-2 * peak, (tag, _rms(e), _rms(e_cpu), peak)


@pytest.mark.gpu
def test_fullsize_backward_hip_fp64_adjudicated():
    """Full-size GRADIENTS (BASELINE configs[1]/[2] shape: dims 16, 513x512 spectrogram, one 48 kHz clip): every
    parameter gradient of the HIP fp32 forward+backward (`SelectiveScanCore.backward` at L = 262 144 inside the real
    graph, model/vmamba.py:347-356), adjudicated by the float64 evaluation (pinned above):
        e_hip = |g_hip - g64|,  e_cpu = |g_cpu-oracle-fp32 - g64|   per tensor and for the whole gradient vector.
    Two fp32 evaluations differ from EACH OTHER by ~5e-3 here (the phase stream ends in exp(i phase) and a
    LayerNorm over two channels: rounding-level forward differences flip O(1) local gradients), so the claim is
    again relative: the HIP gradient is as close to the exact one as the fp32 CPU oracle's — whole vector within
    2x; per tensor (+ a floor of 1e-5 of the largest tensor norm for numerically-zero gradients) 95 % of the ~250
    tensors within 3x and every one within 10x.  (The per-tensor ratio is a ratio of two noise realisations for
    the tensors whose gradient is a cancelling sum over all positions — e.g. the last blocks' biases, where the
    fp32 oracle itself is off by 0.8 absolute — so its maximum moves by a few x whenever ANY upstream kernel
    changes its summation order; each kernel's own accuracy is pinned separately against float64 in
    tests/test_gpu_kernels.py / test_ss2d_fused.py at 1e-4.)"""
    import oracle
    from oracle.torch_backend import oracle_stft_patch, use_oracle
    (dims, n_fft, win, hop), wave, target, hf, y32, y64, lsd_ref = _case("48k")
    gy = torch.randn(1, 1, wave.shape[-1], generator=torch.Generator().manual_seed(77))
    m64 = use_oracle(_model(hop, dims, n_fft, win), f64=True)
    with oracle.float64(), oracle_stft_patch():
        (m64(wave.double(), hf) * gy.double()).sum().backward()
    m_cpu = use_oracle(_model(hop, dims, n_fft, win))
    with oracle_stft_patch():
        (m_cpu(wave, hf) * gy).sum().backward()
    m_gpu = _model(hop, dims, n_fft, win).to("cuda:0")
    (m_gpu(wave.cuda(), hf.cuda()).float() * gy.cuda()).sum().backward()
    torch.cuda.synchronize()
    g64 = {n: p.grad for n, p in m64.named_parameters()}
    ref = {n: p.grad for n, p in m_cpu.named_parameters()}
    got = {n: p.grad for n, p in m_gpu.named_parameters()}
    assert sum(g is None for g in g64.values()) == 129 and all((got[n] is None) == (g64[n] is None) == (ref[n] is None) for n in g64)
    names = [n for n in g64 if g64[n] is not None]
    assert all(torch.isfinite(got[n]).all() for n in names)
    floor = 1e-5 * max(g64[n].norm().item() for n in names)
    e_hip = {n: (got[n].cpu().double() - g64[n]).norm().item() for n in names}
    e_cpu = {n: (ref[n].double() - g64[n]).norm().item() for n in names}
    tot64 = sum(g64[n].pow(2).sum() for n in names).sqrt().item()
    t_hip, t_cpu = (sum(v ** 2 for v in e.values()) ** 0.5 / tot64 for e in (e_hip, e_cpu))
    ratio = {n: e_hip[n] / (e_cpu[n] + floor) for n in names}
    worst = sorted(ratio, key=ratio.get)[-3:]
    print(f"full-size backward vs float64: whole-vector rel L2  hip {t_hip:.2e}  cpu-oracle fp32 {t_cpu:.2e}; worst per-tensor "
          f"ratios " + ", ".join(f"{n} {ratio[n]:.2f}" for n in worst))
    assert t_hip <= 1.0 * t_cpu, (t_hip, t_cpu)          # round 4: measured 0.51x (was 1.39x)
    p95 = float(np.percentile(list(ratio.values()), 95))
    print(f"per-tensor ratio: median {float(np.median(list(ratio.values()))):.2f}, 95th percentile {p95:.2f}, max {ratio[worst[-1]]:.2f}")
    assert p95 <= 1.5, p95                               # measured 0.75
    assert ratio[worst[-1]] <= 3.0, (worst[-1], ratio[worst[-1]], e_hip[worst[-1]], e_cpu[worst[-1]])   # measured 1.24


@pytest.mark.gpu
def test_fullsize_backward_dims32_hip_fp64_adjudicated():
    """configs[4] (vm_asr_48k_16k_MPD_VSSM32.yaml: DIMS 32 -> d_inner 64 .. 512, dt_rank 2 .. 16, KD up to 2048) BACKWARD at full
    size through the HIP path in fp32: every parameter gradient against the float64 evaluation (tests/f64ref.py on the GPU,
    pinned to the reference's float64 run above) and, beside it, the CPU oracle's fp32 backward — the same relative claim as
    test_fullsize_backward_hip_fp64_adjudicated makes for DIMS 16."""
    import f64ref
    from oracle.torch_backend import oracle_stft_patch, use_oracle
    (dims, n_fft, win, hop), wave, target, hf, y32, y64, lsd_ref = _case("d32")
    assert dims == 32
    gy = torch.randn(1, 1, wave.shape[-1], generator=torch.Generator().manual_seed(78))
    m_gpu = _model(hop, dims, n_fft, win).to("cuda:0")
    m64 = f64ref.model64(m_gpu)
    with f64ref.Patch(f64ref.FAMILIES):
        (m64(wave.double().cuda(), hf.cuda()) * gy.double().cuda()).sum().backward()
    (m_gpu(wave.cuda(), hf.cuda()).float() * gy.cuda()).sum().backward()
    torch.cuda.synchronize()
    m_cpu = use_oracle(_model(hop, dims, n_fft, win))
    with oracle_stft_patch():
        (m_cpu(wave, hf) * gy).sum().backward()
    g64 = {n: p.grad for n, p in m64.named_parameters()}
    got = {n: p.grad for n, p in m_gpu.named_parameters()}
    ref = {n: p.grad for n, p in m_cpu.named_parameters()}
    assert all((got[n] is None) == (g64[n] is None) == (ref[n] is None) for n in g64)
    names = [n for n in g64 if g64[n] is not None]
    assert all(torch.isfinite(got[n]).all() for n in names)
    floor = 1e-5 * max(g64[n].norm().item() for n in names)
    e_hip = {n: (got[n].double() - g64[n]).norm().item() for n in names}
    e_cpu = {n: (ref[n].double().cuda() - g64[n]).norm().item() for n in names}
    tot64 = sum(g64[n].pow(2).sum() for n in names).sqrt().item()
    t_hip, t_cpu = (sum(v ** 2 for v in e.values()) ** 0.5 / tot64 for e in (e_hip, e_cpu))
    ratio = sorted(e_hip[n] / (e_cpu[n] + floor) for n in names)
    print(f"[d32] full-size backward vs float64: whole-vector rel L2  hip {t_hip:.2e}  cpu-oracle fp32 {t_cpu:.2e}; per-tensor ratio "
          f"median {ratio[len(ratio) // 2]:.2f}, 95th percentile {ratio[int(0.95 * len(ratio))]:.2f}, max {ratio[-1]:.2f}")
    assert t_hip <= 1.0 * t_cpu, (t_hip, t_cpu)          # round 4: measured 0.57x
    assert ratio[int(0.95 * len(ratio))] <= 1.5 and ratio[-1] <= 3.0, (ratio[int(0.95 * len(ratio))], ratio[-1])     # measured 0.67, 0.95


@pytest.mark.gpu
def test_dstate32_nfft2048_model_forward_hip_vs_oracle():
    """SURVEY.md 0.1 / BASELINE configs[4] as BASELINE.json words it ("d_state=32 n_fft=2048"): the reference's own override
    `--opts MODEL.VSSM.SSM_D_STATE 32 DATA.STFT.N_FFT 2048` on vm_asr_48k_16k_MPD_VSSM32.yaml (DIMS 32) — the long-sequence
    stress at MODEL level (L up to 524 288 with 32 states per row: the general-N scan path inside the real network), one
    full-size clip, fp32: HIP forward against the CPU oracle's forward of the same module and against float64."""
    import f64ref
    import oracle
    import vm_asr_amd
    from oracle.torch_backend import oracle_stft_patch, use_oracle
    from synth import synth_state
    from vm_asr_amd.config import get_config
    cfg = get_config(opts=["MODEL.NAME", "DualStreamInteractiveMambaUNet", "MODEL.VSSM.DIMS", 32, "MODEL.VSSM.SSM_D_STATE", 32,
                           "DATA.STFT.N_FFT", 2048, "DATA.TARGET_SR", 48000, "TRAIN.LOW_FREQ_REPLACEMENT", True])
    assert cfg.DATA.STFT.HOP_LENGTH == 240 and cfg.DATA.STFT.WIN_LENGTH == 1024     # config.py:313-320, :55 (win stays 1024)
    T = int(cfg.DATA.SEGMENT * cfg.DATA.TARGET_SR)
    wave = 0.1 * torch.randn(1, 1, T, generator=torch.Generator().manual_seed(4242))
    hf = torch.full((1,), int((2048 // 2 + 1) * 16000 / 48000), dtype=torch.int64)

    def build():
        torch.manual_seed(123)
        return synth_state(vm_asr_amd.get_model(cfg)["generator"]).eval()
    m = build().to("cuda:0")
    assert any(mod.d_state == 32 for mod in m.modules() if hasattr(mod, "d_state"))
    with torch.no_grad():
        y = m(wave.cuda(), hf.cuda()).float().cpu().numpy()
    y64 = f64ref.forward64(m, wave, hf)
    with oracle_stft_patch(), torch.no_grad():
        y_cpu = use_oracle(build())(wave, hf).float().numpy()
    peak = np.abs(y64).max()
    e, ec = np.abs(y - y64), np.abs(y_cpu - y64)
    print(f"[d_state 32, n_fft 2048, dims 32] hip fp32: max {e.max() / peak:.2e} rms {_rms(e) / peak:.2e} | cpu-oracle fp32: max "
          f"{ec.max() / peak:.2e} rms {_rms(ec) / peak:.2e} | hip vs cpu-oracle max {np.abs(y - y_cpu).max() / peak:.2e}")
    assert y.shape == (1, 1, T) and np.isfinite(y).all()
    assert e.max() <= K_MAX * ec.max() + 1e-4 * peak and _rms(e) <= K_RMS * _rms(ec) + 1e-5 * peak
    assert e.max() <= 5e-3 * peak and _rms(e) <= 2e-4 * peak


@pytest.mark.gpu
def test_dstate32_model_backward_hip_fp64_adjudicated():
    """BASELINE configs[4] as worded (`MODEL.VSSM.SSM_D_STATE 32`, config.py:100) at MODEL level, BACKWARD: the general-N scan
    (csrc/sscan_n.hip: cus/selective_scan_bwd_kernel.cuh:125-241 with its loop over 32 states) inside the real network — dims 32
    (d_inner 4 .. 512, dt_rank 1 .. 16), n_fft 256 so that one clip is 128 x 128 bins (L = 16 384 .. 16 for the scan calls).
    Every parameter gradient of the HIP fp32 path against the float64 evaluation of the same module (tests/f64ref.py on the GPU),
    with the CPU oracle's fp32 backward beside it: the HIP path must be as close to float64 as the sequential fp32 reference."""
    import f64ref
    from oracle.torch_backend import oracle_stft_patch, use_oracle
    n_fft, hop, dims = 256, 64, 32
    T = hop * 127
    g = torch.Generator().manual_seed(3232)
    wave = 0.1 * torch.randn(1, 1, T, generator=g)
    gy = torch.randn(1, 1, T, generator=g)
    hf = torch.full((1,), int((n_fft // 2 + 1) * 16000 / 48000), dtype=torch.int64)
    m_gpu = _model(hop, dims, n_fft, n_fft, d_state=32).to("cuda:0")
    assert all(mod.d_state == 32 for mod in m_gpu.modules() if hasattr(mod, "d_state"))
    m64 = f64ref.model64(m_gpu)
    with f64ref.Patch(f64ref.FAMILIES):
        (m64(wave.double().cuda(), hf.cuda()) * gy.double().cuda()).sum().backward()
    y = m_gpu(wave.cuda(), hf.cuda()).float()
    (y * gy.cuda()).sum().backward()
    torch.cuda.synchronize()
    m_cpu = use_oracle(_model(hop, dims, n_fft, n_fft, d_state=32))
    with oracle_stft_patch():
        (m_cpu(wave, hf) * gy).sum().backward()
    g64 = {n: p.grad for n, p in m64.named_parameters()}
    got = {n: p.grad for n, p in m_gpu.named_parameters()}
    ref = {n: p.grad for n, p in m_cpu.named_parameters()}
    assert all((got[n] is None) == (g64[n] is None) == (ref[n] is None) for n in g64)
    names = [n for n in g64 if g64[n] is not None]
    assert any("A_logs" in n for n in names) and all(torch.isfinite(got[n]).all() for n in names)
    floor = 1e-5 * max(g64[n].norm().item() for n in names)
    e_hip = {n: (got[n].double() - g64[n]).norm().item() for n in names}
    e_cpu = {n: (ref[n].double().cuda() - g64[n]).norm().item() for n in names}
    tot64 = sum(g64[n].pow(2).sum() for n in names).sqrt().item()
    t_hip, t_cpu = (sum(v ** 2 for v in e.values()) ** 0.5 / tot64 for e in (e_hip, e_cpu))
    ratio = sorted(e_hip[n] / (e_cpu[n] + floor) for n in names)
    print(f"[d_state 32, dims 32, 128x128] backward vs float64: whole-vector rel L2  hip {t_hip:.2e}  cpu-oracle fp32 {t_cpu:.2e}; "
          f"per-tensor ratio median {ratio[len(ratio) // 2]:.2f}, 95th percentile {ratio[int(0.95 * len(ratio))]:.2f}, max {ratio[-1]:.2f}")
    assert t_hip <= 3e-3 and t_hip <= 1.0 * t_cpu, (t_hip, t_cpu)          # measured 6.0e-4 vs 1.2e-3 (0.5x)
    assert ratio[int(0.95 * len(ratio))] <= 2.0 and ratio[-1] <= 4.0, (ratio[int(0.95 * len(ratio))], ratio[-1])
