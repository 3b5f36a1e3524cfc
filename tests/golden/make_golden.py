"""Generate golden input/output vectors by running the *reference itself* on CPU.

Run in the build container only:   python tests/golden/make_golden.py
(needs /root/reference; see _refload.py).  Outputs: tests/golden/*.npz — data only
(inputs + expected outputs); no reference source is stored.

Fixture families:
  scan_*.npz      selective_scan_ref fwd + autograd bwd on the reference's own test
                  distribution (kernels/selective_scan/test_selective_scan.py:593-654)
  csm.npz         CrossScan / CrossMerge fwd+bwd           (model/vmamba.py:27-73)
  dwconv.npz      depthwise conv3x3 + bias + SiLU fwd+bwd  (model/vmamba.py:859-868,1543-1545)
  ss2d.npz        one SS2D + one VSSBlock, d_model 16, 16x16, fwd + all grads
  stft.npz        wav2spectro / spectro2wav                (utils/stft.py:22-115)
  model_tiny.npz  DualStreamInteractiveMambaUNet dims=8, n_fft=128 fwd + grads + LSD
  metric.npz      LSD / SNR / LSD-HF / LSD-LF on fixed pairs (model/metric.py)
  loss.npz        MultiResolutionSTFTLoss values + d/dx (model/loss.py:137-184)
  model_variants.npz  the m2p / p2m / single ablation forwards on model_tiny's weights: output, LSD, grad norms
  fullsize.npz    full-size generator forward (dims 16, n_fft 1024) at 16 kHz and 48 kHz: clip, output, LSD
  fullsize2.npz   float64 evaluation of the full-size forwards (the adjudicator for fp32 differences) and the
                  dims-32 / n_fft-2048 full-size forwards (configs/vm_asr_48k_16k_MPD_VSSM32.yaml, ..._nfft2048.yaml)
  trainstep.npz   one reference `_get_losses` evaluation (trainer/trainer.py:318-399) with the MPD in train mode
  mpd.npz         MultiPeriodDiscriminator hidden=2 scores / feature maps / LSGAN losses / grads
                  (model/discriminator.py:21-147, model/loss.py:188-235)
"""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from _refload import hann_in_double, load_reference, patch_ss2d_to_cpu, patch_ss2d_to_cpu64  # noqa: E402

torch.set_num_threads(8)


def _np(t):
    if t is None:
        return None
    t = t.detach()
    if t.dtype == torch.bfloat16 or t.dtype == torch.float16:
        t = t.float()
    return t.cpu().numpy().copy()   # a copy: later in-place updates (power iteration) must not leak into the fixture


def save(name, **arrs):
    arrs = {k: v for k, v in arrs.items() if v is not None}
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **arrs)
    print(f"  wrote {name}: {os.path.getsize(path) / 1024:.1f} KiB, {len(arrs)} arrays")


# --------------------------------------------------------------------------------------
def gen_scan(ns):
    """(B, KD, G, N, L) cases x flag combos; seed 0; reference distribution."""
    cases = [
        # tag,        B, KD, G, N,  L,   itype
        ("b2d24g2n1l64", 2, 24, 2, 1, 64, torch.float32),
        ("b2d24g4n1l65", 2, 24, 4, 1, 65, torch.float32),
        ("b1d8g4n1l4113", 1, 8, 4, 1, 4113, torch.float32),
        ("b2d16g4n8l300", 2, 16, 4, 8, 300, torch.float32),
        ("b1d8g4n32l520", 1, 8, 4, 32, 520, torch.float32),
        ("b2d8g1n1l1", 2, 8, 1, 1, 1, torch.float32),
        ("b1d4g4n1l1023", 1, 4, 4, 1, 1023, torch.float32),
        ("b2d24g2n1l256bf16", 2, 24, 2, 1, 256, torch.bfloat16),
        ("b2d24g2n1l256fp16", 2, 24, 2, 1, 256, torch.float16),
    ]
    for tag, Bn, KD, G, N, L, itype in cases:
        combos = [(True, True, True)]
        if itype == torch.float32 and L <= 300 and N == 1:
            combos = [(d, b, s) for d in (False, True) for b in (False, True) for s in (False, True)]
        elif itype == torch.float32 and L <= 300:
            combos = [(True, True, True), (False, False, False)]
        out = {}
        for has_D, has_bias, softplus in combos:
            torch.random.manual_seed(0)
            A = (-0.5 * torch.rand(KD, N)).requires_grad_()
            Bm = torch.randn(Bn, G, N, L).to(itype).requires_grad_()
            Cm = torch.randn(Bn, G, N, L).to(itype).requires_grad_()
            D = torch.randn(KD).requires_grad_() if has_D else None
            bias = (0.5 * torch.rand(KD)).requires_grad_() if has_bias else None
            u = torch.randn(Bn, KD, L).to(itype).requires_grad_()
            delta = (0.5 * torch.rand(Bn, KD, L)).to(itype).requires_grad_()
            y, last = ns.selective_scan_ref(u, delta, A, Bm, Cm, D, None, bias, softplus,
                                            return_last_state=True)
            g = torch.randn(y.shape).to(itype)
            y.backward(g)
            key = f"D{int(has_D)}b{int(has_bias)}s{int(softplus)}"
            if not out:
                out.update(u=_np(u), delta=_np(delta), A=_np(A), B=_np(Bm), C=_np(Cm), dout=_np(g))
                out["meta"] = np.array([Bn, KD, G, N, L, {torch.float32: 0, torch.float16: 1,
                                                          torch.bfloat16: 2}[itype]])
            # inputs are identical across combos only if RNG draws match; store per combo
            out[f"{key}_u"] = _np(u)
            out[f"{key}_delta"] = _np(delta)
            out[f"{key}_dout"] = _np(g)
            if has_D:
                out[f"{key}_D"] = _np(D)
                out[f"{key}_dD"] = _np(D.grad)
            if has_bias:
                out[f"{key}_bias"] = _np(bias)
                out[f"{key}_dbias"] = _np(bias.grad)
            out[f"{key}_out"] = _np(y)
            out[f"{key}_last"] = _np(last)
            out[f"{key}_du"] = _np(u.grad)
            out[f"{key}_ddelta"] = _np(delta.grad)
            out[f"{key}_dA"] = _np(A.grad)
            out[f"{key}_dB"] = _np(Bm.grad)
            out[f"{key}_dC"] = _np(Cm.grad)
        # drop redundant top-level copies of combo-varying inputs
        for k in ("u", "delta", "dout"):
            out.pop(k)
        save(f"scan_{tag}.npz", **out)


def gen_csm(ns):
    out = {}
    for tag, shape in (("a", (2, 3, 5, 7)), ("b", (1, 2, 64, 64)), ("c", (2, 4, 16, 48))):
        torch.manual_seed(1)
        x = torch.randn(*shape, requires_grad=True)
        xs = ns.vmamba.CrossScan.apply(x)
        g = torch.randn_like(xs)
        xs.backward(g)
        Bn, C, H, W = shape
        ys = torch.randn(Bn, 4, C, H, W, requires_grad=True)
        y = ns.vmamba.CrossMerge.apply(ys)
        gy = torch.randn_like(y)
        y.backward(gy)
        out.update({f"{tag}_x": _np(x), f"{tag}_xs": _np(xs), f"{tag}_gxs": _np(g),
                    f"{tag}_dx": _np(x.grad), f"{tag}_ys": _np(ys), f"{tag}_y": _np(y),
                    f"{tag}_gy": _np(gy), f"{tag}_dys": _np(ys.grad)})
    save("csm.npz", **out)


def gen_dwconv(ns):
    out = {}
    for tag, shape in (("a", (2, 4, 9, 11)), ("b", (1, 32, 16, 16)), ("c", (2, 2, 40, 24))):
        torch.manual_seed(2)
        Bn, C, H, W = shape
        conv = torch.nn.Conv2d(C, C, 3, padding=1, groups=C, bias=True)
        x = torch.randn(*shape, requires_grad=True)
        y = F.silu(conv(x))
        g = torch.randn_like(y)
        y.backward(g)
        out.update({f"{tag}_x": _np(x), f"{tag}_w": _np(conv.weight), f"{tag}_b": _np(conv.bias),
                    f"{tag}_y": _np(y), f"{tag}_g": _np(g), f"{tag}_dx": _np(x.grad),
                    f"{tag}_dw": _np(conv.weight.grad), f"{tag}_db": _np(conv.bias.grad)})
    save("dwconv.npz", **out)


def _module_fixture(prefix, module, x, out):
    module.zero_grad()
    x = x.clone().requires_grad_()
    y = module(x)
    torch.manual_seed(99)
    g = torch.randn_like(y)
    y.backward(g)
    out[f"{prefix}_x"] = _np(x)
    out[f"{prefix}_y"] = _np(y)
    out[f"{prefix}_g"] = _np(g)
    out[f"{prefix}_dx"] = _np(x.grad)
    for k, v in module.state_dict().items():
        out[f"{prefix}_sd::{k}"] = _np(v)
    for k, p in module.named_parameters():
        if p.grad is not None:
            out[f"{prefix}_grad::{k}"] = _np(p.grad)


def gen_ss2d(ns):
    out = {}
    torch.manual_seed(3)
    for tag, d_model, d_state, H, W in (("ss2d16", 16, 1, 16, 16), ("ss2d8n4", 8, 4, 8, 12),
                                        ("ss2d1", 1, 1, 32, 32)):
        m = ns.vmamba.SS2D(d_model=d_model, d_state=d_state, ssm_ratio=2.0, dt_rank="auto",
                           act_layer=torch.nn.SiLU, d_conv=3, conv_bias=True, dropout=0.0,
                           initialize="v0", forward_type="v5", channel_first=False)
        patch_ss2d_to_cpu(ns, m)
        # make Ds / A_logs / biases non-trivial so parity is meaningful
        with torch.no_grad():
            m.Ds.add_(0.1 * torch.randn_like(m.Ds))
            m.A_logs.add_(0.3 * torch.randn_like(m.A_logs))
            m.out_norm.weight.add_(0.1 * torch.randn_like(m.out_norm.weight))
            m.out_norm.bias.add_(0.1 * torch.randn_like(m.out_norm.bias))
        x = torch.randn(2, H, W, d_model)
        _module_fixture(tag, m, x, out)
    blk = ns.vmamba.VSSBlock(hidden_dim=16, drop_path=0.0, norm_layer=torch.nn.LayerNorm,
                             channel_first=False, ssm_d_state=1, ssm_ratio=2.0,
                             ssm_dt_rank="auto", ssm_act_layer=torch.nn.SiLU, ssm_conv=3,
                             ssm_conv_bias=True, ssm_drop_rate=0.0, ssm_init="v0",
                             forward_type="v5", mlp_ratio=4.0, mlp_act_layer=torch.nn.GELU,
                             mlp_drop_rate=0.0, gmlp=False)
    patch_ss2d_to_cpu(ns, blk)
    x = torch.randn(2, 16, 16, 16)
    _module_fixture("vssblock16", blk, x, out)
    save("ss2d.npz", **out)


def gen_stft(ns):
    out = {}
    torch.manual_seed(4)
    # (tag, T, n_fft, hop, win)
    for tag, T, n_fft, hop, win in (("s", 2400, 1024, 240, 1024), ("t", 2016, 128, 32, 128),
                                    ("w", 4800, 2048, 240, 1024), ("r", 1000, 256, 80, 256)):
        wav = 0.1 * torch.randn(2, 1, T)
        mag, ph = ns.stft.wav2spectro(wav, n_fft, hop, win, "log2")
        out.update({f"{tag}_wav": _np(wav), f"{tag}_mag": _np(mag), f"{tag}_phase": _np(ph),
                    f"{tag}_cfg": np.array([n_fft, hop, win])})
        # inverse with gradients wrt mag/phase (H3 needs backward)
        m2 = (mag + 0.05 * torch.randn_like(mag)).requires_grad_()
        p2 = (ph + 0.05 * torch.randn_like(ph)).requires_grad_()
        rec = ns.stft.spectro2wav(m2, p2, n_fft, hop, win, "log2")
        g = torch.randn_like(rec)
        rec.backward(g)
        out.update({f"{tag}_imag": _np(m2), f"{tag}_iphase": _np(p2), f"{tag}_rec": _np(rec),
                    f"{tag}_grec": _np(g), f"{tag}_dmag": _np(m2.grad), f"{tag}_dphase": _np(p2.grad)})
    # full-size clip: store checksums + 4 frames only
    wav = 0.1 * torch.randn(1, 1, 122640)
    mag, ph = ns.stft.wav2spectro(wav, 1024, 240, 1024, "log2")
    rec = ns.stft.spectro2wav(mag, ph, 1024, 240, 1024, "log2")
    fr = [0, 1, 255, 511]
    out.update(big_wav=_np(wav), big_mag_frames=_np(mag[..., fr]), big_phase_frames=_np(ph[..., fr]),
               big_frames=np.array(fr), big_mag_sum=np.array(mag.double().sum().item()),
               big_mag_abs=np.array(mag.double().abs().sum().item()),
               big_cos_sum=np.array(torch.cos(ph.double()).sum().item()),
               big_rec_err=np.array((rec - wav).abs().max().item()),
               big_shape=np.array(mag.shape))
    save("stft.npz", **out)


def gen_model(ns):
    out = {}
    torch.manual_seed(123)
    # dims >= 8: with dims=4 the second output PatchExpanding ends in LayerNorm(1), which outputs its
    # bias and cuts every gradient upstream (a degenerate model)
    kw = dict(in_chans=1, patch_size=4, depths=[2, 2, 2, 2], dims=8, ssm_d_state=1, ssm_ratio=2.0,
              ssm_dt_rank="auto", ssm_act_layer="silu", ssm_conv=3, ssm_conv_bias=True,
              ssm_drop_rate=0.0, ssm_init="v0", forward_type="v5", mlp_ratio=4.0,
              mlp_act_layer="gelu", mlp_drop_rate=0.0, gmlp=False, drop_path_rate=0.1,
              patch_norm=True, norm_layer="LN", patchembed_version="v2", downsample_version="v1",
              upsample_version="v1", output_version="v3", concat_skip=True, interact="dual",
              n_fft=128, hop_length=32, win_length=128, spectro_scale="log2",
              low_freq_replacement=True)
    m = ns.model.DualStreamInteractiveMambaUNet(**kw)
    patch_ss2d_to_cpu(ns, m)
    # de-symmetrise the two streams (deepcopy init makes them identical) and make the
    # zero-init / one-init tensors non-trivial so every op is exercised
    g = torch.Generator().manual_seed(7)
    with torch.no_grad():
        for n, p in m.named_parameters():
            p.add_(0.02 * torch.randn(p.shape, generator=g))
    m.eval()  # DropPath off (stochastic)
    T = 32 * 63
    wave = 0.1 * torch.randn(2, 1, T, generator=g)
    target = 0.1 * torch.randn(2, 1, T, generator=g)
    hf = torch.full((2,), int(65 * 16000 / 48000), dtype=torch.int64)
    # the spectrogram the reference fed to the network: frame 0 of a reflect-padded STFT is exactly
    # real, so its phase is +-pi by FFT rounding noise; parity tests of the network inject THIS
    # spectrogram and test the STFT separately (angles compared on the circle)
    mag_in, phase_in = m._mag_phase(wave)
    out.update(mag_in=_np(mag_in), phase_in=_np(phase_in))
    y = m(wave, hf)
    gy = torch.randn(y.shape, generator=g)
    y.backward(gy)
    out.update(wave=_np(wave), target=_np(target), hf=_np(hf), y=_np(y), gy=_np(gy),
               lsd=np.array(ns.metric.lsd(y.detach().squeeze(1), target.squeeze(1))),
               snr=np.array(ns.metric.snr(y.detach().squeeze(1), target.squeeze(1))))
    n_unused = 0
    for k, v in m.state_dict().items():
        out[f"sd::{k}"] = _np(v)
    for k, p in m.named_parameters():
        if p.grad is None:
            n_unused += 1
        elif p.numel() <= 2048:
            out[f"grad::{k}"] = _np(p.grad)
        else:  # large tensors: 256 evenly spaced elements + the L2 norm keep the fixture small
            flat = p.grad.flatten()
            idx = torch.linspace(0, flat.numel() - 1, 256).long()
            out[f"gradsample::{k}"] = _np(flat[idx])
            out[f"gradnorm::{k}"] = np.array(flat.double().norm().item())
    out["n_unused"] = np.array(n_unused)
    print(f"  tiny model: {sum(p.numel() for p in m.parameters())} params, "
          f"{n_unused} tensors without grad")
    save("model_tiny.npz", **out)


def gen_variants(ns):
    """The ablation forwards (configs/vm_asr_48k_MPD_{M2P,P2M,SINGLE}.yaml -> interact = m2p / p2m / single,
    model/model.py:1229-1552) on the weights, inputs and injected spectrogram of model_tiny.npz: output,
    LSD and the L2 norm of every parameter gradient (small fixture: no second copy of the weights)."""
    z = np.load(os.path.join(HERE, "model_tiny.npz"))
    kw = dict(in_chans=1, patch_size=4, depths=[2, 2, 2, 2], dims=8, ssm_d_state=1, ssm_ratio=2.0,
              ssm_dt_rank="auto", ssm_act_layer="silu", ssm_conv=3, ssm_conv_bias=True,
              ssm_drop_rate=0.0, ssm_init="v0", forward_type="v5", mlp_ratio=4.0,
              mlp_act_layer="gelu", mlp_drop_rate=0.0, gmlp=False, drop_path_rate=0.1,
              patch_norm=True, norm_layer="LN", patchembed_version="v2", downsample_version="v1",
              upsample_version="v1", output_version="v3", concat_skip=True,
              n_fft=128, hop_length=32, win_length=128, spectro_scale="log2", low_freq_replacement=True)
    sd = {k[4:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("sd::")}
    wave, target, hf, gy = (torch.from_numpy(z[k]) for k in ("wave", "target", "hf", "gy"))
    mag_in, phase_in = torch.from_numpy(z["mag_in"]), torch.from_numpy(z["phase_in"])
    out = {}
    for variant in ("m2p", "p2m", "single"):
        torch.manual_seed(123)
        m = ns.model.DualStreamInteractiveMambaUNet(interact=variant, **kw)
        patch_ss2d_to_cpu(ns, m)
        missing, unexpected = m.load_state_dict(sd, strict=False)
        assert not missing, missing
        m.eval()
        m._mag_phase = lambda x, _a=mag_in, _b=phase_in: (_a.clone(), _b.clone())   # the spectrogram of model_tiny.npz
        y = m(wave, hf)
        y.backward(gy)
        out[f"{variant}_y"] = _np(y)
        out[f"{variant}_lsd"] = np.array(ns.metric.lsd(y.detach().squeeze(1), target.squeeze(1)))
        out[f"{variant}_n_unexpected"] = np.array(len(unexpected))
        n_unused = 0
        for k, p in m.named_parameters():
            if p.grad is None:
                n_unused += 1
            else:
                out[f"{variant}_gradnorm::{k}"] = np.array(p.grad.double().norm().item())
        out[f"{variant}_n_unused"] = np.array(n_unused)
        print(f"  {variant}: {n_unused} tensors without grad, {len(unexpected)} unexpected keys")
    save("model_variants.npz", **out)


def gen_fullsize(ns):
    """BASELINE.json configs[0] / configs[1]: the full-size generator (dims 16, n_fft 1024) forward on one
    synthetic clip, 16 kHz (hop 80, T 40 880) and 48 kHz (hop 240, T 122 640), through the reference's CPU
    selective_scan_ref path; weights from synth.synth_state (not stored), phase indeterminacies pinned
    (synth.canonical_phase).  Stored: the clip, the output wave and its LSD against a second clip."""
    from synth import canonical_phase, synth_state
    out = {}
    for tag, hop, T in (("16k", 80, 40880), ("48k", 240, 122640)):
        torch.manual_seed(123)
        m = ns.model.DualStreamInteractiveMambaUNet(
            in_chans=1, patch_size=4, depths=[2, 2, 2, 2], dims=16, ssm_d_state=1, ssm_ratio=2.0,
            ssm_dt_rank="auto", ssm_act_layer="silu", ssm_conv=3, ssm_conv_bias=True, ssm_drop_rate=0.0,
            ssm_init="v0", forward_type="v5", mlp_ratio=4.0, mlp_act_layer="gelu", mlp_drop_rate=0.0, gmlp=False,
            drop_path_rate=0.1, patch_norm=True, norm_layer="LN", patchembed_version="v2", downsample_version="v1",
            upsample_version="v1", output_version="v3", concat_skip=True, interact="dual",
            n_fft=1024, hop_length=hop, win_length=1024, spectro_scale="log2", low_freq_replacement=True)
        patch_ss2d_to_cpu(ns, m)
        synth_state(m)
        m.eval()
        g = torch.Generator().manual_seed(31)
        wave = 0.1 * torch.randn(1, 1, T, generator=g)
        target = 0.1 * torch.randn(1, 1, T, generator=g)
        hf = torch.full((1,), 171, dtype=torch.int64)
        orig = m._mag_phase

        def pinned(x, _f=orig):
            mag, phase = _f(x)
            return mag, canonical_phase(phase)
        m._mag_phase = pinned
        import time
        t0 = time.time()
        with torch.no_grad():
            y = m(wave, hf)
        print(f"  {tag}: reference forward {time.time() - t0:.1f} s, |y|max {y.abs().max().item():.4f}")
        out.update({f"{tag}_wave": _np(wave), f"{tag}_target": _np(target), f"{tag}_hf": _np(hf), f"{tag}_y": _np(y),
                    f"{tag}_lsd": np.array(ns.metric.lsd(y.squeeze(1), target.squeeze(1)))})
    save("fullsize.npz", **out)


FULLSIZE_CASES = {
    # tag: (dims, n_fft, win, hop, T, input seed)   — 16k / 48k reuse the inputs of fullsize.npz (seed 31)
    "16k": (16, 1024, 1024, 80, 40880, 31),
    "48k": (16, 1024, 1024, 240, 122640, 31),
    "d32": (32, 1024, 1024, 240, 122640, 32),      # configs/vm_asr_48k_16k_MPD_VSSM32.yaml:4-5
    "n2048": (16, 2048, 1024, 240, 122640, 33),    # configs/vm_asr_48k_16k_nfft2048.yaml:17-19 (WIN_LENGTH stays 1024)
}


def _fullsize_model(ns, dims, n_fft, win, hop):
    torch.manual_seed(123)
    return ns.model.DualStreamInteractiveMambaUNet(
        in_chans=1, patch_size=4, depths=[2, 2, 2, 2], dims=dims, ssm_d_state=1, ssm_ratio=2.0,
        ssm_dt_rank="auto", ssm_act_layer="silu", ssm_conv=3, ssm_conv_bias=True, ssm_drop_rate=0.0,
        ssm_init="v0", forward_type="v5", mlp_ratio=4.0, mlp_act_layer="gelu", mlp_drop_rate=0.0, gmlp=False,
        drop_path_rate=0.1, patch_norm=True, norm_layer="LN", patchembed_version="v2", downsample_version="v1",
        upsample_version="v1", output_version="v3", concat_skip=True, interact="dual",
        n_fft=n_fft, hop_length=hop, win_length=win, spectro_scale="log2", low_freq_replacement=True)


def gen_fullsize2(ns):
    """(1) float64 evaluation of the reference's full-size forward for every case (module.double(),
    selective_scan_ref with its `.float()` casts turned into `.double()`, float64 hann window): stored as
    d64 = y64 - y32 in fp32, so tests can ask "is the HIP result as close to the exact answer as the reference's
    own fp32 run".  (2) the dims-32 and n_fft-2048 full-size fp32 forwards + LSD (BASELINE configs[4]).
    Inputs are regenerated from their seed in the tests; a checksum of the clip is stored."""
    from synth import canonical_phase, synth_state
    import time
    old = np.load(os.path.join(HERE, "fullsize.npz"))
    out = {}
    for tag, (dims, n_fft, win, hop, T, seed) in FULLSIZE_CASES.items():
        g = torch.Generator().manual_seed(seed)
        wave = 0.1 * torch.randn(1, 1, T, generator=g)
        target = 0.1 * torch.randn(1, 1, T, generator=g)
        hf = torch.full((1,), int((n_fft // 2 + 1) * 16000 / 48000), dtype=torch.int64)
        if f"{tag}_y" in old.files:
            assert np.array_equal(old[f"{tag}_wave"], wave.numpy())
            y32 = torch.from_numpy(old[f"{tag}_y"])
        else:
            m = synth_state(patch_ss2d_to_cpu(ns, _fullsize_model(ns, dims, n_fft, win, hop))).eval()
            f32 = m._mag_phase
            m._mag_phase = lambda x, _f=f32: (lambda mp: (mp[0], canonical_phase(mp[1])))(_f(x))
            t0 = time.time()
            with torch.no_grad():
                y32 = m(wave, hf)
            print(f"  {tag}: reference fp32 forward {time.time() - t0:.1f} s, |y|max {y32.abs().max().item():.4f}")
            out[f"{tag}_y"] = _np(y32)
            out[f"{tag}_lsd"] = np.array(ns.metric.lsd(y32.squeeze(1), target.squeeze(1)))
        # float64: weights are the SAME fp32 values (synth_state before .double())
        m = synth_state(_fullsize_model(ns, dims, n_fft, win, hop))
        patch_ss2d_to_cpu64(ns, m).eval()
        f64 = m._mag_phase
        m._mag_phase = lambda x, _f=f64: (lambda mp: (mp[0], canonical_phase(mp[1])))(_f(x))
        t0 = time.time()
        with torch.no_grad(), hann_in_double():
            y64 = m(wave.double(), hf)
        assert y64.dtype == torch.float64
        d = y64 - y32.double()
        print(f"  {tag}: reference fp64 forward {time.time() - t0:.1f} s; |y32 - y64| max {d.abs().max().item():.3e} "
              f"rms {d.pow(2).mean().sqrt().item():.3e} (peak {y64.abs().max().item():.4f})")
        out[f"{tag}_d64"] = d.float().numpy()
        out[f"{tag}_wave_sum"] = np.array(wave.double().sum().item())
        out[f"{tag}_cfg"] = np.array([dims, n_fft, win, hop, T, seed])
    save("fullsize2.npz", **out)


def gen_loss(ns):
    """MultiResolutionSTFTLoss (model/loss.py:137-184) values and d/dx on a fixed pair."""
    g = torch.Generator().manual_seed(6)
    x = (0.1 * torch.randn(2, 6000, generator=g)).requires_grad_()
    y = 0.1 * torch.randn(2, 6000, generator=g)
    out = {}
    for tag, emph in (("plain", False), ("emph", True)):
        L = ns.loss.MultiResolutionSTFTLoss(factor_sc=0.5, factor_mag=0.5, emphasize_high_freq=emph)
        x.grad = None
        sc, mag = L(x, y)
        (sc + mag).backward()
        out.update({f"{tag}_sc": np.array(sc.item()), f"{tag}_mag": np.array(mag.item()), f"{tag}_dx": _np(x.grad)})
    out.update(x=_np(x), y=_np(y))
    save("loss.npz", **out)


def gen_mpd(ns):
    """MultiPeriodDiscriminator (model/discriminator.py:21-147) hidden=2 on a 2 x 1 x 1 201 pair (not a
    multiple of any period -> reflect pad), with the HiFi-GAN LSGAN losses (model/loss.py:188-235).
    eval mode: spectral norm uses the stored u, v (no power iteration): scores, all 30 feature maps,
    the three losses and d(gen+feat)/dy_hat, d(disc)/d(two weights).  train mode: forward(y, y_hat) from
    the same state = one power iteration per call (real, then fake), scores only."""
    torch.manual_seed(11)
    D = ns.discriminator.MultiPeriodDiscriminator(hidden=2)
    g = torch.Generator().manual_seed(12)
    y = 0.3 * torch.randn(2, 1, 1201, generator=g)
    y_hat = (0.3 * torch.randn(2, 1, 1201, generator=g)).requires_grad_()
    out = {f"sd::{k}": _np(v) for k, v in D.state_dict().items()}
    out.update(y=_np(y), y_hat=_np(y_hat))
    L = ns.loss.HiFiGANLoss("lsgan")
    D.eval()
    rs, gs, fr, fg = D(y, y_hat)
    for i, (a, b) in enumerate(zip(rs, gs)):
        out[f"eval_real{i}"], out[f"eval_gen{i}"] = _np(a), _np(b)
    for i, (a, b) in enumerate(zip(fr, fg)):
        for j, (u, v) in enumerate(zip(a, b)):
            out[f"eval_fmap_real{i}_{j}"], out[f"eval_fmap_gen{i}_{j}"] = _np(u), _np(v)
    d_loss = L.discriminator_loss(rs, gs)
    g_loss = L.generator_loss(gs)
    f_loss = L.feature_loss(fr, fg)
    out.update(d_loss=np.array(d_loss.item()), g_loss=np.array(g_loss.item()), f_loss=np.array(f_loss.item()))
    (g_loss + f_loss).backward(retain_graph=True)
    out["d_gf_dyhat"] = _np(y_hat.grad)
    names = ["discriminators.0.layers.0.parametrizations.weight.original", "discriminators.4.layers.3.parametrizations.weight.original",
             "discriminators.2.conv_post.bias"]
    params = dict(D.named_parameters())
    for p in params.values():
        p.grad = None
    d_loss.backward()
    for n in names:
        out[f"d_disc::{n}"] = _np(params[n].grad)
    D.train()
    rs, gs, _, _ = D(y, y_hat.detach())
    for i, (a, b) in enumerate(zip(rs, gs)):
        out[f"train_real{i}"], out[f"train_gen{i}"] = _np(a), _np(b)
    save("mpd.npz", **out)


def _reference_get_losses():
    """`Trainer._get_losses`, `_get_mpd_loss`, `_get_stft_loss` taken from trainer/trainer.py by AST (the module
    itself imports tensorboard / tqdm / the data pipeline) and bound to a bare object."""
    import ast
    import types
    from _refload import REF
    src = open(os.path.join(REF, "trainer/trainer.py")).read()
    cls = [n for n in ast.parse(src).body if isinstance(n, ast.ClassDef) and n.name == "Trainer"][0]
    fns = [n for n in cls.body if isinstance(n, ast.FunctionDef) and n.name in ("_get_losses", "_get_mpd_loss", "_get_stft_loss")]
    assert len(fns) == 3
    return fns


def gen_trainstep(ns):
    """One evaluation of the reference's loss function for a train step, trainer/trainer.py:318-399
    (`_get_losses` -> `_get_stft_loss`, `_get_mpd_loss`), MPD (hidden 2) in TRAIN mode: two discriminator
    calls = four spectral-norm power iterations per weight (real, fake, real, fake).  `wave_out` is a leaf
    that requires grad, so d(total generator loss)/d(wave_out) is exactly what flows back into the generator.
    Two states: "cold" (freshly initialised u, v — the sigma of each of the four passes differs) and "warm"
    (the same weights after 2000 train-mode power iterations: u, v converged, the four sigmas agree to rounding).
    Stored per state: MPD state_dict before, the five losses, d total_g / d wave_out, the gradient of the
    discriminator loss wrt three weights, and u/v after the step."""
    import ast
    import types
    fns = _reference_get_losses()
    g = dict(torch=torch, mae_loss=ns.loss.mae_loss, mse_loss=ns.loss.mse_loss)
    exec(compile(ast.Module(body=fns, type_ignores=[]), "reference_trainer_losses", "exec"), g)
    NS = types.SimpleNamespace
    cfg = NS(TRAIN=NS(LOSSES=NS(GEN=["multi_resolution_stft"]),
                      ADVERSARIAL=NS(DISCRIMINATORS=["mpd"], ONLY_FEATURE_LOSS=False, ONLY_ADVERSARIAL_LOSS=False,
                                     FEATURE_LOSS_LAMBDA=100, GAN_LOSS_TYPE="lsgan")))
    gen = torch.Generator().manual_seed(21)
    T = 6000
    wave_target = 0.1 * torch.randn(2, 1, T, generator=gen)
    wave_out0 = wave_target + 0.03 * torch.randn(2, 1, T, generator=gen)
    out = dict(wave_target=_np(wave_target), wave_out=_np(wave_out0))
    names = ["discriminators.0.layers.0.parametrizations.weight.original",
             "discriminators.4.layers.3.parametrizations.weight.original", "discriminators.2.conv_post.bias"]
    for state in ("cold", "warm"):
        torch.manual_seed(11)
        D = ns.discriminator.MultiPeriodDiscriminator(hidden=2)
        D.train()
        if state == "warm":
            with torch.no_grad():
                for _ in range(1000):   # 1000 calls x (real, fake) = 2000 power iterations per weight
                    D(wave_target, wave_out0)
        for k, v in D.state_dict().items():
            out[f"{state}_sd::{k}"] = _np(v)
        me = NS(config=cfg, gan=True, models={"mpd": D},
                multi_resolution_stft=ns.loss.MultiResolutionSTFTLoss(factor_sc=0.5, factor_mag=0.5, emphasize_high_freq=False),
                higi_gan_loss=ns.loss.HiFiGANLoss("lsgan"))
        for f in ("_get_losses", "_get_mpd_loss", "_get_stft_loss"):
            setattr(me, f, types.MethodType(g[f], me))
        wave_out = wave_out0.clone().requires_grad_()
        losses = me._get_losses(wave_out, wave_target)
        for k, v in losses["generator"].items():
            out[f"{state}_g::{k}"] = np.array(v.item())
        for k, v in losses["discriminator"].items():
            out[f"{state}_d::{k}"] = np.array(v.item())
        total_g = sum(losses["generator"].values())
        total_d = sum(losses["discriminator"].values())
        params = dict(D.named_parameters())
        total_g.backward(retain_graph=True)          # reference order: _optimize(G) then _optimize_adversarial(D)
        out[f"{state}_dwave"] = _np(wave_out.grad)
        for p in params.values():
            p.grad = None                            # optimizer_D.zero_grad() (trainer/trainer.py:435)
        total_d.backward()
        for n in names:
            out[f"{state}_dD::{n}"] = _np(params[n].grad)
        out[f"{state}_gradnorm_D"] = np.array(torch.sqrt(sum((p.grad.double() ** 2).sum() for p in params.values() if p.grad is not None)).item())
        for k, v in D.state_dict().items():
            if k.endswith("._u") or k.endswith("._v"):
                out[f"{state}_after::{k}"] = _np(v)
        print(f"  {state}: " + " ".join(f"{k}={v.item():.5f}" for k, v in {**losses['generator'], **losses['discriminator']}.items()))
    save("trainstep.npz", **out)


def gen_metric(ns):
    out = {}
    g = torch.Generator().manual_seed(5)
    a = 0.1 * torch.randn(3, 8192, generator=g)
    b = a + 0.02 * torch.randn(3, 8192, generator=g)
    hf = torch.tensor([171, 300, 512])
    out.update(a=_np(a), b=_np(b), hf=_np(hf),
               lsd=np.array(ns.metric.lsd(a, b)), snr=np.array(ns.metric.snr(a, b)),
               lsd_hf=np.array(ns.metric.lsd_hf(a, b, hf)), lsd_lf=np.array(ns.metric.lsd_lf(a, b, hf)))
    save("metric.npz", **out)


if __name__ == "__main__":
    ns = load_reference()
    which = sys.argv[1:] or ["scan", "csm", "dwconv", "ss2d", "stft", "model", "metric", "loss", "mpd", "variants", "fullsize", "fullsize2", "trainstep"]
    for w in which:
        print(f"[{w}]")
        globals()[f"gen_{w}"](ns)
