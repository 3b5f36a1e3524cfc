"""Evaluation harness (vm_asr_amd/tester.py <- trainer/tester.py:15-240, utils/post_processing.py:4-34), the oflex
operator surface (cusoflex/selective_scan_oflex.cpp) and the CLI's flag handling (main.py:28-93)."""
import csv
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_unfold_fold_roundtrip_and_overlap_average():
    from vm_asr_amd.tester import fold_audio, unfold_audio
    x = torch.randn(2, 1, 1000)
    seg = unfold_audio(x, 400, 100)                     # step 300 -> 3 segments cover [0, 1000)
    assert seg.shape == (2, 1, 3, 400)
    assert torch.allclose(fold_audio(seg, 1000, 400, 100), x, atol=1e-6)
    # overlaps are averaged: two different estimates of the overlapped samples
    seg2 = seg.clone()
    seg2[:, :, 1] += 1.0
    y = fold_audio(seg2, 1000, 400, 100)
    assert torch.allclose(y[..., 300:400], x[..., 300:400] + 0.5, atol=1e-6)       # covered by segments 0 and 1
    assert torch.allclose(y[..., 400:600], x[..., 400:600] + 1.0, atol=1e-6)       # only segment 1
    # samples no segment reaches stay zero (reference behaviour)
    assert float(fold_audio(unfold_audio(torch.ones(1, 1, 1050), 400, 100), 1050, 400, 100)[..., 1000:].abs().sum()) == 0.0


def _eval_setup(tmp_path, device, T_long=None):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_trainer import _resumable, _tiny_config
    from vm_asr_amd import metric
    from vm_asr_amd.trainer import SyntheticVCTK
    cfg = _tiny_config()
    cfg.defrost()
    cfg.OUTPUT = str(tmp_path / "ckpt")
    cfg.freeze()
    tr = _resumable(cfg, tmp_path, device)
    tr._save_checkpoint(1, save_best=True)
    ev = cfg.clone()
    ev.MODEL.RESUME_PATH, ev.OUTPUT, ev.TAG, ev.EVAL_MODE = str(tmp_path / "ckpt"), str(tmp_path / "out"), "8000_16000", True
    ev.TEST.OVERLAP = 160
    ev.freeze()

    class _DS(SyntheticVCTK):       # one regular clip and one 2.2x longer one (the overlap-fold branch)
        def __getitem__(self, i):
            inp, tgt, hc, name, pad = super().__getitem__(i)
            if i % 2:
                inp, tgt = inp.repeat(1, 3)[:, :T_long], tgt.repeat(1, 3)[:, :T_long]
            return inp, tgt, hc, name, 16 if i % 2 else 0
    ds = _DS(ev, length=2, sr_in=8000)
    if T_long is None:
        T_long = int(2.2 * ds.T) // 80 * 80
    loader = torch.utils.data.DataLoader(ds, batch_size=1)
    mets = [metric.snr, metric.lsd, metric.lsd_hf, metric.lsd_lf]
    return tr, ev, loader, mets


def _check_eval(tmp_path, device):
    import vm_asr_amd
    from vm_asr_amd.tester import Tester
    tr, ev, loader, mets = _eval_setup(tmp_path, device)
    gen = vm_asr_amd.get_model(ev)["generator"]
    if device == "cpu":
        from oracle.torch_backend import use_oracle
        use_oracle(gen)
    cwd = os.getcwd()
    os.chdir(tmp_path)
    try:
        t = Tester({"generator": gen}, mets, ev, torch.device(device), loader)
        # weights came from the checkpoint the trainer wrote
        for (k, a), (_, b) in zip(t.models["generator"].state_dict().items(), tr.models["generator"].state_dict().items()):
            assert torch.equal(a.cpu(), b.cpu()), k
        res = t.evaluate()
        rows = list(csv.reader(open("results_16kHz.csv")))
    finally:
        os.chdir(cwd)
    assert rows[0] == ["SAMPLE_RATE", "SNR", "LSD", "LSD_HF", "LSD_LF", "RTF", "RTF_RECIPROCAL"] and len(rows) == 2
    assert int(rows[1][0]) == 8000 and all(np.isfinite(float(v)) for v in rows[1][1:])
    assert set(res) == {"snr", "lsd", "lsd_hf", "lsd_lf", "rtf", "rtf_reciprocal", "sample_rate"} and res["rtf"] > 0
    wavs = sorted(os.listdir(ev.OUTPUT))
    assert len(wavs) == 6 and wavs[0].endswith("_down.wav")
    import wave
    with wave.open(os.path.join(ev.OUTPUT, wavs[0])) as f:
        assert f.getframerate() == 16000 and f.getsampwidth() == 2 and f.getnframes() > 0
    return res


def test_tester_cpu_oracle_backend(tmp_path):
    from oracle.torch_backend import oracle_stft_patch
    with oracle_stft_patch():
        _check_eval(tmp_path, "cpu")


@pytest.mark.gpu
def test_tester_hip(tmp_path):
    _check_eval(tmp_path, "cuda:0")


def test_cli_flags_follow_the_reference():
    """main.py:28-93 flags -> config (config.py:267-334): eval mode, tag, resume path as output, batch size, opts."""
    sys.path.insert(0, ROOT)
    import main
    cfgdir = os.path.join(ROOT, "tests", "golden")
    yml = os.path.join(cfgdir, "_cli_test.yaml")
    open(yml, "w").write("MODEL:\n  VSSM:\n    DIMS: 16\nTRAIN:\n  ADVERSARIAL:\n    ENABLE: True\n    DISCRIMINATORS: ['mpd']\nDATA:\n  BATCH_SIZE: 4\n  TARGET_SR: 48000\n")
    try:
        a, c = main.parse_option(["--cfg", yml, "--eval", "--tag", "16000_48000", "--resume", "/tmp/ck", "--batch-size", "2",
                                  "--opts", "MODEL.VSSM.SSM_D_STATE", "32", "DATA.STFT.N_FFT", "2048"])
        assert c.EVAL_MODE and c.TAG == "16000_48000" and c.OUTPUT == "/tmp/ck" and c.DATA.BATCH_SIZE == 2
        assert c.MODEL.VSSM.SSM_D_STATE == 32 and c.DATA.STFT.N_FFT == 2048 and c.DATA.STFT.HOP_LENGTH == 240 and c.is_frozen()
        a, c = main.parse_option(["--cfg", yml, "--disable_amp", "--accumulation-steps", "2", "--target_sr", "16000"])
        assert not c.AMP_ENABLE and c.TRAIN.ACCUMULATION_STEPS == 2 and c.DATA.STFT.HOP_LENGTH == 80 and not c.EVAL_MODE
    finally:
        os.remove(yml)


@pytest.mark.gpu
def test_oflex_output_dtype():
    """`selective_scan_cuda_oflex.fwd(..., out_float)` (cusoflex/selective_scan_oflex.cpp:163-164,218-239): 16-bit inputs,
    fp32 output — bit-identical to the fp32 kernel on the up-converted inputs and within 1e-4 of the oracle on them;
    backward with an fp32 dout returns 16-bit du / ddelta.  out_float=False is the plain operator."""
    import oracle
    from vm_asr_amd import selective_scan as ss
    g = torch.Generator().manual_seed(0)
    Bn, KD, G, N, L = 2, 16, 4, 2, 700
    u = torch.randn(Bn, KD, L, generator=g).to(torch.bfloat16).cuda()
    dl = (0.5 * torch.rand(Bn, KD, L, generator=g)).to(torch.bfloat16).cuda()
    A = (-0.5 * torch.rand(KD, N, generator=g)).cuda()
    Bm, Cm = (torch.randn(Bn, G, N, L, generator=g).to(torch.bfloat16).cuda() for _ in range(2))
    D, bias = torch.randn(KD, generator=g).cuda(), (0.5 * torch.rand(KD, generator=g)).cuda()
    out, x = ss.fwd_oflex(u, dl, A, Bm, Cm, D, bias, True, 1, True)
    assert out.dtype == torch.float32
    ref, _ = ss.fwd(u.float(), dl.float(), A, Bm.float(), Cm.float(), D, bias, True, 1)
    assert torch.equal(out, ref)
    want = oracle.sscan_fwd(*[t.float().cpu().numpy() for t in (u, dl, A, Bm, Cm, D, bias)], True)
    assert np.abs(out.cpu().numpy() - want).max() <= 1e-4 * max(1.0, np.abs(want).max())
    assert ss.fwd_oflex(u, dl, A, Bm, Cm, D, bias, True, 1, False)[0].dtype == torch.bfloat16
    dout = torch.randn(Bn, KD, L, generator=g).cuda()
    du, dd, dA, dB, dC, dD, db = ss.bwd_oflex(u, dl, A, Bm, Cm, D, bias, dout, x, True, 1)
    assert du.dtype == dd.dtype == dB.dtype == dC.dtype == torch.bfloat16 and dA.dtype == torch.float32
    # autograd surface (model/vmamba.py:358-392)
    uu = u.clone().requires_grad_()
    y = ss.SelectiveScanOflex.apply(uu, dl, A, Bm, Cm, D, bias, True, 1, 1, True)
    assert y.dtype == torch.float32
    y.backward(dout)
    assert torch.equal(uu.grad, du)
