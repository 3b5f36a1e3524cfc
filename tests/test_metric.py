"""vm_asr_amd.metric (SNR, LSD, LSD-HF, LSD-LF on the HIP STFT) vs the reference's model/metric.py:5-67
(golden tests/golden/metric.npz, made by the reference on CPU).  LSD is the parity metric of BASELINE.json."""
import os

import numpy as np
import pytest
import torch

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def _run(device, tol):
    from vm_asr_amd import metric
    z = np.load(os.path.join(GOLDEN, "metric.npz"))
    a, b = torch.from_numpy(z["a"]).to(device), torch.from_numpy(z["b"]).to(device)
    hf = torch.from_numpy(z["hf"])
    got = {"lsd": metric.lsd(a, b), "snr": metric.snr(a, b), "lsd_hf": metric.lsd_hf(a, b, hf), "lsd_lf": metric.lsd_lf(a, b, hf)}
    for k, v in got.items():
        assert torch.is_tensor(v) and v.ndim == 0          # no host sync inside the metric
        want = float(z[k])
        assert abs(float(v) - want) <= tol * max(1.0, abs(want)), (k, float(v), want)
    # |STFT| itself: (B, 1025, frames), non-normalised hann, n_fft 2048 / hop 512 (model/metric.py:5-12)
    s = metric.stft(a)
    assert s.shape == (3, 1025, 1 + 8192 // 512)
    # metric(x, x): LSD 0; and the trainer's call form (keyword hf) works for every metric
    assert float(metric.lsd(a, a)) == 0.0
    for f in (metric.lsd, metric.snr, metric.lsd_hf, metric.lsd_lf):
        assert torch.isfinite(f(a, b, hf=hf))


def test_metrics_cpu_oracle_backend():
    from oracle.torch_backend import oracle_stft_patch
    with oracle_stft_patch():
        _run("cpu", 1e-4)


@pytest.mark.gpu
def test_metrics_hip():
    _run("cuda:0", 1e-4)


@pytest.mark.gpu
def test_lsd_full_clip_hip_vs_oracle():
    """LSD / SNR at the benchmark's clip length (B=4, 122 640 samples) against the oracle's C restatement."""
    import oracle
    from vm_asr_amd import metric
    g = torch.Generator().manual_seed(9)
    a = 0.1 * torch.randn(4, 122640, generator=g)
    b = a + 0.03 * torch.randn(4, 122640, generator=g)
    assert abs(float(metric.lsd(a.cuda(), b.cuda())) - oracle.lsd(a.numpy(), b.numpy())) < 1e-4
    assert abs(float(metric.snr(a.cuda(), b.cuda())) - oracle.snr(a.numpy(), b.numpy())) < 1e-3
