"""MR-STFT loss on the HIP STFT front-end vs the reference's loss (golden: tests/golden/loss.npz,
produced by model/loss.py on CPU).  CPU variant runs the same module with the oracle STFT."""
import os

import numpy as np
import pytest
import torch

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def _run(device):
    from vm_asr_amd.loss import MultiResolutionSTFTLoss
    z = np.load(os.path.join(GOLDEN, "loss.npz"))
    y = torch.from_numpy(z["y"]).to(device)
    for tag, emph in (("plain", False), ("emph", True)):
        L = MultiResolutionSTFTLoss(factor_sc=0.5, factor_mag=0.5, emphasize_high_freq=emph).to(device)
        x = torch.from_numpy(z["x"]).to(device).requires_grad_()
        sc, mag = L(x, y)
        (sc + mag).backward()
        assert abs(sc.item() - float(z[f"{tag}_sc"])) < 1e-4 * max(1.0, abs(float(z[f"{tag}_sc"])))
        assert abs(mag.item() - float(z[f"{tag}_mag"])) < 1e-4 * max(1.0, abs(float(z[f"{tag}_mag"])))
        want = z[f"{tag}_dx"]
        err = np.abs(x.grad.cpu().numpy() - want).max()
        assert err <= 1e-4 * max(1e-3, np.abs(want).max()) + 1e-7, (tag, err, np.abs(want).max())


def test_mr_stft_loss_cpu_oracle_backend():
    from oracle.torch_backend import oracle_stft_patch
    with oracle_stft_patch():
        _run("cpu")


@pytest.mark.gpu
def test_mr_stft_loss_hip():
    _run("cuda:0")


@pytest.mark.gpu
def test_stft_bwd_vs_oracle_full_clip():
    import oracle
    from vm_asr_amd import stft
    g = torch.Generator().manual_seed(3)
    for n_fft, hop, win in ((1024, 120, 600), (2048, 240, 1200), (512, 50, 240)):
        x = (0.1 * torch.randn(2, 122640, generator=g)).to("cuda:0").requires_grad_()
        re, im = stft.stft_reim(x, n_fft, hop, win)
        gr, gi = torch.randn(re.shape, generator=g), torch.randn(re.shape, generator=g)
        (re * gr.to("cuda:0") + im * gi.to("cuda:0")).sum().backward()
        want = oracle.stft_bwd(gr.numpy(), gi.numpy(), 122640, n_fft, hop, win)
        err = np.abs(x.grad.cpu().numpy() - want).max()
        assert err < 1e-4 * np.abs(want).max(), (n_fft, err)
