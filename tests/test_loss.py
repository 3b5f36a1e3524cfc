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


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(2, 513, 37), (1, 257, 5), (3, 1025, 64)])
def test_stft_loss_kernels_match_float64(shape):
    """csrc/stftloss.hip (one pass for the three sums of sc / mag, one backward pass) against the float64 torch expression of
    model/loss.py:17-45,137-184 on random spectra — including bins below the clamp (exact zero gradient there), sizes that are
    not multiples of 4, and only one of the two upstream gradients."""
    from vm_asr_amd.loss import _STFTLossFn
    torch.manual_seed(shape[1])
    rx, ix, ry, iy = (torch.randn(shape, device="cuda:0") for _ in range(4))
    rx[0, :3, 0] = 1e-5; ix[0, :3, 0] = -2e-5                                   # below the clamp: re^2 + im^2 < 1e-7
    ry[0, 5, 1] = 0.0; iy[0, 5, 1] = 0.0
    for wsc, wml in ((0.7, 1.3), (1.0, 0.0), (0.0, 2.0)):
        a, b = rx.clone().requires_grad_(), ix.clone().requires_grad_()
        sc, ml = _STFTLossFn.apply(a, b, ry, iy)
        terms = ([wsc * sc] if wsc else []) + ([wml * ml] if wml else [])
        sum(terms).backward()
        a64, b64 = rx.double().requires_grad_(), ix.double().requires_grad_()
        mx = torch.sqrt(torch.clamp(a64 ** 2 + b64 ** 2, min=1e-7))
        my = torch.sqrt(torch.clamp(ry.double() ** 2 + iy.double() ** 2, min=1e-7))
        sc64 = torch.norm(my - mx, p="fro") / torch.norm(my, p="fro")
        ml64 = torch.nn.functional.l1_loss(torch.log(my), torch.log(mx))
        (wsc * sc64 + wml * ml64).backward()
        assert abs(sc.item() - sc64.item()) <= 2e-6 * abs(sc64.item()) and abs(ml.item() - ml64.item()) <= 2e-6 * abs(ml64.item())
        for got, want in ((a.grad, a64.grad), (b.grad, b64.grad)):
            err = (got.double() - want).abs().max().item()
            assert err <= 2e-5 * want.abs().max().item(), (shape, wsc, wml, err, want.abs().max().item())
        assert not a.grad[0, :3, 0].any() and not b.grad[0, :3, 0].any()


@pytest.mark.gpu
def test_lsgan_terms_kernel_matches_torch():
    """HiFiGANLoss('lsgan') discriminator / generator losses through csrc/featloss.hip's one-launch LSGAN kernels against the
    per-tensor torch expression of model/loss.py:190-213 in float64: values and the gradient of every score tensor."""
    from vm_asr_amd.loss import HiFiGANLoss
    torch.manual_seed(0)
    L = HiFiGANLoss("lsgan")
    sizes = [(4, 7013), (4, 4675), (4, 2805), (4, 2004), (4, 1276)]
    real = [torch.randn(s, device="cuda:0").requires_grad_() for s in sizes]
    gen = [torch.randn(s, device="cuda:0").requires_grad_() for s in sizes]
    d = L.discriminator_loss(real, gen)
    g = L.generator_loss(gen)
    (1.5 * d + 0.5 * g).backward()
    r64 = [t.detach().double().requires_grad_() for t in real]
    g64 = [t.detach().double().requires_grad_() for t in gen]
    d64 = sum(torch.mean((a - 1) ** 2) + torch.mean(b ** 2) for a, b in zip(r64, g64))
    gg64 = sum(torch.mean((1 - b) ** 2) for b in g64)
    (1.5 * d64 + 0.5 * gg64).backward()
    assert abs(d.item() - d64.item()) <= 1e-6 * d64.item() and abs(g.item() - gg64.item()) <= 1e-6 * gg64.item()
    for a, b in zip(real + gen, r64 + g64):       # (the two terms' gradients of a generated score partly cancel: absolute bound)
        assert (a.grad.double() - b.grad).abs().max().item() <= 2e-6 * b.grad.abs().max().item()
