"""Evaluation metrics on the HIP STFT: SNR, LSD, LSD-HF, LSD-LF (model/metric.py:5-67).

LSD (the parity metric of BASELINE.json) = mean_t sqrt(mean_f (log10|S_out|^2 - log10|S_tgt|^2)^2)
with a non-normalised hann STFT, n_fft 2048 / hop 512.  Unlike the reference these return
tensors (no per-metric .item() host sync, trainer/trainer.py:179-182); call float() to read.
"""
import torch

from .stft import stft_complex

__all__ = ["stft", "snr", "lsd", "lsd_hf", "lsd_lf"]


def stft(audio, n_fft=2048, hop_length=512):
    """|STFT| of (B,T) audio -> (B, n_fft/2+1, frames)."""
    re, im = stft_complex(audio, n_fft, hop_length, n_fft, normalized=False)
    return torch.sqrt(re.pow(2) + im.pow(2))


def snr(output, target, **kwargs):
    return (20 * torch.log10(torch.norm(target, dim=-1) / torch.norm(output - target, dim=-1).clamp(min=1e-8))).mean()


def _logspec(x):
    return torch.log10(stft(x).square().clamp(1e-8))


def lsd(output, target, **kwargs):
    return (_logspec(output) - _logspec(target)).square().mean(dim=1).sqrt().mean()


def _lsd_band(output, target, hf, high):
    sp, st = _logspec(output), _logspec(target)
    vals = []
    for i in range(output.size(0)):
        h = int(hf[i])
        d = (sp[i, h:] - st[i, h:]) if high else (sp[i, :h] - st[i, :h])
        vals.append(d.square().mean(dim=0).sqrt().mean())
    return torch.stack(vals).mean()


def lsd_hf(output, target, hf):
    return _lsd_band(output, target, hf, True)


def lsd_lf(output, target, hf):
    return _lsd_band(output, target, hf, False)
