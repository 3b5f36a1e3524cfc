"""STFT / iSTFT front-end (HIP).  Same names, arguments and return values as the
reference's utils/stft.py:

    wav2spectro(waveform, n_fft, hop_length, win_length, spectro_scale) -> (mag, phase)   # :22-68
    spectro2wav(mag, phase, n_fft, hop_length, win_length, spectro_scale) -> waveform      # :71-115

Only spectro_scale == "log2" (every shipped config: config.py:58) is implemented; the
"dB" branch of the reference needs torchaudio and is unreachable from the yamls.
spectro2wav is differentiable wrt (mag, phase) (training back-propagates through it);
wav2spectro is not differentiable (its input is the network input).

Compute: vm_asr_amd/csrc/stft.hip.  No CPU fallback.
"""
import ctypes
from typing import Tuple

import torch

from . import _lib

__all__ = ["wav2spectro", "spectro2wav", "stft_complex", "stft_reim", "ISTFTFunction", "STFTReImFunction"]


def _p(t):
    return ctypes.c_void_p(t.data_ptr())


def _stft(waveform, n_fft, hop, win, normalized, logmag):
    if not waveform.is_cuda:
        raise RuntimeError("wav2spectro: expected a CUDA (HIP) tensor; vm_asr_amd has no CPU path")
    *other, length = waveform.shape
    w = waveform.reshape(-1, length).float().contiguous()
    Bn = w.shape[0]
    F, M = n_fft // 2 + 1, 1 + length // hop
    with torch.cuda.device(w.device):
        o0 = torch.empty((Bn, F, M), dtype=torch.float32, device=w.device)
        o1 = torch.empty_like(o0)
        _lib.check(_lib.lib().vmasr_stft(_p(w), _p(o0), _p(o1), Bn, length, n_fft, hop, win, int(normalized),
                                         int(logmag), _lib.current_stream(w.device)), "stft")
    return o0.view(*other, F, M), o1.view(*other, F, M)


@torch.no_grad()
def wav2spectro(waveform: torch.Tensor, n_fft: int, hop_length: int, win_length: int,
                spectro_scale: str) -> Tuple[torch.Tensor, torch.Tensor]:
    if spectro_scale != "log2":
        raise NotImplementedError("only spectro_scale='log2' is supported (reference dB branch needs torchaudio)")
    return _stft(waveform, n_fft, hop_length, win_length, True, True)


@torch.no_grad()
def stft_complex(waveform, n_fft, hop_length, win_length, normalized=False):
    """(re, im) of torch.stft(center=True, window=hann(win_length)) — used by loss/metrics."""
    return _stft(waveform, n_fft, hop_length, win_length, normalized, False)


class STFTReImFunction(torch.autograd.Function):
    """Differentiable (re, im) of torch.stft(center=True, window=hann(win_length)) for the losses:
    forward = vmasr_stft(logmag=0), backward = vmasr_stft_bwd.  Unlike torch.stft (rocFFT) both
    directions are plain kernel launches and therefore HIP-graph capturable."""

    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(ctx, wave, n_fft, hop, win, normalized):
        if not wave.is_cuda:
            raise RuntimeError("stft_reim: expected a CUDA (HIP) tensor; vm_asr_amd has no CPU path")
        ctx.cfg = (wave.shape, n_fft, hop, win, bool(normalized))
        return _stft(wave, n_fft, hop, win, normalized, False)

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(ctx, gre, gim):
        shape, n_fft, hop, win, normalized = ctx.cfg
        T = shape[-1]
        F, M = n_fft // 2 + 1, 1 + T // hop
        gre, gim = gre.reshape(-1, F, M).float().contiguous(), gim.reshape(-1, F, M).float().contiguous()
        Bn = gre.shape[0]
        lib = _lib.lib()
        with torch.cuda.device(gre.device):
            gw = torch.empty((Bn, T), dtype=torch.float32, device=gre.device)
            wsb = lib.vmasr_stft_bwd_workspace(Bn, T, n_fft, hop)
            ws = torch.empty(wsb // 4, dtype=torch.float32, device=gre.device)
            _lib.check(lib.vmasr_stft_bwd(_p(gre), _p(gim), _p(gw), Bn, T, n_fft, hop, win, int(normalized), _p(ws), wsb,
                                          _lib.current_stream(gre.device)), "stft_bwd")
        return gw.view(shape), None, None, None, None


def stft_reim(waveform, n_fft, hop_length, win_length, normalized=False):
    """(re, im), each (..., n_fft/2+1, 1+T//hop); differentiable wrt `waveform`."""
    return STFTReImFunction.apply(waveform, n_fft, hop_length, win_length, normalized)


class ISTFTFunction(torch.autograd.Function):
    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(ctx, mag, phase, hop, win):
        Bn, F, M = mag.shape
        mag, phase = mag.contiguous(), phase.contiguous()
        lib = _lib.lib()
        with torch.cuda.device(mag.device):
            wav = torch.empty((Bn, hop * (M - 1)), dtype=torch.float32, device=mag.device)
            wsb = lib.vmasr_istft_workspace(Bn, F, M, hop)
            ws = torch.empty(wsb // 4, dtype=torch.float32, device=mag.device)
            _lib.check(lib.vmasr_istft(_p(mag), _p(phase), _p(wav), Bn, F, M, hop, win, _p(ws), wsb,
                                       _lib.current_stream(mag.device)), "istft")
        ctx.save_for_backward(mag, phase)
        ctx.cfg = (hop, win)
        return wav

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(ctx, g):
        mag, phase = ctx.saved_tensors
        hop, win = ctx.cfg
        Bn, F, M = mag.shape
        g = g.float().contiguous()
        with torch.cuda.device(mag.device):
            dmag, dphase = torch.empty_like(mag), torch.empty_like(phase)
            _lib.check(_lib.lib().vmasr_istft_bwd(_p(mag), _p(phase), _p(g), _p(dmag), _p(dphase), Bn, F, M, hop,
                                                  win, _lib.current_stream(mag.device)), "istft_bwd")
        return dmag, dphase, None, None


def spectro2wav(mag: torch.Tensor, phase: torch.Tensor, n_fft: int, hop_length: int, win_length: int,
                spectro_scale: str) -> torch.Tensor:
    if spectro_scale != "log2":
        raise NotImplementedError("only spectro_scale='log2' is supported (reference dB branch needs torchaudio)")
    if not mag.is_cuda:
        raise RuntimeError("spectro2wav: expected a CUDA (HIP) tensor; vm_asr_amd has no CPU path")
    *other, freqs, frames = mag.shape
    # the reference derives n_fft from the bin count (utils/stft.py:89)
    wav = ISTFTFunction.apply(mag.reshape(-1, freqs, frames), phase.reshape(-1, freqs, frames), hop_length,
                              win_length)
    return wav.view(*other, wav.shape[-1])
