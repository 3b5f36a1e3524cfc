"""TEST INFRASTRUCTURE ONLY (part of oracle/).  Glue: torch.autograd Functions backed by the CPU oracle (oracle/), and a helper
that plugs them into vm_asr_amd modules through the same keyword hooks the reference exposes
on SS2D.forward_corev2 (model/vmamba.py:1398-1400).  Lets the host logic (module wiring,
state_dict layout, autograd plumbing) be checked on a machine without a GPU, and gives
bench.py its `cpu_baseline` leg.  Never imported by the product."""
from functools import partial

import torch

from . import oracle as oracle


def _tdt():
    """torch dtype of the active oracle build (float32, or float64 inside `oracle.float64()`)."""
    return torch.float64 if oracle._F64[0] else torch.float32


def _n(t):
    return None if t is None else t.detach().to(_tdt()).cpu().numpy()


class OracleSelectiveScan(torch.autograd.Function):
    @staticmethod
    def forward(ctx, u, delta, A, B, C, D=None, delta_bias=None, delta_softplus=False, nrows=1,
                backnrows=1, oflex=True):
        ctx.sp = bool(delta_softplus)
        ctx.save_for_backward(u, delta, A, B, C, D, delta_bias)
        out = oracle.sscan_fwd(_n(u), _n(delta), _n(A), _n(B), _n(C), _n(D), _n(delta_bias), ctx.sp)
        return torch.from_numpy(out).to(u.dtype)

    @staticmethod
    def backward(ctx, dout):
        u, delta, A, B, C, D, bias = ctx.saved_tensors
        du, dd, dA, dB, dC, dD, db = oracle.sscan_bwd(_n(u), _n(delta), _n(A), _n(B), _n(C), _n(D), _n(bias),
                                                      _n(dout), ctx.sp)
        t = torch.from_numpy
        return (t(du).to(u.dtype), t(dd).to(delta.dtype), t(dA), t(dB).to(B.dtype), t(dC).to(C.dtype),
                None if D is None else t(dD), None if bias is None else t(db), None, None, None, None)


class OracleCrossScan(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        ctx.shape = x.shape
        return torch.from_numpy(oracle.cross_scan(_n(x))).to(x.dtype)

    @staticmethod
    def backward(ctx, g):
        B, C, H, W = ctx.shape
        return torch.from_numpy(oracle.cross_merge(_n(g).reshape(B, 4, C, H, W))).to(g.dtype).view(B, C, H, W)


class OracleCrossMerge(torch.autograd.Function):
    @staticmethod
    def forward(ctx, ys):
        ctx.shape = ys.shape
        return torch.from_numpy(oracle.cross_merge(_n(ys))).to(ys.dtype)

    @staticmethod
    def backward(ctx, g):
        B, K, C, H, W = ctx.shape
        return torch.from_numpy(oracle.cross_scan(_n(g).reshape(B, C, H, W))).to(g.dtype).view(B, 4, C, H, W)


class OracleDWConvSiLU(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, b):
        ctx.save_for_backward(x, w, b)
        return torch.from_numpy(oracle.dwconv_silu_fwd(_n(x), _n(w).reshape(-1, 3, 3), _n(b))).to(x.dtype)

    @staticmethod
    def backward(ctx, g):
        x, w, b = ctx.saved_tensors
        dx, dw, db = oracle.dwconv_silu_bwd(_n(x), _n(w).reshape(-1, 3, 3), _n(b), _n(g))
        return (torch.from_numpy(dx).to(x.dtype), torch.from_numpy(dw).view_as(w).to(w.dtype),
                torch.from_numpy(db).to(b.dtype))


class OracleISTFT(torch.autograd.Function):
    @staticmethod
    def forward(ctx, mag, phase, hop, win):
        ctx.save_for_backward(mag, phase)
        ctx.cfg = (hop, win)
        return torch.from_numpy(oracle.istft(_n(mag), _n(phase), hop, win))

    @staticmethod
    def backward(ctx, g):
        mag, phase = ctx.saved_tensors
        dm, dp = oracle.istft_bwd(_n(mag), _n(phase), _n(g), *ctx.cfg)
        return torch.from_numpy(dm), torch.from_numpy(dp), None, None


def oracle_wav2spectro(waveform, n_fft, hop_length, win_length, spectro_scale):
    assert spectro_scale == "log2"
    mag, ph = oracle.stft(_n(waveform), n_fft, hop_length, win_length)
    return torch.from_numpy(mag), torch.from_numpy(ph)


def oracle_spectro2wav(mag, phase, n_fft, hop_length, win_length, spectro_scale):
    assert spectro_scale == "log2"
    *other, F, M = mag.shape
    wav = OracleISTFT.apply(mag.reshape(-1, F, M).to(_tdt()), phase.reshape(-1, F, M).to(_tdt()), hop_length, win_length)
    return wav.view(*other, wav.shape[-1])


class OracleSTFTReIm(torch.autograd.Function):
    @staticmethod
    def forward(ctx, wave, n_fft, hop, win, normalized):
        ctx.cfg = (wave.shape, n_fft, hop, win, bool(normalized))
        re, im = oracle.stft(_n(wave), n_fft, hop, win, normalized=normalized, logmag=False)
        return torch.from_numpy(re), torch.from_numpy(im)

    @staticmethod
    def backward(ctx, gre, gim):
        shape, n_fft, hop, win, normalized = ctx.cfg
        gw = oracle.stft_bwd(_n(gre), _n(gim), shape[-1], n_fft, hop, win, normalized)
        return torch.from_numpy(gw).view(shape), None, None, None, None


def oracle_stft_reim(waveform, n_fft, hop_length, win_length, normalized=False):
    return OracleSTFTReIm.apply(waveform, n_fft, hop_length, win_length, normalized)


def use_oracle(module, f64=False):
    """Rewire every SS2D in `module` to the CPU oracle (the reference's own hook mechanism).
    f64: the float64 evaluation (adjudicator): module.double(), no fp32 cast in front of the scan; run the model
    inside `with oracle.float64(), oracle_stft_patch():`."""
    from vm_asr_amd.vmamba import SS2D
    if f64:
        module.double()
    for m in module.modules():
        if isinstance(m, SS2D):
            m.forward_core = partial(m.forward_corev2, force_fp32=(not m.disable_force32) and not f64,
                                     SelectiveScan=OracleSelectiveScan, CrossScan=OracleCrossScan,
                                     CrossMerge=OracleCrossMerge)
            m.conv_act_fn = OracleDWConvSiLU.apply
    return module


class oracle_stft_patch:
    """Context manager: route vm_asr_amd.model's STFT front-end to the oracle."""

    def __enter__(self):
        import vm_asr_amd.metric as Q
        import vm_asr_amd.model as M
        import vm_asr_amd.stft as S
        self._M, self._S, self._Q = M, S, Q
        self._saved = (M.wav2spectro, M.spectro2wav, S.stft_reim, Q.stft_complex)
        M.wav2spectro, M.spectro2wav, S.stft_reim = oracle_wav2spectro, oracle_spectro2wav, oracle_stft_reim
        Q.stft_complex = lambda w, n_fft, hop, win, normalized=False: tuple(
            torch.from_numpy(t) for t in oracle.stft(_n(w), n_fft, hop, win, normalized=normalized, logmag=False))
        return self

    def __exit__(self, *exc):
        self._M.wav2spectro, self._M.spectro2wav, self._S.stft_reim, self._Q.stft_complex = self._saved
        return False
