"""ctypes/numpy binding of oracle/libvmasr_oracle.so (see vmasr_oracle.c for citations).

Parity pinning: checked against the reference-generated goldens in tests/test_oracle.py.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.environ.get("VMASR_ORACLE_LIB") or os.path.join(_HERE, "libvmasr_oracle.so")   # VMASR_ORACLE_LIB: the ASan build (make asan)
_LIB64_PATH = os.path.join(_HERE, "libvmasr_oracle64.so")
_lib = None
_lib64 = None
_F64 = [False]


class float64:
    """Context manager: inside it every function of this module runs the float64 build of the same C source
    (libvmasr_oracle64.so) on float64 arrays — the adjudicator of fp32-vs-fp32 differences (vmasr_oracle.c header)."""

    def __enter__(self):
        self._saved = _F64[0]
        _F64[0] = True
        return self

    def __exit__(self, *exc):
        _F64[0] = self._saved
        return False


def _dt():
    return np.float64 if _F64[0] else np.float32

__all__ = ["build", "lib", "num_threads", "float64", "sscan_fwd", "sscan_bwd", "cross_scan", "cross_merge",
           "dwconv_silu_fwd", "dwconv_silu_bwd", "stft", "stft_bwd", "istft", "istft_bwd", "lsd", "snr"]


def build(force=False):
    """Compile the C restatement with gcc (no GPU, no reference needed)."""
    src = os.path.join(_HERE, "vmasr_oracle.c")
    for path in (_LIB_PATH, _LIB64_PATH):
        if os.path.dirname(os.path.abspath(path)) != _HERE:
            continue          # an explicitly given library outside the tree is used as it is
        if force or not os.path.exists(path) or os.path.getmtime(path) < os.path.getmtime(src):
            subprocess.check_call(["make", "-C", _HERE, "-B", os.path.basename(path)], stdout=subprocess.DEVNULL)
    return _LIB_PATH


def lib():
    global _lib, _lib64
    if _lib is None:
        build()
        _lib, _lib64 = ctypes.CDLL(_LIB_PATH), ctypes.CDLL(_LIB64_PATH)
        for l in (_lib, _lib64):
            l.vmasr_oracle_num_threads.restype = ctypes.c_int
            l.vmasr_oracle_stft_frames.restype = ctypes.c_int
    return _lib64 if _F64[0] else _lib


def num_threads():
    return int(lib().vmasr_oracle_num_threads())


def _f(a):
    return None if a is None else np.ascontiguousarray(a, dtype=_dt())


def _p(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


def sscan_fwd(u, delta, A, B, C, D=None, bias=None, softplus=False, return_last_state=False):
    u, delta, A, B, C, D, bias = map(_f, (u, delta, A, B, C, D, bias))
    batch, dim, L = u.shape
    if B.ndim == 3:
        B = B[:, None]
        C = C[:, None]
    G, N = B.shape[1], B.shape[2]
    out = np.empty_like(u)
    last = np.empty((batch, dim, N), _dt())
    lib().vmasr_oracle_sscan_fwd(_p(u), _p(delta), _p(A), _p(B), _p(C), _p(D), _p(bias),
                                 int(softplus), batch, dim, G, N, L, _p(out), _p(last))
    return (out, last) if return_last_state else out


def sscan_bwd(u, delta, A, B, C, D, bias, dout, softplus=False):
    u, delta, A, B, C, D, bias, dout = map(_f, (u, delta, A, B, C, D, bias, dout))
    batch, dim, L = u.shape
    squeeze = B.ndim == 3
    if squeeze:
        B = B[:, None]
        C = C[:, None]
    G, N = B.shape[1], B.shape[2]
    du, dd = np.empty_like(u), np.empty_like(u)
    dA = np.empty_like(A)
    dB, dC = np.empty_like(B), np.empty_like(C)
    dD = np.empty(dim, _dt()) if D is not None else None
    db = np.empty(dim, _dt()) if bias is not None else None
    lib().vmasr_oracle_sscan_bwd(_p(u), _p(delta), _p(A), _p(B), _p(C), _p(D), _p(bias), _p(dout),
                                 int(softplus), batch, dim, G, N, L, _p(du), _p(dd), _p(dA),
                                 _p(dB), _p(dC), _p(dD), _p(db))
    if squeeze:
        dB, dC = dB[:, 0], dC[:, 0]
    return du, dd, dA, dB, dC, dD, db


def cross_scan(x):
    x = _f(x)
    Bn, C, H, W = x.shape
    xs = np.empty((Bn, 4, C, H * W), _dt())
    lib().vmasr_oracle_cross_scan(_p(x), Bn, C, H, W, _p(xs))
    return xs


def cross_merge(ys):
    ys = _f(ys)
    Bn, K, C, H, W = ys.shape
    assert K == 4
    y = np.empty((Bn, C, H * W), _dt())
    lib().vmasr_oracle_cross_merge(_p(ys), Bn, C, H, W, _p(y))
    return y


def dwconv_silu_fwd(x, w, b=None):
    x, w, b = _f(x), _f(w), _f(b)
    Bn, C, H, W = x.shape
    y = np.empty_like(x)
    lib().vmasr_oracle_dwconv_silu_fwd(_p(x), _p(w), _p(b), Bn, C, H, W, _p(y), None)
    return y


def dwconv_silu_bwd(x, w, b, g):
    x, w, b, g = _f(x), _f(w), _f(b), _f(g)
    Bn, C, H, W = x.shape
    dx, dw = np.empty_like(x), np.empty_like(w)
    db = np.empty(C, _dt())
    lib().vmasr_oracle_dwconv_silu_bwd(_p(x), _p(w), _p(b), _p(g), Bn, C, H, W, _p(dx), _p(dw), _p(db))
    return dx, dw, db


def stft(wav, n_fft, hop, win, normalized=True, logmag=True):
    """wav (..., T) -> (mag, phase) or (re, im), each (..., n_fft/2+1, frames)."""
    wav = _f(wav)
    lead, T = wav.shape[:-1], wav.shape[-1]
    w2 = wav.reshape(-1, T)
    F, M = n_fft // 2 + 1, 1 + T // hop
    o0 = np.empty((w2.shape[0], F, M), _dt())
    o1 = np.empty_like(o0)
    lib().vmasr_oracle_stft(_p(w2), w2.shape[0], T, n_fft, hop, win, int(normalized), int(logmag),
                            _p(o0), _p(o1))
    return o0.reshape(*lead, F, M), o1.reshape(*lead, F, M)


def stft_bwd(gre, gim, T, n_fft, hop, win, normalized=False):
    """adjoint of stft(..., logmag=False): (gre, gim) (..., F, M) -> gwav (..., T)."""
    gre, gim = _f(gre), _f(gim)
    lead, (F, M) = gre.shape[:-2], gre.shape[-2:]
    g0, g1 = gre.reshape(-1, F, M), gim.reshape(-1, F, M)
    gw = np.empty((g0.shape[0], T), _dt())
    lib().vmasr_oracle_stft_bwd(_p(g0), _p(g1), g0.shape[0], T, n_fft, hop, win, int(normalized), _p(gw))
    return gw.reshape(*lead, T)


def istft(mag, phase, hop, win):
    mag, phase = _f(mag), _f(phase)
    lead, (F, M) = mag.shape[:-2], mag.shape[-2:]
    m2, p2 = mag.reshape(-1, F, M), phase.reshape(-1, F, M)
    wav = np.empty((m2.shape[0], hop * (M - 1)), _dt())
    lib().vmasr_oracle_istft(_p(m2), _p(p2), m2.shape[0], F, M, hop, win, _p(wav))
    return wav.reshape(*lead, -1)


def istft_bwd(mag, phase, g, hop, win):
    mag, phase, g = _f(mag), _f(phase), _f(g)
    lead, (F, M) = mag.shape[:-2], mag.shape[-2:]
    m2, p2 = mag.reshape(-1, F, M), phase.reshape(-1, F, M)
    g2 = g.reshape(m2.shape[0], -1)
    dm, dp = np.empty_like(m2), np.empty_like(p2)
    lib().vmasr_oracle_istft_bwd(_p(m2), _p(p2), _p(g2), m2.shape[0], F, M, hop, win, _p(dm), _p(dp))
    return dm.reshape(mag.shape), dp.reshape(phase.shape)


def lsd(output, target, n_fft=2048, hop=512):
    """Log-spectral distance, model/metric.py:5-12,26-29 (non-normalised hann STFT)."""
    def spec(a):
        re, im = stft(a, n_fft, hop, n_fft, normalized=False, logmag=False)
        return np.sqrt(re.astype(_dt()) ** 2 + im.astype(_dt()) ** 2)
    sp = np.log10(np.maximum(spec(output) ** 2, 1e-8))
    st = np.log10(np.maximum(spec(target) ** 2, 1e-8))
    return float(np.mean(np.sqrt(np.mean((sp - st) ** 2, axis=-2))))


def snr(output, target):
    """model/metric.py:15-23."""
    output, target = _f(output), _f(target)
    num = np.linalg.norm(target, axis=-1)
    den = np.maximum(np.linalg.norm(output - target, axis=-1), 1e-8)
    return float(np.mean(20 * np.log10(num / den)))
