"""Multi-period discriminator (PyTorch / MIOpen convs): the adversary of the MPD configs.

Re-statement of model/discriminator.py:21-147 (HiFi-GAN style, periods 2,3,5,7,11, hidden 32
-> 41.09 M parameters).  The reference's inverted ternary (`weight_norm if use_spectral_norm
else spectral_norm`, :37) means the default `use_spectral_norm=False` yields SPECTRAL norm;
that is reproduced so state_dicts (parametrizations.weight.original + power-iteration
buffers) stay compatible.  MSD (:174-337) is not enabled by any yaml and is not built.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F
from torch.nn.utils import parametrize
from torch.nn.utils.parametrizations import weight_norm

from .linear import linear as _linear
from .streams import parallel as _parallel

__all__ = ["PeriodDiscriminator", "MultiPeriodDiscriminator", "spectral_norm"]


class _SpectralNorm(nn.Module):
    """Spectral-norm parametrization with the state_dict layout of
    torch.nn.utils.parametrizations.spectral_norm (`parametrizations.weight.original`,
    `parametrizations.weight.0._u/_v`) and the same algorithm (one power iteration per training
    forward, sigma = u^T W v on cloned vectors).  The matrix-vector products are written as
    (N,1) matmuls in fp32: on ROCm 7.2 `torch.mv` (aten::addmv_ -> rocBLAS gemv) costs ~4 ms of
    HOST time per call, 1.1 s per training step for the 30 MPD layers (profiles/r01_*)."""

    def __init__(self, weight, n_power_iterations=1, eps=1e-12):
        super().__init__()
        self.n_power_iterations, self.eps = n_power_iterations, eps
        w = weight.detach().flatten(1)
        u = F.normalize(w.new_empty(w.size(0)).normal_(0, 1), dim=0, eps=eps)
        v = F.normalize(w.new_empty(w.size(1)).normal_(0, 1), dim=0, eps=eps)
        self.register_buffer("_u", u)
        self.register_buffer("_v", v)
        self._power_method(w, 15)

    @torch.autograd.no_grad()
    def _power_method(self, w, n):
        if w.is_cuda and n > 0 and w.dtype == torch.float32:
            import ctypes
            from . import _lib
            w = w.contiguous()
            R, C = w.shape
            ws = torch.empty(R + C, dtype=torch.float32, device=w.device)
            p = lambda t: ctypes.c_void_p(t.data_ptr())  # noqa: E731
            with torch.cuda.device(w.device):
                _lib.check(_lib.lib().vmasr_spectral_power_iter(p(w), p(self._u), p(self._v), p(ws), R, C, int(n),
                                                                float(self.eps), _lib.current_stream(w.device)),
                           "spectral_power_iter")
            return
        for _ in range(n):
            self._u = F.normalize((w @ self._v.unsqueeze(1)).squeeze(1), dim=0, eps=self.eps, out=self._u)
            self._v = F.normalize((w.t() @ self._u.unsqueeze(1)).squeeze(1), dim=0, eps=self.eps, out=self._v)

    def forward(self, weight):
        with torch.autocast(device_type=weight.device.type, enabled=False):
            w = weight.float().flatten(1)
            if self.training:
                self._power_method(w, self.n_power_iterations)
            u, v = self._u.clone(), self._v.clone()
            sigma = (u * (w @ v.unsqueeze(1)).squeeze(1)).sum()
            return weight / sigma


def spectral_norm(module, name="weight", n_power_iterations=1, eps=1e-12):
    parametrize.register_parametrization(module, name, _SpectralNorm(getattr(module, name), n_power_iterations, eps))
    return module


def _conv_kx1_cl(x, weight, bias, stride, pad):
    """Conv2d with a (k,1) kernel, stride (s,1), zero padding (pad,0) on CHANNEL-LAST input
    x (B, P, T, Cin) -> (B, P, T_out, Cout), evaluated as unfold + GEMM.  MIOpen runs these
    (5,1)/(3,1) bf16 convolutions with its `naive_conv_*` fallback (40+ ms per call on MI355X);
    as GEMMs (K = Cin*k up to 5120) they run on the MFMA pipes through hipBLASLt."""
    k = weight.shape[2]
    if pad:
        x = F.pad(x, (0, 0, pad, pad))
    cols = x.unfold(2, k, stride)                      # (B, P, T_out, Cin, k) view
    Bn, P, To, Cin, _ = cols.shape
    w = weight[:, :, :, 0].reshape(weight.shape[0], Cin * k)   # (Cout, Cin*k), (c,k) order; cast inside linear()
    y = _linear(cols.reshape(Bn, P, To, Cin * k), w, bias)
    return y


class PeriodDiscriminator(nn.Module):
    def __init__(self, period, kernel_size=5, stride=3, use_spectral_norm=False, hidden=32):
        super().__init__()
        self.period = period
        norm = weight_norm if use_spectral_norm else spectral_norm  # sic
        pad = (kernel_size - 1) // 2
        chans = [1, hidden, hidden * 4, hidden * 16, hidden * 32]
        layers = [norm(nn.Conv2d(chans[i], chans[i + 1], (kernel_size, 1), (stride, 1), padding=(pad, 0)))
                  for i in range(4)]
        layers.append(norm(nn.Conv2d(hidden * 32, hidden * 32, (kernel_size, 1), 1, padding=(2, 0))))
        self.layers = nn.ModuleList(layers)
        self.conv_post = norm(nn.Conv2d(hidden * 32, 1, (3, 1), 1, padding=(1, 0)))

    def forward(self, x, detach_weights=False):
        """`detach_weights`: use the (spectrally normalised) weights as constants — the generator's pass
        through the discriminator, where no discriminator gradient is wanted.
        Feature maps are returned channel-last (B, period, T', C): the losses that consume them
        (L1 feature matching, LSGAN means) are layout-agnostic; the flattened score matches the
        reference's element set."""
        fmap = []
        b, c, t = x.shape
        if t % self.period != 0:
            n_pad = self.period - (t % self.period)
            x = F.pad(x, (0, n_pad), "reflect")
            t = t + n_pad
        wb = (lambda l: (l.weight.detach(), l.bias.detach())) if detach_weights else (lambda l: (l.weight, l.bias))
        if not x.is_cuda:  # host/CPU runs (tests, cpu_baseline) keep the plain convolutions
            x = x.view(b, c, t // self.period, self.period)
            for layer in list(self.layers) + [self.conv_post]:
                w, bias = wb(layer)
                x = F.conv2d(x, w, bias, layer.stride, layer.padding)
                if layer is not self.conv_post:
                    x = F.gelu(x)
                fmap.append(x)
            return torch.flatten(x, 1, -1), fmap
        x = x.view(b, c, t // self.period, self.period).permute(0, 3, 2, 1)  # (B, P, T', C=1)
        for layer in self.layers:
            w, bias = wb(layer)
            x = F.gelu(_conv_kx1_cl(x, w, bias, layer.stride[0], layer.padding[0]))
            fmap.append(x)
        w, bias = wb(self.conv_post)
        x = _conv_kx1_cl(x, w, bias, 1, self.conv_post.padding[0])
        fmap.append(x)
        return torch.flatten(x, 1, -1), fmap


class MultiPeriodDiscriminator(nn.Module):
    def __init__(self, hidden=32, periods=(2, 3, 5, 7, 11)):
        super().__init__()
        self.discriminators = nn.ModuleList([PeriodDiscriminator(p, hidden=hidden) for p in periods])

    def forward_single(self, x, detach_weights=False):
        """scores and feature maps of ONE signal batch (used for the generator pass, where the
        real-signal features of the discriminator pass are reused instead of recomputed)."""
        res = _parallel([(lambda d=d: d(x, detach_weights)) for d in self.discriminators], x.device, "d")
        return [r[0] for r in res], [r[1] for r in res]

    def forward_pair(self, y, y_hat):
        """Same results as forward(y, y_hat) with ONE pass per discriminator over the stacked batch
        [y; y_hat] (same weights, identical per-sample arithmetic, half the kernel launches)."""
        n = y.shape[0]
        y_real, y_gen, fmap_real, fmap_gen = [], [], [], []
        both = torch.cat((y, y_hat), dim=0)
        for s, f in _parallel([(lambda d=d: d(both)) for d in self.discriminators], y.device, "d"):
            y_real.append(s[:n]); y_gen.append(s[n:])
            fmap_real.append([t[:n] for t in f]); fmap_gen.append([t[n:] for t in f])
        return y_real, y_gen, fmap_real, fmap_gen

    def spectral_norms(self):
        return [m for m in self.modules() if isinstance(m, _SpectralNorm)]

    def forward(self, y, y_hat):
        y_real, y_gen, fmap_real, fmap_gen = [], [], [], []
        for disc in self.discriminators:
            r, fr = disc(y)
            y_real.append(r)
            fmap_real.append(fr)
            if y_hat is not None:
                g, fg = disc(y_hat)
                y_gen.append(g)
                fmap_gen.append(fg)
            else:
                y_gen.append(0)
                fmap_gen.append(0)
        return y_real, y_gen, fmap_real, fmap_gen
