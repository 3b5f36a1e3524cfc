"""Multi-period discriminator: the adversary of the MPD configs.

Re-statement of model/discriminator.py:21-147 (HiFi-GAN style, periods 2,3,5,7,11, hidden 32
-> 41.09 M parameters).  The reference's inverted ternary (`weight_norm if use_spectral_norm
else spectral_norm`, :37) means the default `use_spectral_norm=False` yields SPECTRAL norm;
that is reproduced so state_dicts (parametrizations.weight.original + power-iteration
buffers) stay compatible.  MSD (:174-337) is not enabled by any yaml and is not built.

How it runs on the GPU (MIOpen has only `naive_conv_*` fallbacks for these (k,1) convolutions):

  * signals folded to channel-last sequences (B, period, T/period, C); every convolution is
    `vmasr_im2col_kx1` (HIP gather) + a hipBLASLt GEMM + `vmasr_col2im_kx1` in the backward;
  * the five period discriminators run layer by layer on stacked operands — one batched GEMM per
    layer (`_forward_batched`); the one-by-one path (`PeriodDiscriminator.forward`) is the same
    arithmetic and is what `forward(y, y_hat)` (the reference's call) and the CPU use;
  * spectral norm: power iteration and sigma for all 30 weights in one launch per phase
    (`SpectralBatch` -> `vmasr_spectral_power_iter_batched`), W / sigma as one autograd function;
  * `_ConvKx1Fn` (GEMMs on shifted views, no im2col) is an exact alternative kept opt-in
    (`VMASR_MPD_CONV=gemm`): measured slower than im2col + one GEMM.
"""
import os

import torch
import torch.nn as nn
import torch.nn.functional as F
from torch.nn.utils import parametrize
from torch.nn.utils.parametrizations import weight_norm

from . import _lib
from .linear import _mm_acc, linear as _linear, weight_grad as _weight_grad

__all__ = ["PeriodDiscriminator", "MultiPeriodDiscriminator", "spectral_norm", "plain_torch_ops"]

_PLAIN_OPS = [False]


class plain_torch_ops:
    """Inside this context the discriminator runs on plain torch operators (F.conv2d) on any device instead of the
    HIP im2col + GEMM functions.  Those launch raw kernels in their backward and are therefore differentiable
    ONCE; the WGAN-GP gradient penalty (model/loss.py:237-260) differentiates the discriminator TWICE
    (`autograd.grad(create_graph=True)`), which only the plain operators support."""

    def __enter__(self):
        self._saved = _PLAIN_OPS[0]
        _PLAIN_OPS[0] = True
        return self

    def __exit__(self, *exc):
        _PLAIN_OPS[0] = self._saved
        return False


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
            pre = getattr(self, "_sigma_pre", None)
            if pre is not None and self.training and self.n_power_iterations == 0 and weight.dtype == torch.float32:
                # u, v and sigma = u^T W v of this step come from the batched launch (SpectralBatch.run)
                return _SNDivFn.apply(weight, self._u, self._v, pre)
            w = (weight if weight.dtype == torch.float64 else weight.float()).flatten(1)   # float64: tests' adjudicator
            if self.training:
                self._power_method(w, self.n_power_iterations)
            u, v = self._u.clone(), self._v.clone()
            sigma = (u * (w @ v.unsqueeze(1)).squeeze(1)).sum()
            return weight / sigma


class _SNDivFn(torch.autograd.Function):
    """W / sigma with sigma = u^T W v precomputed (u, v constants, as in torch's spectral_norm):
    dL/dW = (g - <g, W/sigma> u v^T) / sigma  — two passes over the weight instead of the GEMV + outer-product
    GEMM + five elementwise kernels autograd needs for the same expression."""

    @staticmethod
    def forward(ctx, weight, u, v, sigma):
        out = weight / sigma
        ctx.save_for_backward(out, u, v, sigma)
        return out

    @staticmethod
    def backward(ctx, g):
        out, u, v, sigma = ctx.saved_tensors
        g2, o2 = g.reshape(u.numel(), -1), out.reshape(u.numel(), -1)
        s = torch.dot(g2.reshape(-1), o2.reshape(-1))
        gw = torch.addcmul(g2, (u * (-s)).unsqueeze(1), v.unsqueeze(0)) / sigma
        return gw.view_as(out), None, None, None


def _ptr_array(tensors):
    import ctypes
    return (ctypes.c_void_p * len(tensors))(*[ctypes.c_void_p(t.data_ptr()) for t in tensors])


class _SNStackFn(torch.autograd.Function):
    """The spectrally normalised weights of one layer of all n period discriminators as ONE (n, N, k*Cin) GEMM operand in
    (tap, channel) column order: out[s] = permute(W_s / sigma_s) with sigma_s, u_s, v_s from the batched power iteration
    (constants, as in torch's spectral_norm).  One launch forward, two backward (csrc/spectral.hip) instead of n divisions +
    a stack and, per weight, a dot product, an outer-product update and a division."""

    @staticmethod
    def forward(ctx, n, *args):
        sig, us, vs, ws = args[:n], args[n:2 * n], args[2 * n:3 * n], args[3 * n:]
        N, Cin, k = ws[0].shape[0], ws[0].shape[1], ws[0].shape[2]
        dev = ws[0].device
        wc = [w.detach().contiguous() for w in ws]
        with torch.cuda.device(dev):
            out = torch.empty((n, N, k * Cin), dtype=torch.float32, device=dev)
            _lib.check(_lib.lib().vmasr_sn_stack_fwd(_ptr_array(wc), _ptr_array(sig), n, out.data_ptr(), N, Cin, k,
                                                     _lib.current_stream(dev)), "sn_stack_fwd")
        ctx.save_for_backward(out, *sig, *us, *vs)
        ctx.geom = (n, N, Cin, k, [w.shape for w in ws])
        return out

    @staticmethod
    def backward(ctx, dW):
        n, N, Cin, k, shapes = ctx.geom
        out, *rest = ctx.saved_tensors
        sig, us, vs = rest[:n], rest[n:2 * n], rest[2 * n:3 * n]
        dW = dW.float().contiguous()
        lib, dev = _lib.lib(), dW.device
        with torch.cuda.device(dev):
            gws = [torch.empty(shp, dtype=torch.float32, device=dev) for shp in shapes]
            partials = torch.empty(n * lib.vmasr_sn_dot_blocks(), dtype=torch.float64, device=dev)
            _lib.check(lib.vmasr_sn_stack_bwd(dW.data_ptr(), out.data_ptr(), _ptr_array(gws), _ptr_array(sig), _ptr_array(us), _ptr_array(vs),
                                              n, partials.data_ptr(), N, Cin, k, _lib.current_stream(dev)), "sn_stack_bwd")
        return (None, *([None] * (3 * n)), *gws)


def _sn_stack(layers):
    """_SNStackFn over the n same-shaped spectrally normalised convolutions `layers`, or None when they do not qualify
    (sigmas not precomputed by SpectralBatch.run, eval mode, other dtypes / devices): the caller then stacks `l.weight`."""
    sns, origs = [], []
    for l in layers:
        if not (isinstance(l, nn.Conv2d) and parametrize.is_parametrized(l, "weight")):
            return None
        sn, w = l.parametrizations.weight[0], l.parametrizations.weight.original
        if not (isinstance(sn, _SpectralNorm) and getattr(sn, "_sigma_pre", None) is not None and sn.training and sn.n_power_iterations == 0
                and w.is_cuda and w.dtype == torch.float32 and w.dim() == 4 and w.shape[3] == 1):
            return None
        sns.append(sn); origs.append(w)
    n = len(layers)
    if n > 8 or any(w.shape != origs[0].shape for w in origs) or origs[0].shape[1] * origs[0].shape[2] * 4 > 60 * 1024:
        return None
    if n > 1 and (origs[0].numel() % 4):
        return None
    return _SNStackFn.apply(n, *[sn._sigma_pre for sn in sns], *[sn._u for sn in sns], *[sn._v for sn in sns], *origs)


class SpectralBatch:
    """Power iteration of MANY _SpectralNorm modules in one launch per phase
    (vmasr_spectral_power_iter_batched): the descriptor table (pointers to the fp32 weights, u, v and
    scratch) is built once on the device; the pointers are those of parameters and buffers, which live
    at fixed addresses for the life of the model on its device."""

    def __init__(self, modules, weights):
        import numpy as np
        assert 0 < len(modules) <= 64
        dev = weights[0].device
        self.modules, self.eps = list(modules), float(modules[0].eps)
        mats = [w.detach() for w in weights]
        assert all(w.is_cuda and w.dtype == torch.float32 and w.is_contiguous() for w in mats)
        shapes = [(w.shape[0], w[0].numel()) for w in mats]
        self.ws = torch.zeros(sum(r + c for r, c in shapes), dtype=torch.float32, device=dev)
        item = np.dtype([("W", "u8"), ("u", "u8"), ("v", "u8"), ("t", "u8"), ("s", "u8"),
                         ("R", "i4"), ("C", "i4"), ("rb", "i4"), ("ct", "i4")])
        tab = np.zeros(len(mats), dtype=item)
        off = rb = ct = 0
        for i, (m, w, (r, c)) in enumerate(zip(self.modules, mats, shapes)):
            tab[i] = (w.data_ptr(), m._u.data_ptr(), m._v.data_ptr(), self.ws.data_ptr() + 4 * off,
                      self.ws.data_ptr() + 4 * (off + r), r, c, rb, ct)
            off += r + c
            rb += -(-r // 4)
            ct += -(-c // 1024) * -(-r // 32)
        self.n, self.row_blocks, self.col_tiles = len(mats), rb, ct
        self.weight_bytes = sum(4 * r * c for r, c in shapes)
        self.ptrs = [(w.data_ptr(), m._u.data_ptr(), m._v.data_ptr()) for m, w in zip(self.modules, mats)]
        self.sigma = torch.ones(len(mats), dtype=torch.float32, device=dev)
        self.table = torch.from_numpy(tab.view(np.uint8).copy()).to(dev)

    def matches(self, weights):
        return len(weights) == self.n and all(
            (w.data_ptr(), m._u.data_ptr(), m._v.data_ptr()) == p for m, w, p in zip(self.modules, weights, self.ptrs))

    @torch.no_grad()
    def run(self, n_iter, with_sigma=False):
        """n_iter power iterations of every matrix; with_sigma: also sigma_m = u^T W v, handed to the modules
        (`_sigma_pre`, a view of self.sigma) until clear_sigma()."""
        dev = self.table.device
        with torch.cuda.device(dev):
            _lib.check(_lib.lib().vmasr_spectral_power_iter_batched(
                self.table.data_ptr(), self.n, self.row_blocks, self.col_tiles, self.weight_bytes, int(n_iter), self.eps,
                self.sigma.data_ptr() if with_sigma else None, _lib.current_stream(dev)), "spectral_power_iter_batched")
        if with_sigma:
            for i, m in enumerate(self.modules):
                object.__setattr__(m, "_sigma_pre", self.sigma[i])

    def clear_sigma(self):
        for m in self.modules:
            object.__setattr__(m, "_sigma_pre", None)


def spectral_norm(module, name="weight", n_power_iterations=1, eps=1e-12):
    parametrize.register_parametrization(module, name, _SpectralNorm(getattr(module, name), n_power_iterations, eps))
    return module


class _Im2ColFn(torch.autograd.Function):
    """x (B, P, H, C) channel-last -> columns (B, P, H1, k*C), (tap, channel) order, zero padding implicit:
    the HIP gather / adjoint-gather kernels of vm_asr_amd/csrc/im2col.hip."""

    @staticmethod
    def forward(ctx, x, k, stride, pad):
        B, P, H, C = x.shape
        xc = x.contiguous()
        H1 = (H + 2 * pad - k) // stride + 1
        with torch.cuda.device(x.device):
            cols = torch.empty((B, P, H1, k * C), dtype=x.dtype, device=x.device)
            _lib.check(_lib.lib().vmasr_im2col_kx1(xc.data_ptr(), cols.data_ptr(), B * P, H, C, k, stride, pad, 0,
                                                   _lib.torch_dtype_code(x.dtype), _lib.current_stream(x.device)), "im2col_kx1")
        ctx.geom = (B, P, H, C, k, stride, pad)
        return cols

    @staticmethod
    def backward(ctx, g):
        B, P, H, C, k, stride, pad = ctx.geom
        g = g.contiguous()
        with torch.cuda.device(g.device):
            dx = torch.empty((B, P, H, C), dtype=g.dtype, device=g.device)
            _lib.check(_lib.lib().vmasr_col2im_kx1(g.data_ptr(), dx.data_ptr(), B * P, H, C, k, stride, pad,
                                                   _lib.torch_dtype_code(g.dtype), _lib.current_stream(g.device)), "col2im_kx1")
        return dx, None, None, None


def _conv_kx1_cl(x, weight, bias, stride, pad):
    """Conv2d with a (k,1) kernel, stride (s,1), zero padding (pad,0) on CHANNEL-LAST input
    x (B, P, T, Cin) -> (B, P, T_out, Cout), evaluated as unfold + GEMM.  MIOpen runs these
    (5,1)/(3,1) bf16 convolutions with its `naive_conv_*` fallback (40+ ms per call on MI355X);
    as GEMMs (K = Cin*k up to 5120) they run on the MFMA pipes through hipBLASLt."""
    k = weight.shape[2]
    if x.is_cuda and x.dtype in (torch.float32, torch.float16, torch.bfloat16) and x.shape[2] + 2 * pad >= k:
        cols = _Im2ColFn.apply(x, k, stride, pad)      # (B, P, T_out, k*Cin), (tap, c) order, HIP gather
        w = weight[:, :, :, 0].permute(0, 2, 1).reshape(weight.shape[0], -1)
        return _linear(cols, w, bias)
    if pad:
        x = F.pad(x, (0, 0, pad, pad))
    cols = x.unfold(2, k, stride)                      # (B, P, T_out, Cin, k) view
    Bn, P, To, Cin, _ = cols.shape
    w = weight[:, :, :, 0].reshape(weight.shape[0], Cin * k)   # (Cout, Cin*k), (c,k) order; cast inside linear()
    y = _linear(cols.reshape(Bn, P, To, Cin * k), w, bias)
    return y


class _ConvKx1Fn(torch.autograd.Function):
    """The same convolution with NO im2col: on the (B*P sequences, T, C) channel-last layout the taps of a
    (k,1) kernel are row-shifted views of the zero-padded input, so the convolution is a few GEMMs on views.

      stride 3, k 5:  Xp (N, 3*Hq, C) viewed as V (N*Hq, 3C):   Y = V @ W[taps 0-2]^T ;  Y[:-1] += V[1:, :2C] @ W[taps 3-4]^T
      stride 1, k:    Xp (N*Hp, C):                              Y[:R] = sum_j Xp[j:j+R] @ W[tap j]^T,   R = N*Hp - (k-1)

    Rows that straddle two sequences are junk and lie exactly in the rows the valid-output view drops.
    unfold + GEMM reads and writes 5/3x (stride 3) or 5x (stride 1) the input as columns and scatters it back
    in the backward (`_unfold_backward`: 2.7 ms/step); here the only copy is the zero padding (1x).
    The backward is the same GEMMs transposed; dW is accumulated in fp32 (split over rows where it is one tile)."""

    @staticmethod
    def forward(ctx, x, weight, bias, stride, pad, cdt):
        B, P, H, C = x.shape
        Cout, _, k, _ = weight.shape
        N = B * P
        H1 = (H + 2 * pad - k) // stride + 1
        w = weight.detach()[:, :, :, 0].to(cdt)                       # (Cout, C, k)
        bc = None if bias is None else bias.detach().to(cdt)
        if stride == 3:
            Hq = H1 + 1
            Hp = 3 * Hq
            xp = F.pad(x.detach().reshape(N, H, C).to(cdt), (0, 0, pad, Hp - pad - H))   # negative = crop unused tail
            V = xp.view(N * Hq, 3 * C)
            Wa = w[:, :, 0:3].permute(0, 2, 1).reshape(Cout, 3 * C)   # (tap, c) order = V's column order
            Wb = w[:, :, 3:5].permute(0, 2, 1).reshape(Cout, 2 * C)
            Y = torch.addmm(bc, V, Wa.t()) if bc is not None else V @ Wa.t()
            Y[:-1].addmm_(V[1:, :2 * C], Wb.t())
            ctx.save_for_backward(xp, Wa, Wb)
            rows = Hq
        else:
            Hp = H + 2 * pad
            xp = F.pad(x.detach().reshape(N, H, C).to(cdt), (0, 0, pad, pad)).view(N * Hp, C)
            R = N * Hp - (k - 1)
            Wt = w.permute(2, 0, 1).contiguous()                       # (k, Cout, C)
            Y = torch.empty((N * Hp, Cout), dtype=cdt, device=x.device)
            Y[R:].zero_()
            if bc is not None:
                torch.addmm(bc, xp[0:R], Wt[0].t(), out=Y[:R])
            else:
                torch.mm(xp[0:R], Wt[0].t(), out=Y[:R])
            for j in range(1, k):
                Y[:R].addmm_(xp[j:j + R], Wt[j].t())
            ctx.save_for_backward(xp, Wt)
            rows = Hp
        ctx.meta = (x.shape, x.dtype, weight.dtype, None if bias is None else bias.dtype, stride, pad, k, H1, rows)
        return Y.view(B, P, rows, Cout)[:, :, :H1]

    @staticmethod
    def backward(ctx, gy):
        (B, P, H, C), xdt, wdt, bdt, stride, pad, k, H1, rows = ctx.meta
        N = B * P
        Cout = gy.shape[-1]
        cdt = ctx.saved_tensors[0].dtype
        g = F.pad(gy.reshape(N, H1, Cout).to(cdt), (0, 0, 0, rows - H1)).view(N * rows, Cout)   # junk rows: zero gradient
        need_x, need_w, need_b = ctx.needs_input_grad[0], ctx.needs_input_grad[1], ctx.needs_input_grad[2]
        dx = dw = db = None
        if stride == 3:
            xp, Wa, Wb = ctx.saved_tensors
            V = xp.view(N * rows, 3 * C)
            if need_x:
                dV = g @ Wa
                dV[1:, :2 * C].addmm_(g[:-1], Wb)
                dxp = dV.view(N, 3 * rows, C)
            if need_w:
                dWa = _weight_grad(g, V).view(Cout, 3, C)
                dWb = _weight_grad(g[:-1], V[1:, :2 * C]).view(Cout, 2, C)
                dw = torch.cat((dWa, dWb), dim=1).permute(0, 2, 1).unsqueeze(-1).to(wdt)
            Hp = 3 * rows
        else:
            xp, Wt = ctx.saved_tensors
            Hp = rows
            R = N * Hp - (k - 1)
            if need_x:
                dX = torch.zeros((N * Hp, C), dtype=cdt, device=gy.device)
                for j in range(k):
                    dX[j:j + R].addmm_(g[:R], Wt[j])
                dxp = dX.view(N, Hp, C)
            if need_w:
                dw = torch.stack([_weight_grad(g[:R], xp[j:j + R]) for j in range(k)], dim=2).unsqueeze(-1).to(wdt)
        if need_x:
            avail = min(H, Hp - pad)
            dx = dxp[:, pad:pad + avail]
            if avail < H:
                dx = F.pad(dx, (0, 0, 0, H - avail))
            dx = dx.reshape(B, P, H, C).to(xdt)
        if need_b and bdt is not None:
            db = g.sum(0, dtype=torch.float32 if cdt in (torch.float16, torch.bfloat16) else None).to(bdt)
        return dx, dw, db, None, None, None


def conv_kx1(x, weight, bias, stride, pad):
    """(k,1) convolution of channel-last x (B, P, T, Cin) -> (B, P, T_out, Cout): the im2col-free GEMM form
    for the discriminator's two shapes (k 5 / stride 3, and stride 1), unfold + GEMM otherwise."""
    k = weight.shape[2]
    mode = os.environ.get("VMASR_MPD_CONV", "unfold")    # gemm | s3 (stride-3 layers only) | unfold (default: measured fastest)
    ok = (stride == 3 and k == 5 and mode in ("gemm", "s3")) or (stride == 1 and mode == "gemm")
    if x.is_cuda and ok and x.shape[2] + 2 * pad >= k:
        cdt = torch.get_autocast_dtype("cuda") if torch.is_autocast_enabled("cuda") else x.dtype
        return _ConvKx1Fn.apply(x, weight, bias, stride, pad, cdt)
    return _conv_kx1_cl(x, weight, bias, stride, pad)


# ---- all period discriminators, layer by layer (stacked GEMM operands) ---------------------------
# The five period discriminators have the same layer shapes and nearly the same number of GEMM rows
# (B*p*T'_p ~ B*T/3^l for every p), but run one after the other each of their GEMMs fills a fraction of the
# 256 CUs (M ~ 4.7 k rows x N = 1024: 76 tiles of 256x256).  Stacked into one batched GEMM per layer they
# fill the chip, and GELU / bias / weight casts run once per layer instead of once per discriminator.

def _round_up(v, m):
    return -(-v // m) * m


def _slot_arrays(ptrs, Ns, Hs=None):
    """ctypes host arrays (device pointers, per-slot sizes) of the multi-slot entry points (include/vmasr_hip.h)."""
    import ctypes
    n = len(ptrs)
    a = (ctypes.c_void_p * n)(*[ctypes.c_void_p(p) if p else None for p in ptrs])
    b = (ctypes.c_int64 * n)(*Ns)
    c = (ctypes.c_int32 * n)(*Hs) if Hs is not None else None
    return a, b, c


class _StackedIm2ColFn(torch.autograd.Function):
    """n channel-last inputs (B, P_i, H_i, C) -> one (n, rows, k*C) column tensor, slot i holding the im2col
    of input i in its first B*P_i*H1_i rows and zeros below (vmasr_im2col_kx1 with rows_out).
    geom = ((N_i, H_i), ...) : the n inputs are the slots of ONE stacked tensor xs[0] (n, rows_in, C), slot i holding N_i
    sequences of H_i positions in its first rows (the previous layer's stacked output); the backward then writes the
    stacked gradient directly (no per-slot tensors, no re-stacking)."""

    @staticmethod
    def forward(ctx, k, stride, pad, rows, geom, *xs):
        lib = _lib.lib()
        if geom is not None:
            x = xs[0].contiguous()
            n, rows_in, C = x.shape
            dt, dev, step = x.dtype, x.device, rows_in * C * x.element_size()
            srcs = [(x.data_ptr() + i * step, N, H) for i, (N, H) in enumerate(geom)]
            ctx.geom = (k, stride, pad, None, tuple(geom), (n, rows_in, C))
        else:
            C, dt, dev = xs[0].shape[3], xs[0].dtype, xs[0].device
            xcs = [x.contiguous() for x in xs]
            srcs = [(x.data_ptr(), x.shape[0] * x.shape[1], x.shape[2]) for x in xcs]
            ctx.geom = (k, stride, pad, [tuple(x.shape) for x in xs], None, None)
        with torch.cuda.device(dev):
            cols = torch.empty((len(srcs), rows, k * C), dtype=dt, device=dev)
            for i, (ptr, N, H) in enumerate(srcs):
                _lib.check(lib.vmasr_im2col_kx1(ptr, cols[i].data_ptr(), N, H, C, k, stride, pad, rows,
                                                _lib.torch_dtype_code(dt), _lib.current_stream(dev)), "im2col_kx1")
        return cols

    @staticmethod
    def backward(ctx, g):
        k, stride, pad, shapes, geom, sshape = ctx.geom
        g = g.contiguous()
        lib = _lib.lib()
        with torch.cuda.device(g.device):
            if geom is not None:
                n, rows_in, C = sshape
                dx = torch.empty(sshape, dtype=g.dtype, device=g.device)
                _, Ns, Hs = _slot_arrays([0] * n, [N for N, _ in geom], [H for _, H in geom])
                _lib.check(lib.vmasr_col2im_kx1_stacked(g.data_ptr(), dx.data_ptr(), Ns, Hs, n, C, k, stride, pad, g.shape[1], rows_in,
                                                        _lib.torch_dtype_code(g.dtype), _lib.current_stream(g.device)), "col2im_kx1_stacked")
                return (None, None, None, None, None, dx)
            dxs = [torch.empty(shp, dtype=g.dtype, device=g.device) for shp in shapes]
            ptrs, Ns, Hs = _slot_arrays([d.data_ptr() for d in dxs], [B * P for B, P, _, _ in shapes], [H for _, _, H, _ in shapes])
            _lib.check(lib.vmasr_col2im_kx1_multi(g.data_ptr(), ptrs, Ns, Hs, len(shapes), shapes[0][3], k, stride, pad, g.shape[1],
                                                  _lib.torch_dtype_code(g.dtype), _lib.current_stream(g.device)), "col2im_kx1_multi")
        return (None, None, None, None, None, *dxs)


# Backward-phase switch of the trainer's shared fake pass: while the GENERATOR loss is back-propagated through
# the discriminator's graph only the column / input gradients are wanted; the weight gradients belong to the
# discriminator loss' own backward through the same graph.
_PHASE = {"skip_weight_grads": False, "scores_only": False}


class skip_weight_grads:
    def __enter__(self):
        _PHASE["skip_weight_grads"] = True

    def __exit__(self, *exc):
        _PHASE["skip_weight_grads"] = False


class scores_only:
    """with scores_only(): the loss being back-propagated reads the discriminator's SCORES only (the discriminator loss of
    model/loss.py:190-213), no feature map: a map's only consumer is then the layer above it, which may finish the layer's activation
    backward — GELU', bf16 split, bias-gradient column sums — in its input-gradient epilogue (_StackedConvMfmaFn._fuse_below)."""

    def __enter__(self):
        _PHASE["scores_only"] = True

    def __exit__(self, *exc):
        _PHASE["scores_only"] = False


class _BatchedLinearFn(torch.autograd.Function):
    """y[i] = cols[i] @ W[i]^T + b[i] for the n stacked discriminators (one batched GEMM); backward: one batched
    GEMM for the column gradient, the weight gradient split over the rows into a larger batch (fp32 sum)."""

    @staticmethod
    def forward(ctx, cols, weight, bias, cdt, act=False):
        """act: GELU on the output; for fp32 operands on the GPU the bias + GELU epilogue and, in the backward, GELU' + the
        bias gradient are single passes (csrc/split.hip) instead of add_, gelu, gelu_backward and a column sum."""
        wc = weight.detach().to(cdt)                                   # (n, N, K): the operand of dcols = gy @ W
        # The forward operand is a CONTIGUOUS (n, K, N) copy: batched bf16 GEMMs with a transposed-view B operand
        # fault the GPU on ROCm 7.2 / hipBLASLt for e.g. (5, 36608, 640) x (5, 640, 512)^T (tools/bmm_probe.py);
        # contiguous-B ("NN") and transposed-A ("TN", the weight gradient) forms are fine at every MPD shape.
        y = torch.bmm(cols, wc.transpose(1, 2).contiguous())
        n, M, N = y.shape
        fused = act and y.is_cuda and y.dtype == torch.float32 and N % 4 == 0 and N <= 1024
        pre = None
        if fused:
            pre = y
            with torch.cuda.device(y.device):
                y = torch.empty_like(pre)
                _lib.check(_lib.lib().vmasr_bias_gelu_fwd(pre.data_ptr(), bias.detach().float().contiguous().data_ptr(), y.data_ptr(),
                                                          n, M, N, 1, _lib.current_stream(y.device)), "bias_gelu_fwd")
        else:
            y.add_(bias.detach().to(cdt).unsqueeze(1))
            if act:
                pre = y
                y = F.gelu(pre)
        ctx.save_for_backward(cols, wc, *([pre] if pre is not None else []))
        ctx.meta = (weight.dtype, bias.dtype, act, fused)
        return y

    @staticmethod
    def backward(ctx, gy):
        cols, wc, *rest = ctx.saved_tensors
        wdt, bdt, act, fused = ctx.meta
        gy = gy.contiguous()
        n, M, N = gy.shape
        K = cols.shape[2]
        skip_w = _PHASE["skip_weight_grads"]
        db = None
        if fused:
            want_db = ctx.needs_input_grad[2] and not skip_w
            with torch.cuda.device(gy.device):
                gx = torch.empty_like(gy)
                db32, = _lib.zeros_f32(gy.device, (n, N) if want_db else None)
                _lib.check(_lib.lib().vmasr_gelu_bwd(rest[0].data_ptr(), gy.data_ptr(), gx.data_ptr(), db32.data_ptr() if want_db else None,
                                                     n, M, N, _lib.current_stream(gy.device)), "gelu_bwd")
            gy = gx
            db = db32.to(bdt) if want_db else None
        elif act:
            gy = torch.ops.aten.gelu_backward(gy, rest[0])
        dcols = torch.bmm(gy, wc) if ctx.needs_input_grad[0] else None
        dw = None
        if skip_w:
            return dcols, None, None, None, None
        if ctx.needs_input_grad[1]:
            acc = torch.float32 if gy.dtype in (torch.float16, torch.bfloat16) else gy.dtype
            tiles = n * -(-N // 64) * -(-K // 64)
            want = min(M // 2048, max(1, 512 // tiles))
            S = max(d for d in range(1, max(1, want) + 1) if (M // 256) % d == 0) if M % 256 == 0 else 1
            if S > 1:   # (n, S, M/S, .) -> batch n*S: the row split is a free view because S divides M
                part = _mm_acc(gy.view(n * S, M // S, N).transpose(1, 2), cols.view(n * S, M // S, K), acc)
                dw = part.view(n, S, N, K).sum(1)
            else:
                dw = _mm_acc(gy.transpose(1, 2), cols, acc)
            dw = dw.to(wdt)
        if ctx.needs_input_grad[2] and not fused:
            db = gy.sum(1, dtype=torch.float32 if gy.dtype in (torch.float16, torch.bfloat16) else None).to(bdt)
        return dcols, dw, db, None, None


def split_bf16(x):
    """fp32 tensor -> (hi, lo) bf16 with x = hi + lo up to 2^-17 |x| (vm_asr_amd/csrc/split.hip)."""
    x = x.contiguous()
    with torch.cuda.device(x.device):
        hi = torch.empty(x.shape, dtype=torch.bfloat16, device=x.device)
        lo = torch.empty(x.shape, dtype=torch.bfloat16, device=x.device)
        _lib.check(_lib.lib().vmasr_split_bf16(x.data_ptr(), hi.data_ptr(), lo.data_ptr(), x.numel(),
                                               _lib.current_stream(x.device)), "split_bf16")
    return hi, lo


def _bmm3(ah, al, bh, bl):
    """(ah + al) @ (bh + bl) without the lo*lo term: three bf16 MFMA GEMMs, fp32 output and accumulation."""
    f32 = torch.float32
    y = torch.bmm(ah, bh, out_dtype=f32)
    y += torch.bmm(al, bh, out_dtype=f32)
    y += torch.bmm(ah, bl, out_dtype=f32)
    return y


def _split_k(n, N, K, M):
    """Split factor S of the weight-gradient GEMM's contraction (M rows): hipBLASLt runs these as 256x256 macro
    tiles, so a (N, K) output with few tiles leaves most of the 256 CUs idle unless the contraction is spread over
    S batches.  Measured on MI355X (tools/bench_gemm.py, profiles/r02_gemm_layouts.log): 512x640 (30 tiles for five
    slots) 203 us at S=1, 80 us at S=8; 1024x5120 (400 tiles = 1.56 waves of CUs) 491 us at S=1, 362 us at S=3;
    1024x2560 is flat (190 / 182 us)."""
    if M % 256:
        return 1
    tiles = n * -(-N // 256) * -(-K // 256)
    blocks = M // 256
    if tiles <= 64:
        want = 8
    elif 256 < tiles < 512:
        want = 3
    else:
        return 1
    return max(d for d in range(1, want + 1) if blocks % d == 0)


def _dw3(gh, gl, ch, cl, wdt):
    """dW = (gh + gl)^T (ch + cl) without lo*lo over the stacked rows: the three products of all S contraction
    slabs land in ONE (3, n*S, N, K) buffer that a single reduction sums (instead of two read-modify-write passes
    plus a slab sum)."""
    n, M, N = gh.shape
    K = ch.shape[2]
    S = _split_k(n, N, K, M)
    v = (lambda t: t.view(n * S, M // S, t.shape[2])) if S > 1 else (lambda t: t)
    ght, glt = v(gh).transpose(1, 2), v(gl).transpose(1, 2)
    parts = torch.empty((3, n * S, N, K), dtype=torch.float32, device=gh.device)
    torch.bmm(ght, v(ch), out_dtype=torch.float32, out=parts[0])
    torch.bmm(glt, v(ch), out_dtype=torch.float32, out=parts[1])
    torch.bmm(ght, v(cl), out_dtype=torch.float32, out=parts[2])
    if (N * K) % 4 == 0 and n <= 65535:
        with torch.cuda.device(gh.device):
            dw = torch.empty((n, N, K), dtype=torch.float32, device=gh.device)
            _lib.check(_lib.lib().vmasr_sum_parts(parts.data_ptr(), dw.data_ptr(), 3, n, S, N * K, _lib.current_stream(gh.device)), "sum_parts")
        return dw.to(wdt)
    return parts.view(3, n, S, N, K).sum((0, 2)).to(wdt)


def _split_mode(K, N, cdt):
    """Which GEMMs of the fp32 discriminator run as error-compensated bf16 triples: the compute-bound ones
    (K*N >= 2^18: the 128->512, 512->1024 and 1024->1024 convolutions, 98 % of the FLOPs); the two small-K layers
    are memory-bound and stay plain fp32 GEMMs.  VMASR_MPD_GEMM=fp32 switches the triples off."""
    return (cdt == torch.float32 and K * N >= int(os.environ.get("VMASR_MPD_SPLIT_MIN", str(1 << 18)))
            and os.environ.get("VMASR_MPD_GEMM", "bf16x3") == "bf16x3")


class _BatchedLinearSplitFn(torch.autograd.Function):
    """_BatchedLinearFn for fp32 operands on the bf16 matrix cores: every GEMM (y, dcols, dW) is the
    error-compensated triple hi*hi + lo*hi + hi*lo of bf16 splits (csrc/split.hip), accumulated in fp32 —
    the fp32 result to ~1e-6 relative at 16/3 of the fp32 MFMA rate."""

    @staticmethod
    def forward(ctx, cols, weight, bias):
        ch, cl = split_bf16(cols)                                        # (n, M, K)
        w = weight.detach().float()
        wh, wl = split_bf16(w)                                           # (n, N, K): operand of dcols = gy @ W
        wth, wtl = split_bf16(w.transpose(1, 2).contiguous())            # (n, K, N): contiguous B operand (see _BatchedLinearFn)
        y = _bmm3(ch, cl, wth, wtl).add_(bias.detach().float().unsqueeze(1))
        ctx.save_for_backward(ch, cl, wh, wl)
        ctx.meta = (weight.dtype, bias.dtype)
        return y

    @staticmethod
    def backward(ctx, gy):
        ch, cl, wh, wl = ctx.saved_tensors
        wdt, bdt = ctx.meta
        gy = gy.float().contiguous()
        n, M, N = gy.shape
        K = ch.shape[2]
        gh, gl = split_bf16(gy)
        dcols = _bmm3(gh, gl, wh, wl) if ctx.needs_input_grad[0] else None
        if _PHASE["skip_weight_grads"]:
            return dcols, None, None
        dw = db = None
        if ctx.needs_input_grad[1]:
            dw = _dw3(gh, gl, ch, cl, wdt)
        if ctx.needs_input_grad[2]:
            db = gy.sum(1).to(bdt)
        return dcols, dw, db


class _StackedConvSplitFn(torch.autograd.Function):
    """_StackedIm2ColFn + _BatchedLinearSplitFn as one function for fp32 inputs: the im2col kernel writes the bf16
    hi / lo operands directly (no fp32 column tensor, no separate split pass); the column gradient is ONE GEMM over
    the concatenated contraction [gh | gl | gh] @ [wh; wh; wl] (the three products accumulate inside the GEMM
    instead of two read-modify-write passes over the (rows, k*C) gradient), then col2im per slot."""

    @staticmethod
    def forward(ctx, k, stride, pad, rows, act, geom, weight, bias, *xs):
        """act: apply GELU to the output inside (epilogue kernel; the backward then fuses GELU', the bias gradient
        and the bf16 split of the incoming gradient into one pass, csrc/split.hip).
        geom = ((N_i, H_i), ...): the inputs are the slots of ONE stacked fp32 tensor xs[0] (n, rows_in, C) — the previous
        layer's stacked output — and the backward returns its stacked gradient (see _StackedIm2ColFn)."""
        lib = _lib.lib()
        if geom is not None:
            xs0 = xs[0].float().contiguous()
            n, rows_in, C = xs0.shape
            dev, K = xs0.device, k * C
            xcs = [xs0]
            ptrs, Ns, Hs = _slot_arrays([xs0.data_ptr() + i * rows_in * C * 4 for i in range(n)], [N for N, _ in geom], [H for _, H in geom])
        else:
            C, dev = xs[0].shape[3], xs[0].device
            n, K = len(xs), k * C
            xcs = [x.float().contiguous() for x in xs]
            ptrs, Ns, Hs = _slot_arrays([x.data_ptr() for x in xcs], [x.shape[0] * x.shape[1] for x in xcs], [x.shape[2] for x in xcs])
        w = weight.detach().float().contiguous()
        N = w.shape[1]
        fused = act and N % 4 == 0 and N <= 1024
        kcat = fused and os.environ.get("VMASR_MPD_KCAT", "0") == "1"
        with torch.cuda.device(dev):
            if kcat:
                # (opt-in, VMASR_MPD_KCAT=1 — measured SLOWER in round 3: 38.8 vs 38.0 ms per step.  The epilogue gains 0.36 ms
                #  (one partial product to read instead of three), but im2col writes a third operand block (+0.25 ms) and
                #  hipBLASLt's kernels for K' = 3K with M = 4.7k .. 36k are slower than three K-sized products (+0.9 ms).)
                # ONE operand [hi | lo | hi] (n, rows, 3K): the forward triple as a single GEMM over the concatenated contraction;
                # hi / lo stay addressable as column blocks (ld = 3K) for the weight-gradient GEMMs
                acat = torch.empty((n, rows, 3 * K), dtype=torch.bfloat16, device=dev)
                _lib.check(lib.vmasr_im2col_kx1_split3_multi(ptrs, Ns, Hs, n, acat.data_ptr(), C, k, stride, pad, rows,
                                                             _lib.current_stream(dev)), "im2col_kx1_split3_multi")
                ch, cl = acat[:, :, :K], acat[:, :, K:2 * K]
            else:
                ch = torch.empty((n, rows, K), dtype=torch.bfloat16, device=dev)
                cl = torch.empty((n, rows, K), dtype=torch.bfloat16, device=dev)
                _lib.check(lib.vmasr_im2col_kx1_split_multi(ptrs, Ns, Hs, n, ch.data_ptr(), cl.data_ptr(), C, k, stride, pad, rows,
                                                            _lib.current_stream(dev)), "im2col_kx1_split_multi")
        # weights: one pass to the (n, K, 3N) bf16 operand [hi^T | hi^T | lo^T] (csrc/split.hip): column blocks 0 and 2 are
        # the forward B operands; all of it, transposed, is the [wh; wh; wl] operand of the column-gradient GEMM
        # (kept as the transpose of a contiguous tensor: hipBLASLt's kernels for that layout are ~9 % faster here)
        with torch.cuda.device(dev):
            wcat = torch.empty((n, K, 3 * N), dtype=torch.bfloat16, device=dev)
            _lib.check(lib.vmasr_weight_prep_split(w.data_ptr(), wcat.data_ptr(), n, N, K, _lib.current_stream(dev)), "weight_prep_split")
        wth, wtl = wcat[:, :, :N], wcat[:, :, 2 * N:]
        b32 = bias.detach().float().contiguous()
        pre = None
        if kcat:
            f32 = torch.float32
            with torch.cuda.device(dev):
                wk = torch.cat((wth, wth, wtl), dim=1)                       # (n, 3K, N) = [w_hi^T; w_hi^T; w_lo^T]
                pre = torch.bmm(acat, wk, out_dtype=f32)
                y = torch.empty_like(pre)
                _lib.check(lib.vmasr_bias_gelu_fwd(pre.data_ptr(), b32.data_ptr(), y.data_ptr(), n, rows, N, 1,
                                                   _lib.current_stream(dev)), "bias_gelu_fwd")
        elif fused:
            # the three products side by side; the epilogue sums them, adds the bias (-> pre, in place in part 0) and applies GELU
            f32 = torch.float32
            with torch.cuda.device(dev):
                parts = torch.empty((3, n, rows, N), dtype=f32, device=dev)
                torch.bmm(ch, wth, out_dtype=f32, out=parts[0])
                torch.bmm(cl, wth, out_dtype=f32, out=parts[1])
                torch.bmm(ch, wtl, out_dtype=f32, out=parts[2])
                pre = parts[0]
                y = torch.empty_like(pre)
                _lib.check(lib.vmasr_bias_gelu_fwd(parts.data_ptr(), b32.data_ptr(), y.data_ptr(), n, rows, N, 3,
                                                   _lib.current_stream(dev)), "bias_gelu_fwd")
        else:
            y = _bmm3(ch, cl, wth, wtl)
            y.add_(b32.unsqueeze(1))
            if act:
                pre = y
                y = F.gelu(pre)
        ctx.save_for_backward(ch, cl, wcat, *([pre] if pre is not None else []))
        ctx.geom = (k, stride, pad, [tuple(x.shape) for x in xs], weight.dtype, bias.dtype, [x.dtype for x in xs], act, fused,
                    tuple(geom) if geom is not None else None)
        return y

    @staticmethod
    def backward(ctx, gy):
        ch, cl, wcat, *rest = ctx.saved_tensors
        k, stride, pad, shapes, wdt, bdt, xdts, act, fused, sgeom = ctx.geom
        gy = gy.float().contiguous()
        n, M, N = gy.shape
        K = ch.shape[2]
        lib = _lib.lib()
        want_db = ctx.needs_input_grad[7] and not _PHASE["skip_weight_grads"]
        want_dx = any(ctx.needs_input_grad[8:])
        db32 = gcat = None
        if N % 4 == 0 and N <= 1024:
            # one pass: (GELU' *) gradient -> bf16 split (+ bias gradient); the fp32 gradient is never written
            with torch.cuda.device(gy.device):
                db32, = _lib.zeros_f32(gy.device, (n, N) if want_db else None)
                if want_dx:    # [gh | gl | gh]: the weight-gradient GEMMs read gh, gl as column blocks of it (lda = 3N)
                    gcat = torch.empty((n, M, 3 * N), dtype=torch.bfloat16, device=gy.device)
                    gh, gl = gcat[:, :, :N], gcat[:, :, N:2 * N]
                else:
                    gh = torch.empty((n, M, N), dtype=torch.bfloat16, device=gy.device)
                    gl = torch.empty((n, M, N), dtype=torch.bfloat16, device=gy.device)
                _lib.check(lib.vmasr_gelu_bwd_split(rest[0].data_ptr() if act else None, gy.data_ptr(),
                                                    None if want_dx else gh.data_ptr(), None if want_dx else gl.data_ptr(),
                                                    gcat.data_ptr() if want_dx else None, db32.data_ptr() if want_db else None,
                                                    n, M, N, _lib.current_stream(gy.device)), "gelu_bwd_split")
        else:
            if act:
                gy = torch.ops.aten.gelu_backward(gy, rest[0])
            gh, gl = split_bf16(gy)
            db32 = gy.sum(1) if want_db else None
        dxs = [None] * len(shapes)
        if want_dx:
            if gcat is None:
                gcat = torch.cat((gh, gl, gh), dim=2)
            dcols = torch.bmm(gcat, wcat.transpose(1, 2), out_dtype=torch.float32)
            if sgeom is not None:      # stacked input: its stacked gradient in one launch (zero rows below each slot's data)
                with torch.cuda.device(gy.device):
                    dxs_ = torch.empty(shapes[0], dtype=torch.float32, device=gy.device)
                    _, Ns, Hs = _slot_arrays([0] * n, [N for N, _ in sgeom], [H for _, H in sgeom])
                    _lib.check(lib.vmasr_col2im_kx1_stacked(dcols.data_ptr(), dxs_.data_ptr(), Ns, Hs, n, shapes[0][2], k, stride, pad, M,
                                                            shapes[0][1], _lib.F32, _lib.current_stream(gy.device)), "col2im_kx1_stacked")
                dxs = [dxs_.to(xdts[0])]
        if want_dx and sgeom is None:
            with torch.cuda.device(gy.device):
                outs = [torch.empty(shp, dtype=torch.float32, device=gy.device) if ctx.needs_input_grad[8 + i] else None
                        for i, shp in enumerate(shapes)]
                ptrs, Ns, Hs = _slot_arrays([o.data_ptr() if o is not None else 0 for o in outs],
                                            [B * P for B, P, _, _ in shapes], [H for _, _, H, _ in shapes])
                _lib.check(lib.vmasr_col2im_kx1_multi(dcols.data_ptr(), ptrs, Ns, Hs, n, shapes[0][3], k, stride, pad, M, _lib.F32,
                                                      _lib.current_stream(gy.device)), "col2im_kx1_multi")
                dxs = [o.to(xdts[i]) if o is not None else None for i, o in enumerate(outs)]
        dw = db = None
        if not _PHASE["skip_weight_grads"]:
            if ctx.needs_input_grad[6]:
                dw = _dw3(gh, gl, ch, cl, wdt)
            if ctx.needs_input_grad[7]:
                db = db32.to(bdt)
        return (None, None, None, None, None, None, dw, db, *dxs)


class _StackedConvMfmaFn(torch.autograd.Function):
    """One stacked (k,1) convolution + bias + GELU of the n period discriminators as ONE implicit bf16x3 MFMA GEMM launch
    each way (csrc/convgemm.hip, vm_asr_amd/convgemm.py) — no im2col operand, no partial products, no col2im.
    x (n, rows_in, C) fp32 stacked input (slot i: N_i sequences of H_i positions), pair = its bf16 (hi, lo) split if the
    producing layer already wrote it; W (n, Cout, k*C) fp32 in (tap, channel) order; returns y = GELU(conv + bias) stacked
    (n, rows, Cout) and leaves the pair of y in `out_pair` for the next layer.  wcache: dict shared by the passes of a step
    while the weights are frozen (the split / transposed-split operands of W are built once per step)."""

    @staticmethod
    def forward(ctx, k, stride, pad, rows, geom, wcache, out_pair, weight, bias, x, xh, xl, link=None, below=None):
        """link / below: plain dicts shared with the layer above / below (None: no fusion across this boundary).  The backward of the
        layer ABOVE may finish this layer's activation backward in its input-gradient epilogue (csrc/convgemm.hip EPI 2) and leave the
        result in link["stash"]; this layer's backward then starts from it (see _fuse_below)."""
        from . import convgemm as cg
        x_req = x.requires_grad
        x = x.float().contiguous()
        w = weight.detach().float().contiguous()
        f32 = x.shape[2] < 128 and _l1_mode() == "f32"       # the 32 -> 128 layer: exact-f32 products, forward and input gradient
        ops = wcache.get("ops") if wcache is not None else None
        if ops is None:
            ops = {} if f32 else {"w": split_bf16(w)}
            if wcache is not None:
                wcache["ops"] = ops
        if f32:
            pre, y, yh, yl = cg.conv_fwd_f32(x, w, bias.detach().float().contiguous(), geom, k, stride, pad, rows, act=True)
            xh = xl = x                                       # (the bf16 pair of x is made in the backward, where the weight gradient wants it)
        else:
            if xh is None:
                xh, xl = split_bf16(x)
            wh, wl = ops["w"]
            pre, y, yh, yl = cg.conv_fwd(xh, xl, wh, wl, bias.detach().float().contiguous(), geom, k, stride, pad, rows, act=True)
        out_pair.append((yh, yl))
        ctx.save_for_backward(xh, xl, pre, w)
        ctx.f32 = f32
        ctx.meta = (k, stride, pad, tuple(geom), ops, weight.dtype, bias.dtype, x.shape)
        ctx.link, ctx.below = link, below
        if link is not None:
            link.update(pre=pre, C=x.shape[2], x_req=x_req, w_req=weight.requires_grad, b_req=bias.requires_grad)
        return y

    @staticmethod
    def _fuse_below(ctx, gh, gl, wth, wtl, geom, k, stride, pad, rows_in, skip_w):
        """Input gradient of this layer + the activation backward of the layer below in one launch, if that layer can start from it:
        -> True (result left in below["stash"]; the caller returns a poisoned placeholder as dx) or False (nothing done).
        The bias gradient's column sums and the feature-matching term of the map between the two layers are part of the epilogue."""
        from . import convgemm as cg
        b = ctx.below
        if b is None or "pre" not in b or os.environ.get("VMASR_MPD_FUSE_GELU_BWD", "1") != "1" or det_mode():
            return False
        want_db = b["b_req"] and not skip_w
        want_f32 = b["C"] < 128 and b["x_req"]
        want_pair = (not want_f32 and b["x_req"]) or (b["w_req"] and not skip_w)
        if not (want_f32 or want_pair):
            return False
        # The map between the two layers must have no other consumer (it would receive the poisoned placeholder in autograd's sum):
        # (a) the generator-loss pass with the stacked feature-matching loss — the map's tap holds the sign map AND the loss' backward has
        #     left the upstream gradient: the term goes into the epilogue too; or
        # (b) a pass the caller declared to read scores only (scores_only(): the discriminator loss) — no term, taps pass through.
        h = b.get("tap")
        kw = {}
        if h is not None and h.get("sgn") is not None and h.get("gtok") is not None:
            kw = dict(sgn=h["sgn"], gtok=h["gtok"], scale=h["scale"], valid=h["valid"])
            h["consumed"] = True
        elif not _PHASE["scores_only"]:
            return False
        db32 = None
        if want_db:
            db32, = _lib.zeros_f32(gh.device, (gh.shape[0], wth.shape[1]))
            kw["db"] = db32
        b["stash_db"] = db32
        b["stash"] = cg.conv_dgrad_gelu(gh, gl, wth, wtl, geom, k, stride, pad, rows_in, b["pre"], want_f32=want_f32,
                                        want_pair=want_pair, **kw)
        return True

    @staticmethod
    def backward(ctx, gy):
        from . import convgemm as cg
        xh, xl, pre, w = ctx.saved_tensors
        k, stride, pad, geom, ops, wdt, bdt, xshape = ctx.meta
        n, M, N = gy.shape
        C = xshape[2]
        lib = _lib.lib()
        skip_w = _PHASE["skip_weight_grads"]
        want_db = ctx.needs_input_grad[8] and not skip_w
        # The input gradient of the 32 -> 128 layer stays FP32 arithmetic: it is the last GEMM in front of d(loss)/d(wave), a sum with heavy
        # cancellation, where the pair's 16-17 bits per product showed (2.5e-3 of the gradient's scale from float64 against 4e-4 for fp32 —
        # tests/test_mpd.py holds 5e-4).  Default (ctx.f32): the exact-f32 MFMA implicit GEMM; VMASR_MPD_CONV_L1=1: fp32 library GEMM + col2im.
        # The weight gradient takes the bf16x3 kernel like the other layers
        fp32_dgrad = C < 128 and ctx.needs_input_grad[9]
        need_pair = (not fp32_dgrad and ctx.needs_input_grad[9]) or (ctx.needs_input_grad[7] and not skip_w)
        gh = gl = gx = None
        stash = ctx.link.pop("stash", None) if ctx.link is not None else None
        if stash is not None:      # the layer above has already applied GELU' (and the feature-matching term): gy is a placeholder
            gx, pair_ = stash
            gh, gl = pair_ if pair_ is not None else (None, None)
            db32 = ctx.link.pop("stash_db", None)
            if (want_db and db32 is None) or (need_pair and gh is None) or (fp32_dgrad and gx is None):
                raise RuntimeError("MPD: the fused activation backward left less than this layer's backward needs")
        with torch.cuda.device(gy.device):
            if stash is None:
                gy = gy.float().contiguous()
                db32, = _lib.zeros_f32(gy.device, (n, N) if want_db else None)
            if stash is not None:
                pass
            elif need_pair:
                gh = torch.empty((n, M, N), dtype=torch.bfloat16, device=gy.device)
                gl = torch.empty((n, M, N), dtype=torch.bfloat16, device=gy.device)
                _lib.check(lib.vmasr_gelu_bwd_split(pre.data_ptr(), gy.data_ptr(), gh.data_ptr(), gl.data_ptr(), None,
                                                    db32.data_ptr() if want_db else None, n, M, N, _lib.current_stream(gy.device)),
                           "gelu_bwd_split")
            if fp32_dgrad and stash is None:
                gx = torch.empty_like(gy)
                _lib.check(lib.vmasr_gelu_bwd(pre.data_ptr(), gy.data_ptr(), gx.data_ptr(),
                                              db32.data_ptr() if (want_db and not need_pair) else None, n, M, N,
                                              _lib.current_stream(gy.device)), "gelu_bwd")
        dx = dw = db = None
        if fp32_dgrad and ctx.f32:
            if "wt32" not in ops:      # (n, Cout, k, C) -> (n, C, k*Cout) fp32: the input gradient's B operand, (tap, output channel) order
                ops["wt32"] = w.view(n, N, k, C).permute(0, 3, 2, 1).reshape(n, C, k * N).contiguous()
            dx = cg.conv_dgrad_f32(gx, ops["wt32"], geom, k, stride, pad, xshape[1])      # exact-f32 implicit GEMM: no column operand, no col2im
        elif fp32_dgrad:
            dcols = torch.bmm(gx, w)                                      # (n, M, k*C) fp32
            with torch.cuda.device(gy.device):
                dx = torch.empty(xshape, dtype=torch.float32, device=gy.device)
                _, Ns, Hs = _slot_arrays([0] * n, [ns for ns, _ in geom], [h for _, h in geom])
                _lib.check(lib.vmasr_col2im_kx1_stacked(dcols.data_ptr(), dx.data_ptr(), Ns, Hs, n, C, k, stride, pad, M, xshape[1], _lib.F32,
                                                        _lib.current_stream(gy.device)), "col2im_kx1_stacked")
        elif ctx.needs_input_grad[9]:
            if "wt" not in ops:      # (n, Cout, k, C) -> (n, C, k*Cout): the dgrad GEMM's B operand, (tap, output channel) order
                ops["wt"] = split_bf16(w.view(n, N, k, C).permute(0, 3, 2, 1).reshape(n, C, k * N).contiguous())
            wth, wtl = ops["wt"]
            if _StackedConvMfmaFn._fuse_below(ctx, gh, gl, wth, wtl, geom, k, stride, pad, xshape[1], skip_w):
                dx = _poison(gy.device).expand(xshape)      # nobody may read it: the layer below starts from below["stash"]
            else:
                dx = cg.conv_dgrad(gh, gl, wth, wtl, geom, k, stride, pad, xshape[1])
        if not skip_w:
            if ctx.needs_input_grad[7]:
                if ctx.f32:
                    xh, xl = split_bf16(xh)                                # (saved as the fp32 input)
                dw = cg.conv_wgrad(gh, gl, xh, xl, geom, k, stride, pad).to(wdt)
            if want_db:
                db = db32.to(bdt)
        return (None, None, None, None, None, None, None, dw, db, dx, None, None, None, None)


_POISON = {}


def _poison(device):
    """A NaN scalar: expanded to the shape of a gradient that must not be read (its content travelled another way)."""
    t = _POISON.get(device)
    if t is None:
        t = _POISON[device] = torch.full((), float("nan"), dtype=torch.float32, device=device)
    return t


def _l1_mode():
    """how the 32 -> 128 layer runs: "f32" (default) exact-f32 MFMA implicit GEMM, "1" bf16x3 pairs (forward below the accuracy gate), "0" library GEMMs"""
    return os.environ.get("VMASR_MPD_CONV_L1", "f32")


def det_mode():
    """deterministic-reduction mode of the library (VMASR_DETERMINISTIC=1 / vmasr_set_deterministic): the fused epilogue's bias-gradient
    atomics have no ordered form, so the unfused chain (gelu_bwd_split with its tickets) runs there"""
    return os.environ.get("VMASR_DETERMINISTIC", "0") == "1" or bool(_lib.lib().vmasr_get_deterministic())


class _StackedConvFirstFn(torch.autograd.Function):
    """The first convolution (1 -> 32 channels, kernel (5,1), stride (3,1), padding 2) + GELU of all n period discriminators
    in one launch on the folded signals xs[i] (B, p, H, 1) — csrc/convfirst.hip — instead of a 5-column im2col operand, a
    K = 5 GEMM and an epilogue pass.  W (n, 32, 5), bias (n, 32).  Returns the stacked activations (n, rows, 32)."""

    @staticmethod
    def forward(ctx, rows, W, bias, *xs):
        n, dev = len(xs), xs[0].device
        lib = _lib.lib()
        xcs = [x.float().contiguous() for x in xs]
        w32, b32 = W.detach().float().contiguous(), bias.detach().float().contiguous()
        ptrs, Ns, Hs = _slot_arrays([x.data_ptr() for x in xcs], [x.shape[0] * x.shape[1] for x in xcs], [x.shape[2] for x in xcs])
        with torch.cuda.device(dev):
            pre = torch.empty((n, rows, 32), dtype=torch.float32, device=dev)
            act = torch.empty((n, rows, 32), dtype=torch.float32, device=dev)
            _lib.check(lib.vmasr_conv_first_fwd(ptrs, Ns, Hs, n, w32.data_ptr(), b32.data_ptr(), pre.data_ptr(), act.data_ptr(), rows,
                                                _lib.current_stream(dev)), "conv_first_fwd")
        ctx.save_for_backward(pre, w32, *xcs)
        ctx.meta = (W.dtype, bias.dtype, [x.dtype for x in xs], [tuple(x.shape) for x in xs])
        return act

    @staticmethod
    def backward(ctx, gy):
        pre, w32, *xcs = ctx.saved_tensors
        wdt, bdt, xdts, shapes = ctx.meta
        n, rows, _ = pre.shape
        lib, dev = _lib.lib(), pre.device
        gy = gy.float().contiguous()
        skip_w = _PHASE["skip_weight_grads"]
        want_dw, want_db = ctx.needs_input_grad[1] and not skip_w, ctx.needs_input_grad[2] and not skip_w
        want_dx = any(ctx.needs_input_grad[3:])
        ptrs, Ns, Hs = _slot_arrays([x.data_ptr() for x in xcs], [x.shape[0] * x.shape[1] for x in xcs], [x.shape[2] for x in xcs])
        dxs = [None] * n
        with torch.cuda.device(dev):
            dcols = torch.empty((n, rows, 5), dtype=torch.float32, device=dev) if want_dx else None
            dw, db = _lib.zeros_f32(dev, (n, 32, 5) if want_dw else None, (n, 32) if want_db else None)
            _lib.check(lib.vmasr_conv_first_bwd(ptrs, Ns, Hs, n, w32.data_ptr(), pre.data_ptr(), gy.data_ptr(),
                                                dcols.data_ptr() if want_dx else None, dw.data_ptr() if want_dw else None,
                                                db.data_ptr() if want_db else None, rows, _lib.current_stream(dev)), "conv_first_bwd")
            if want_dx:
                outs = [torch.empty(shp, dtype=torch.float32, device=dev) if ctx.needs_input_grad[3 + i] else None for i, shp in enumerate(shapes)]
                optrs, Ns2, Hs2 = _slot_arrays([o.data_ptr() if o is not None else 0 for o in outs], [B * P for B, P, _, _ in shapes],
                                               [H for _, _, H, _ in shapes])
                _lib.check(lib.vmasr_col2im_kx1_multi(dcols.data_ptr(), optrs, Ns2, Hs2, n, 1, 5, 3, 2, rows, _lib.F32,
                                                      _lib.current_stream(dev)), "col2im_kx1_multi")
                dxs = [o.to(xdts[i]) if o is not None else None for i, o in enumerate(outs)]
        return (None, dw.to(wdt) if want_dw else None, db.to(bdt) if want_db else None, *dxs)


class _StackedConvPostFn(torch.autograd.Function):
    """conv_post (C -> 1 channels, kernel (3,1), stride 1, padding 1) of all n period discriminators directly on the previous
    layer's stacked output x (n, rows, C) — csrc/convpost.hip: one streaming pass forward, one backward, instead of a
    (rows, 3C) im2col operand feeding a GEMV.  W (n, 1, 3C) in (tap, channel) order, bias (n, 1); Ms[i] valid rows =
    whole sequences of Hs[i] positions.  Returns (n, rows, 1)."""

    @staticmethod
    def forward(ctx, Ms, Hs, W, bias, x):
        import ctypes
        n, rows, C = x.shape
        lib, dev = _lib.lib(), x.device
        xc, w32, b32 = x.contiguous(), W.detach().float().contiguous(), bias.detach().float().contiguous()
        ms, hs = (ctypes.c_int64 * n)(*Ms), (ctypes.c_int32 * n)(*Hs)
        with torch.cuda.device(dev):
            y = torch.empty((n, rows, 1), dtype=torch.float32, device=dev)
            _lib.check(lib.vmasr_conv_post_fwd(xc.data_ptr(), w32.data_ptr(), b32.data_ptr(), y.data_ptr(), ms, hs, n, rows, C, 3,
                                               _lib.current_stream(dev)), "conv_post_fwd")
        ctx.save_for_backward(xc, w32)
        ctx.meta = (tuple(Ms), tuple(Hs), W.dtype, bias.dtype, tuple(bias.shape))
        return y

    @staticmethod
    def backward(ctx, gy):
        import ctypes
        xc, w32 = ctx.saved_tensors
        Ms, Hs, wdt, bdt, bshape = ctx.meta
        n, rows, C = xc.shape
        lib, dev = _lib.lib(), xc.device
        gy = gy.float().contiguous()
        skip_w = _PHASE["skip_weight_grads"]
        want_dx, want_dw, want_db = ctx.needs_input_grad[4], ctx.needs_input_grad[2] and not skip_w, ctx.needs_input_grad[3] and not skip_w
        ms, hs = (ctypes.c_int64 * n)(*Ms), (ctypes.c_int32 * n)(*Hs)
        with torch.cuda.device(dev):
            dx = torch.empty_like(xc) if want_dx else None
            dw, db = _lib.zeros_f32(dev, (n, 1, 3 * C) if want_dw else None, (n,) if want_db else None)
            _lib.check(lib.vmasr_conv_post_bwd(xc.data_ptr(), w32.data_ptr(), gy.data_ptr(), dx.data_ptr() if want_dx else None,
                                               dw.data_ptr() if want_dw else None, db.data_ptr() if want_db else None, ms, hs, n, rows, C, 3,
                                               _lib.current_stream(dev)), "conv_post_bwd")
        return (None, None, dw.to(wdt) if want_dw else None, db.view(bshape).to(bdt) if want_db else None, dx)


class _UnstackRowsFn(torch.autograd.Function):
    """(n, rows, N) -> n views y[i, :M_i]; the backward assembles the stacked gradient with one copy per slot
    (autograd's own select/slice backward would zero-fill a full-size tensor per slot)."""

    @staticmethod
    def forward(ctx, y, *Ms):
        ctx.shape = tuple(y.shape)
        ctx.Ms = Ms
        return tuple(y[i, :m] for i, m in enumerate(Ms))

    @staticmethod
    def backward(ctx, *gs):
        ref = next(g for g in gs if g is not None)
        if ref.is_cuda and len(gs) <= 8:
            with torch.cuda.device(ref.device):
                full = torch.empty(ctx.shape, dtype=ref.dtype, device=ref.device)
                gc = [g.contiguous() if g is not None else None for g in gs]
                ptrs, Ms, _ = _slot_arrays([g.data_ptr() if g is not None else 0 for g in gc], list(ctx.Ms))
                _lib.check(_lib.lib().vmasr_stack_rows(ptrs, Ms, len(gs), full.data_ptr(), ctx.shape[1],
                                                       ctx.shape[2] * ref.element_size(), _lib.current_stream(ref.device)), "stack_rows")
            return (full, *([None] * len(ctx.Ms)))
        full = torch.empty(ctx.shape, dtype=ref.dtype, device=ref.device)
        for i, (g, m) in enumerate(zip(gs, ctx.Ms)):
            if g is None:
                full[i].zero_()
            else:
                full[i, :m].copy_(g)
                if m < ctx.shape[1]:
                    full[i, m:].zero_()
        return (full, *([None] * len(ctx.Ms)))


class StackedFeatures(list):
    """Feature maps of the batched discriminator pass: the usual list (per discriminator) of lists (per layer) of
    channel-last views, plus the stacked per-layer tensors they are views of — `stacks[l]` is (n, rows_l, N_l)
    and `valid[l][i]` says how many leading rows of slot i belong to this signal — so that losses over ALL
    discriminators can run as one kernel per layer (feature_loss_stacked) instead of one per feature map."""

    def __init__(self, per_disc, stacks, valid, taps=None):
        super().__init__(per_disc)
        self.stacks, self.valid = stacks, valid
        self.taps = taps if taps is not None else [None] * len(stacks)     # (token, holder) of _FeatTapFn per layer, or None

    def detach(self):
        return StackedFeatures([[f.detach() for f in fs] for fs in self], [y.detach() for y in self.stacks], self.valid)


class _FeatTapFn(torch.autograd.Function):
    """Identity on a stacked feature map that also hands out a one-element TOKEN.  The feature-matching loss takes the token —
    not the map — as its differentiable input (_MaskedL1Fn with `tap`) and leaves sign(gen - real) in `holder`; the map's
    gradient is then formed HERE as  gy + g_loss * scale[s] * sign  in one pass (vmasr_masked_l1_bwd_add: r 5 B, w 4 B per
    element) instead of the loss's own backward pass (r 1, w 4) + autograd's sum of the two gradients (r 8, w 4): the maps are
    0.8 GB per generator step.  A tap nobody feeds (the discriminator phase) passes gy through."""

    @staticmethod
    def forward(ctx, y, holder):
        ctx.holder = holder
        ctx.set_materialize_grads(False)
        return y.view_as(y), y.new_zeros(1)

    @staticmethod
    def backward(ctx, gy, gtok):
        import ctypes
        h = ctx.holder
        sgn = h.get("sgn")          # (kept: with a shared discriminator pass the graph is walked once per loss phase)
        h.pop("gtok", None)         # (left by the loss' backward for the layer above: _StackedConvMfmaFn._fuse_below)
        if h.pop("consumed", False):    # the layer above has formed gy + g_loss * scale * sign (and GELU') in its epilogue
            return gy, None
        if gtok is None or sgn is None:
            return gy, None
        valid, scale = h["valid"], h["scale"]
        n, rows_g, N = sgn.shape
        v = (ctypes.c_int64 * n)(*valid)
        sc = (ctypes.c_float * n)(*scale)
        add = None if gy is None else gy.float().contiguous()
        gtok = gtok.float().contiguous()
        with torch.cuda.device(sgn.device):
            out = torch.empty((n, rows_g, N), dtype=torch.float32, device=sgn.device)
            _lib.check(_lib.lib().vmasr_masked_l1_bwd_add(sgn.data_ptr(), gtok.data_ptr(), add.data_ptr() if add is not None else None,
                                                          out.data_ptr(), v, sc, n, rows_g, N, _lib.current_stream(sgn.device)),
                       "masked_l1_bwd_add")
        return out, None


_FEAT_MASKS = {}


class _MaskedL1Fn(torch.autograd.Function):
    """sum_s scale[s] * sum_{r < valid[s]} |gen[s, r] - real[s, r]| over two stacked fp32 feature tensors in one pass
    (csrc/featloss.hip), gradient with respect to `gen` only (the real-signal features are constants of the
    generator phase); the forward leaves sign(gen - real) as int8 for the one-pass backward."""

    @staticmethod
    def forward(ctx, real, gen, valid, scale, token, holder):
        """token / holder: of the map's _FeatTapFn — then `gen` is the DETACHED map, the gradient goes to the token and the
        tap forms the map's gradient from the sign left in `holder`."""
        import ctypes
        n, rows_g, N = gen.shape
        lib, dev = _lib.lib(), gen.device
        nb = lib.vmasr_masked_l1_blocks()
        v = (ctypes.c_int64 * n)(*valid)
        sc = (ctypes.c_float * n)(*scale)
        tapped = token is not None and ctx.needs_input_grad[4]
        with torch.cuda.device(dev):
            partials = torch.empty(n * nb, dtype=torch.float64, device=dev)
            sgn = torch.empty((n, rows_g, N), dtype=torch.int8, device=dev) if (ctx.needs_input_grad[1] or tapped) else None
            _lib.check(lib.vmasr_masked_l1_fwd(real.data_ptr(), gen.data_ptr(), sgn.data_ptr() if sgn is not None else None,
                                               partials.data_ptr(), v, sc, n, real.shape[1], rows_g, N, _lib.current_stream(dev)), "masked_l1_fwd")
        ctx.meta = (valid, scale, gen.shape)
        ctx.tapped = tapped
        ctx.holder = holder if tapped else None
        if tapped:
            holder.update(sgn=sgn, valid=valid, scale=scale)
        elif sgn is not None:
            ctx.save_for_backward(sgn)
        return partials.sum().float()

    @staticmethod
    def backward(ctx, g):
        import ctypes
        if ctx.tapped:
            # this node runs before the discriminator's layers (it was created after them); the layer above the tapped map folds
            # g * scale * sign into its input-gradient epilogue when it finds the upstream gradient here (_fuse_below)
            ctx.holder["gtok"] = g.detach().reshape(1).float().contiguous()
            return None, None, None, None, g.reshape(1), None
        (sgn,) = ctx.saved_tensors
        valid, scale, (n, rows_g, N) = ctx.meta
        v = (ctypes.c_int64 * n)(*valid)
        sc = (ctypes.c_float * n)(*scale)
        g = g.float().contiguous()
        with torch.cuda.device(sgn.device):
            dgen = torch.empty((n, rows_g, N), dtype=torch.float32, device=sgn.device)
            _lib.check(_lib.lib().vmasr_masked_l1_bwd(sgn.data_ptr(), g.data_ptr(), dgen.data_ptr(), v, sc, n, rows_g, N,
                                                      _lib.current_stream(sgn.device)), "masked_l1_bwd")
        return None, dgen, None, None, None, None


def _masked_l1_ok(yr, yg):
    return (yg.is_cuda and yr.is_cuda and yg.dtype == torch.float32 and yr.dtype == torch.float32 and yg.is_contiguous()
            and yr.is_contiguous() and not yr.requires_grad and yg.shape[0] <= 8 and yg.shape[0] == yr.shape[0]
            and (yg.shape[1] * yg.shape[2]) % 4 == 0 and (yr.shape[1] * yr.shape[2]) % 4 == 0
            and os.environ.get("VMASR_FEAT_L1", "1") == "1")


def feature_loss_stacked(real, gen):
    """HiFi-GAN feature-matching loss (model/loss.py:227-235: mean over feature maps of mean |r - g|) from two
    StackedFeatures with the same per-slot row counts; None if the inputs do not qualify."""
    if not (isinstance(real, StackedFeatures) and isinstance(gen, StackedFeatures)) or real.valid != gen.valid:
        return None
    n_maps = sum(len(fs) for fs in gen)
    total = None
    for yr, yg, valid, tap in zip(real.stacks, gen.stacks, real.valid, gen.taps):
        R, N = min(yr.shape[1], yg.shape[1]), yg.shape[2]
        if max(valid) > R:
            return None
        if _masked_l1_ok(yr, yg):
            scale = tuple(1.0 / (m * N * n_maps) for m in valid)
            if tap is not None and tap[0].requires_grad and "sgn" not in tap[1]:
                term = _MaskedL1Fn.apply(yr, yg.detach(), tuple(valid), scale, *tap)
            else:
                term = _MaskedL1Fn.apply(yr, yg, tuple(valid), scale, None, None)
            total = term if total is None else total + term
            continue
        key = (yg.device, valid, R, N, n_maps)
        mask = _FEAT_MASKS.get(key)
        if mask is None:       # 1 / (elements of the feature map * number of maps) on its rows, 0 on padding rows
            rows = torch.arange(R, device=yg.device).unsqueeze(0)
            m = torch.tensor(valid, device=yg.device).unsqueeze(1)
            mask = ((rows < m).float() / (m.float() * N * n_maps)).unsqueeze(2)
            _FEAT_MASKS[key] = mask
        term = ((yg[:, :R] - yr[:, :R]).abs() * mask).sum()
        total = term if total is None else total + term
    return total


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
        if not x.is_cuda or _PLAIN_OPS[0]:  # host runs (tests, cpu_baseline) and double backward: plain convolutions
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
            x = F.gelu(conv_kx1(x, w, bias, layer.stride[0], layer.padding[0]))
            fmap.append(x)
        w, bias = wb(self.conv_post)
        x = conv_kx1(x, w, bias, 1, self.conv_post.padding[0])
        fmap.append(x)
        return torch.flatten(x, 1, -1), fmap


class MultiPeriodDiscriminator(nn.Module):
    def __init__(self, hidden=32, periods=(2, 3, 5, 7, 11)):
        super().__init__()
        self.discriminators = nn.ModuleList([PeriodDiscriminator(p, hidden=hidden) for p in periods])
        self._frozen = None      # per-layer stacked weights while frozen_weights() is active

    def frozen_weights(self):
        """Context in which the caller guarantees that the (normalised) weights do not change — inside
        torch.nn.utils.parametrize.cached() within one training step: the batched passes then share the stacked
        weight of each layer instead of rebuilding it per pass."""
        import contextlib

        @contextlib.contextmanager
        def ctx():
            object.__setattr__(self, "_frozen", {})
            try:
                yield
            finally:
                object.__setattr__(self, "_frozen", None)
        return ctx()

    def _forward_batched(self, x, detach_weights=False):
        """All discriminators layer by layer on stacked GEMM operands (GPU path).  Same scores and feature maps
        (channel-last (B, p, T', C) views) as running the PeriodDiscriminators one by one."""
        discs = list(self.discriminators)
        n, (B, _, T) = len(discs), x.shape
        cdt = torch.get_autocast_dtype("cuda") if torch.is_autocast_enabled("cuda") else x.dtype
        cur = []
        for d in discs:
            xp, p = x, d.period
            if T % p:
                xp = F.pad(xp, (0, p - T % p), "reflect")
            cur.append(xp.view(B, 1, -1, p).permute(0, 3, 2, 1).to(cdt))          # (B, p, T/p, 1)
        fmaps, stacks, valid, taps = [[] for _ in discs], [], [], []
        next_pair = this_link = None
        for li in range(len(discs[0].layers) + 1):
            layers = [d.layers[li] if li < len(d.layers) else d.conv_post for d in discs]
            k, stride, pad = layers[0].kernel_size[0], layers[0].stride[0], layers[0].padding[0]
            P = [c.shape[1] for c in cur]
            H1 = [(c.shape[2] + 2 * pad - k) // stride + 1 for c in cur]
            Ms = [B * p * h for p, h in zip(P, H1)]
            W = None
            if os.environ.get("VMASR_SN_STACK", "1") == "1":
                # normalisation, stack and (tap, c) permutation of the layer's n weights in one launch; while the trainer
                # holds the weights fixed for the step (frozen_weights()) the passes share it — one gradient path back
                key = (li, bool(detach_weights))
                W = self._frozen.get(key) if self._frozen is not None else None
                if W is None:
                    W = _sn_stack(layers)
                    if W is not None and detach_weights:
                        W = W.detach()
                    if W is not None and self._frozen is not None:
                        self._frozen[key] = W
            if W is not None:
                ws = [(None, l.bias.detach() if detach_weights else l.bias) for l in layers]
            else:
                ws = [(l.weight.detach(), l.bias.detach()) if detach_weights else (l.weight, l.bias) for l in layers]
                # (n, Cout, k, Cin) -> (n, Cout, k*Cin): (tap, c) column order, gathered by the stack's own copy
                W = torch.stack([w.squeeze(3).transpose(1, 2) for w, _ in ws])
                W = W.reshape(n, W.shape[1], -1)
            act = li < len(discs[0].layers)
            bstack = torch.stack([b for _, b in ws])
            # from the second layer on the inputs are the slots of the previous layer's stacked output: hand that tensor over
            # (slot i = B*p_i sequences of H_i positions) so that its gradient comes back stacked, in one launch
            sgeom, src = None, cur
            pair, next_pair = next_pair, None     # the bf16 (hi, lo) pair of stacks[-1], if the previous layer's epilogue wrote it
            prev_link, this_link = this_link, None   # set by an MFMA layer: the layer above may finish its activation backward
            if stacks and stacks[-1].dtype == cdt and os.environ.get("VMASR_STACK_INPUT", "1") == "1":
                sgeom, src = tuple((B * p, c.shape[2]) for c, p in zip(cur, P)), (stacks[-1],)
            if (not act and stacks and cdt == torch.float32 and k == 3 and stride == 1 and pad == 1 and W.shape[1] == 1
                    and stacks[-1].dtype == torch.float32 and os.environ.get("VMASR_CONV_POST", "1") == "1"
                    and _lib.lib().vmasr_conv_post_supported(stacks[-1].shape[2], k)):
                # the 1-channel output convolution straight on the previous layer's stacked maps (no column operand)
                y = _StackedConvPostFn.apply(tuple(valid[-1]), tuple(c.shape[2] for c in cur), W, bstack, stacks[-1])
            elif (act and li == 0 and cdt == torch.float32 and k == 5 and stride == 3 and pad == 2 and W.shape[1] == 32 and W.shape[2] == 5
                  and all(c.shape[3] == 1 for c in cur) and os.environ.get("VMASR_CONV_FIRST", "1") == "1"):
                # the 1 -> 32 channel input convolution + GELU straight from the folded signals (no 5-column operand / K = 5 GEMM)
                y = _StackedConvFirstFn.apply(_round_up(max(Ms), 256), W, bstack, *cur)
            elif (cdt == torch.float32 and os.environ.get("VMASR_MPD_GEMM", "bf16x3") == "bf16x3" and act and sgeom is not None
                  and os.environ.get("VMASR_MPD_CONV", "mfma") == "mfma"
                  and (cur[0].shape[3] >= 128 or _l1_mode() != "0")
                  # (shape, slot count and row count of the whole stacked launch: an MPD with more periods or a longer segment than
                  #  the launchers address falls through to the split-GEMM path below)
                  and _lib.lib().vmasr_conv_mfma_supported_launch(cur[0].shape[3], W.shape[1], k, stride, n,
                                                                  max(_round_up(max(Ms), 256), max(g[0] * g[1] for g in sgeom)))):
                # the three compute-bound layers (128 -> 512 -> 1024 -> 1024): one implicit-GEMM launch each way (csrc/convgemm.hip); the
                # layer's epilogue leaves the bf16 pair of its activation for the next layer.  The 32 -> 128 layer takes the same kernels
                # in their EXACT-F32 form (round 6, VMASR_MPD_CONV_L1=f32, the default: fp32 operands, v_mfma_f32_32x32x2_f32, forward and
                # input gradient; the weight gradient as a bf16x3 triple): it is the first GEMM behind the signal, and with its forward at
                # the pair's 16-17 bits (VMASR_MPD_CONV_L1=1) the input gradient d(loss)/d(wave) of an |f|-type loss moved to 2.5e-3 of its
                # scale from float64 (fp32: 4e-4; gate 5e-4, tests/test_mpd.py) — near-zero GELU outputs change sign.  =0: the round-5
                # path (im2col + fp32 library GEMM + bias / GELU pass; GEMM + col2im)
                wcache = None
                if self._frozen is not None:
                    wcache = self._frozen.setdefault(("mfma_ops", li), {})
                out_pair = []
                xh, xl = pair if pair is not None else (None, None)
                link = {}
                y = _StackedConvMfmaFn.apply(k, stride, pad, _round_up(max(Ms), 256), sgeom, wcache, out_pair, W, bstack, src[0], xh, xl,
                                             link, prev_link)
                next_pair, this_link = out_pair[0], link
            elif _split_mode(W.shape[2], W.shape[1], cdt) and cur[0].shape[3] % 4 == 0:
                y = _StackedConvSplitFn.apply(k, stride, pad, _round_up(max(Ms), 256), act, sgeom, W, bstack, *src)
            else:
                cols = _StackedIm2ColFn.apply(k, stride, pad, _round_up(max(Ms), 256), sgeom, *src)
                y = _BatchedLinearFn.apply(cols, W, bstack, cdt, act)
            tap = None
            if (act and y.requires_grad and y.dtype == torch.float32 and x.requires_grad
                    and os.environ.get("VMASR_FEAT_TAP", "1") == "1"):
                # generator phase: the map's gradient (next layer's + feature-matching loss's) is formed in one pass (_FeatTapFn)
                holder = {}
                y, token = _FeatTapFn.apply(y, holder)
                tap = (token, holder)
                if this_link is not None:
                    this_link["tap"] = holder
            outs = _UnstackRowsFn.apply(y, *Ms)
            cur = [o.view(B, p, h, -1) for o, p, h in zip(outs, P, H1)]
            for f, c in zip(fmaps, cur):
                f.append(c)
            stacks.append(y)
            valid.append(tuple(Ms))
            taps.append(tap)
        return [torch.flatten(c, 1, -1) for c in cur], StackedFeatures(fmaps, stacks, valid, taps)

    def _use_batched(self, x):
        return (x.is_cuda and not _PLAIN_OPS[0] and os.environ.get("VMASR_MPD_BATCHED", "1") == "1"
                and len(self.discriminators) > 1)

    def forward_single(self, x, detach_weights=False):
        """scores and feature maps of ONE signal batch (used for the generator pass, where the
        real-signal features of the discriminator pass are reused instead of recomputed)."""
        if self._use_batched(x):
            return self._forward_batched(x, detach_weights)
        res = [d(x, detach_weights) for d in self.discriminators]
        return [r[0] for r in res], [r[1] for r in res]

    def forward_pair(self, y, y_hat):
        """Same results as forward(y, y_hat) from ONE pass over the stacked batch [y; y_hat] (same weights,
        identical per-sample arithmetic); on the GPU all discriminators advance layer by layer together
        (_forward_batched) and the real-signal features come back as StackedFeatures."""
        n = y.shape[0]
        y_real, y_gen, fmap_real, fmap_gen = [], [], [], []
        both = torch.cat((y, y_hat), dim=0)
        if self._use_batched(both):
            scores, feats = self._forward_batched(both)
            y_real, y_gen = [s[:n] for s in scores], [s[n:] for s in scores]
            # the real half of every slot is its leading rows (batch-major row order)
            fmap_real = StackedFeatures([[t[:n] for t in f] for f in feats], feats.stacks, [tuple(m // 2 for m in v) for v in feats.valid])
            return y_real, y_gen, fmap_real, [[t[n:] for t in f] for f in feats]
        else:
            res = [d(both) for d in self.discriminators]
        for s, f in res:
            y_real.append(s[:n]); y_gen.append(s[n:])
            fmap_real.append([t[:n] for t in f]); fmap_gen.append([t[n:] for t in f])
        return y_real, y_gen, fmap_real, fmap_gen

    def spectral_norms(self):
        return [m for m in self.modules() if isinstance(m, _SpectralNorm)]

    def power_iterate_all(self, n_iter, with_sigma=False):
        """n_iter power iterations of every spectrally normalised weight, batched into one launch per phase
        (with_sigma: the normalising sigmas too; call clear_sigmas() when the weights may change).
        Returns False (nothing done) when the model is not on the GPU in fp32."""
        pairs = [(mod.parametrizations.weight[0], mod.parametrizations.weight.original)
                 for mod in self.modules() if isinstance(mod, nn.Conv2d) and parametrize.is_parametrized(mod, "weight")
                 and isinstance(mod.parametrizations.weight[0], _SpectralNorm)]
        if not pairs or not all(w.is_cuda and w.dtype == torch.float32 for _, w in pairs) or len(pairs) > 64:
            return False
        weights = [w.detach() for _, w in pairs]
        batch = getattr(self, "_sn_batch", None)
        if batch is None or not batch.matches(weights):
            batch = SpectralBatch([m for m, _ in pairs], weights)
            object.__setattr__(self, "_sn_batch", batch)
        batch.run(n_iter, with_sigma)
        return True

    def clear_sigmas(self):
        batch = getattr(self, "_sn_batch", None)
        if batch is not None:
            batch.clear_sigma()

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
