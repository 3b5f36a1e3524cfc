"""The period discriminator's (k,1) convolutions as implicit bf16x3 MFMA GEMMs (csrc/convgemm.hip).

Reference: model/discriminator.py:21-147 (Conv2d (k,1), stride (s,1), zero padding, GELU), run in fp32 there.
Operands are bf16 PAIRS (hi, lo) with x = hi + lo up to 2^-17 |x| (csrc/split.hip); each layer's epilogue writes the
pair of its activation, so im2col / col2im / partial products never exist.  Stacked layout as in discriminator.py:
(n slots, rows, C) channel-last, slot i holding nseq_i sequences of H_i positions in its leading rows, zeros below.
All functions launch on the current stream; tensors must be contiguous and on the GPU (no CPU path)."""
import ctypes

import torch

from . import _lib


def supported(Cin, Cout, k, stride):
    return bool(_lib.lib().vmasr_conv_mfma_supported(int(Cin), int(Cout), int(k), int(stride)))


_LIMIT = {"cus": 0, "min_channels": 0}


class cu_limit:
    """with cu_limit(n[, min_channels]): the conv_mfma kernels launched inside occupy at most n CUs (persistent tile loops) — those of
    layers with max(Cin, Cout) >= min_channels, the others keep the whole chip; 0 / None = no limit.
    Host-side launch geometry only — the results do not depend on it."""

    def __init__(self, cus, min_channels=0):
        self.new = {"cus": int(cus or 0), "min_channels": int(min_channels or 0)}

    def __enter__(self):
        self.prev = dict(_LIMIT)
        _LIMIT.update(self.new)
        return self

    def __exit__(self, *exc):
        _LIMIT.update(self.prev)
        _lib.lib().vmasr_conv_set_cu_limit(0)
        return False


def _apply_limit(Cin, Cout):
    lim = _LIMIT["cus"] if max(Cin, Cout) >= _LIMIT["min_channels"] else 0
    _lib.lib().vmasr_conv_set_cu_limit(lim)


def _slots(n):
    return (_lib.CgSlot * n)()


def _ptr(t, i, step):
    return ctypes.c_void_p(t.data_ptr() + i * step) if t is not None else None


def _need(*ts):
    for t in ts:
        if t is not None and not (t.is_cuda and t.is_contiguous()):
            raise RuntimeError("convgemm: tensors must be contiguous CUDA tensors (there is no CPU path)")


def out_positions(H, k, stride, pad):
    return (H + 2 * pad - k) // stride + 1


def conv_fwd(xh, xl, wh, wl, bias, geom, k, stride, pad, rows_out, act=True, want_pair=True):
    """xh, xl (n, rows_in, Cin) bf16; wh, wl (n, Cout, k*Cin) bf16, (tap, channel) column order; bias (n, Cout) fp32;
    geom = ((nseq_i, H_i), ...).  -> pre (n, rows_out, Cout) fp32, and with act: y = GELU(pre) fp32, (yh, yl) its bf16
    pair (want_pair); rows below nseq_i * H1_i are zero in every output."""
    _need(xh, xl, wh, wl, bias)
    n, rows_in, Cin = xh.shape
    Cout = wh.shape[1]
    dev = xh.device
    with torch.cuda.device(dev):
        pre = torch.empty((n, rows_out, Cout), dtype=torch.float32, device=dev)
        y = torch.empty_like(pre) if act else None
        yh = torch.empty((n, rows_out, Cout), dtype=torch.bfloat16, device=dev) if act and want_pair else None
        yl = torch.empty_like(yh) if yh is not None else None
        sl = _slots(n)
        for i, (nseq, H) in enumerate(geom):
            s = sl[i]
            s.ah, s.al = _ptr(xh, i, rows_in * Cin * 2), _ptr(xl, i, rows_in * Cin * 2)
            s.bh, s.bl = _ptr(wh, i, Cout * k * Cin * 2), _ptr(wl, i, Cout * k * Cin * 2)
            s.c0, s.c1 = _ptr(pre, i, rows_out * Cout * 4), _ptr(y, i, rows_out * Cout * 4)
            s.ch, s.cl = _ptr(yh, i, rows_out * Cout * 2), _ptr(yl, i, rows_out * Cout * 2)
            s.bias = _ptr(bias, i, Cout * 4)
            s.nseq, s.H = int(nseq), int(H)
        _apply_limit(Cin, Cout)
        _lib.check(_lib.lib().vmasr_conv_mfma_fwd(sl, n, Cin, Cout, k, stride, pad, rows_out, int(bool(act)),
                                                  _lib.current_stream(dev)), "conv_mfma_fwd")
    return pre, y, yh, yl


def conv_dgrad(gh, gl, wth, wtl, geom, k, stride, pad, rows_in):
    """gh, gl (n, rows_out, Cout) bf16 pair of the output gradient; wth, wtl (n, Cin, k*Cout) bf16: the weight in
    (tap, output channel) column order; geom = ((nseq_i, H_i), ...) with H_i the INPUT positions.
    -> dx (n, rows_in, Cin) fp32 (zero below nseq_i * H_i)."""
    _need(gh, gl, wth, wtl)
    n, rows_out, Cout = gh.shape
    Cin = wth.shape[1]
    dev = gh.device
    with torch.cuda.device(dev):
        dx = torch.empty((n, rows_in, Cin), dtype=torch.float32, device=dev)
        sl = _slots(n)
        for i, (nseq, H) in enumerate(geom):
            s = sl[i]
            s.ah, s.al = _ptr(gh, i, rows_out * Cout * 2), _ptr(gl, i, rows_out * Cout * 2)
            s.bh, s.bl = _ptr(wth, i, Cin * k * Cout * 2), _ptr(wtl, i, Cin * k * Cout * 2)
            s.c0 = _ptr(dx, i, rows_in * Cin * 4)
            s.nseq, s.H = int(nseq), int(H)
        _apply_limit(Cin, Cout)
        _lib.check(_lib.lib().vmasr_conv_mfma_dgrad(sl, n, Cin, Cout, k, stride, pad, rows_in, _lib.current_stream(dev)),
                   "conv_mfma_dgrad")
    return dx


def conv_fwd_f32(x, w, bias, geom, k, stride, pad, rows_out, act=True, want_pair=True):
    """conv_fwd with FP32 operands and exact-f32 products (csrc/convgemm.hip OPS 1; Cin = 32): x (n, rows_in, Cin) fp32, w (n, Cout, k*Cin)
    fp32 in (tap, channel) order.  Same outputs as conv_fwd."""
    _need(x, w, bias)
    n, rows_in, Cin = x.shape
    Cout = w.shape[1]
    dev = x.device
    assert x.dtype == torch.float32 and w.dtype == torch.float32
    with torch.cuda.device(dev):
        pre = torch.empty((n, rows_out, Cout), dtype=torch.float32, device=dev)
        y = torch.empty_like(pre) if act else None
        yh = torch.empty((n, rows_out, Cout), dtype=torch.bfloat16, device=dev) if act and want_pair else None
        yl = torch.empty_like(yh) if yh is not None else None
        sl = _slots(n)
        for i, (nseq, H) in enumerate(geom):
            s = sl[i]
            s.ah, s.bh = _ptr(x, i, rows_in * Cin * 4), _ptr(w, i, Cout * k * Cin * 4)
            s.c0, s.c1 = _ptr(pre, i, rows_out * Cout * 4), _ptr(y, i, rows_out * Cout * 4)
            s.ch, s.cl = _ptr(yh, i, rows_out * Cout * 2), _ptr(yl, i, rows_out * Cout * 2)
            s.bias = _ptr(bias, i, Cout * 4)
            s.nseq, s.H = int(nseq), int(H)
        _apply_limit(Cin, Cout)
        _lib.check(_lib.lib().vmasr_conv_f32_fwd(sl, n, Cin, Cout, k, stride, pad, rows_out, int(bool(act)), _lib.current_stream(dev)),
                   "conv_f32_fwd")
    return pre, y, yh, yl


def conv_dgrad_f32(g, wt, geom, k, stride, pad, rows_in):
    """conv_dgrad with FP32 operands and exact-f32 products (Cin = 32): g (n, rows_out, Cout) fp32, wt (n, Cin, k*Cout) fp32 in
    (tap, output channel) order.  -> dx (n, rows_in, Cin) fp32."""
    _need(g, wt)
    n, rows_out, Cout = g.shape
    Cin = wt.shape[1]
    dev = g.device
    assert g.dtype == torch.float32 and wt.dtype == torch.float32
    with torch.cuda.device(dev):
        dx = torch.empty((n, rows_in, Cin), dtype=torch.float32, device=dev)
        sl = _slots(n)
        for i, (nseq, H) in enumerate(geom):
            s = sl[i]
            s.ah, s.bh = _ptr(g, i, rows_out * Cout * 4), _ptr(wt, i, Cin * k * Cout * 4)
            s.c0 = _ptr(dx, i, rows_in * Cin * 4)
            s.nseq, s.H = int(nseq), int(H)
        _apply_limit(Cin, Cout)
        _lib.check(_lib.lib().vmasr_conv_f32_dgrad(sl, n, Cin, Cout, k, stride, pad, rows_in, _lib.current_stream(dev)), "conv_f32_dgrad")
    return dx


def conv_dgrad_gelu(gh, gl, wth, wtl, geom, k, stride, pad, rows_in, pre, want_f32=False, want_pair=True, sgn=None, gtok=None,
                    scale=None, valid=None, db=None):
    """conv_dgrad with the activation backward of the layer BELOW in the epilogue: g = (dx + gtok * scale_i * sgn) * GELU'(pre), where
    pre (n, rows_in, Cin) fp32 is that layer's pre-activation and sgn (n, rows_in, Cin) int8 (optional) the sign map of its
    feature-matching term (rows < valid_i of slot i; gtok a device scalar).  -> (g fp32 or None, (gh, gl) bf16 pair or None).
    db (n, Cin) fp32, zeroed by the caller: the column sums of g are added to it (the layer below's bias gradient).
    Replaces vmasr_masked_l1_bwd_add + vmasr_gelu_bwd_split over the feature map (csrc/convgemm.hip EPI 2)."""
    _need(gh, gl, wth, wtl, pre, sgn, gtok, db)
    assert db is None or (db.shape == (gh.shape[0], wth.shape[1]) and db.dtype == torch.float32)
    n, rows_out, Cout = gh.shape
    Cin = wth.shape[1]
    dev = gh.device
    assert pre.shape == (n, rows_in, Cin) and pre.dtype == torch.float32 and (want_f32 or want_pair)
    assert sgn is None or (sgn.shape == pre.shape and sgn.dtype == torch.int8 and gtok is not None and gtok.dtype == torch.float32)
    with torch.cuda.device(dev):
        g32 = torch.empty((n, rows_in, Cin), dtype=torch.float32, device=dev) if want_f32 else None
        oh = torch.empty((n, rows_in, Cin), dtype=torch.bfloat16, device=dev) if want_pair else None
        ol = torch.empty_like(oh) if want_pair else None
        sl = _slots(n)
        ep = (_lib.CgGeluBwd * n)()
        for i, (nseq, H) in enumerate(geom):
            s = sl[i]
            s.ah, s.al = _ptr(gh, i, rows_out * Cout * 2), _ptr(gl, i, rows_out * Cout * 2)
            s.bh, s.bl = _ptr(wth, i, Cin * k * Cout * 2), _ptr(wtl, i, Cin * k * Cout * 2)
            s.c0 = _ptr(g32, i, rows_in * Cin * 4)
            s.ch, s.cl = _ptr(oh, i, rows_in * Cin * 2), _ptr(ol, i, rows_in * Cin * 2)
            s.nseq, s.H = int(nseq), int(H)
            ep[i].pre = _ptr(pre, i, rows_in * Cin * 4)
            ep[i].sgn = _ptr(sgn, i, rows_in * Cin)
            ep[i].valid = int(valid[i]) if (sgn is not None and valid is not None) else 0
            ep[i].scale = float(scale[i]) if (sgn is not None and scale is not None) else 0.0
            ep[i].db = _ptr(db, i, Cin * 4)
        _apply_limit(Cin, Cout)
        _lib.check(_lib.lib().vmasr_conv_mfma_dgrad_gelu(sl, ep, ctypes.c_void_p(gtok.data_ptr()) if sgn is not None else None, n, Cin, Cout,
                                                         k, stride, pad, rows_in, _lib.current_stream(dev)), "conv_mfma_dgrad_gelu")
    return g32, ((oh, ol) if want_pair else None)


def wgrad_splits(n, Cin, Cout, k, M):
    """Split factor of the weight gradient's contraction (the M rows).  256 x 256 tiles (one workgroup per CU) when both channel counts
    allow: no split (200 / 400 tiles on the 512 -> 1024 / 1024 -> 1024 layers: a split would add a sum pass over 50-100 MB per slab for
    <= 20 % better chip filling); 128 x 128 tiles otherwise, split until ~768 workgroups."""
    if Cin % 256 == 0 and Cout % 256 == 0:
        return 1
    T = 128
    tiles = n * (Cout // T) * (k * Cin // T)
    s = 1
    while tiles * s < 768 and s < 16 and M // (2 * s) >= 1024:
        s *= 2
    return s


def conv_wgrad(gh, gl, xh, xl, geom, k, stride, pad, splits=None):
    """dW (n, Cout, k*Cin) fp32, (tap, channel) column order = sum over the rows of g^T x_cols; geom as in conv_fwd (INPUT H)."""
    _need(gh, gl, xh, xl)
    n, rows_out, Cout = gh.shape
    _, rows_in, Cin = xh.shape
    dev = gh.device
    if splits is None:
        M = max(nseq * out_positions(H, k, stride, pad) for nseq, H in geom)
        splits = wgrad_splits(n, Cin, Cout, k, M)
    with torch.cuda.device(dev):
        parts = torch.empty((n, splits, Cout, k * Cin), dtype=torch.float32, device=dev)
        sl = _slots(n)
        for i, (nseq, H) in enumerate(geom):
            s = sl[i]
            s.ah, s.al = _ptr(gh, i, rows_out * Cout * 2), _ptr(gl, i, rows_out * Cout * 2)
            s.bh, s.bl = _ptr(xh, i, rows_in * Cin * 2), _ptr(xl, i, rows_in * Cin * 2)
            s.c0 = _ptr(parts, i, splits * Cout * k * Cin * 4)
            s.nseq, s.H = int(nseq), int(H)
        _apply_limit(Cin, Cout)
        _lib.check(_lib.lib().vmasr_conv_mfma_wgrad(sl, n, Cin, Cout, k, stride, pad, splits, _lib.current_stream(dev)),
                   "conv_mfma_wgrad")
        if splits == 1:
            return parts.view(n, Cout, k * Cin)
        dw = torch.empty((n, Cout, k * Cin), dtype=torch.float32, device=dev)
        _lib.check(_lib.lib().vmasr_sum_parts(parts.data_ptr(), dw.data_ptr(), 1, n, splits, Cout * k * Cin,
                                              _lib.current_stream(dev)), "sum_parts")
    return dw
