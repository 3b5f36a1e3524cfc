"""Selective-scan operator boundary: drop-in for the reference's `selective_scan_cuda_core`.

`fwd` / `bwd` keep the pybind module's names, positional order, shape/dtype checks and
error type (RuntimeError) — kernels/selective_scan/csrc/selective_scan/cus/
selective_scan.cpp:157-239 (fwd), :241-349 (bwd), exported at :351-354 — and
`SelectiveScanCore` mirrors model/vmamba.py:323-356, so `SS2D.forward_corev2` can use
it verbatim through its `SelectiveScan=` hook.

Differences from the reference, both internal to the fwd->bwd hand-off:
  * x holds one saved state per 256 steps (vmasr_sscan_chunk()), not per 2048;
  * bwd needs a scratch buffer for long sequences (allocated here with torch.empty).

Compute is libvmasr_hip.so (vm_asr_amd/csrc/sscan.hip).  No CPU fallback.
"""
import ctypes

import torch

from . import _lib

__all__ = ["fwd", "bwd", "fwd_oflex", "bwd_oflex", "SelectiveScanCore", "SelectiveScanOflex", "selective_scan_fn", "tune"]


def _chk(cond, msg):
    if not cond:
        raise RuntimeError(msg)


def _ptr(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def _check_common(u, delta, A, B, C, D, delta_bias):
    _chk(u.dtype in (torch.float32, torch.float16, torch.bfloat16), "u must be float32/float16/bfloat16")
    _chk(A.dtype == torch.float32, "A must be float32")
    _chk(delta.dtype == u.dtype and B.dtype == u.dtype and C.dtype == u.dtype,
         "delta, B, C must have the dtype of u")
    for name, t in (("u", u), ("delta", delta), ("A", A), ("B", B), ("C", C)):
        _chk(t.is_cuda, f"{name} must be a CUDA (HIP) tensor")
    _chk(u.dim() == 3, "u must be (batch, dim, seqlen)")
    batch, dim, seqlen = u.shape
    _chk(u.stride(-1) == 1 or seqlen == 1, "u must have contiguous last dimension")
    _chk(delta.stride(-1) == 1 or seqlen == 1, "delta must have contiguous last dimension")
    _chk(A.dim() == 2, "A must be (dim, dstate)")
    dstate = A.shape[1]
    _chk(B.dim() == 4 and C.dim() == 4, "B, C must be (batch, n_groups, dstate, seqlen)")
    n_groups = B.shape[1]
    _chk(dim % n_groups == 0, "dims should be dividable by n_groups")
    _chk(dstate <= 256, "selective_scan only supports state dimension <= 256")
    _chk(tuple(delta.shape) == (batch, dim, seqlen), "delta must have shape (batch, dim, seqlen)")
    _chk(tuple(A.shape) == (dim, dstate), "A must have shape (dim, dstate)")
    _chk(tuple(B.shape) == (batch, n_groups, dstate, seqlen), "B must have shape (batch, n_groups, dstate, seqlen)")
    _chk(tuple(C.shape) == (batch, n_groups, dstate, seqlen), "C must have shape (batch, n_groups, dstate, seqlen)")
    _chk(B.stride(-1) == 1 or seqlen == 1, "B must have contiguous last dimension")
    _chk(C.stride(-1) == 1 or seqlen == 1, "C must have contiguous last dimension")
    for name, t in (("D", D), ("delta_bias", delta_bias)):
        if t is not None:
            _chk(t.dtype == torch.float32, f"{name} must be float32")
            _chk(t.is_cuda, f"{name} must be a CUDA (HIP) tensor")
            _chk(tuple(t.shape) == (dim,), f"{name} must have shape (dim,)")
            _chk(t.stride(-1) == 1 or dim == 1, f"{name} must be contiguous")
    return batch, dim, seqlen, dstate, n_groups


def _fill(p, u, delta, A, B, C, D, delta_bias, out, x, delta_softplus, n_chunks):
    batch, dim, seqlen = u.shape
    p.batch, p.dim, p.seqlen, p.dstate, p.n_groups, p.n_chunks = batch, dim, seqlen, A.shape[1], B.shape[1], n_chunks
    p.dtype = _lib.torch_dtype_code(u.dtype)
    p.delta_softplus = int(bool(delta_softplus))
    p.A_d_stride, p.A_dstate_stride = A.stride(0), A.stride(1)
    p.B_batch_stride, p.B_group_stride, p.B_dstate_stride = B.stride(0), B.stride(1), B.stride(2)
    p.C_batch_stride, p.C_group_stride, p.C_dstate_stride = C.stride(0), C.stride(1), C.stride(2)
    p.u_batch_stride, p.u_d_stride = u.stride(0), u.stride(1)
    p.delta_batch_stride, p.delta_d_stride = delta.stride(0), delta.stride(1)
    if out is not None:
        p.out_batch_stride, p.out_d_stride = out.stride(0), out.stride(1)
    p.A_ptr, p.B_ptr, p.C_ptr = _ptr(A), _ptr(B), _ptr(C)
    p.D_ptr, p.delta_bias_ptr = _ptr(D), _ptr(delta_bias)
    p.u_ptr, p.delta_ptr, p.out_ptr, p.x_ptr = _ptr(u), _ptr(delta), _ptr(out), _ptr(x)


def fwd(u, delta, A, B, C, D=None, delta_bias=None, delta_softplus=False, nrows=1):
    """-> [out (batch,dim,L) u.dtype, x (batch,dim,n_chunks,2*dstate) fp32].
    `nrows` is accepted and ignored, as in the reference (cus/selective_scan.cpp:235)."""
    batch, dim, seqlen, dstate, n_groups = _check_common(u, delta, A, B, C, D, delta_bias)
    lib = _lib.lib()
    chunk = lib.vmasr_sscan_chunk()
    n_chunks = (seqlen + chunk - 1) // chunk
    with torch.cuda.device(u.device):
        out = torch.empty_like(delta)
        if out.stride(-1) != 1:
            out = torch.empty(delta.shape, dtype=delta.dtype, device=delta.device)
        x = torch.empty((batch, dim, n_chunks, dstate * 2), dtype=torch.float32, device=u.device)
        p = _lib.SScanParams()
        _fill(p, u, delta, A, B, C, D, delta_bias, out, x, delta_softplus, n_chunks)
        _lib.check(lib.vmasr_sscan_fwd(ctypes.byref(p), _lib.current_stream(u.device)), "selective_scan_fwd")
    return [out, x]


def bwd(u, delta, A, B, C, D, delta_bias, dout, x, delta_softplus, nrows=1):
    """-> [du, ddelta, dA, dB, dC, dD, ddelta_bias] (cus/selective_scan.cpp:241-349)."""
    batch, dim, seqlen, dstate, n_groups = _check_common(u, delta, A, B, C, D, delta_bias)
    _chk(dout.dtype == u.dtype and dout.is_cuda, "dout must match u")
    _chk(tuple(dout.shape) == (batch, dim, seqlen), "dout must have shape (batch, dim, seqlen)")
    _chk(dout.stride(-1) == 1 or seqlen == 1, "dout must have contiguous last dimension")
    lib = _lib.lib()
    chunk = lib.vmasr_sscan_chunk()
    n_chunks = (seqlen + chunk - 1) // chunk
    if n_chunks > 1:
        _chk(x is not None, "x is required when seqlen spans more than one chunk")
    if x is not None:
        _chk(x.dtype == torch.float32 and x.is_cuda and x.is_contiguous(), "x must be contiguous float32")
        _chk(tuple(x.shape) == (batch, dim, n_chunks, 2 * dstate), "x must have shape (batch, dim, n_chunks, 2*dstate)")
    with torch.cuda.device(u.device):
        du = torch.empty_like(u)
        ddelta = torch.empty_like(delta)
        if du.stride(-1) != 1:
            du = torch.empty(u.shape, dtype=u.dtype, device=u.device)
        if ddelta.stride(-1) != 1:
            ddelta = torch.empty(u.shape, dtype=u.dtype, device=u.device)
        f32 = dict(dtype=torch.float32, device=u.device)
        dB, dC, dA, dD, ddelta_bias = _lib.zeros_f32(u.device, B.shape, C.shape, A.shape,
                                                     (dim,) if D is not None else None,
                                                     (dim,) if delta_bias is not None else None)
        q = _lib.SScanBwdParams()
        _fill(q.f, u, delta, A, B, C, D, delta_bias, None, x, delta_softplus, n_chunks)
        q.dout_batch_stride, q.dout_d_stride = dout.stride(0), dout.stride(1)
        q.du_batch_stride, q.du_d_stride = du.stride(0), du.stride(1)
        q.ddelta_batch_stride, q.ddelta_d_stride = ddelta.stride(0), ddelta.stride(1)
        q.dA_d_stride, q.dA_dstate_stride = dA.stride(0), dA.stride(1)
        q.dout_ptr, q.du_ptr, q.ddelta_ptr = _ptr(dout), _ptr(du), _ptr(ddelta)
        q.dA_ptr, q.dB_ptr, q.dC_ptr, q.dD_ptr, q.ddelta_bias_ptr = _ptr(dA), _ptr(dB), _ptr(dC), _ptr(dD), _ptr(ddelta_bias)
        ws_bytes = lib.vmasr_sscan_bwd_workspace(ctypes.byref(q))
        ws = torch.empty(max(ws_bytes, 4) // 4, **f32) if ws_bytes else None
        q.ws_ptr, q.ws_bytes = _ptr(ws), ws_bytes
        _lib.check(lib.vmasr_sscan_bwd(ctypes.byref(q), _lib.current_stream(u.device)), "selective_scan_bwd")
    # reference casts dB/dC back to the input dtype (cus/selective_scan.cpp:347)
    return [du, ddelta, dA, dB.to(B.dtype), dC.to(C.dtype), dD, ddelta_bias]


def tune(rows=-1, split=-1):
    """Override the launch heuristics (benchmarks/tests only); -1 = automatic."""
    _lib.lib().vmasr_sscan_tune(int(rows), int(split))


class SelectiveScanCore(torch.autograd.Function):
    """model/vmamba.py:323-356 with the HIP operator underneath."""

    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda")
    def forward(ctx, u, delta, A, B, C, D=None, delta_bias=None, delta_softplus=False, nrows=1,
                backnrows=1, oflex=True):
        ctx.delta_softplus = delta_softplus
        out, x, *rest = fwd(u, delta, A, B, C, D, delta_bias, delta_softplus, 1)
        ctx.save_for_backward(u, delta, A, B, C, D, delta_bias, x)
        return out

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(ctx, dout, *args):
        u, delta, A, B, C, D, delta_bias, x = ctx.saved_tensors
        if dout.stride(-1) != 1:
            dout = dout.contiguous()
        du, ddelta, dA, dB, dC, dD, ddelta_bias, *rest = bwd(
            u, delta, A, B, C, D, delta_bias, dout, x, ctx.delta_softplus, 1)
        return (du, ddelta, dA, dB, dC, dD, ddelta_bias, None, None, None, None)


# ---- `selective_scan_cuda_oflex` surface (cusoflex/selective_scan_oflex.cpp:157-239 fwd, :241-352 bwd) ----------------
# oflex = "output flexible": with `out_float` the forward returns fp32 for 16-bit inputs (:218-219, :235-239) and the
# backward takes an fp32 `dout`.  The kernels of sscan.hip convert every operand to fp32 on load and compute in fp32, so
# running them on the (exactly) up-converted inputs gives bit-for-bit what an out_float store would: that is how the
# option is provided — two conversion passes instead of more kernel instantiations, since no shipped config selects a
# forward type that uses it (model/vmamba.py:785-841).  `out_float=False` is `fwd` / `bwd` unchanged.
def fwd_oflex(u, delta, A, B, C, D=None, delta_bias=None, delta_softplus=False, nrows=1, out_float=True):
    """-> [out (fp32 if out_float else u.dtype), x]"""
    if not out_float or u.dtype == torch.float32:
        return fwd(u, delta, A, B, C, D, delta_bias, delta_softplus, nrows)
    _check_common(u, delta, A, B, C, D, delta_bias)
    return fwd(u.float(), delta.float(), A, B.float(), C.float(), D, delta_bias, delta_softplus, nrows)


def bwd_oflex(u, delta, A, B, C, D, delta_bias, dout, x, delta_softplus, nrows=1):
    """dout may be fp32 for 16-bit inputs; du, ddelta, dB, dC come back in the input dtype (:335-347)."""
    if dout.dtype == u.dtype:
        return bwd(u, delta, A, B, C, D, delta_bias, dout, x, delta_softplus, nrows)
    _chk(dout.dtype == torch.float32, "dout must be float32 or match u")
    du, dd, dA, dB, dC, dD, db = bwd(u.float(), delta.float(), A, B.float(), C.float(), D, delta_bias, dout, x, delta_softplus, nrows)
    return [du.to(u.dtype), dd.to(delta.dtype), dA, dB.to(B.dtype), dC.to(C.dtype), dD, db]


class SelectiveScanOflex(torch.autograd.Function):
    """model/vmamba.py:358-392 with the HIP operator underneath."""

    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda")
    def forward(ctx, u, delta, A, B, C, D=None, delta_bias=None, delta_softplus=False, nrows=1, backnrows=1, oflex=True):
        ctx.delta_softplus = delta_softplus
        out, x, *rest = fwd_oflex(u, delta, A, B, C, D, delta_bias, delta_softplus, 1, oflex)
        ctx.save_for_backward(u, delta, A, B, C, D, delta_bias, x)
        return out

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(ctx, dout, *args):
        u, delta, A, B, C, D, delta_bias, x = ctx.saved_tensors
        if dout.stride(-1) != 1:
            dout = dout.contiguous()
        du, ddelta, dA, dB, dC, dD, ddelta_bias, *rest = bwd_oflex(u, delta, A, B, C, D, delta_bias, dout, x, ctx.delta_softplus, 1)
        return (du, ddelta, dA, dB, dC, dD, ddelta_bias, None, None, None, None)


def selective_scan_fn(u, delta, A, B, C, D=None, delta_bias=None, delta_softplus=False):
    return SelectiveScanCore.apply(u, delta, A, B, C, D, delta_bias, delta_softplus)
