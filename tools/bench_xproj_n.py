"""Micro-benchmark of the general-d_state x_proj / dt_proj kernels (csrc/xproj_n.hip) on the call shapes of BASELINE configs[4]
(DIMS 32, d_state 32, n_fft 2048, per-GPU batch 8): device time of forward / backward (library HIP events) and the HBM bytes
they move (xs + outputs) per second.   N=32 B=8 python tools/bench_xproj_n.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from vm_asr_amd import _lib  # noqa: E402
from vm_asr_amd.xproj import x_proj_dt  # noqa: E402

N = int(os.environ.get("N", 32))
B = int(os.environ.get("B", 8))
SHAPES = [(2, 1, 524288), (32, 1, 131072), (64, 2, 32768), (128, 4, 8192), (256, 8, 2048), (512, 16, 512)]   # (d_inner, dt_rank, L)
dev = "cuda:0"
tot = {"f": 0.0, "b": 0.0}
for D, R, L in SHAPES:
    K, C = 4, R + 2 * N
    g = torch.Generator(device=dev).manual_seed(0)
    xs = torch.randn(B, K, D, L, device=dev, generator=g).requires_grad_()
    Wx = (torch.randn(K, C, D, device=dev, generator=g) / D ** 0.5).requires_grad_()
    Wdt = torch.randn(K, D, R, device=dev, generator=g).requires_grad_()
    gd = torch.randn(B, K * D, L, device=dev, generator=g)
    gB, gC = torch.randn(B, K, N, L, device=dev, generator=g), torch.randn(B, K, N, L, device=dev, generator=g)
    iters = 5
    for it in range(iters + 1):
        if it == 1:
            torch.cuda.synchronize()
            _lib.prof_reset()
            _lib.prof_enable(True)
        dts, Bs, Cs = x_proj_dt(xs, Wx, Wdt, N)
        torch.autograd.backward([dts, Bs, Cs], [gd, gB, gC])
        xs.grad = Wx.grad = Wdt.grad = None
    torch.cuda.synchronize()
    _lib.prof_enable(False)
    pr = _lib.prof_collect()
    tf = pr["xproj_fwd"]["ms"] / iters
    ta, tb = pr["xproj_bwd_a"]["ms"] / iters, pr["xproj_bwd_b"]["ms"] / iters
    bf = B * K * L * (2 * D + C) * 4
    print(f"D={D:4d} R={R:2d} L={L:7d}  fwd {tf*1e3:8.1f} us {bf/tf/1e9:6.2f} TB/s | bwd_a {ta*1e3:8.1f} us | bwd_b {tb*1e3:8.1f} us", flush=True)
    tot["f"] += tf
    tot["b"] += ta + tb
print(f"sum over the six shapes: fwd {tot['f']:.2f} ms, bwd {tot['b']:.2f} ms")
