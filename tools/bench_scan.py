"""Micro-benchmark of the selective-scan kernels on the SS2D call shapes (dev tool)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vm_asr_amd import selective_scan as ss

SHAPES = [(8, 262144), (64, 65536), (128, 16384), (256, 4096), (512, 1024), (1024, 256)]
B = int(os.environ.get("B", 4))
SP = os.environ.get("SOFTPLUS", "1") == "1"
dev = "cuda:0"


def run(KD, L, tune, iters=20):
    g = torch.Generator(device=dev).manual_seed(0)
    u = torch.randn(B, KD, L, device=dev, generator=g)
    delta = 0.5 * torch.rand(B, KD, L, device=dev, generator=g)
    A = -0.5 * torch.rand(KD, 1, device=dev, generator=g)
    Bm = torch.randn(B, 4, 1, L, device=dev, generator=g)
    Cm = torch.randn(B, 4, 1, L, device=dev, generator=g)
    D = torch.randn(KD, device=dev, generator=g)
    bias = 0.5 * torch.rand(KD, device=dev, generator=g)
    dout = torch.randn(B, KD, L, device=dev, generator=g)
    ss.tune(*tune)
    out, x = ss.fwd(u, delta, A, Bm, Cm, D, bias, SP, 1)
    ss.bwd(u, delta, A, Bm, Cm, D, bias, dout, x, SP, 1)
    torch.cuda.synchronize()
    e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
    e0.record()
    for _ in range(iters):
        out, x = ss.fwd(u, delta, A, Bm, Cm, D, bias, SP, 1)
    e1.record()
    for _ in range(iters):
        ss.bwd(u, delta, A, Bm, Cm, D, bias, dout, x, SP, 1)
    e2.record()
    torch.cuda.synchronize()
    tf, tb = e0.elapsed_time(e1) / iters * 1e-3, e1.elapsed_time(e2) / iters * 1e-3
    if os.environ.get("KTIME", "1") == "1":  # device time of the kernels themselves (library HIP events)
        from vm_asr_amd import _lib
        _lib.prof_reset(); _lib.prof_enable(True)
        for _ in range(iters):
            out, x = ss.fwd(u, delta, A, Bm, Cm, D, bias, SP, 1)
        for _ in range(iters):
            ss.bwd(u, delta, A, Bm, Cm, D, bias, dout, x, SP, 1)
        _lib.prof_enable(False)
        pr = _lib.prof_collect()
        tf = sum(v["ms"] for k, v in pr.items() if k.startswith("sscan_fwd")) / iters * 1e-3
        tb = sum(v["ms"] for k, v in pr.items() if k.startswith("sscan_bwd")) / iters * 1e-3
    bf = (3 * KD + 2 * 4) * L * 4 * B
    bb = (5 * KD + 4 * 4) * L * 4 * B
    return tf, bf / tf / 1e12, tb, bb / tb / 1e12


if __name__ == "__main__":
    print(f"B={B}")
    for KD, L in SHAPES:
        for tune in [(-1, -1)] + ([(1, 0), (1, 1), (1, 2), (2, 0), (2, 1), (2, 2)] if os.environ.get("SWEEP", "1") == "1" else []):
            if tune[0] > 0 and (KD // 4) % tune[0]:
                continue
            tf, gf, tb, gb = run(KD, L, tune)
            print(f"KD={KD:5d} L={L:7d} tune={tune!s:8s} fwd {tf*1e6:8.1f} us {gf:5.2f} TB/s | bwd {tb*1e6:8.1f} us {gb:5.2f} TB/s", flush=True)
