"""Microbenchmark: the streaming few-feature Linear kernel (csrc/skinny.hip) against F.linear (hipBLASLt) at the generator's shapes (dev tool)."""
import os
import sys

os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F

from vm_asr_amd import linear as L

dev = torch.device("cuda:0")


def timeit(fn, n=40):
    """us per call inside a replayed HIP graph of n calls (eager calls of this size only measure the launch path)"""
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n):
            fn()
    g.replay()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(5):
        g.replay()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / (5 * n) * 1e3


for rows, K, N in ((262144, 9, 8), (262144, 16, 8), (262144, 8, 16), (65536, 16, 32), (65536, 16, 72), (65536, 72, 16), (65536, 32, 16),
                   (65536, 16, 64), (65536, 64, 16), (32768, 32, 64), (32768, 64, 32), (16384, 64, 32), (16384, 32, 64)):
    x = torch.randn(rows, K, device=dev).bfloat16()
    w = torch.randn(N, K, device=dev)
    wb = w.bfloat16()
    b = torch.randn(N, device=dev)
    bb = b.bfloat16()
    t_sk = timeit(lambda: L._skinny(x, w, b, torch.bfloat16))
    t_lt = timeit(lambda: F.linear(x, wb, bb))
    byts = rows * (K + N) * 2
    print(f"rows {rows:7d} K {K:3d} N {N:3d}: skinny {t_sk:6.1f} us ({byts / t_sk / 1e6:5.2f} TB/s)   F.linear {t_lt:6.1f} us")
