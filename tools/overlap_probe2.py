"""Do two branches of ONE captured HIP graph run concurrently?  (dev tool)  A chain of tiny kernels on the capture stream and the
discriminator's layer-4 convolutions on a forked stream, captured serially and as a fork/join."""
import os
import sys

os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from vm_asr_amd import convgemm as cg
from vm_asr_amd import discriminator as D

dev = torch.device("cuda:0")
B = 4
T, periods, k, pad, stride, C = 122640, (2, 3, 5, 7, 11), 5, 2, 1, 1024
geom = []
for p in periods:
    h = -(-T // p)
    for _ in range(4):
        h = (h + 4 - 5) // 3 + 1
    geom.append((2 * B * p, h))
rows = -(-max(ns * h for ns, h in geom) // 256) * 256
n = len(periods)
x = torch.randn(n, rows, C, device=dev)
W = torch.randn(n, C, k * C, device=dev) / (k * C) ** 0.5
bias = torch.randn(n, C, device=dev)
xh, xl = D.split_bf16(x)
wh, wl = D.split_bf16(W)
gh, gl = D.split_bf16(torch.randn(n, rows, C, device=dev))
small = torch.zeros(1024, device=dev)


def convs():
    cg.conv_fwd(xh, xl, wh, wl, bias, geom, k, stride, pad, rows, act=True)
    cg.conv_dgrad(gh, gl, wh, wl, geom, k, stride, pad, rows)
    cg.conv_wgrad(gh, gl, xh, xl, geom, k, stride, pad)


def chain(nk):
    for _ in range(nk):
        small.add_(1.0)


def capture(mode, nk):
    side = torch.cuda.Stream(dev)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, capture_error_mode="thread_local"):
        cur = torch.cuda.current_stream()
        if mode == "fork":
            side.wait_stream(cur)
            with torch.cuda.stream(side):
                convs()
            chain(nk)
            cur.wait_stream(side)
        elif mode == "serial":
            convs()
            chain(nk)
        elif mode == "convs":
            convs()
        else:
            chain(nk)
    return g


def timeg(g, n=5):
    ts = []
    for it in range(n + 2):
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        g.replay()
        b.record()
        torch.cuda.synchronize()
        if it >= 2:
            ts.append(a.elapsed_time(b))
    return sum(ts) / len(ts)


convs(); chain(3); torch.cuda.synchronize()
for nk in (300, 1000):
    r = {m: timeg(capture(m, nk)) for m in ("chain", "convs", "serial", "fork")}
    print(f"{nk} tiny kernels: chain {r['chain']:.2f} ms  convs {r['convs']:.2f} ms  serial {r['serial']:.2f} ms  fork/join {r['fork']:.2f} ms")
