"""The two-stream step's fork/join topology with synthetic kernels inside ONE captured graph (dev tool): does the SECOND side-stream
phase (D-loss backward beside the generator's backward) overlap on replay?"""
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
small2 = torch.zeros(1024, device=dev)


def convs(reps=1):
    for _ in range(reps):
        cg.conv_fwd(xh, xl, wh, wl, bias, geom, k, stride, pad, rows, act=True)


def chain(nk, t=small):
    for _ in range(nk):
        t.add_(1.0)


def capture(mode):
    side = torch.cuda.Stream(dev)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, capture_error_mode="thread_local"):
        cur = torch.cuda.current_stream()
        if mode == "two":
            side.wait_stream(cur)
            with torch.cuda.stream(side):
                convs(2)                 # D(real)
            chain(800)                   # G forward
            side.wait_stream(cur)
            with torch.cuda.stream(side):
                convs(4)                 # D(fake) + G-loss dgrad
            cur.wait_stream(side)
            chain(1600)                  # G backward
            with torch.cuda.stream(side):
                chain(5, small2)
                convs(6)                 # D-loss backward
            cur.wait_stream(side)
        else:
            convs(2); chain(800); convs(4); chain(1600); chain(5, small2); convs(6)
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


convs(); chain(3); chain(3, small2); torch.cuda.synchronize()
print(f"serial {timeg(capture('serial')):.2f} ms   two-stream {timeg(capture('two')):.2f} ms   (convs ~0.8 ms each x 12, tiny kernels ~2.5 us each x 2400)")
