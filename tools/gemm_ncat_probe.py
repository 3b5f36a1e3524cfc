"""The forward bf16x3 triple of the MPD layers: three (M x K).(K x N) products (as now) vs two launches with the weight operand
concatenated along N, A_hi.[B_hi | B_lo] (N' = 2N: more output tiles per launch, better wave quantisation) + A_lo.B_hi.  Device
time under graph replay at the three layer shapes (dev tool)."""
import os
os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")
import torch

def timed(fn, rep=20):
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3): fn()
    torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(rep): fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3): g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (3 * rep) * 1e3

n = 5
f32 = torch.float32
for (M, K, N) in [(36608, 640, 512), (12288, 2560, 1024), (4864, 5120, 1024), (9728, 5120, 1024)]:
    ch = torch.randn(n, M, K, device="cuda").to(torch.bfloat16); cl = torch.randn_like(ch)
    wcat = torch.randn(n, K, 3 * N, device="cuda").to(torch.bfloat16)
    wth, wtl, w2 = wcat[:, :, :N], wcat[:, :, 2 * N:], wcat[:, :, N:]
    parts = torch.empty(3, n, M, N, dtype=f32, device="cuda")
    o2 = torch.empty(n, M, 2 * N, dtype=f32, device="cuda")
    def three():
        torch.bmm(ch, wth, out_dtype=f32, out=parts[0]); torch.bmm(cl, wth, out_dtype=f32, out=parts[1]); torch.bmm(ch, wtl, out_dtype=f32, out=parts[2])
    def two():
        torch.bmm(ch, w2, out_dtype=f32, out=o2); torch.bmm(cl, wth, out_dtype=f32, out=parts[1])
    t3, t2 = timed(three), timed(two)
    fl = 3 * 2.0 * n * M * K * N
    print(f"M={M} K={K} N={N}: three products {t3:8.1f} us ({fl / t3 / 1e6:6.0f} TFLOP/s)   N-concatenated {t2:8.1f} us ({fl / t2 / 1e6:6.0f} TFLOP/s)", flush=True)
