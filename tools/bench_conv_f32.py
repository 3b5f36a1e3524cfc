"""The 32 -> 128 layer of the MPD at configs[2] sizes: exact-f32 implicit GEMM (csrc/convgemm.hip OPS 1) vs the round-5 path
(im2col + fp32 library GEMM + bias/GELU pass + split; GEMM + col2im).  usage: python tools/bench_conv_f32.py [batch=4]"""
import os, sys
os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vm_asr_amd import convgemm as cg
from vm_asr_amd import discriminator as D


def timeit(fn, n=20, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
dev = torch.device("cuda:0")
T, periods, k, stride, pad, Cin, Cout = 122640, (2, 3, 5, 7, 11), 5, 3, 2, 32, 128
geom = []
for p in periods:
    h = -(-T // p)
    h1 = (h + 4 - 5) // 3 + 1            # positions after the first layer
    geom.append((B * p, h1))
H1 = [cg.out_positions(h, k, stride, pad) for _, h in geom]
Ms = [ns * h1 for (ns, _), h1 in zip(geom, H1)]
rows_in = -(-max(ns * h for ns, h in geom) // 256) * 256
rows_out = -(-max(Ms) // 256) * 256
n = len(periods)
x = torch.randn(n, rows_in, Cin, device=dev)
W = torch.randn(n, Cout, k * Cin, device=dev) / (k * Cin) ** 0.5
bias = torch.randn(n, Cout, device=dev)
gy = torch.randn(n, rows_out, Cout, device=dev)
Wt = W.view(n, Cout, k, Cin).permute(0, 3, 2, 1).reshape(n, Cin, k * Cout).contiguous()
flops = sum(2.0 * M * k * Cin * Cout for M in Ms)
print(f"32 -> 128 layer, batch {B}: rows_out {rows_out} x {n} slots, {flops / 1e9:.1f} GFLOP per pass, output {n * rows_out * Cout * 12 / 1e6:.0f} MB (pre + act + pair)")
t = timeit(lambda: cg.conv_fwd_f32(x, W, bias, geom, k, stride, pad, rows_out, act=True))
print(f"   f32 fwd    {t:8.1f} us  {flops / t / 1e6:7.1f} TFLOP/s f32")
t = timeit(lambda: cg.conv_dgrad_f32(gy, Wt, geom, k, stride, pad, rows_in))
print(f"   f32 dgrad  {t:8.1f} us  {flops / t / 1e6:7.1f} TFLOP/s f32")
xh, xl = D.split_bf16(x)
gh, gl = D.split_bf16(gy)
t = timeit(lambda: cg.conv_wgrad(gh, gl, xh, xl, geom, k, stride, pad))
print(f"   bf16x3 wgrad {t:6.1f} us")
wh, wl = D.split_bf16(W)
t = timeit(lambda: cg.conv_fwd(xh, xl, wh, wl, bias, geom, k, stride, pad, rows_out, act=True))
print(f"   bf16x3 fwd {t:8.1f} us (for comparison: below the accuracy gate)")
sgeom = tuple(geom)
def old_fwd():
    cols = D._StackedIm2ColFn.apply(k, stride, pad, rows_out, sgeom, x)
    y = D._BatchedLinearFn.apply(cols, W, bias, torch.float32, True)
    return D.split_bf16(y)
t = timeit(old_fwd)
print(f"   round-5 fwd (im2col + fp32 GEMM + bias/GELU + split) {t:8.1f} us")
