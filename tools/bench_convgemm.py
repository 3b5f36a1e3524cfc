"""Microbenchmark: one stacked MPD convolution layer (5 period discriminators, configs[2] sizes) as
  (a) the implicit bf16x3 MFMA kernels of csrc/convgemm.hip (forward+epilogue, dgrad, wgrad), and
  (b) round 3's path: im2col_split + weight_prep + 3 hipBLASLt GEMMs + bias_gelu; gelu_bwd_split + cat-GEMM + col2im + 3 wgrad GEMMs + sum.
Usage: python tools/bench_convgemm.py [batch=8]   (8 = the discriminator pass over [real; fake] at per-GPU batch 4)"""
import os
import sys

os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from vm_asr_amd import convgemm as cg
from vm_asr_amd import discriminator as D


def timeit(fn, n=10, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3      # us


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    dev = torch.device("cuda:0")
    T = 122640
    periods = (2, 3, 5, 7, 11)
    # positions per sequence entering each layer
    H = []
    for p in periods:
        h = -(-T // p)
        hs = [h]
        for _ in range(4):
            hs.append((hs[-1] + 4 - 5) // 3 + 1)
        H.append(hs)                       # hs[l] = input positions of layer l (0-based); layer 4 has stride 1
    layers = [(2, 128, 512, 3), (3, 512, 1024, 3), (4, 1024, 1024, 1)]
    torch.manual_seed(0)
    for li, Cin, Cout, stride in layers:
        k, pad = 5, 2
        geom = [(B * p, H[i][li]) for i, p in enumerate(periods)]
        H1 = [cg.out_positions(h, k, stride, pad) for _, h in geom]
        Ms = [ns * h1 for (ns, _), h1 in zip(geom, H1)]
        rows_in = -(-max(ns * h for ns, h in geom) // 256) * 256
        rows_out = -(-max(Ms) // 256) * 256
        n = len(periods)
        x = torch.randn(n, rows_in, Cin, device=dev)
        for i, (ns, h) in enumerate(geom):
            x[i, ns * h:] = 0
        W = torch.randn(n, Cout, k * Cin, device=dev) / (k * Cin) ** 0.5
        bias = torch.randn(n, Cout, device=dev)
        gy = torch.randn(n, rows_out, Cout, device=dev)
        flops = sum(2.0 * M * k * Cin * Cout for M in Ms)
        xh, xl = D.split_bf16(x)
        wh, wl = D.split_bf16(W)
        Wt = W.view(n, Cout, k, Cin).permute(0, 3, 2, 1).reshape(n, Cin, k * Cout).contiguous()
        wth, wtl = D.split_bf16(Wt)
        gh, gl = D.split_bf16(gy)
        t_f = timeit(lambda: cg.conv_fwd(xh, xl, wh, wl, bias, geom, k, stride, pad, rows_out, act=True))
        t_d = timeit(lambda: cg.conv_dgrad(gh, gl, wth, wtl, geom, k, stride, pad, rows_in))
        t_w = timeit(lambda: cg.conv_wgrad(gh, gl, xh, xl, geom, k, stride, pad))
        sp = cg.wgrad_splits(n, Cin, Cout, k, max(Ms))
        print(f"layer {li} ({Cin}->{Cout}, stride {stride}) rows_out {rows_out} rows_in {rows_in}: {flops / 1e9:.0f} GFLOP fp32-equivalent per pass")
        for nm, t in (("fwd", t_f), ("dgrad", t_d), (f"wgrad(S={sp})", t_w)):
            print(f"   mfma {nm:12s} {t:8.1f} us   {3 * flops / t / 1e6:7.1f} TFLOP/s bf16 ({flops / t / 1e6:6.1f} fp32-eq)")
        # round-3 path
        xg = x.clone().requires_grad_(True)
        Wg = W.clone().requires_grad_(True)
        bg = bias.clone().requires_grad_(True)
        sgeom = tuple(geom)

        def old_fwd():
            return D._StackedConvSplitFn.apply(k, stride, pad, rows_out, True, sgeom, Wg, bg, xg)
        t_of = timeit(lambda: old_fwd())

        def old_fb():
            y = old_fwd()
            y.backward(gy)
            xg.grad = Wg.grad = bg.grad = None
        t_ofb = timeit(old_fb)
        print(f"   r03  fwd          {t_of:8.1f} us;  fwd+bwd {t_ofb:8.1f} us   (mfma fwd+dgrad+wgrad {t_f + t_d + t_w:8.1f} us + gelu_bwd_split)")


if __name__ == "__main__":
    main()
