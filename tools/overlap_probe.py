"""Do the generator's small kernels and the period discriminator's chip-filling MFMA kernels overlap when they are issued on two
streams?  (dev tool: the measurement behind the two-stream train step.)
  A: the generator-only train step (HIP graph replay) alone
  B: a run of discriminator convolution kernels (layer 4 forward / dgrad / wgrad) alone
  C: both at once, B on a side stream
usage: python tools/overlap_probe.py [batch=4]"""
import os
import sys

os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
from vm_asr_amd import convgemm as cg
from vm_asr_amd import discriminator as D


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    dev = torch.device("cuda:0")
    cfg = bench.make_config("vm_asr_48k", B)
    tr = bench.build_trainer(cfg, dev, amp=True, capturable=True)
    for m in tr.models.values():
        if m is not None:
            m.train()
    batch = bench.synth_batch(cfg, dev, 0)
    assert tr.enable_graphs(batch, warmup=2), getattr(tr, "graph_error", None)

    # layer 4 of the stacked discriminators at the bench's size
    T, periods, k, pad, stride, C = 122640, (2, 3, 5, 7, 11), 5, 2, 1, 1024
    geom = []
    for p in periods:
        h = -(-T // p)
        for _ in range(4):
            h = (h + 4 - 5) // 3 + 1
        geom.append((2 * B * p, h))
    Ms = [ns * h for ns, h in geom]
    rows = -(-max(Ms) // 256) * 256
    n = len(periods)
    x = torch.randn(n, rows, C, device=dev)
    W = torch.randn(n, C, k * C, device=dev) / (k * C) ** 0.5
    bias = torch.randn(n, C, device=dev)
    xh, xl = D.split_bf16(x)
    wh, wl = D.split_bf16(W)
    gh, gl = D.split_bf16(torch.randn(n, rows, C, device=dev))

    def convs(reps):
        for _ in range(reps):
            cg.conv_fwd(xh, xl, wh, wl, bias, geom, k, stride, pad, rows, act=True)
            cg.conv_dgrad(gh, gl, wh, wl, geom, k, stride, pad, rows)
            cg.conv_wgrad(gh, gl, xh, xl, geom, k, stride, pad)

    side = torch.cuda.Stream(dev)
    s2 = torch.cuda.Stream(dev)
    main_s = torch.cuda.current_stream(dev)

    def run(g_steps, conv_reps, n=5):
        ts = []
        for it in range(n + 2):
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(main_s)
            side.wait_stream(main_s)
            if conv_reps:
                with torch.cuda.stream(side):
                    convs(conv_reps)
            if os.environ.get("PROBE_G_STREAM") == "1":
                s2.wait_stream(main_s)
                with torch.cuda.stream(s2):
                    for _ in range(g_steps):
                        tr.train_step(*batch)
                main_s.wait_stream(s2)
            else:
                for _ in range(g_steps):
                    tr.train_step(*batch)
            main_s.wait_stream(side)
            b.record(main_s)
            torch.cuda.synchronize()
            if it >= 2:
                ts.append(a.elapsed_time(b))
        return sum(ts) / len(ts)

    # control: a chain of tiny eager kernels instead of the graph replay
    small = torch.zeros(1024, device=dev)
    mid = torch.zeros(4 << 20, device=dev)

    def run2(kind, conv_reps, n=5):
        ts = []
        for it in range(n + 2):
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(main_s)
            side.wait_stream(main_s)
            if conv_reps:
                with torch.cuda.stream(side):
                    convs(conv_reps)
            if kind == "small":
                for _ in range(600):
                    small.add_(1.0)
            elif kind == "mid":
                for _ in range(300):
                    mid.add_(1.0)
            main_s.wait_stream(side)
            b.record(main_s)
            torch.cuda.synchronize()
            if it >= 2:
                ts.append(a.elapsed_time(b))
        return sum(ts) / len(ts)

    for kind in (() if os.environ.get("PROBE_SKIP_EAGER") else ("small", "mid")):
        tA, tB, tC = run2(kind, 0), run2("none", 1), run2(kind, 1)
        print(f"eager {kind} chain alone {tA:6.2f} ms   convs alone {tB:6.2f} ms   both {tC:6.2f} ms")
    for reps in (1, 2):
        tA, tB, tC = run(1, 0), run(0, reps), run(1, reps)
        print(f"conv reps {reps}:  G step alone {tA:6.2f} ms   convs alone {tB:6.2f} ms   both {tC:6.2f} ms   (sum {tA + tB:6.2f}, max {max(tA, tB):6.2f})")


if __name__ == "__main__":
    main()
