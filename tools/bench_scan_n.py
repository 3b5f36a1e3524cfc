"""Micro-benchmark of the general-d_state selective-scan kernels (csrc/sscan_n.hip) on the call shapes of BASELINE configs[4]
(DIMS 32, d_state 32, n_fft 2048, per-GPU batch 8).  Prints per call: device time of the forward / backward kernels (library HIP
events), contract-algorithmic TB/s and lane-instruction slots per state-step (time x 16 384 lanes x 2.4 GHz /
(B KD N L)) — the VALU roofline of this operator: ~10 (forward) / ~25 (backward) packed instructions per state-step.
  N=32 B=8 python tools/bench_scan_n.py            (VMASR_SSCAN_N_LEGACY=1: the one-state-at-a-time kernels of sscan.hip)
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from vm_asr_amd import _lib, selective_scan as ss  # noqa: E402

N = int(os.environ.get("N", 32))
B = int(os.environ.get("B", 8))
SHAPES = [(8, 524288), (128, 131072), (256, 32768), (512, 8192), (1024, 2048), (2048, 512)]
if os.environ.get("SHAPES"):
    SHAPES = [tuple(int(v) for v in s.split("x")) for s in os.environ["SHAPES"].split(",")]
TUNES = [(-1, -1)] + ([(1, -1), (2, -1), (4, -1)] if os.environ.get("SWEEP", "0") == "1" else [])
LANE_RATE = 256 * 4 * 16 * 2.4e9   # lane-instructions per second of the chip
dev = "cuda:0"


def run(KD, L, tune, iters):
    g = torch.Generator(device=dev).manual_seed(0)
    u = torch.randn(B, KD, L, device=dev, generator=g)
    delta = 0.5 * torch.rand(B, KD, L, device=dev, generator=g) * (0.2 if N > 8 else 1.0)
    A = -(1.0 + torch.arange(N, device=dev, dtype=torch.float32))[None].repeat(KD, 1) * (0.5 + torch.rand(KD, N, device=dev, generator=g))
    Bm = torch.randn(B, 4, N, L, device=dev, generator=g)
    Cm = torch.randn(B, 4, N, L, device=dev, generator=g)
    D = torch.randn(KD, device=dev, generator=g)
    bias = 0.5 * torch.rand(KD, device=dev, generator=g) - 3.0
    dout = torch.randn(B, KD, L, device=dev, generator=g)
    ss.tune(*tune)
    out, x = ss.fwd(u, delta, A, Bm, Cm, D, bias, True, 1)
    ss.bwd(u, delta, A, Bm, Cm, D, bias, dout, x, True, 1)
    torch.cuda.synchronize()
    _lib.prof_reset()
    _lib.prof_enable(True)
    for _ in range(iters):
        out, x = ss.fwd(u, delta, A, Bm, Cm, D, bias, True, 1)
    for _ in range(iters):
        ss.bwd(u, delta, A, Bm, Cm, D, bias, dout, x, True, 1)
    torch.cuda.synchronize()
    _lib.prof_enable(False)
    pr = _lib.prof_collect()
    tf = sum(v["ms"] for k, v in pr.items() if k.startswith("sscan_fwd")) / iters * 1e-3
    tb = sum(v["ms"] for k, v in pr.items() if k.startswith("sscan_bwd")) / iters * 1e-3
    parts = {k: round(v["ms"] / iters * 1e3, 1) for k, v in sorted(pr.items()) if k.startswith("sscan")}
    return tf, tb, parts


if __name__ == "__main__":
    print(f"N={N} B={B} legacy={os.environ.get('VMASR_SSCAN_N_LEGACY', '0')}")
    tot_f = tot_b = 0.0
    for KD, L in SHAPES:
        for tune in TUNES:
            if tune[0] > 0 and (KD // 4) % tune[0]:
                continue
            steps = B * KD * N * L
            tf, tb, parts = run(KD, L, tune, 3 if steps > 5e8 else 10)
            bf, bb = (3 * KD + 2 * 4 * N) * L * 4 * B, (5 * KD + 4 * 4 * N) * L * 4 * B
            print(f"KD={KD:5d} L={L:7d} tune={tune!s:8s} fwd {tf*1e6:9.1f} us {bf/tf/1e12:5.2f} TB/s {tf*LANE_RATE/steps:6.1f} slots | "
                  f"bwd {tb*1e6:9.1f} us {bb/tb/1e12:5.2f} TB/s {tb*LANE_RATE/steps:6.1f} slots   {parts}", flush=True)
            if tune == TUNES[0]:
                tot_f += tf
                tot_b += tb
    print(f"sum over the six shapes: fwd {tot_f*1e3:.2f} ms, bwd {tot_b*1e3:.2f} ms")
