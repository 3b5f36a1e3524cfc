"""x_proj / dt_proj map (csrc/xproj.hip) at the deep-stage shapes of vm_asr_48k (bf16 autocast feeds it fp32 scan
streams): device time per launch from the library's HIP events (dev tool)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vm_asr_amd import _lib  # noqa: E402
from vm_asr_amd.xproj import x_proj_dt  # noqa: E402

REP = 20
for B, D, L, R in ((4, 64, 4096, 2), (8, 64, 4096, 2), (4, 128, 1024, 4), (8, 128, 1024, 4), (4, 256, 256, 8), (4, 32, 16384, 1)):
    K, N = 4, 1
    xs = torch.randn(B, K, D, L, device="cuda", requires_grad=True)
    Wx = torch.randn(K, R + 2 * N, D, device="cuda", requires_grad=True)
    Wdt = torch.randn(K, D, R, device="cuda", requires_grad=True)

    def step():
        dts, Bs, Cs = x_proj_dt(xs, Wx, Wdt, N)
        (dts.sum() + Bs.sum() + Cs.sum()).backward()
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    _lib.prof_reset(); _lib.prof_enable(True)
    for _ in range(REP):
        step()
    torch.cuda.synchronize()
    _lib.prof_enable(False)
    for k, v in sorted(_lib.prof_collect().items()):
        n = v["launches"] // REP
        print(f"B={B} D={D:3d} L={L:5d} R={R}: {k:14s} {n} launches/iter, {v['ms'] / v['launches'] * 1e3:7.1f} us avg, {v['alg_bytes'] / v['ms'] / 1e6:8.1f} GB/s")
