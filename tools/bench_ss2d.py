"""SS2D core: fused operator (csrc/ss2d.hip) vs the unfused HIP chain, forward + backward, at the three fused call
shapes of vm_asr_48k (B from $B, default 4; activations bf16 as under autocast).  Device time per kernel from the
library's HIP events; the chain's ATen glue (casts, contiguous) is NOT in those numbers, so wall time per call (HIP
graph-free, synchronised) is printed too.   usage: B=4 python tools/bench_ss2d.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vm_asr_amd import _lib  # noqa: E402
from vm_asr_amd.vmamba import SS2D  # noqa: E402

B = int(os.environ.get("B", "4"))
REP = int(os.environ.get("REP", "20"))
for D, H in ((32, 128), (16, 256), (2, 512)):
    torch.manual_seed(0)
    m = SS2D(d_model=D // 2, d_state=1, ssm_ratio=2.0, dt_rank=1, forward_type="v5").cuda()
    x = torch.randn(B, D, H, H, device="cuda").to(torch.bfloat16)
    gy = torch.randn(B, H, H, D, device="cuda")
    for flag in ("1", "0"):
        os.environ["VMASR_SS2D_FUSED"] = flag

        def step():
            xi = x.clone().requires_grad_()
            with torch.autocast("cuda", dtype=torch.bfloat16):
                y = m.forward_core(xi)
            y.backward(gy.to(y.dtype))
        for _ in range(3):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(REP):
            step()
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) / REP * 1e3
        _lib.prof_reset(); _lib.prof_enable(True)
        for _ in range(REP):
            step()
        torch.cuda.synchronize()
        _lib.prof_enable(False)
        prof = _lib.prof_collect()
        dev = sum(v["ms"] for v in prof.values()) / REP
        el = B * D * H * H
        print(f"D={D:3d} {H}x{H} B={B} {'fused  ' if flag == '1' else 'unfused'}: lib kernels {dev * 1e3:8.1f} us/call  "
              f"wall {wall * 1e3:8.1f} us/call  ({el * 4 * 4 / dev / 1e6:7.1f} GB/s per 16 B/(row,pos))")
        for k, v in sorted(prof.items()):
            print(f"      {k:18s} {v['launches'] // REP:3d} x {v['ms'] / v['launches'] * 1e3:8.1f} us   {v['alg_bytes'] / v['ms'] / 1e6:8.1f} GB/s")
