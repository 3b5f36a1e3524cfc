"""Deep-stage SS2D core (csrc/ss2d_deep.hip) vs the unfused HIP chain it replaces, forward + backward, at the three deep call
shapes of vm_asr_48k (B from $B, default 4; activations bf16 as under autocast).  Device time per kernel from the library's
HIP events (the chain's ATen glue is not in those numbers) and wall time per call.   usage: B=4 python tools/bench_ss2d_deep.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vm_asr_amd import _lib  # noqa: E402
from vm_asr_amd.vmamba import SS2D  # noqa: E402

B = int(os.environ.get("B", "4"))
REP = int(os.environ.get("REP", "20"))
for D, H in ((64, 64), (128, 32), (256, 16)):
    torch.manual_seed(0)
    m = SS2D(d_model=D // 2, d_state=1, ssm_ratio=2.0, dt_rank="auto", forward_type="v5").cuda()
    x = torch.randn(B, D, H, H, device="cuda").to(torch.bfloat16)
    gy = torch.randn(B, H, H, D, device="cuda")
    for flag in ("1", "0"):
        os.environ["VMASR_SS2D_DEEP"] = flag

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
        print(f"D={D:3d} {H}x{H} B={B} {'deep core' if flag == '1' else 'unfused  '}: lib kernels {dev * 1e3:8.1f} us/call  wall {wall * 1e3:8.1f} us/call")
        for k, v in sorted(prof.items()):
            print(f"      {k:18s} {v['launches'] // REP:3d} x {v['ms'] / v['launches'] * 1e3:8.1f} us")
os.environ.pop("VMASR_SS2D_DEEP", None)
