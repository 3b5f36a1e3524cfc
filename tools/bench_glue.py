"""ss2d_pre / ln_gate (csrc/ss2d_glue.hip) per SS2D call shape of vm_asr_48k at B=$B (default 4), bf16 activations:
device time per launch from the library's HIP events (dev tool)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vm_asr_amd import _lib  # noqa: E402
from vm_asr_amd.ss2d_glue import ln_gate, ss2d_pre  # noqa: E402

B = int(os.environ.get("B", "4"))
REP = 20
for D, H, Bm in ((2, 512, 1), (16, 256, 1), (32, 128, 1), (64, 64, 1), (128, 32, 1), (256, 16, 1), (64, 64, 2), (128, 32, 2), (256, 16, 2)):
    b = B * Bm
    xz = torch.randn(b, H, H, 2 * D, device="cuda").to(torch.bfloat16).requires_grad_()
    y = torch.randn(b, D, H * H, device="cuda").requires_grad_()
    w, bb = torch.ones(D, device="cuda", requires_grad=True), torch.zeros(D, device="cuda", requires_grad=True)
    for tag in ("pre", "ln_gate"):
        def step():
            if tag == "pre":
                xT, sz = ss2d_pre(xz)
                (xT.float().sum() + sz.float().sum()).backward()
            else:
                sz = torch.ones(b, H, H, D, device="cuda", dtype=torch.bfloat16, requires_grad=True)
                ln_gate(y, sz, w, bb, 1e-5).float().sum().backward()
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
            print(f"D={D:3d} {H:3d}x{H:<3d} B={b}: {k:10s} {n} launches/iter, {v['ms'] / v['launches'] * 1e3:7.1f} us avg, {v['alg_bytes'] / v['ms'] / 1e6:8.1f} GB/s")
