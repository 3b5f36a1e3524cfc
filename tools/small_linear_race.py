"""Is vmasr_small_linear_bwd (1 -> 4, fp32 x, bf16 gy, 262144 rows) bit-reproducible on its own, and beside work on a second stream?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("TENSILE_STREAMK_DATA_PARALLEL", "1")
import torch
from vm_asr_amd import linear, _lib

dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
rows = 262144
for (i_f, o_f, xdt, gdt) in [(1, 4, torch.float32, torch.bfloat16), (4, 1, torch.bfloat16, torch.bfloat16), (2, 1, torch.float32, torch.bfloat16), (1, 4, torch.float32, torch.float32)]:
    x = torch.randn(rows, i_f, generator=g).to(dev).to(xdt).requires_grad_(True)
    w = torch.randn(o_f, i_f, generator=g).to(dev).requires_grad_(True)
    b = torch.randn(o_f, generator=g).to(dev).requires_grad_(True)
    gy = (1e-3 * torch.randn(rows, o_f, generator=g)).to(dev).to(gdt)
    side = torch.cuda.Stream()
    big = torch.randn(4096, 4096, device=dev)

    def run():
        for t in (x, w, b):
            t.grad = None
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=gdt == torch.bfloat16):
            y = linear._SmallLinearFn.apply(x, w, b, gdt)
        y.backward(gy)
        return w.grad.clone(), b.grad.clone(), x.grad.clone()
    ref = run()
    for mode in ("alone", "beside_gemm", "beside_fill", "churn"):
        bad = torch.zeros(3, device=dev)
        keep = []
        for it in range(3000):
            if mode == "beside_gemm" and it % 4 == 0:
                side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side):
                    big2 = big @ big
            if mode == "beside_fill" and it % 2 == 0:
                with torch.cuda.stream(side):
                    t = torch.empty(1 << 22, device=dev).fill_(float(it))
            if mode == "churn":      # allocator churn on the main stream: small blocks allocated and freed around the call
                keep = [torch.full((n,), 1e30, device=dev) for n in (3, 7, 2048, 8, 64, 2048, 1)]
                if it % 3 == 0:
                    keep = keep[::2]
            got = run()
            for k in range(3):
                bad[k] += (got[k] != ref[k]).any()
        torch.cuda.synchronize()
        print(f"in {i_f} out {o_f} x {xdt} gy {gdt} {mode}: runs with dw/db/dx differing = {bad.tolist()}", flush=True)
