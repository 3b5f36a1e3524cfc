"""fp32 Linear on the GPU (hipBLASLt via F.linear, the HIP small-linear row map) vs the same product on the CPU (what the
reference's fp32 run uses) — both measured against float64.  dev tool."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, torch.nn.functional as F
from vm_asr_amd.linear import linear
print("allow_tf32", torch.backends.cuda.matmul.allow_tf32, "blas", torch.backends.cuda.preferred_blas_library())
g = torch.Generator().manual_seed(0)
def rel(a, b):
    return ((a.double() - b).pow(2).mean().sqrt() / b.pow(2).mean().sqrt()).item(), ((a.double() - b).abs().max() / b.abs().max()).item()
for rows, K, N in [(16384, 16, 64), (16384, 32, 16), (16384, 16, 64), (16384, 64, 16), (4096, 32, 128), (4096, 128, 32), (1024, 64, 256),
                   (1024, 256, 64), (256, 128, 512), (256, 512, 128), (256, 256, 128), (65536, 8, 32), (65536, 32, 8), (262144, 1, 4),
                   (262144, 4, 1), (262144, 2, 1), (16384, 72, 16), (65536, 9, 8), (4096, 64, 32), (16384, 32, 16)]:
    x = torch.randn(rows, K, generator=g); w = torch.randn(N, K, generator=g) / K ** 0.5; b = 0.1 * torch.randn(N, generator=g)
    y64 = F.linear(x.double(), w.double(), b.double())
    yc = F.linear(x, w, b)
    with torch.no_grad():
        yg = linear(x.cuda(), w.cuda(), b.cuda()).cpu()
        yg2 = F.linear(x.cuda(), w.cuda(), b.cuda()).cpu()
        yg3 = (x.cuda() @ w.cuda().t() + b.cuda()).cpu()
    print(f"rows {rows:7d} K {K:4d} N {N:4d}: cpu rms/max {rel(yc, y64)[0]:.2e}/{rel(yc, y64)[1]:.2e}  vmasr.linear {rel(yg, y64)[0]:.2e}/{rel(yg, y64)[1]:.2e}"
          f"  F.linear(gpu) {rel(yg2, y64)[0]:.2e}/{rel(yg2, y64)[1]:.2e}  mm+b {rel(yg3, y64)[0]:.2e}")
# elementwise families ATen
x = torch.randn(1 << 20, generator=g)
for name, f in [("gelu", F.gelu), ("silu", F.silu), ("exp2", torch.exp2)]:
    y64 = f(x.double()); print(name, "cpu", rel(f(x), y64), "gpu", rel(f(x.cuda()).cpu(), y64))
