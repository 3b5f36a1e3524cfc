"""One forward + backward selective-scan call per SS2D shape (B=4) for PMC collection (dev tool).
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d out -- python tools/pmc_scan.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vm_asr_amd import selective_scan as ss
dev = "cuda:0"
B = int(os.environ.get("B", 4))
for KD, L in [(8, 262144), (64, 65536), (128, 16384), (256, 4096), (512, 1024), (1024, 256)]:
    g = torch.Generator(device=dev).manual_seed(0)
    u = torch.randn(B, KD, L, device=dev, generator=g)
    delta = 0.5 * torch.rand(B, KD, L, device=dev, generator=g)
    A = -0.5 * torch.rand(KD, 1, device=dev, generator=g)
    Bm = torch.randn(B, 4, 1, L, device=dev, generator=g)
    Cm = torch.randn(B, 4, 1, L, device=dev, generator=g)
    D = torch.randn(KD, device=dev, generator=g)
    bias = 0.5 * torch.rand(KD, device=dev, generator=g)
    dout = torch.randn(B, KD, L, device=dev, generator=g)
    for _ in range(3):
        out, x = ss.fwd(u, delta, A, Bm, Cm, D, bias, True, 1)
        ss.bwd(u, delta, A, Bm, Cm, D, bias, dout, x, True, 1)
    torch.cuda.synchronize()
