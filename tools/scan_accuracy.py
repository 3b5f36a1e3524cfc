"""Distance of the selective-scan kernels' fp32 results from the float64 oracle, next to the fp32 oracle's own distance (dev tool).
   python tools/scan_accuracy.py  B KD G N L      (VMASR_SSCAN_N_LEGACY=1: the one-state-at-a-time general-N kernels)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import torch

import oracle
from vm_asr_amd import selective_scan as ss

shape = tuple(int(v) for v in sys.argv[1:6]) if len(sys.argv) >= 6 else (1, 64, 4, 32, 2048)
Bn, KD, G, N, L = shape
g = torch.Generator().manual_seed(11)
A = -0.5 * torch.rand(KD, N, generator=g)
Bm, Cm = torch.randn(Bn, G, N, L, generator=g), torch.randn(Bn, G, N, L, generator=g)
D, bias = torch.randn(KD, generator=g), 0.5 * torch.rand(KD, generator=g)
u, delta = torch.randn(Bn, KD, L, generator=g), 0.5 * torch.rand(Bn, KD, L, generator=g)
dout = torch.randn(Bn, KD, L, generator=g)
cpu = (u, delta, A, Bm, Cm, D, bias, dout)
args = [t.numpy() for t in cpu]
with oracle.float64():
    w64 = (oracle.sscan_fwd(*args[:7], True),) + tuple(oracle.sscan_bwd(*args[:7], args[7], True))
w32 = (oracle.sscan_fwd(*args[:7], True),) + tuple(oracle.sscan_bwd(*args[:7], args[7], True))
dev = [t.cuda() for t in cpu]
names = ("out", "du", "ddelta", "dA", "dB", "dC", "dD", "dbias")
for tune in ((-1, -1), (1, 1)):
    ss.tune(*tune)
    out, x = ss.fwd(*dev[:7], True, 1)
    got = (out,) + tuple(ss.bwd(*dev[:7], dev[7], x, True, 1))
    print(f"shape {shape} tune {tune} legacy={os.environ.get('VMASR_SSCAN_N_LEGACY', '0')}")
    for name, gg, r32, r64 in zip(names, got, w32, w64):
        ref = np.asarray(r64, np.float64)
        e_hip = np.linalg.norm(gg.double().cpu().numpy().reshape(ref.shape) - ref) / np.linalg.norm(ref)
        e_cpu = np.linalg.norm(np.asarray(r32, np.float64) - ref) / np.linalg.norm(ref)
        print(f"  {name:7s} rel L2 from float64: hip {e_hip:.2e}  oracle fp32 {e_cpu:.2e}  ratio {e_hip / max(e_cpu, 1e-12):6.2f}")
