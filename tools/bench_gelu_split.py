"""Microbenchmark of the g * GELU'(pre) -> bf16 pair pass (csrc/split.hip: gelu_bwd_split) at the discriminator's layer shapes (dev tool)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from vm_asr_amd import _lib

dev = torch.device("cuda:0")
lib = _lib.lib()
for n, M, N in ((5, 36864, 512), (5, 12288, 1024), (5, 109056, 128)):
    pre = torch.randn(n, M, N, device=dev)
    g = torch.randn(n, M, N, device=dev)
    hi = torch.empty(n, M, N, dtype=torch.bfloat16, device=dev)
    lo = torch.empty_like(hi)
    db = torch.zeros(n, N, device=dev)

    def run(with_db=True):
        _lib.check(lib.vmasr_gelu_bwd_split(pre.data_ptr(), g.data_ptr(), hi.data_ptr(), lo.data_ptr(), None, db.data_ptr() if with_db else None,
                                            n, M, N, _lib.current_stream(dev)), "gelu_bwd_split")
    for wdb in (True, False):
        for _ in range(3):
            run(wdb)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(20):
            run(wdb)
        b.record()
        torch.cuda.synchronize()
        us = a.elapsed_time(b) / 20 * 1e3
        print(f"n {n} M {M} N {N} db {wdb}: {us:7.1f} us  {12.0 * n * M * N / us / 1e6:6.2f} TB/s")
