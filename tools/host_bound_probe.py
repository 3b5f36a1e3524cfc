"""Is the replayed step bound by the host's enqueue rate?  Times the host side of N train_step() calls (graph replays, no sync) against the
GPU's time for the same N steps (dev tool).  usage: python tools/host_bound_probe.py [workload] [batch]"""
import os
import sys
import time

os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")
os.environ.setdefault("TENSILE_STREAMK_DATA_PARALLEL", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench

dev = torch.device("cuda:0")
wl = sys.argv[1] if len(sys.argv) > 1 else "vm_asr_48k_MPD"
cfg = bench.make_config(wl, int(sys.argv[2]) if len(sys.argv) > 2 else 0)
tr = bench.build_trainer(cfg, dev, amp=True, capturable=True)
for m in tr.models.values():
    m.train()
batch = bench.synth_batch(cfg, dev, 0)
assert tr.enable_graphs(batch, warmup=2), getattr(tr, "graph_error", None)
for _ in range(5):
    tr.train_step(*batch)
torch.cuda.synchronize()
N = 20
t0 = time.perf_counter()
for _ in range(N):
    tr.train_step(*batch)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"{wl}: host enqueue {1e3 * (t1 - t0) / N:.2f} ms/step, until the GPU is done {1e3 * (t2 - t0) / N:.2f} ms/step "
      f"(host idle at the end: {1e3 * (t2 - t1):.1f} ms of {1e3 * (t2 - t0):.1f})")

# one step at a time from an idle GPU: the host's own time to enqueue a step
hs, gs = [], []
for _ in range(10):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    tr.train_step(*batch)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    hs.append(1e3 * (t1 - t0)); gs.append(1e3 * (t2 - t0))
hs.sort(); gs.sort()
print(f"  from an idle GPU: host returns after {hs[len(hs) // 2]:.2f} ms (median), step done after {gs[len(gs) // 2]:.2f} ms")
