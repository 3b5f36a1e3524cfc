"""The two-stream step's real timeline: device-clock marks captured into the step's graph at the phase boundaries (VMASR_PHASE_EVENTS=1,
vmasr_mark_time), read after a replay outside any profiler (dev tool).  usage: python tools/phase_probe.py [batch]"""
import os
import sys

os.environ["VMASR_PHASE_EVENTS"] = "1"
os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench

dev = torch.device("cuda:0")
cfg = bench.make_config("vm_asr_48k_MPD", int(sys.argv[1]) if len(sys.argv) > 1 else 0)
tr = bench.build_trainer(cfg, dev, amp=True, capturable=True)
for m in tr.models.values():
    m.train()
batch = bench.synth_batch(cfg, dev, 0)
assert tr.enable_graphs(batch, warmup=2), getattr(tr, "graph_error", None)
for _ in range(5):
    tr.train_step(*batch)
torch.cuda.synchronize()
names = ["start", "d_real_fwd_end", "g_fwd_end", "d_fake_fwd_losses_end", "g_dgrad_end", "g_bwd_end", "d_bwd_end", "join"]
acc = {n: 0.0 for n in names}
R = 10
for _ in range(R):
    tr.train_step(*batch)
    torch.cuda.synchronize()
    t = tr._phase_buf.cpu()
    for n in names:
        acc[n] += (int(t[tr.phase_marks[n]]) - int(t[tr.phase_marks["start"]])) / 1e5      # 100 MHz ticks -> ms
for n in names:
    print(f"{n:24s} {acc[n] / R:7.2f} ms after the step's start")
