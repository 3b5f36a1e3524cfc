"""Loss values and gradients of the replayed graphs vs the eager step on the same weights and batch (dev tool)."""
import os
import sys

os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")   # vm_asr_amd/hip_env.py

import torch  # noqa: E402

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

cfg = bench.make_config("vm_asr_48k_MPD", 0)
cfg.defrost() if hasattr(cfg, "defrost") else None
dev = torch.device("cuda", 0)
tr = bench.build_trainer(cfg, dev, amp=True, capturable=True)
for m in tr.models.values():
    m.train()
batch = bench.synth_batch(cfg, dev, 0)
for _ in range(2):
    out, logs = tr.train_step(*batch)
print("eager :", {k: round(float(v), 4) for k, v in logs.items()})
ok = tr.enable_graphs(batch, warmup=2)
print("graphs:", ok)
for i in range(3):
    out, logs = tr.train_step(*batch)
    print(f"graph {i}:", {k: round(float(v), 4) for k, v in logs.items()})
for i in range(3):
    b2 = bench.synth_batch(cfg, dev, 100 + i)          # a different batch every step (what a data loader delivers)
    out, logs = tr.train_step(*b2)
    print(f"graph new batch {i}:", {k: round(float(v), 4) for k, v in logs.items()})
    if os.environ.get("WITH_METRICS"):
        print("   metrics:", {k: round(float(v), 4) for k, v in tr._metrics(out, b2[1], b2[2]).items()})
tr._graphed = None
out, logs = tr.train_step(*batch)
print("eager :", {k: round(float(v), 4) for k, v in logs.items()})
