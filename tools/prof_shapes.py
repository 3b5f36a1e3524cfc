"""Which ATen ops / input shapes own the GPU time of a train step? (dev tool)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from torch.profiler import profile, ProfilerActivity
cfg = bench.make_config("vm_asr_48k_MPD", 0)
dev = torch.device("cuda", 0)
tr = bench.build_trainer(cfg, dev, amp=True)
for m in tr.models.values():
    m.train()
batch = bench.synth_batch(cfg, dev, 0)
for _ in range(3):
    tr.train_step(*batch)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    tr.train_step(*batch)
    torch.cuda.synchronize()
rows = prof.key_averages(group_by_input_shape=True)
rows = sorted(rows, key=lambda r: -r.self_device_time_total)
tot = sum(r.self_device_time_total for r in rows)
print(f"total device time {tot/1e3:.1f} ms")
for r in rows[:int(os.environ.get("TOP", 40))]:
    print(f"{r.self_device_time_total/1e3:8.2f} ms {r.count:5d}x  {r.key[:40]:40s} {str(r.input_shapes)[:150]}")
