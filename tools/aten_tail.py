"""Which ATen ops make up the elementwise/copy tail of one eager train step (dev tool): torch.profiler, grouped by op and
input shape, sorted by device time.   usage: python tools/aten_tail.py [top]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

top = int(sys.argv[1]) if len(sys.argv) > 1 else 60
cfg = bench.make_config(os.environ.get("WORKLOAD", "vm_asr_48k_MPD"), int(os.environ.get("BATCH", "0")))
dev = torch.device("cuda", 0)
bench_mode = os.environ.get("TAIL_BENCH_MODE", "1") == "1"   # the step exactly as bench.py runs it (flat buffers, HIP AdamW, bf16 shadows), executed eagerly
tr = bench.build_trainer(cfg, dev, amp=True, capturable=bench_mode)
for m in tr.models.values():
    m.train()
batch = bench.synth_batch(cfg, dev, 0)
if bench_mode:
    assert tr.enable_graphs(batch, warmup=2), getattr(tr, "graph_error", None)
    tr._graphed = None
for _ in range(3):
    tr.train_step(*batch)
torch.cuda.synchronize()
from torch.profiler import ProfilerActivity, profile  # noqa: E402
fwd_only = os.environ.get("TAIL_FWD_ONLY") == "1"      # the generator's forward alone (autocast as in the step), autograd graph recorded
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    if fwd_only:
        with torch.autocast(device_type="cuda", dtype=torch.bfloat16):
            out = tr.models["generator"](batch[0], batch[2])
    else:
        tr.train_step(*batch)
    torch.cuda.synchronize()
rows = []
from torch.autograd import DeviceType  # noqa: E402
mode = os.environ.get("TAIL_ROWS", "ops")     # ops: host ops with the device time of the kernels they launched; kernels: by kernel
for e in prof.key_averages(group_by_input_shape=True):
    dt = getattr(e, "self_device_time_total", None) or getattr(e, "self_cuda_time_total", 0)
    is_kernel = e.device_type != DeviceType.CPU
    if dt > 0 and is_kernel == (mode == "kernels"):
        rows.append((dt, e.count, e.key, str(e.input_shapes)[:110]))
rows.sort(reverse=True)
tot = sum(r[0] for r in rows)
print(f"total self device time {tot / 1e3:.2f} ms over {sum(r[1] for r in rows)} op calls")
flt = os.environ.get("TAIL_FILTER")        # comma-separated op names: list only those, all shapes
for dt, n, k, sh in (rows[:top] if not flt else [r for r in rows if r[2] in flt.split(",")][:top]):
    print(f"{dt / 1e3:8.3f} ms {n:5d}  {k:40s} {sh}")

if os.environ.get("TAIL_BY_NAME"):
    from collections import defaultdict
    agg = defaultdict(lambda: [0.0, 0])
    for dt, n, k, _ in rows:
        agg[k[:90]][0] += dt
        agg[k[:90]][1] += n
    print("---- by name (all shapes) ----")
    for k, (dt, n) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:int(os.environ["TAIL_BY_NAME"])]:
        print(f"{n:6d} calls {dt / 1e3:8.3f} ms  {k}")
