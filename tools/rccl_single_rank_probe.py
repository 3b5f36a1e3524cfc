"""The N > 1 training-step path on ONE GPU with the real RCCL backend (dev tool): a 1-rank "nccl" process group, the trainer
told that world = 2 so that it takes the flat-buffer all-reduce path (async RCCL all-reduce with ReduceOp.AVG between the
replayed graphs, RCCL's watchdog thread alive during graph capture).  With one rank the collective is the identity, so the
losses must match a plain single-process run; what this shows is that RCCL + HIP-graph capture + replay coexist on this stack.
    python tools/rccl_single_rank_probe.py"""
import os
import sys

os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=os.environ.get("MASTER_PORT", "29577"), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1)
cfg = bench.make_config("vm_asr_48k_MPD", 0)
dev = torch.device("cuda", 0)
tr = bench.build_trainer(cfg, dev, amp=True, capturable=True)
tr.world = 2                      # take the multi-rank path: flat buffers + all-reduce (identity on one rank; AVG over 1 rank)
for m in tr.models.values():
    m.train()
batches = [bench.synth_batch(cfg, dev, s) for s in range(4)]
_, logs = tr.train_step(*batches[0])
print("eager + RCCL all-reduce:", {k: round(float(v), 4) for k, v in logs.items()})
ok = tr.enable_graphs(batches[0], warmup=2)
print("graph capture with the RCCL process group alive:", ok)
for b in batches[1:]:
    _, logs = tr.train_step(*b)
    print("replay + async RCCL all-reduce:", {k: round(float(v), 4) for k, v in logs.items()})
torch.cuda.synchronize()
import time  # noqa: E402
t0 = time.time()
for i in range(20):
    tr.train_step(*batches[i % 4])
torch.cuda.synchronize()
print(f"{(time.time() - t0) / 20 * 1e3:.1f} ms/step with the two all-reduces (164 MB + 9 MB, one rank) between the graphs")
dist.destroy_process_group()
