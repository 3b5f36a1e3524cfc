"""The N > 1 training-step path on ONE GPU with the real RCCL backend (dev tool): a 1-rank "nccl" process group, the trainer
told that world = 2 so that it takes the flat-buffer all-reduce path (async RCCL all-reduce with ReduceOp.AVG between the
replayed graphs — the default —, or, VMASR_GRAPH_COLLECTIVES=1, captured INTO the step's graph as a third branch through RCCL's C API, with rccl.CollectiveWatchdog
armed behind every replay; RCCL's own watchdog thread alive during graph capture).  With one rank the collective is the identity, so the
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
batches = [bench.synth_batch(cfg, dev, s) for s in range(4)]
import time  # noqa: E402


def run(tag, in_graph):
    """One trainer, world pretended 2 (flat buffers + all-reduce; AVG over the one real rank = identity): eager step, capture, three
    replays on new batches (losses printed: must agree between the layouts), then 20 timed steps."""
    os.environ["VMASR_GRAPH_COLLECTIVES"] = "1" if in_graph else "0"
    tr = bench.build_trainer(cfg, dev, amp=True, capturable=True)
    tr.world = 2
    for m in tr.models.values():
        m.train()
    _, logs = tr.train_step(*batches[0])
    print(f"[{tag}] eager + RCCL all-reduce (issued behind the D backward on the side stream):", {k: round(float(v), 4) for k, v in logs.items()})
    ok = tr.enable_graphs(batches[0], warmup=2)
    print(f"[{tag}] graph capture with the RCCL process group alive: {ok}; collectives captured into the graph: "
          f"{tr._graphed.collectives_in_graph}; MPD wire dtype {tr._comm_dtype('mpd')}, generator {tr._comm_dtype('generator')}")
    assert ok and tr._graphed.collectives_in_graph == in_graph
    seen = []
    for b in batches[1:]:
        _, logs = tr.train_step(*b)
        seen.append({k: round(float(v), 4) for k, v in logs.items()})
        print(f"[{tag}] replay:", seen[-1])
    torch.cuda.synchronize()
    t0 = time.time()
    for i in range(20):
        tr.train_step(*batches[i % 4])
    torch.cuda.synchronize()
    print(f"[{tag}] {(time.time() - t0) / 20 * 1e3:.2f} ms/step")
    del tr
    torch.cuda.empty_cache()
    return seen


a = run("collectives = branches of graph A", True)
b = run("collectives between the graphs", False)
worst = max(abs(x[k] - y[k]) / max(1.0, abs(y[k])) for x, y in zip(a, b) for k in x)
print(f"largest relative difference of a replayed loss between the two layouts: {worst:.2e}")
assert worst < 2e-2
dist.destroy_process_group()
