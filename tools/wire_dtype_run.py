"""bf16 vs fp32 gradient wire: 200 optimisation steps of vm_asr_48k_MPD at batch 4 on synthetic clips, from one
initial state, with the MPD gradient (a) untouched (fp32 wire) and (b) rounded to bf16 every step (VMASR_GRAD_COMM_EMULATE=mpd-bf16: what a
bf16 all-reduce does to the values, on one rank), and (c) fp32 again with another stochastic-depth seed — the run-to-run noise the
difference has to be read against.  Prints the loss curves' summary."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")
os.environ.setdefault("TENSILE_STREAMK_DATA_PARALLEL", "1")
os.environ.setdefault("VMASR_STEP_VARIANT", "lane:0.75")
import numpy as np
import torch
import bench

STEPS = int(sys.argv[1]) if len(sys.argv) > 1 else 200
cfg = bench.make_config("vm_asr_48k_MPD", 4)
dev = torch.device("cuda:0")
batches = [bench.synth_batch(cfg, dev, r) for r in range(8)]


def run(emulate, seed):
    if emulate:
        os.environ["VMASR_GRAD_COMM_EMULATE"] = emulate
    else:
        os.environ.pop("VMASR_GRAD_COMM_EMULATE", None)
    tr = bench.build_trainer(cfg, dev, amp=True, capturable=True)
    for m in tr.models.values():
        m.train()
    tr.train_step(*batches[0])
    assert tr.enable_graphs(batches[0], warmup=2)          # (state restored afterwards: every run starts from the same weights)
    torch.manual_seed(seed)
    torch.cuda.manual_seed_all(seed)
    g, d = [], []
    for i in range(STEPS):
        _, logs = tr.train_step(*batches[i % len(batches)])
        g.append(logs["total_loss"].clone())
        d.append(logs["total_disc_loss"].clone())
    torch.cuda.synchronize()
    w = torch.cat([p.detach().flatten().double() for p in tr.models["mpd"].parameters()])
    wg = torch.cat([p.detach().flatten().double() for p in tr.models["generator"].parameters()])
    del tr
    torch.cuda.empty_cache()
    return np.array([float(x) for x in g]), np.array([float(x) for x in d]), w, wg


a = run(None, 1)
b = run("mpd-bf16", 1)
c = run(None, 2)


def summary(tag, x, y):
    k = STEPS // 4
    rel = lambda u, v: float(np.abs(u - v).mean() / np.abs(v).mean())      # noqa: E731
    print(f"{tag:34s} G loss: mean |diff| / mean {rel(x[0], y[0]):.3e} (last quarter {rel(x[0][-k:], y[0][-k:]):.3e}; means {x[0][-k:].mean():.4f} vs {y[0][-k:].mean():.4f})   "
          f"D loss: {rel(x[1], y[1]):.3e} (last quarter {rel(x[1][-k:], y[1][-k:]):.3e}; means {x[1][-k:].mean():.4f} vs {y[1][-k:].mean():.4f})   "
          f"|dW_mpd| / |W| {float((x[2] - y[2]).norm() / y[2].norm()):.3e}   |dW_gen| / |W| {float((x[3] - y[3]).norm() / y[3].norm()):.3e}", flush=True)


print(f"{STEPS} steps, batch 4, 8 synthetic batches cycled, AdamW as configured")
summary("bf16 wire vs fp32 wire (same seed)", b, a)
summary("fp32 vs fp32 (other DropPath seed)", c, a)
print("first / last G losses fp32:", a[0][:3].round(4), a[0][-3:].round(4), " bf16:", b[0][:3].round(4), b[0][-3:].round(4))
