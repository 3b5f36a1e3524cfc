"""Do eager and graph-replayed training agree? (dev tool)  Compares weights after 6 steps: eager vs eager,
eager vs graphs, on the tiny test config."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
import test_trainer as tt

def run(graphs, amp=False, steps=6):
    cfg = tt._tiny_config()
    batch = [t.cuda() for t in tt._batch(cfg, 2)]
    tr = tt._gpu_trainer(cfg, amp=amp, capturable=True)
    for m in tr.models.values():
        m.train()
    hist = []
    if graphs:
        assert tr.enable_graphs(batch, warmup=3)
        n = steps - 3
    else:
        n = steps
    for _ in range(n):
        out, logs = tr.train_step(*batch)
        hist.append(float(logs["total_loss"]))
    torch.cuda.synchronize()
    return {k: v.detach().float().clone() for k, v in tr.models["generator"].state_dict().items()}, hist

def cmp(a, b, tag):
    d = torch.cat([(a[k] - b[k]).abs().flatten() for k in a])
    print(f"{tag}: max {d.max().item():.3e}  frac>5e-4 {(d > 5e-4).float().mean().item():.4f}  frac>1e-5 {(d > 1e-5).float().mean().item():.4f}")

e1, h1 = run(False); e2, h2 = run(False); g1, h3 = run(True)
print("eager losses", h1); print("eager2 losses", h2); print("graph losses", h3)
cmp(e1, e2, "eager vs eager"); cmp(e1, g1, "eager vs graph")
e1s, _ = run(False, steps=1); e2s, _ = run(False, steps=1)
cmp(e1s, e2s, "eager vs eager, 1 step")
