"""deterministic mode vs default mode on the generator alone (bf16 autocast): forward output and every parameter gradient, in module order"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")
os.environ.setdefault("TENSILE_STREAMK_DATA_PARALLEL", "1")
import torch
import bench
from vm_asr_amd import _lib
lib = _lib.lib()
cfg = bench.make_config("vm_asr_48k_MPD", 1)
dev = torch.device("cuda:0")
tr = bench.build_trainer(cfg, dev, amp=True, capturable=False)
gen = tr.models["generator"].train()
batch = bench.synth_batch(cfg, dev, 0)
gsig = torch.randn(1, 1, 122640, generator=torch.Generator().manual_seed(5)).to(dev) * 1e-3
res = {}
for det in (0, 1, 0, 1):
    lib.vmasr_set_deterministic(det)
    for p in gen.parameters():
        p.grad = None
    torch.manual_seed(77); torch.cuda.manual_seed_all(77)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        y = gen(batch[0], batch[2])
    (y.float() * gsig).sum().backward()
    torch.cuda.synchronize()
    key = f"det{det}" + ("b" if f"det{det}" in res else "")
    res[key] = (y.detach().clone(), {n: p.grad.clone() for n, p in gen.named_parameters() if p.grad is not None})
print("forward det0 vs det1 max diff", float((res["det0"][0] - res["det1"][0]).abs().max()), "max", float(res["det0"][0].abs().max()))
def d(a, b):
    out = []
    for n in a:
        r = float(a[n].abs().max())
        out.append((n, float((a[n] - b[n]).abs().max()) / max(r, 1e-30), r))
    return out
same = d(res["det0"][1], res["det0b"][1])
print("atomics run vs atomics run: worst", max(same, key=lambda t: t[1]))
print("det run vs det run: worst", max(d(res["det1"][1], res["det1b"][1]), key=lambda t: t[1]))
cross = d(res["det0"][1], res["det1"][1])
print("atomics vs det: worst", max(cross, key=lambda t: t[1]), " n>1e-3:", sum(t[1] > 1e-3 for t in cross), "of", len(cross))
for n, v, r in cross:
    if v > 1e-3:
        print(f"   {n:70s} {v:.3e}  |g|max {r:.2e}")
