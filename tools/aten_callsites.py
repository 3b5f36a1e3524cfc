"""Which Python call sites issue the small ATen kernels of one train step (fill / zero / copy / add / cat / sum ...)?
TorchDispatchMode over one eager step; counts per (op, innermost vm_asr_amd frame)."""
import collections, os, sys, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")
os.environ.setdefault("TENSILE_STREAMK_DATA_PARALLEL", "1")
import torch
from torch.utils._python_dispatch import TorchDispatchMode
import bench

COUNT = collections.Counter()
ELEMS = collections.Counter()


class Spy(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        out = func(*args, **(kwargs or {}))
        name = str(func).replace("aten.", "")
        st = traceback.extract_stack(limit=40)
        site = "?"
        for fr in reversed(st):
            if "/vm_asr_amd/" in fr.filename or fr.filename.endswith("bench.py"):
                site = f"{os.path.basename(fr.filename)}:{fr.lineno} {fr.name}"
                break
        COUNT[(name, site)] += 1
        t = out if torch.is_tensor(out) else (args[0] if args and torch.is_tensor(args[0]) else None)
        if t is not None:
            ELEMS[(name, site)] += t.numel()
        return out


cfg = bench.make_config("vm_asr_48k_MPD", 4)
dev = torch.device("cuda:0")
tr = bench.build_trainer(cfg, dev, amp=True, capturable=True)
for m in tr.models.values():
    m.train()
batch = bench.synth_batch(cfg, dev, 0)
for _ in range(3):
    tr.train_step(*batch)
torch.cuda.synchronize()
with Spy():
    tr.train_step(*batch)
torch.cuda.synchronize()
skip = ("view", "reshape", "detach", "alias", "t.default", "transpose", "permute", "expand", "slice", "select", "unsqueeze", "squeeze", "as_strided", "_unsafe_view", "split", "unbind", "unflatten", "empty", "is_", "stride", "size", "_local_scalar", "record_stream", "chunk", "set_")
rows = [(k, v) for k, v in COUNT.items() if not any(k[0].startswith(s) for s in skip)]
byop = collections.Counter()
for (op, site), v in rows:
    byop[op] += v
print("ops per step (backward-thread ops are NOT seen by a dispatch mode: forward + trainer glue only):")
for op, v in byop.most_common(25):
    print(f"  {op:40s} {v}")
print("top call sites:")
for (op, site), v in sorted(rows, key=lambda kv: -kv[1])[:60]:
    print(f"  {v:4d}  {op:34s} {site:60s} elems {ELEMS[(op, site)]}")
