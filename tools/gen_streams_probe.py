"""Where does the generator with its phase branch on a second stream (VMASR_GEN_STREAMS=2) stall?  (dev tool)
usage: python tools/gen_streams_probe.py BATCH [fwd|bwd] [amp]"""
import os
import sys
import time

os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench

dev = torch.device("cuda:0")
B = int(sys.argv[1])
what = sys.argv[2] if len(sys.argv) > 2 else "bwd"
amp = len(sys.argv) > 3 and sys.argv[3] == "amp"
cfg = bench.make_config("vm_asr_48k", B)
tr = bench.build_trainer(cfg, dev, amp=amp, capturable=False)
G = tr.models["generator"].train()
wave_in, wave_tgt, highcut = bench.synth_batch(cfg, dev, 0)
for it in range(3):
    t0 = time.time()
    with torch.autocast(device_type="cuda", dtype=torch.bfloat16, enabled=amp):
        out = G(wave_in, highcut)
    torch.cuda.synchronize()
    print(f"iter {it}: forward done {time.time() - t0:.3f}s", flush=True)
    if what == "bwd":
        t0 = time.time()
        out.float().square().mean().backward()
        print(f"iter {it}: backward issued {time.time() - t0:.3f}s", flush=True)
        torch.cuda.synchronize()
        print(f"iter {it}: backward done {time.time() - t0:.3f}s", flush=True)
print("ok")
