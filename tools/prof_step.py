"""torch.profiler view of one train step (dev tool): where does host time go?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from torch.profiler import profile, ProfilerActivity

wl = sys.argv[1] if len(sys.argv) > 1 else "vm_asr_48k_MPD"
cfg = bench.make_config(wl, int(sys.argv[2]) if len(sys.argv) > 2 else 0)
dev = torch.device("cuda", 0)
tr = bench.build_trainer(cfg, dev, amp=True)
for m in tr.models.values():
    m.train()
batch = bench.synth_batch(cfg, dev, 0)
for _ in range(3):
    tr.train_step(*batch)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(3):
    tr.train_step(*batch)
torch.cuda.synchronize()
print("ms/step", (time.perf_counter() - t0) / 3 * 1e3)
# coarse wall-clock split
def timed(fn):
    torch.cuda.synchronize(); t = time.perf_counter(); r = fn(); torch.cuda.synchronize(); return r, (time.perf_counter() - t) * 1e3
with torch.autocast("cuda", dtype=torch.bfloat16):
    out, t_g = timed(lambda: tr.models["generator"](batch[0], batch[2]))
    print("G forward ms", t_g)
    if tr.gan:
        (dl, fr), t_d = timed(lambda: tr._discriminator_losses(out, batch[1]))
        print("D losses fwd ms", t_d)
        gl, t_gl = timed(lambda: tr._generator_losses(out, batch[1], fr))
    else:
        gl, t_gl = timed(lambda: tr._generator_losses(out, batch[1]))
    print("G losses fwd ms", t_gl)
_, t_b = timed(lambda: sum(gl.values()).backward())
print("G backward ms", t_b)
if tr.gan:
    _, t_db = timed(lambda: sum(dl.values()).backward())
    print("D backward ms", t_db)
_, t_o = timed(lambda: tr.optimizer_G.step())
print("G opt ms", t_o)
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    tr.train_step(*batch)
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="self_cpu_time_total", row_limit=25, max_name_column_width=60))
