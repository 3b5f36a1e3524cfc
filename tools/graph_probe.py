"""Which part of the step survives HIP graph capture? (dev tool)"""
import os, sys, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

cfg = bench.make_config("vm_asr_48k_MPD", 0)
dev = torch.device("cuda", 0)
tr = bench.build_trainer(cfg, dev, amp=True, capturable=True)
for m in tr.models.values():
    m.train()
batch = bench.synth_batch(cfg, dev, 0)
for _ in range(2):
    tr._forward_backward(*batch); tr._reduce_and_step()
torch.cuda.synchronize()
G, D = tr.models["generator"], tr.models["mpd"]

def probe(name, fn):
    torch.cuda.synchronize()
    side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        fn()
    torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    try:
        with torch.cuda.graph(g):
            fn()
        g.replay(); torch.cuda.synchronize()
        print(f"[ok]   {name}", flush=True)
    except Exception as e:
        print(f"[FAIL] {name}: {str(e).splitlines()[0]}", flush=True)
        traceback.print_exc(limit=6)
        try:
            torch.cuda.synchronize()
        except Exception:
            pass

from vm_asr_amd import stft as S
wave = batch[0]
ac = lambda: torch.autocast("cuda", dtype=torch.bfloat16)
probe("hip stft", lambda: S.wav2spectro(wave, 1024, 240, 1024, "log2"))
mag, ph = S.wav2spectro(wave, 1024, 240, 1024, "log2")
probe("hip istft", lambda: S.spectro2wav(mag, ph, 1024, 240, 1024, "log2"))
def gen_fwd():
    with ac():
        return G(batch[0], batch[2])
probe("generator forward", gen_fwd)
out = gen_fwd().detach().float()
probe("mr-stft loss fwd", lambda: tr._get_stft_loss(out, batch[1]))
def mr_bwd():
    o = out.clone().requires_grad_()
    tr._get_stft_loss(o, batch[1]).backward()
probe("mr-stft loss fwd+bwd", mr_bwd)
def mpd_fwd():
    with ac():
        return D(batch[1], out)
probe("mpd forward", mpd_fwd)
def gen_fb():
    with ac():
        y = G(batch[0], batch[2])
    y.float().square().mean().backward()
probe("generator fwd+bwd", gen_fb)
probe("full _forward_backward", lambda: tr._forward_backward(*batch))
probe("optimizer steps", lambda: (tr.optimizer_G.step(), tr.optimizer_D.step()))
