import os, sys, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from vm_asr_amd import stft as S
dev = "cuda:0"
x = 0.1 * torch.randn(4, 122640, device=dev)
y = 0.1 * torch.randn(4, 122640, device=dev)

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
        try: torch.cuda.synchronize()
        except Exception: pass

for n, h, w in ((1024, 120, 600), (2048, 240, 1200), (512, 50, 240)):
    probe(f"stft_reim fwd {n}", lambda: S.stft_reim(x, n, h, w))
re, im = S.stft_reim(x, 1024, 120, 600)
mag = torch.sqrt(torch.clamp(re ** 2 + im ** 2, min=1e-7))
probe("sqrt clamp", lambda: torch.sqrt(torch.clamp(re ** 2 + im ** 2, min=1e-7)).transpose(2, 1))
probe("norm fro", lambda: torch.norm(mag - 0.5 * mag, p="fro") / torch.norm(mag, p="fro"))
probe("l1 log", lambda: F.l1_loss(torch.log(mag), torch.log(mag * 1.1)))
def fb():
    xx = x.clone().requires_grad_()
    r, i = S.stft_reim(xx, 2048, 240, 1200)
    (r.square() + i.square()).sum().backward()
probe("stft_reim fwd+bwd 2048", fb)
