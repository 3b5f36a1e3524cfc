"""Does hipGraph replay keep ATen reductions correct on NEW data?  (dev tool; vm_asr_amd/hip_env.py)
  python tools/graph_replay_probe.py                                   -> 0 wrong results (packet capture off: the package default)
  DEBUG_CLR_GRAPH_PACKET_CAPTURE=1 python tools/graph_replay_probe.py  -> wrong results from the second replay on (ROCm 7.2)"""
import os, sys

os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vm_asr_amd import stft as S
dev = torch.device("cuda", 0)
n = 4 * 513 * 1023
T = 122640
def run(name, fn, make):
    ins = make()
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3): fn(*ins)
    torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        o = fn(*ins)
    bad = 0
    for trial in range(4):
        for t, new in zip(ins, make()):
            t.copy_(new)
        g.replay(); torch.cuda.synchronize()
        ref = fn(*ins)
        bad += sum(1 for a, b in zip(o, ref) if abs(float(a) - float(b)) > 1e-4 * max(1.0, abs(float(b))))
    print(f"{name:46s} wrong outputs over 4 replays: {bad}")
mk2 = lambda: [torch.rand(n, device=dev) + 0.5, torch.rand(n, device=dev) + 0.5]
run("10 ATen reductions", lambda a, b: [a.min(), a.max(), b.min(), b.max(), a.sum(), b.sum(), (a - b).abs().sum(), a.log().min(), b.log().max(), (a * b).mean()], mk2)
mkw = lambda: [0.1 * torch.randn(4, T, device=dev)]
def f_stft(w):
    re, im = S.stft_reim(w, 1024, 120, 600)
    return [re.min(), re.max(), im.min(), im.max(), re.sum(), im.sum(), (re * re + im * im).sum()]
run("HIP stft + 7 ATen reductions", f_stft, mkw)
def f_stft2(w):
    re, im = S.stft_reim(w, 1024, 120, 600)
    m = torch.sqrt(torch.clamp(re ** 2 + im ** 2, min=1e-7)).transpose(2, 1)
    return [m.min(), m.max(), m.sum(), torch.log(m).min(), torch.log(m).max()]
run("HIP stft + magnitude (transposed view) + 5 red.", f_stft2, mkw)
def f_plain(w):
    m = torch.sqrt(torch.clamp(w ** 2, min=1e-7)).view(4, 420, 292).transpose(2, 1)
    return [m.min(), m.max(), m.sum(), torch.log(m).min(), torch.log(m).max()]
run("ATen only: transposed view + 5 reductions", f_plain, mkw)
