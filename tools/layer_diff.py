"""Where does the HIP fp32 forward drift from the CPU-oracle forward? (dev tool)  Full-size 16 kHz clip,
synthetic weights; prints max|diff|/max|ref| after every top-level stage and every VSSBlock."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import numpy as np, torch
import test_fullsize as t
from oracle.torch_backend import use_oracle, oracle_stft_patch
from vm_asr_amd.vmamba import VSSBlock, SS2D

z = np.load(os.path.join(ROOT, "tests/golden/fullsize.npz"))
wave, hf = torch.from_numpy(z["16k_wave"]), torch.from_numpy(z["16k_hf"])
def trace(m, dev):
    acts = {}
    def hook(name):
        def f(mod, inp, out):
            o = out[0] if isinstance(out, (tuple, list)) else out
            if torch.is_tensor(o):
                acts.setdefault(name, []).append(o.detach().float().cpu())
        return f
    for n, mod in m.named_modules():
        if isinstance(mod, (VSSBlock, SS2D)) or n.count(".") == 0 and n:
            mod.register_forward_hook(hook(n))
    with torch.no_grad():
        y = m(wave.to(dev), hf.to(dev))
    return acts, y.float().cpu()
m = t._model(80); use_oracle(m)
with oracle_stft_patch():
    a_cpu, y_cpu = trace(m, "cpu")
m2 = t._model(80).cuda()
a_gpu, y_gpu = trace(m2, "cuda")
for k in a_cpu:
    for i, (c, g) in enumerate(zip(a_cpu[k], a_gpu.get(k, []))):
        print(f"{k:55s} #{i} {tuple(c.shape)!s:24s} rel {((c - g).abs().max() / c.abs().max()).item():.2e}")
print("wave", ((y_cpu - y_gpu).abs().max() / y_cpu.abs().max()).item())
