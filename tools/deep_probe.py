"""dev probe: deep core vs oracle chain vs float64, per tensor; WX0=1 zeroes x_proj's weights (dx = the scan's own du)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
import oracle
from test_ss2d_deep import _params, _oracle_core, _run, NAMES, SHAPES
from vm_asr_amd.ss2d_deep import ss2d_deep
for shape in SHAPES:
    B, D, H, W, R = shape
    g = torch.Generator().manual_seed(D * 7 + H)
    x = torch.randn(B, D, H, W, generator=g)
    gy = torch.randn(B, D, H * W, generator=g)
    params = _params(D, R, D + 1)
    p64 = _params(D, R, D + 1, torch.float64)
    if os.environ.get("WX0") == "1":
        params[0].zero_(); p64[0].zero_()
    if os.environ.get("WX0") == "2":     # only the B / C rows of x_proj
        params[0][:, :R].zero_(); p64[0][:, :R].zero_()
    got = _run(ss2d_deep, x.cuda(), [p.cuda() for p in params], gy)
    ref = _run(_oracle_core, x, params, gy)
    with oracle.float64():
        r64 = _run(_oracle_core, x.double(), p64, gy.double())
    for n, a, b, c in zip(NAMES, got, ref, r64):
        a, b = a.double().cpu(), b.double()
        scale = max(c.abs().max().item(), 1e-12)
        e_hip, e_cpu = (a - c).abs().max().item() / scale, (b - c).abs().max().item() / scale
        r_hip, r_cpu = (a - c).pow(2).mean().sqrt().item() / scale, (b - c).pow(2).mean().sqrt().item() / scale
        print(f"{shape} {n:8s}: hip max {e_hip:.2e} rms {r_hip:.2e} | oracle max {e_cpu:.2e} rms {r_cpu:.2e} | ratio {e_hip / max(e_cpu, 1e-30):5.2f} {r_hip / max(r_cpu, 1e-30):5.2f}", flush=True)
