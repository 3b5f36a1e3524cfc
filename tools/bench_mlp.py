"""Fused Mlp branch (csrc/mlp.hip) vs the unfused module path (LayerNorm -> Linear -> GELU -> Linear -> add, hipBLASLt + ATen +
ln.hip) at the generator's shapes (B = $B, default 4), bf16 autocast: device time per call, forward and forward+backward,
REP calls captured in a HIP graph and replayed (dev tool)."""
import os
import sys

os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vm_asr_amd.layernorm import LayerNorm  # noqa: E402
from vm_asr_amd.mlp import fused_mlp_residual  # noqa: E402
from vm_asr_amd.vmamba import Mlp  # noqa: E402

B, REP = int(os.environ.get("B", "4")), int(os.environ.get("REP", "50"))
SHAPES = [(8, 256, torch.float32), (16, 128, torch.float32), (16, 128, torch.bfloat16), (32, 64, torch.bfloat16), (64, 32, torch.bfloat16),
          (128, 16, torch.bfloat16)]


def timed(fn):
    """REP calls captured into ONE HIP graph (how the training step runs them), replayed 5 times: device us per call."""
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3):
            fn()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(REP):
            fn()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (5 * REP) * 1e3


ONLY = os.environ.get("ONLY")            # e.g. ONLY=16 MODE=fused : one shape, one path (for rocprofv3 --kernel-trace --stats)
MODE = os.environ.get("MODE")
for d, H, xdt in SHAPES:
    if ONLY and (int(ONLY) != d or xdt != (torch.float32 if d == 8 else torch.bfloat16)):
        continue
    torch.manual_seed(0)
    norm, mlp = LayerNorm(d).cuda(), Mlp(d, 4 * d).cuda()
    norm.feeds_gemm = True
    x = torch.randn(B, H, H, d, device="cuda").to(xdt)
    gy = torch.randn(B, H, H, d, device="cuda").to(xdt)
    res = {}
    from vm_asr_amd import _lib
    offered = bool(_lib.lib().vmasr_mlp_supported(d, 4 * d))
    for name, f in (("fused", lambda xi: fused_mlp_residual(xi, norm, mlp)), ("unfused", lambda xi: xi + mlp(norm(xi)))):
        if (MODE and MODE != name) or (name == "fused" and not offered):      # d = 128: built, not offered
            res[name] = (0.0, 0.0)
            continue
        def fwd():
            with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
                f(x)

        def fwdbwd():
            xi = x.detach().requires_grad_()
            with torch.autocast("cuda", dtype=torch.bfloat16):
                y = f(xi)
            y.backward(gy)
        res[name] = (timed(fwd), timed(fwdbwd))
    rows = B * H * H
    print(f"d={d:4d} rows={rows:7d} x {str(xdt)[6:]:8s}: fused fwd {res['fused'][0]:7.1f} us  fwd+bwd {res['fused'][1]:7.1f} us | unfused fwd "
          f"{res['unfused'][0]:7.1f} us  fwd+bwd {res['unfused'][1]:7.1f} us   (x in + y out: {rows * d * 2 * x.element_size() / 1e6:.1f} MB)")
