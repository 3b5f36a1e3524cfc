"""Which hardware queue does the runtime give each branch of a captured graph?  (dev tool; run under rocprofv3 --kernel-trace, read with
tools/branch_probe_report.py).  Chains are told apart by kernel: M = cumsum over a large tensor (long kernels, 'the discriminator'),
A = add (main chain), B = mul (a lane forked from the main chain), C = sub (a second lane).
usage: python tools/branch_probe.py TOPOLOGY    (see TOPOLOGIES below)"""
import os
import sys

os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")
import torch

dev = torch.device("cuda:0")
big = torch.randn(64, 1 << 20, device=dev)
xa = torch.randn(1 << 16, device=dev)
xb = torch.randn(1 << 16, device=dev)
xc = torch.randn(1 << 16, device=dev)
main = torch.cuda.current_stream()
sm, sb, sc = torch.cuda.Stream(), torch.cuda.Stream(), torch.cuda.Stream()


def M(n=6):
    for _ in range(n):
        torch.cumsum(big, 1)


def A(n=40):
    global xa
    for _ in range(n):
        xa = xa + 1.0


def B(n=40):
    global xb
    for _ in range(n):
        xb = xb * 1.0001


def C(n=40):
    global xc
    for _ in range(n):
        xc = xc - 1.0


def fork(s, frm=None):
    s.wait_stream(frm or torch.cuda.current_stream())


def join(s, into=None):
    (into or torch.cuda.current_stream()).wait_stream(s)


def topo_step3():
    """the GAN step with a phase lane: M forked at the root and alive to the end; B forked from A, joined, forked again ('forward', 'backward')"""
    fork(sm)
    with torch.cuda.stream(sm):
        M(3)
    fork(sb)
    with torch.cuda.stream(sb):
        B()
    A()
    join(sb)
    A(10)                      # phase 2: main alone
    with torch.cuda.stream(sm):
        M(3)                   # 'D-loss backward' continues M's chain
    fork(sb)
    with torch.cuda.stream(sb):
        B()                    # 'backward lane'
    A()
    join(sb)
    join(sm)


def topo_step3_refork_m():
    """same, but the discriminator's second part is a NEW fork from main (its first part joined before)"""
    fork(sm)
    with torch.cuda.stream(sm):
        M(3)
    fork(sb)
    with torch.cuda.stream(sb):
        B()
    A()
    join(sb)
    join(sm)
    A(10)
    fork(sb)
    with torch.cuda.stream(sb):
        B()
    fork(sm)
    with torch.cuda.stream(sm):
        M(3)
    A()
    join(sb)
    join(sm)


def topo_step3_lane_open():
    """the lane is never joined in between: one chain from the first fork to the end, cross edges only"""
    fork(sm)
    with torch.cuda.stream(sm):
        M(3)
    fork(sb)
    with torch.cuda.stream(sb):
        B()
    A()
    A(10)
    with torch.cuda.stream(sm):
        M(3)
    sb.wait_stream(main)        # cross edge main -> lane
    with torch.cuda.stream(sb):
        B()
    A()
    join(sb)
    join(sm)


def topo_step3_dummy():
    """as topo_step3, but a dummy third branch C is forked right before the backward lane (is the lane then the third child?)"""
    fork(sm)
    with torch.cuda.stream(sm):
        M(3)
    fork(sb)
    with torch.cuda.stream(sb):
        B()
    A()
    join(sb)
    A(10)
    with torch.cuda.stream(sm):
        M(3)
    fork(sc)
    with torch.cuda.stream(sc):
        C(1)
    fork(sb)
    with torch.cuda.stream(sb):
        B()
    A()
    join(sb)
    join(sc)
    join(sm)


TOPOLOGIES = {k[5:]: v for k, v in globals().items() if k.startswith("topo_")}
name = sys.argv[1]
M(1); A(2); B(2); C(2)
for s in (sm, sb, sc):
    with torch.cuda.stream(s):
        M(1); A(1)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, capture_error_mode="thread_local"):
    TOPOLOGIES[name]()
for _ in range(3):
    g.replay()
torch.cuda.synchronize()
