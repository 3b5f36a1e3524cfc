"""Process-level HIP runtime settings this package depends on.  Import (or copy the two lines) BEFORE anything initialises
the GPU — entry points do it before `import torch`.

DEBUG_CLR_GRAPH_PACKET_CAPTURE=0: ROCm 7.x replays an instantiated hipGraph from pre-recorded AQL packets.  On this stack
(ROCm 7.2 driver, PyTorch 2.10+rocm7.0) that path does not keep memset nodes ordered with the kernel nodes around them: every
multi-block ATen reduction zeroes its semaphores with hipMemsetAsync, so from the second replay on reductions that share a
(re-used) semaphore buffer return stale or partial results — silently, and invisibly when the graph is replayed on the SAME
data (tools/graph_debug6.py: 9 wrong results in 40 for a graph of ten reductions; 0 with packet capture off).  The training
step contains ~170 such reductions.  Node-by-node replay costs ~1 ms of the 37 ms step.  graph_step.replay_selftest() checks
the behaviour at run time and refuses graph mode when it is broken.

TENSILE_STREAMK_DATA_PARALLEL=1: every hipBLASLt GEMM kernel of this stack for gfx950 is a stream-K kernel (..._SK3_SKXCCM8): when the
tiles of a problem do not divide over the CUs, workgroups of ONE launch wait for each other's partial tiles through flags in memory.
Two such GEMMs running at the same time on two HIP streams (the generator's Linear layers beside the discriminator's GEMMs: the
two-stream step of DESIGN.md 4g) can stop the device for good — measured at batch 35 (the step never finishes; rocgdb shows one or two
Cijk_* dispatches resident and every queue behind them waiting: profiles/r05_streamk_stall.md), not at the batch sizes benchmarked
before, because whether the stream-K part of a kernel is used depends on the shape.  With this setting the kernels split by output
tile only (no cross-workgroup wait), which is also what the small GEMMs of this model want: the headline step is 1 % FASTER with it.
trainer._two_streams() refuses the two-stream layout when the variable is not in force."""
import os
import sys


def _gpu_untouched():
    """No GEMM can have run yet: torch is not imported, or its GPU side is not initialised (hipBLASLt / Tensile read the variable when
    they are first used — a value set after that is not known to be in force)."""
    t = sys.modules.get("torch")
    try:
        return t is None or not t.cuda.is_initialized()
    except Exception:      # noqa: BLE001  (a half-imported torch: be conservative)
        return False


# data-parallel (non stream-K) library GEMMs are known to be in force: the variable was already "1" when this module was imported, or
# it is set HERE before the GPU was touched.  trainer._two_streams() / model._lanes() test THIS flag, not the raw variable: a
# program that ran a GEMM first and imported vm_asr_amd later keeps the one-stream layout (two concurrent stream-K GEMMs can stop
# the device for good: a hang is the failure mode, so the guard is conservative).
STREAMK_DP_IN_FORCE = os.environ.get("TENSILE_STREAMK_DATA_PARALLEL") == "1" or (
    "TENSILE_STREAMK_DATA_PARALLEL" not in os.environ and _gpu_untouched())

os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")
os.environ.setdefault("TENSILE_STREAMK_DATA_PARALLEL", "1")


def streamk_dp_in_force():
    return STREAMK_DP_IN_FORCE and os.environ.get("TENSILE_STREAMK_DATA_PARALLEL") == "1"
