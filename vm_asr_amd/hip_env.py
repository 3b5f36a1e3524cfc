"""Process-level HIP runtime settings this package depends on.  Import (or copy the two lines) BEFORE anything initialises
the GPU — entry points do it before `import torch`.

DEBUG_CLR_GRAPH_PACKET_CAPTURE=0: ROCm 7.x replays an instantiated hipGraph from pre-recorded AQL packets.  On this stack
(ROCm 7.2 driver, PyTorch 2.10+rocm7.0) that path does not keep memset nodes ordered with the kernel nodes around them: every
multi-block ATen reduction zeroes its semaphores with hipMemsetAsync, so from the second replay on reductions that share a
(re-used) semaphore buffer return stale or partial results — silently, and invisibly when the graph is replayed on the SAME
data (tools/graph_debug6.py: 9 wrong results in 40 for a graph of ten reductions; 0 with packet capture off).  The training
step contains ~170 such reductions.  Node-by-node replay costs ~1 ms of the 37 ms step.  graph_step.replay_selftest() checks
the behaviour at run time and refuses graph mode when it is broken."""
import os

os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")
