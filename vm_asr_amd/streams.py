"""Fork/join helper: run independent sub-graphs of the model on separate HIP streams.

VM-ASR's two spectrogram streams (magnitude / phase) are independent between interaction points,
and the five period discriminators are independent of each other; at batch 4 most of their
kernels are far too small to fill 256 CUs (5 us launches).  Forking them onto side streams lets
the GPU overlap them — also inside a captured HIP graph, where the fork/join events become
parallel branches of the graph, and in the backward pass, which autograd runs on the stream
each forward op used.  Pure scheduling: results are identical.
"""
import os

import torch

__all__ = ["parallel", "enabled"]

_POOL = {}


def enabled(tag=""):
    """VMASR_STREAMS: "0" nowhere (default), "1" everywhere, or a string of tags ("g" generator, "d" MPD)."""
    # Default "0" (measured r01, ROCm 7.2): forking the generator streams ("g") is correct in eager
    # mode but replaying the captured graph with parallel branches never returns, and forking the
    # five period discriminators ("d") wedges the GPU even eagerly (six streams of hipBLASLt GEMMs).
    # The graph-replayed single-stream step (64 ms) beats the eager forked one (114 ms), so opt-in only.
    v = os.environ.get("VMASR_STREAMS", "0")
    return v == "1" or (v != "0" and tag in v)


def _streams(device, n):
    pool = _POOL.setdefault((device.type, device.index), [])
    while len(pool) < n:
        pool.append(torch.cuda.Stream(device=device))
    return pool[:n]


def parallel(fns, device, tag=""):
    """Call the zero-argument functions `fns`; the first on the current stream, the others on side
    streams forked from it and joined back before returning.  Returns their results in order."""
    if len(fns) == 1 or device.type != "cuda" or not enabled(tag):
        return [f() for f in fns]
    cur = torch.cuda.current_stream(device)
    side = _streams(device, len(fns) - 1)
    outs = [None] * len(fns)
    for s in side:
        s.wait_stream(cur)
    for i, (f, s) in enumerate(zip(fns[1:], side), start=1):
        with torch.cuda.stream(s):
            outs[i] = f()
    outs[0] = fns[0]()
    for i, s in enumerate(side, start=1):
        cur.wait_stream(s)
        _record(outs[i], cur)
    return outs


def _record(obj, stream):
    """Tensors produced on a side stream are consumed on `stream`: tell the caching allocator."""
    if torch.is_tensor(obj):
        if obj.is_cuda:
            obj.record_stream(stream)
    elif isinstance(obj, (list, tuple)):
        for o in obj:
            _record(o, stream)
