"""HIP-graph capture of the training step (torch.cuda.CUDAGraph = hipGraph on ROCm).

The step issues ~1 600 kernels, most of them a few microseconds long (the generator is 3 M parameters), so eager execution is
bound by host launch overhead, not by the GPU.

Two-stream layout (default on the GPU: trainer._two_streams) — TWO graphs:

    graph A   one graph with a fork / join.  main stream: G forward -> waveform losses -> ... -> G backward -> pack G gradient;
              side stream: D(real) beside the G forward, then D(fake), the losses on its outputs, their backward down to d/d(wave)
              (input gradients only), then the D loss' backward + pack of the MPD gradient BESIDE the G backward.  Branches of one
              captured graph run concurrently on replay (a graph launch as a whole serialises with every other stream: measured,
              tools/overlap_probe*.py), and nodes are enqueued in capture order — which is why _backward_two() issues the D loss'
              backward before the generator's.
              world_size > 1: both all-reduces run between graph A and graph B on torch.distributed's communicator (default).
              VMASR_GRAPH_COLLECTIVES=1 (RCCL, opt-in): the MPD gradient's all-reduce is a THIRD branch of graph A, issued on the side
              stream right behind the D loss' backward; the generator's (9 MB) follows its pack; both join at the end of the graph.
    graph B   AdamW step for G and for D (capturable optimisers) + refresh of the bf16 shadow weights

One-stream layout (VMASR_TWO_STREAM=0, deterministic mode, no shared fake pass) — three graphs:

    graph A1  G forward -> D loss + G losses -> backward(D loss) -> pack the MPD gradient
    (eager)   async RCCL all-reduce of the MPD flat buffer (164 MB)  [world_size > 1 only] — runs on RCCL's
              stream WHILE graph A2 replays, i.e. hidden behind the generator's backward
    graph A2  backward(G loss) -> pack the generator gradient
    (eager)   async all-reduce of the generator flat buffer (9 MB), join both collectives
    graph B   as above

The graphs of a layout share one memory pool and one autograd graph (built while the first is captured, consumed while the
next is — the fwd/bwd split torch.cuda.make_graphed_callables uses).  The learning rate is a device tensor (trainer.lr_to_device),
so a scheduler update between replays takes effect in graph B.

The library's own kernels are launched on the capturing stream through ctypes, so they are part of
graph A like any ATen kernel; the in-library event profiler must be off during capture and replay.
"""
import torch

__all__ = ["GraphedTrainStep", "replay_selftest"]

_SELFTEST = {}


def replay_selftest(device):
    """Does this runtime replay a captured graph faithfully?  Captures a graph of ten multi-block ATen reductions (each zeroes
    its semaphores with a memset node), replays it four times on NEW data and compares every result with the eager value.
    False on the ROCm 7.2 AQL-packet replay path (vm_asr_amd/hip_env.py), where the memset nodes lose their order; cached per
    device."""
    key = str(device)
    if key in _SELFTEST:
        return _SELFTEST[key]
    n = 1 << 21

    def fn(a, b):
        return [a.min(), a.max(), b.min(), b.max(), a.sum(), b.sum(), (a - b).abs().sum(), a.log().min(), b.log().max(), (a * b).mean()]

    gen = torch.Generator(device=device).manual_seed(1234)
    mk = lambda: [torch.rand(n, device=device, generator=gen) + 0.5, torch.rand(n, device=device, generator=gen) + 0.5]   # noqa: E731
    ins = mk()
    cur = torch.cuda.current_stream(device)
    side = torch.cuda.Stream(device)
    side.wait_stream(cur)
    with torch.cuda.stream(side):
        for _ in range(2):
            fn(*ins)
    cur.wait_stream(side)
    torch.cuda.synchronize(device)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, capture_error_mode="thread_local"):
        outs = fn(*ins)
    ok = True
    for _ in range(4):
        for t, new in zip(ins, mk()):
            t.copy_(new)
        g.replay()
        torch.cuda.synchronize(device)
        ref = fn(*ins)
        ok = ok and all(abs(float(a) - float(b)) <= 1e-4 * max(1.0, abs(float(b))) for a, b in zip(outs, ref))
    del g, outs
    _SELFTEST[key] = ok
    return ok


class GraphedTrainStep:
    def __init__(self, trainer, example_batch, warmup=3):
        tr = self.tr = trainer
        if tr.device.type != "cuda":
            raise RuntimeError("graphs need a GPU")
        if tr._acc != 1:
            raise RuntimeError("graph capture needs TRAIN.ACCUMULATION_STEPS == 1")
        if tr.dp_mode != "flat":
            raise RuntimeError("graph capture needs dp_mode='flat' (DDP hooks are not capturable here)")
        for opt in [tr.optimizer_G] + ([tr.optimizer_D] if tr.gan else []):
            if not opt.defaults.get("capturable", False):
                raise RuntimeError("graph capture needs capturable=True optimisers (build_optimizer(..., capturable=True))")
        if not replay_selftest(tr.device):
            raise RuntimeError("this HIP runtime does not replay captured graphs faithfully (memset nodes lose their order: "
                               "ATen reductions return stale results from the second replay on); set "
                               "DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 before the process initialises the GPU (vm_asr_amd/hip_env.py)")
        from . import _lib
        _lib.prof_enable(False)
        self.static_in = [t.clone() for t in example_batch]
        cur = torch.cuda.current_stream()
        side = torch.cuda.Stream()
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            for i in range(max(2, warmup)):
                tr._forward_backward(*self.static_in)
                if i == 0:  # flat buffers exist before anything is captured
                    tr._setup_flat("generator", tr.optimizer_G)
                    if tr.gan:
                        tr._setup_flat("mpd", tr.optimizer_D)
                tr._reduce_and_step()
        cur.wait_stream(side)
        torch.cuda.synchronize()
        for opt in [tr.optimizer_G] + ([tr.optimizer_D] if tr.gan else []):
            from .trainer import lr_to_device
            if any(torch.is_tensor(g["lr"]) and not g["lr"].is_cuda for g in opt.param_groups):
                lr_to_device(opt, tr.device)      # a host lr would be frozen into graph B at capture
        # world_size > 1 on RCCL with VMASR_GRAPH_COLLECTIVES=1: the gradient all-reduces are captured INTO graph A (branches of the same
        # graph run concurrently on replay; a graph launch as a whole serialises with every other stream, so a collective issued between
        # two replays is fully exposed: tools/overlap_probe*.py).  Default (or another backend): collectives between the graphs.
        import os
        import torch.distributed as dist
        self.collectives_in_graph = (tr.world > 1 and tr.dp_mode == "flat" and dist.is_initialized() and dist.get_backend() == "nccl"
                                     and tr.graph_collectives()
                                     # (two-stream layout only: a collective forked in one capture cannot be joined in another,
                                     #  so the one-stream layout's A1 | all-reduce | A2 overlap stays between its graphs)
                                     and tr._two_streams())
        if self.collectives_in_graph:
            # the captured collectives go through RCCL's C API on a communicator of this trainer (vm_asr_amd/rccl.py); one eager
            # round first: RCCL sets up its channels / buffers on the first collective of a communicator, which must not happen
            # inside a capture
            try:
                comm = tr.enable_direct_rccl()      # (raises on EVERY rank if it failed on any: rccl.RcclComm agrees by all-reduce(MIN))
            except RuntimeError as e:
                tr.logger.warning(f"{e}: the gradient all-reduces stay between the graphs")
                tr._graph_collectives = False
                self.collectives_in_graph = False
        self.watchdog = None
        if self.collectives_in_graph:
            for key in (["mpd"] if tr.gan else []) + ["generator"]:
                comm.all_reduce_(torch.zeros_like(tr._flat[key], dtype=tr._comm_dtype(key)), avg=True, stream=tr._comm_stream())
            torch.cuda.synchronize()
            from .rccl import CollectiveWatchdog
            self.watchdog = CollectiveWatchdog()     # nothing else watches a captured collective: the rank exits non-zero if one hangs
        # No cyclic garbage collection while a capture is open: an old CUDAGraph (a trainer <-> GraphedTrainStep cycle left by an earlier
        # trainer of this process) whose destructor runs inside a capture frees device memory there, which the runtime refuses — the
        # process aborts.  torch.cuda.graph() only collects first when torch.compiler.config.force_cudagraph_gc is set (2.9+).
        import gc
        gc.collect()
        gc_was_on = gc.isenabled()
        gc.disable()
        try:
            self._capture(tr)
        finally:
            if gc_was_on:
                gc.enable()

    def _capture(self, tr):
        self.graph_fb = torch.cuda.CUDAGraph()
        # thread_local: only THIS thread's calls are checked during capture — RCCL's watchdog thread polls its events
        # (hipEventQuery) at any time, which in the default global mode would invalidate a capture in progress
        with torch.cuda.graph(self.graph_fb, capture_error_mode="thread_local"):
            st = tr._forward_losses(*self.static_in)
            if st.get("two"):      # two-stream step: ONE graph with a fork / join (the branches run concurrently on replay)
                tr._backward_both(st, reduce=self.collectives_in_graph)
                if self.collectives_in_graph:
                    tr._reduce_grads("generator", async_op=True)
                    tr._wait_reduces()                       # the join of the collective branches: last node(s) of graph A
            else:
                tr._backward_d(st)
        self.graph_g = None
        if not st.get("two"):
            self.graph_g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph_g, pool=self.graph_fb.pool(), capture_error_mode="thread_local"):
                tr._backward_g(st)
        self.static_out, self.static_logs = st["wave_out"].detach(), st["logs"]
        del st
        self.graph_opt = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph_opt, pool=self.graph_fb.pool(), capture_error_mode="thread_local"):
            tr._optimizer_steps()

    def __call__(self, wave_input, wave_target, highcut):
        for dst, src in zip(self.static_in, (wave_input, wave_target, highcut)):
            if dst.data_ptr() != src.data_ptr():
                dst.copy_(src, non_blocking=True)
        self.graph_fb.replay()
        if self.watchdog is not None:
            done = torch.cuda.Event()
            done.record()                       # behind graph A (outside any capture: queryable)
            self.watchdog.arm(done, "the gradient all-reduce captured into the step's graph")
        if not self.collectives_in_graph:
            if self.tr.gan:
                self.tr._reduce_grads("mpd", async_op=True)   # (between the graphs: exposed — a graph launch does not overlap another stream)
            if self.graph_g is not None:
                self.graph_g.replay()
            self.tr._reduce_grads("generator", async_op=True)
            self.tr._wait_reduces()
        self.graph_opt.replay()
        return self.static_out, self.static_logs
