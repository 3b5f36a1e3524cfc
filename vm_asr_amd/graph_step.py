"""HIP-graph capture of the training step (torch.cuda.CUDAGraph = hipGraph on ROCm).

The step issues ~4 000 small kernels (the model is 3 M parameters; most launches are a few
microseconds), so eager execution is bound by host launch overhead, not by the GPU.  Three graphs:

    graph A1  G forward -> D loss + G losses -> backward(D loss) -> pack the MPD gradient
    (eager)   async RCCL all-reduce of the MPD flat buffer (164 MB)  [world_size > 1 only] — runs on RCCL's
              stream WHILE graph A2 replays, i.e. hidden behind the generator's backward
    graph A2  backward(G loss) -> pack the generator gradient
    (eager)   async all-reduce of the generator flat buffer (9 MB), join both collectives
    graph B   AdamW step for G and for D (capturable optimisers) + refresh of the bf16 shadow weights

A1 and A2 share one memory pool and one autograd graph (built while A1 is captured, consumed while A2 is — the
fwd/bwd split torch.cuda.make_graphed_callables uses).  The learning rate is a device tensor (trainer.lr_to_device),
so a scheduler update between replays takes effect in graph B.

The library's own kernels are launched on the capturing stream through ctypes, so they are part of
graph A like any ATen kernel; the in-library event profiler must be off during capture and replay.
"""
import torch

__all__ = ["GraphedTrainStep"]


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
        self.graph_fb = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph_fb):
            st = tr._forward_losses(*self.static_in)
            tr._backward_d(st)
        self.graph_g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph_g, pool=self.graph_fb.pool()):
            tr._backward_g(st)
        self.static_out, self.static_logs = st["wave_out"].detach(), st["logs"]
        del st
        self.graph_opt = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph_opt, pool=self.graph_fb.pool()):
            tr._optimizer_steps()

    def __call__(self, wave_input, wave_target, highcut):
        for dst, src in zip(self.static_in, (wave_input, wave_target, highcut)):
            if dst.data_ptr() != src.data_ptr():
                dst.copy_(src, non_blocking=True)
        self.graph_fb.replay()
        if self.tr.gan:
            self.tr._reduce_grads("mpd", async_op=True)       # overlaps graph A2
        self.graph_g.replay()
        self.tr._reduce_grads("generator", async_op=True)
        self.tr._wait_reduces()
        self.graph_opt.replay()
        return self.static_out, self.static_logs
