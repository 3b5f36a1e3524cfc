"""HIP-graph capture of the training step (torch.cuda.CUDAGraph = hipGraph on ROCm).

The step issues ~10 000 small kernels (the model is 3 M parameters; most launches are a few
microseconds), so eager execution is bound by host launch overhead, not by the GPU.  Two graphs:

    graph A   zero grads -> G forward -> D loss + G losses -> backward(G loss) -> backward(D loss)
    (eager)   one RCCL all-reduce per flat gradient buffer          [world_size > 1 only]
    graph B   AdamW step for G and for D (capturable optimisers) + refresh of the bf16 shadow weights

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
        self.graph_fb = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph_fb):
            self.static_out, self.static_logs = tr._forward_backward(*self.static_in)
        self.graph_opt = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph_opt, pool=self.graph_fb.pool()):
            tr._optimizer_steps()

    def __call__(self, wave_input, wave_target, highcut):
        for dst, src in zip(self.static_in, (wave_input, wave_target, highcut)):
            if dst.data_ptr() != src.data_ptr():
                dst.copy_(src, non_blocking=True)
        self.graph_fb.replay()
        self.tr._reduce_grads("generator")
        if self.tr.gan:
            self.tr._reduce_grads("mpd")
        self.graph_opt.replay()
        return self.static_out, self.static_logs
