"""Fault isolation before enabling the K-concatenated forward triple of the MPD layers (VMASR_MPD_KCAT): the batched bf16 GEMMs it
introduces, at the three layer shapes, each under its own sub-test (run with a short timeout: hipBLASLt has faulted the GPU on
some batched layouts).   A = [hi|lo|hi] (n, M, 3K) contiguous; B = (n, 3K, N) contiguous; and the weight-gradient
products whose B operand becomes a COLUMN BLOCK of A (ld = 3K)."""
import os, sys, time
os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")
import torch
n = 5
for (M, K, N, S) in [(36608, 640, 512, 8), (12288, 2560, 1024, 1), (4864, 5120, 1024, 1)]:
    a = torch.randn(n, M, 3 * K, device="cuda").to(torch.bfloat16)
    b = torch.randn(n, 3 * K, N, device="cuda").to(torch.bfloat16)
    t = time.time(); y = torch.bmm(a, b, out_dtype=torch.float32); torch.cuda.synchronize()
    ref = torch.bmm(a[:1, :256].float(), b[:1].float())
    print(f"fwd  M={M} K=3x{K} N={N}: ok {time.time() - t:.3f}s err {(y[0, :256] - ref).abs().max().item():.3e}", flush=True)
    g = torch.randn(n, M, 3 * N, device="cuda").to(torch.bfloat16)
    gh = g[:, :, :N]
    ch = a[:, :, :K]
    v = (lambda t_: t_.view(n * S, M // S, t_.shape[2])) if S > 1 else (lambda t_: t_)
    t = time.time(); dw = torch.bmm(v(gh).transpose(1, 2), v(ch), out_dtype=torch.float32); torch.cuda.synchronize()
    ref = torch.bmm(v(gh)[:1].transpose(1, 2).float(), v(ch)[:1].float())
    print(f"dW   M={M} K={K} N={N} S={S}: ok {time.time() - t:.3f}s err {(dw[0] - ref[0]).abs().max().item():.3e}", flush=True)
print("all ok")
