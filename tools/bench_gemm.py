"""Microbenchmark of the discriminator's weight-gradient GEMMs (bf16 operands, fp32 output) in the operand layouts
the host side can produce: A = g^T as a transposed VIEW of the (M, N) gradient split (what `_StackedConvSplitFn`
does today) versus a contiguous (N, M) operand (what a transposing split kernel would write), with split-K factors.
Run on the GPU box: python tools/bench_gemm.py"""
import torch


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


def main():
    dev = "cuda"
    f32 = torch.float32
    for name, (n, N, M, K) in {"dW4": (5, 1024, 6144, 5120), "dW3": (5, 1024, 6144, 2560), "dW2": (5, 512, 18432, 640)}.items():
        g = torch.randn(n, M, N, device=dev).bfloat16()
        c = torch.randn(n, M, K, device=dev).bfloat16()
        gT = g.transpose(1, 2).contiguous()
        flops = 2.0 * n * N * M * K
        for S in (1, 2, 3, 4, 6, 8):
            if M % (S * 256):
                continue
            gv = g.view(n * S, M // S, N).transpose(1, 2)
            cv = c.view(n * S, M // S, K)
            # contiguous (N, M/S) slabs: what a transposing split kernel would write for split factor S
            gTc = g.view(n * S, M // S, N).transpose(1, 2).contiguous()
            t_view = timeit(lambda: torch.bmm(gv, cv, out_dtype=f32))
            t_cont = timeit(lambda: torch.bmm(gTc, cv, out_dtype=f32))
            # output-transposed form: dW^T (K, N) = c^T g
            t_outT = timeit(lambda: torch.bmm(cv.transpose(1, 2), g.view(n * S, M // S, N), out_dtype=f32))
            print(f"{name} S={S}: A=g^T view {t_view*1e3:7.1f} us ({flops/t_view/1e9:6.0f} TF/s)   A contiguous {t_cont*1e3:7.1f} us "
                  f"({flops/t_cont/1e9:6.0f} TF/s)   dW^T form {t_outT*1e3:7.1f} us ({flops/t_outT/1e9:6.0f} TF/s)", flush=True)
    # forward / dcols reference points
    for name, (n, M, K, N) in {"fwd4": (5, 6144, 5120, 1024), "dcols4": (5, 6144, 3072, 5120), "fwd2": (5, 18432, 640, 512),
                               "dcols2": (5, 18432, 1536, 640)}.items():
        a = torch.randn(n, M, K, device=dev).bfloat16()
        b = torch.randn(n, K, N, device=dev).bfloat16()
        bT = torch.randn(n, N, K, device=dev).bfloat16()
        flops = 2.0 * n * M * K * N
        t1 = timeit(lambda: torch.bmm(a, b, out_dtype=f32))
        t2 = timeit(lambda: torch.bmm(a, bT.transpose(1, 2), out_dtype=f32))
        print(f"{name}: B (K,N) contiguous {t1*1e3:7.1f} us ({flops/t1/1e9:6.0f} TF/s)   B = (N,K)^T view {t2*1e3:7.1f} us ({flops/t2/1e9:6.0f} TF/s)", flush=True)


if __name__ == "__main__":
    main()
