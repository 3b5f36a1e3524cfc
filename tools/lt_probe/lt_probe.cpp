// lt_probe — how much faster than hipBLASLt's first heuristic choice are its other candidate algorithms at the
// discriminator's GEMM shapes?  (dev tool; bf16 x bf16 -> fp32, strided batched, column-major arguments as torch passes them)
//   usage: lt_probe m n k opA opB lda ldb ldc batch strideA strideB strideC [nalgos]
#include <hip/hip_runtime.h>
#include <hipblaslt/hipblaslt.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { auto e_ = (x); if (e_ != 0) { fprintf(stderr, "%s failed: %d (line %d)\n", #x, (int)e_, __LINE__); exit(2); } } while (0)

int main(int argc, char **argv) {
    if (argc < 13) { fprintf(stderr, "usage\n"); return 1; }
    const long m = atol(argv[1]), n = atol(argv[2]), k = atol(argv[3]);
    const bool tA = argv[4][0] == 'T', tB = argv[5][0] == 'T';
    const long lda = atol(argv[6]), ldb = atol(argv[7]), ldc = atol(argv[8]);
    const int batch = atoi(argv[9]);
    const long sA = atol(argv[10]), sB = atol(argv[11]), sC = atol(argv[12]);
    const int want = argc > 13 ? atoi(argv[13]) : 24;
    hipblasLtHandle_t h; CK(hipblasLtCreate(&h));
    void *A, *B, *C, *ws;
    const size_t wsz = 256u << 20;
    CK(hipMalloc(&A, (size_t)sA * batch * 2 + 4096)); CK(hipMalloc(&B, (size_t)sB * batch * 2 + 4096));
    CK(hipMalloc(&C, (size_t)sC * batch * 4 + 4096)); CK(hipMalloc(&ws, wsz));
    CK(hipMemset(A, 0x3c, (size_t)sA * batch * 2)); CK(hipMemset(B, 0x3c, (size_t)sB * batch * 2));
    hipblasLtMatrixLayout_t la, lb, lc;
    CK(hipblasLtMatrixLayoutCreate(&la, HIP_R_16BF, tA ? k : m, tA ? m : k, lda));
    CK(hipblasLtMatrixLayoutCreate(&lb, HIP_R_16BF, tB ? n : k, tB ? k : n, ldb));
    CK(hipblasLtMatrixLayoutCreate(&lc, HIP_R_32F, m, n, ldc));
    int32_t bc = batch; int64_t so;
    so = sA; CK(hipblasLtMatrixLayoutSetAttribute(la, HIPBLASLT_MATRIX_LAYOUT_BATCH_COUNT, &bc, sizeof(bc))); CK(hipblasLtMatrixLayoutSetAttribute(la, HIPBLASLT_MATRIX_LAYOUT_STRIDED_BATCH_OFFSET, &so, sizeof(so)));
    so = sB; CK(hipblasLtMatrixLayoutSetAttribute(lb, HIPBLASLT_MATRIX_LAYOUT_BATCH_COUNT, &bc, sizeof(bc))); CK(hipblasLtMatrixLayoutSetAttribute(lb, HIPBLASLT_MATRIX_LAYOUT_STRIDED_BATCH_OFFSET, &so, sizeof(so)));
    so = sC; CK(hipblasLtMatrixLayoutSetAttribute(lc, HIPBLASLT_MATRIX_LAYOUT_BATCH_COUNT, &bc, sizeof(bc))); CK(hipblasLtMatrixLayoutSetAttribute(lc, HIPBLASLT_MATRIX_LAYOUT_STRIDED_BATCH_OFFSET, &so, sizeof(so)));
    hipblasLtMatmulDesc_t d; CK(hipblasLtMatmulDescCreate(&d, HIPBLAS_COMPUTE_32F, HIP_R_32F));
    hipblasOperation_t oa = tA ? HIPBLAS_OP_T : HIPBLAS_OP_N, ob = tB ? HIPBLAS_OP_T : HIPBLAS_OP_N;
    CK(hipblasLtMatmulDescSetAttribute(d, HIPBLASLT_MATMUL_DESC_TRANSA, &oa, sizeof(oa)));
    CK(hipblasLtMatmulDescSetAttribute(d, HIPBLASLT_MATMUL_DESC_TRANSB, &ob, sizeof(ob)));
    hipblasLtMatmulPreference_t p; CK(hipblasLtMatmulPreferenceCreate(&p));
    CK(hipblasLtMatmulPreferenceSetAttribute(p, HIPBLASLT_MATMUL_PREF_MAX_WORKSPACE_BYTES, &wsz, sizeof(wsz)));
    std::vector<hipblasLtMatmulHeuristicResult_t> res(want);
    int got = 0;
    CK(hipblasLtMatmulAlgoGetHeuristic(h, d, la, lb, lc, lc, p, want, res.data(), &got));
    const float alpha = 1.f, beta = 0.f;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const double flops = 2.0 * m * n * k * batch;
    float first = 0, best = 1e30f; int besti = -1;
    for (int i = 0; i < got; ++i) {
        auto run = [&]() { return hipblasLtMatmul(h, d, &alpha, A, la, B, lb, &beta, C, lc, C, lc, &res[i].algo, ws, wsz, 0); };
        if (run() != HIPBLAS_STATUS_SUCCESS) { printf("  algo %2d: launch failed\n", i); continue; }
        run();
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0, 0));
        for (int r = 0; r < 10; ++r) run();
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 10;
        if (i == 0) first = ms;
        if (ms < best) { best = ms; besti = i; }
        printf("  algo %2d: %8.1f us  %7.0f TFLOP/s  ws %zu\n", i, ms * 1e3, flops / ms / 1e9, res[i].workspaceSize);
    }
    printf("m=%ld n=%ld k=%ld %c%c batch=%d: %d algos, first %.1f us, best #%d %.1f us (%.1f %% faster)\n", m, n, k, tA ? 'T' : 'N',
           tB ? 'T' : 'N', batch, got, first * 1e3, besti, best * 1e3, 100.0 * (first - best) / first);
    return 0;
}
