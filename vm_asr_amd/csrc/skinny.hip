// skinny.hip — y = x W^T (+ b) for MANY rows and FEW features (rows >= 4096; in, out <= 96): the U-Net glue of the generator at its
// two highest resolutions — patch embedding (im2col columns, K = 9 / 72), patch merging / expanding, skip and output projections
// (model/model.py:57-116,603-633; model/vmamba.py:1826-1837: nn.Linear / nn.Conv2d there) — and the input gradients of the same layers.
//
// These products are pure streams (a 262 144 x 9 operand is 4.7 MB, 38 MFLOP), but a GEMM library runs them as tiled GEMMs: 13-25 us per
// call on hipBLASLt against 2-4 us of HBM time, and in the two-stream step every microsecond of a chip-filling generator kernel costs
// ~3 us of step time.  Here a wave owns 64 consecutive rows = ONE contiguous block of the row-major operand:
//   load   the block with 16-byte vector loads into LDS as fp32 (row pitch K + 1: the per-lane row reads below are conflict-free),
//   compute lane = row: acc[n] += x[k] * W[k][n] with W^T in LDS (same address for all lanes: broadcast reads, four n per read),
//   store  the 64 x N results through LDS as one contiguous block again.
// fp32 accumulation in k order; inputs / outputs fp32 or bf16.  The weight is read through (row, column) strides, so the same kernel
// serves y = x W^T (forward, W (N, K)) and dx = g W (input gradient, W (N_out_of_layer, K): strides swapped).
#include "common.h"

namespace vmasr {
namespace {

constexpr int SK_ROWS = 256;           // rows per workgroup (4 waves x 64)

// NP: padded output count the accumulator array is unrolled over (multiple of 4)
template <typename TI, typename TO, int NP>
__global__ __launch_bounds__(SK_ROWS) void skinny_linear_kernel(const TI *__restrict__ x, const float *__restrict__ w, const float *__restrict__ bias,
                                                                TO *__restrict__ y, const long rows, const int K, const int N,
                                                                const long w_sn, const long w_sk) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float *wt = sm;                                    // [K][NP]  (W^T, zero-padded columns)
    float *xs = sm + K * NP;                           // [SK_ROWS][K + 1]; reused for the output block [SK_ROWS][N + 1]
    const int tid = threadIdx.x;
    const long r0 = (long)blockIdx.x * SK_ROWS;
    const int nrows = (int)((rows - r0) < SK_ROWS ? (rows - r0) : SK_ROWS);
    for (int i = tid; i < K * NP; i += SK_ROWS) {
        const int k = i / NP, n = i - k * NP;
        wt[i] = n < N ? w[n * w_sn + k * w_sk] : 0.f;
    }
    // the workgroup's rows are one contiguous run of nrows * K elements
    const TI *xb = x + r0 * K;
    const int total = nrows * K;
    const int P = K + 1;
    if constexpr (sizeof(TI) == 2) {
        // 8 elements per 16-byte load where the run's start is 16-byte aligned (r0 K 2 bytes: always, since SK_ROWS K 2 % 16 == 0)
        const int nv = total / 8;
        const uint4 *xv = reinterpret_cast<const uint4 *>(xb);
        for (int i = tid; i < nv; i += SK_ROWS) {
            const uint4 v = xv[i];
            const bf16_t *e = reinterpret_cast<const bf16_t *>(&v);
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int idx = i * 8 + q, r = idx / K, k = idx - r * K;
                xs[r * P + k] = (float)e[q];
            }
        }
        for (int idx = nv * 8 + tid; idx < total; idx += SK_ROWS) {
            const int r = idx / K, k = idx - r * K;
            xs[r * P + k] = (float)xb[idx];
        }
    } else {
        const int nv = total / 4;
        const float4 *xv = reinterpret_cast<const float4 *>(xb);
        for (int i = tid; i < nv; i += SK_ROWS) {
            const float4 v = xv[i];
            const float e[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int idx = i * 4 + q, r = idx / K, k = idx - r * K;
                xs[r * P + k] = e[q];
            }
        }
        for (int idx = nv * 4 + tid; idx < total; idx += SK_ROWS) {
            const int r = idx / K, k = idx - r * K;
            xs[r * P + k] = (float)xb[idx];
        }
    }
    __syncthreads();
    float acc[NP];
#pragma unroll
    for (int n = 0; n < NP; ++n) acc[n] = 0.f;
    if (tid < nrows) {
        const float *xr = xs + tid * P;
#pragma unroll 2
        for (int k = 0; k < K; ++k) {
            const float xv = xr[k];
            const float4 *wr = reinterpret_cast<const float4 *>(wt + k * NP);
#pragma unroll
            for (int n4 = 0; n4 < NP / 4; ++n4) {
                const float4 wv = wr[n4];                              // (one address for the whole wave: broadcast)
                acc[4 * n4 + 0] = fmaf(xv, wv.x, acc[4 * n4 + 0]);
                acc[4 * n4 + 1] = fmaf(xv, wv.y, acc[4 * n4 + 1]);
                acc[4 * n4 + 2] = fmaf(xv, wv.z, acc[4 * n4 + 2]);
                acc[4 * n4 + 3] = fmaf(xv, wv.w, acc[4 * n4 + 3]);
            }
        }
    }
    __syncthreads();                                   // every row has been read: the block's LDS becomes the output block
    const int PO = N + 1;
    if (tid < nrows) {
#pragma unroll
        for (int n = 0; n < NP; ++n)
            if (n < N) xs[tid * PO + n] = acc[n] + (bias ? bias[n] : 0.f);
    }
    __syncthreads();
    TO *yb = y + r0 * N;
    const int tout = nrows * N;
    for (int idx = tid; idx < tout; idx += SK_ROWS) {
        const int r = idx / N, n = idx - r * N;
        yb[idx] = (TO)xs[r * PO + n];
    }
}

template <typename TI, typename TO>
int sk_launch(const void *x, const float *w, const float *bias, void *y, long rows, int K, int N, long w_sn, long w_sk, hipStream_t st) {
    const int NP = (N + 3) / 4 * 4;
    const int pitch = (K > N ? K : N) + 1;
    const size_t smem = ((size_t)K * NP + (size_t)SK_ROWS * pitch) * sizeof(float);
    const dim3 grid((unsigned)((rows + SK_ROWS - 1) / SK_ROWS)), block(SK_ROWS);
    const double bytes = (double)rows * (K * sizeof(TI) + N * sizeof(TO)) + (double)K * N * 4;
#define SK_CASE(np)                                                                                                                   \
    if (NP <= np) {                                                                                                                   \
        const size_t sm_ = ((size_t)K * np + (size_t)SK_ROWS * pitch) * sizeof(float);                                                \
        static bool attr_##np = false;                 /* > 64 KB of dynamic LDS needs the opt-in, once per instantiation (136 KB max) */ \
        if (!attr_##np) {                                                                                                             \
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&skinny_linear_kernel<TI, TO, np>),                              \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 140 * 1024);                                        \
            attr_##np = true;                                                                                                         \
        }                                                                                                                             \
        VMASR_LAUNCH(VMASR_K_SKINNY_LINEAR, bytes, (skinny_linear_kernel<TI, TO, np>), grid, block, sm_, st, static_cast<const TI *>(x), w,  \
                     bias, static_cast<TO *>(y), rows, K, N, w_sn, w_sk);                                                             \
        return check_launch("skinny_linear");                                                                                        \
    }
    SK_CASE(8) SK_CASE(16) SK_CASE(32) SK_CASE(64) SK_CASE(96)
#undef SK_CASE
    (void)smem;
    return VMASR_EINVAL;
}

}  // namespace
}  // namespace vmasr

using namespace vmasr;

VMASR_EXPORT int vmasr_skinny_linear_supported(int64_t rows, int32_t in_f, int32_t out_f) {
    // LDS: W^T (in x out padded) + a 256-row block at pitch max(in, out) + 1, fp32: <= 96 x 96 x 4 + 256 x 97 x 4 = 136 KB
    return rows >= 4096 && in_f >= 1 && in_f <= 96 && out_f >= 1 && out_f <= 96;
}

VMASR_EXPORT int vmasr_skinny_linear(const void *x, const float *w, const float *bias, void *y, int64_t rows, int32_t in_f, int32_t out_f,
                                     int64_t w_stride_out, int64_t w_stride_in, int32_t x_dtype, int32_t y_dtype, vmasr_stream_t stream) {
    VMASR_REQUIRE(x && w && y, VMASR_EINVAL, "skinny_linear: null tensor");
    VMASR_REQUIRE(vmasr_skinny_linear_supported(rows, in_f, out_f), VMASR_EINVAL, "skinny_linear: unsupported shape (rows %lld, in %d, out %d)",
                  (long long)rows, in_f, out_f);
    VMASR_REQUIRE(aligned_to(x, 16) && aligned_to(y, 4), VMASR_EINVAL, "skinny_linear: unaligned tensor");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const bool xb = x_dtype == VMASR_BF16, yb = y_dtype == VMASR_BF16;
    VMASR_REQUIRE((xb || x_dtype == VMASR_F32) && (yb || y_dtype == VMASR_F32), VMASR_EINVAL, "skinny_linear: fp32 or bf16 tensors");
    if (xb && yb) return sk_launch<bf16_t, bf16_t>(x, w, bias, y, rows, in_f, out_f, w_stride_out, w_stride_in, st);
    if (xb) return sk_launch<bf16_t, float>(x, w, bias, y, rows, in_f, out_f, w_stride_out, w_stride_in, st);
    if (yb) return sk_launch<float, bf16_t>(x, w, bias, y, rows, in_f, out_f, w_stride_out, w_stride_in, st);
    return sk_launch<float, float>(x, w, bias, y, rows, in_f, out_f, w_stride_out, w_stride_in, st);
}
