// linear.hip — Linear layers with tiny feature counts over very many rows, for gfx950.
//
// The last VSS block of VM-ASR's output layer runs at full spectrogram resolution with
// d_model = 1 (model/model.py:865-885): its in_proj / out_proj / Mlp are nn.Linear(1,4), (2,1),
// (1,4), (4,1) applied to B*512*512 = 10^6 rows, and the 1x1 conv before it is a (4 -> 1) map
// (model/model.py:862-864).  As GEMMs these are degenerate: the weight gradient is a
// (4 x 10^6) @ (10^6 x 1) product that hipBLASLt runs in 1.3-1.5 ms per call, 17 ms per training
// step in total (profiles/r01_*).  They are memory-bound row maps:
//
//   forward : y[r, :] = W x[r, :] + b                        reads x, writes y
//   backward: dx[r, :] = W^T gy[r, :];  dW = sum_r gy[r,:] x[r,:]^T;  db = sum_r gy[r,:]
//
// One thread owns 4 consecutive rows (so every access is a 4-element vector), walks rows
// grid-stride with the OUT*IN + OUT gradient sums in registers, and each workgroup leaves one
// partial row that a small kernel sums (deterministic, no atomics).  IN, OUT in {1,2,4,8},
// IN*OUT <= 32; fp32 accumulation and fp32 weights whatever the activation dtype.
#include "common.h"

namespace vmasr {
namespace {

constexpr int kRowsPerThread = 4;

template <typename T, int N>  // N consecutive elements starting at p[i0] (i0 % 4 == 0 when N >= 4)
__device__ __forceinline__ void load_n(const T *__restrict__ p, long i0, long len, float (&v)[N]) {
    if constexpr (N % 4 == 0) {
#pragma unroll
        for (int k = 0; k < N / 4; ++k) {
            float q[4];
            if (i0 + 4 * k + 3 < len) load4<T, true>(p + i0 + 4 * k, 0, 4, q);
            else {
#pragma unroll
                for (int i = 0; i < 4; ++i) q[i] = (i0 + 4 * k + i < len) ? to_f32(p[i0 + 4 * k + i]) : 0.f;
            }
            v[4 * k] = q[0]; v[4 * k + 1] = q[1]; v[4 * k + 2] = q[2]; v[4 * k + 3] = q[3];
        }
    } else {
#pragma unroll
        for (int i = 0; i < N; ++i) v[i] = (i0 + i < len) ? to_f32(p[i0 + i]) : 0.f;
    }
}

template <typename T, int N>
__device__ __forceinline__ void store_n(T *__restrict__ p, long i0, long len, const float (&v)[N]) {
    if constexpr (N % 4 == 0) {
#pragma unroll
        for (int k = 0; k < N / 4; ++k) {
            const float q[4] = {v[4 * k], v[4 * k + 1], v[4 * k + 2], v[4 * k + 3]};
            if (i0 + 4 * k + 3 < len) store4<T, true>(p + i0 + 4 * k, 0, 4, q);
            else {
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (i0 + 4 * k + i < len) p[i0 + 4 * k + i] = from_f32<T>(q[i]);
            }
        }
    } else {
#pragma unroll
        for (int i = 0; i < N; ++i)
            if (i0 + i < len) p[i0 + i] = from_f32<T>(v[i]);
    }
}

template <typename TX, typename TY, int IN, int OUT>
__global__ __launch_bounds__(256) void small_linear_fwd_kernel(const TX *__restrict__ x, const float *__restrict__ w,
                                                               const float *__restrict__ bias, TY *__restrict__ y,
                                                               const long rows) {
    float wr[OUT][IN], br[OUT];
#pragma unroll
    for (int o = 0; o < OUT; ++o) {
        br[o] = bias ? bias[o] : 0.f;
#pragma unroll
        for (int i = 0; i < IN; ++i) wr[o][i] = w[o * IN + i];
    }
    const long stride = (long)gridDim.x * blockDim.x * kRowsPerThread;
    for (long r0 = ((long)blockIdx.x * blockDim.x + threadIdx.x) * kRowsPerThread; r0 < rows; r0 += stride) {
        float xv[kRowsPerThread * IN], yv[kRowsPerThread * OUT];
        load_n<TX, kRowsPerThread * IN>(x, r0 * IN, rows * IN, xv);
#pragma unroll
        for (int k = 0; k < kRowsPerThread; ++k)
#pragma unroll
            for (int o = 0; o < OUT; ++o) {
                float a = br[o];
#pragma unroll
                for (int i = 0; i < IN; ++i) a = fmaf(wr[o][i], xv[k * IN + i], a);
                yv[k * OUT + o] = a;
            }
        store_n<TY, kRowsPerThread * OUT>(y, r0 * OUT, rows * OUT, yv);
    }
}

template <typename TX, typename TY, int IN, int OUT>
__global__ __launch_bounds__(256) void small_linear_bwd_kernel(const TX *__restrict__ x, const float *__restrict__ w,
                                                               const TY *__restrict__ gy, TX *__restrict__ dx,
                                                               float *__restrict__ part, const long rows) {
    constexpr int NACC = OUT * IN + OUT;
    __shared__ float s_part[4][NACC];
    float wr[OUT][IN], acc[NACC];
#pragma unroll
    for (int o = 0; o < OUT; ++o)
#pragma unroll
        for (int i = 0; i < IN; ++i) wr[o][i] = w[o * IN + i];
#pragma unroll
    for (int k = 0; k < NACC; ++k) acc[k] = 0.f;
    const long stride = (long)gridDim.x * blockDim.x * kRowsPerThread;
    for (long r0 = ((long)blockIdx.x * blockDim.x + threadIdx.x) * kRowsPerThread; r0 < rows; r0 += stride) {
        float xv[kRowsPerThread * IN], gv[kRowsPerThread * OUT], dxv[kRowsPerThread * IN];
        load_n<TX, kRowsPerThread * IN>(x, r0 * IN, rows * IN, xv);
        load_n<TY, kRowsPerThread * OUT>(gy, r0 * OUT, rows * OUT, gv);  // zeros beyond the last row
#pragma unroll
        for (int k = 0; k < kRowsPerThread; ++k) {
#pragma unroll
            for (int i = 0; i < IN; ++i) {
                float a = 0.f;
#pragma unroll
                for (int o = 0; o < OUT; ++o) a = fmaf(wr[o][i], gv[k * OUT + o], a);
                dxv[k * IN + i] = a;
            }
#pragma unroll
            for (int o = 0; o < OUT; ++o) {
#pragma unroll
                for (int i = 0; i < IN; ++i) acc[o * IN + i] = fmaf(gv[k * OUT + o], xv[k * IN + i], acc[o * IN + i]);
                acc[OUT * IN + o] += gv[k * OUT + o];
            }
        }
        if (dx) store_n<TX, kRowsPerThread * IN>(dx, r0 * IN, rows * IN, dxv);
    }
    if (part) {
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
        for (int k = 0; k < NACC; ++k) {
            const float s = wave_sum(acc[k]);
            if (lane == 0) s_part[wave][k] = s;
        }
        __syncthreads();
        if (threadIdx.x < NACC)
            part[(size_t)blockIdx.x * NACC + threadIdx.x] =
                s_part[0][threadIdx.x] + s_part[1][threadIdx.x] + s_part[2][threadIdx.x] + s_part[3][threadIdx.x];
    }
}

// out[c] = sum_k part[k][c]   (one wave per column)
__global__ __launch_bounds__(256) void colsum_kernel(const float *__restrict__ part, const int nblk, const int ncol,
                                                     float *__restrict__ dw, const int n_w, float *__restrict__ db) {
    const int c = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (c >= ncol) return;
    float a = 0.f;
    for (int k = lane; k < nblk; k += 64) a += part[(size_t)k * ncol + c];
    a = wave_sum(a);
    if (lane == 0) {
        if (c < n_w) { if (dw) dw[c] = a; }
        else if (db) db[c - n_w] = a;
    }
}

int grid_for(long rows) {
    long nblk = (rows + 256 * kRowsPerThread - 1) / (256 * kRowsPerThread);
    return (int)(nblk > 1024 ? 1024 : (nblk < 1 ? 1 : nblk));
}

bool supported(int IN, int OUT) {
    auto pow2le8 = [](int v) { return v == 1 || v == 2 || v == 4 || v == 8; };
    return pow2le8(IN) && pow2le8(OUT) && IN * OUT <= 32;
}

template <typename TX, typename TY, int KIND>
int launch_io(int IN, int OUT, hipStream_t st, double bytes, const void *x, const float *w, const float *bias,
              const void *gy, void *out, float *part, long rows) {
    const dim3 grid(grid_for(rows));
#define VMASR_SL(I, O)                                                                                                  \
    if (IN == I && OUT == O) {                                                                                          \
        if (KIND == 0)                                                                                                  \
            VMASR_LAUNCH(VMASR_K_SMALL_LINEAR_FWD, bytes, (small_linear_fwd_kernel<TX, TY, I, O>), grid, dim3(256), 0, st, \
                         (const TX *)x, w, bias, (TY *)out, rows);                                                      \
        else                                                                                                            \
            VMASR_LAUNCH(VMASR_K_SMALL_LINEAR_BWD, bytes, (small_linear_bwd_kernel<TX, TY, I, O>), grid, dim3(256), 0, st, \
                         (const TX *)x, w, (const TY *)gy, (TX *)out, part, rows);                                      \
        return 0;                                                                                                       \
    }
    VMASR_SL(1, 1) VMASR_SL(1, 2) VMASR_SL(1, 4) VMASR_SL(1, 8) VMASR_SL(2, 1) VMASR_SL(2, 2) VMASR_SL(2, 4) VMASR_SL(2, 8)
    VMASR_SL(4, 1) VMASR_SL(4, 2) VMASR_SL(4, 4) VMASR_SL(4, 8) VMASR_SL(8, 1) VMASR_SL(8, 2) VMASR_SL(8, 4)
#undef VMASR_SL
    set_error("small_linear: unsupported (in=%d, out=%d)", IN, OUT);
    return VMASR_EINVAL;
}

template <int KIND>
int dispatch(int xdt, int ydt, int IN, int OUT, hipStream_t st, double bytes, const void *x, const float *w,
             const float *bias, const void *gy, void *out, float *part, long rows) {
#define VMASR_SL_T(TX, TY) launch_io<TX, TY, KIND>(IN, OUT, st, bytes, x, w, bias, gy, out, part, rows)
    if (xdt == VMASR_F32 && ydt == VMASR_F32) return VMASR_SL_T(float, float);
    if (xdt == VMASR_BF16 && ydt == VMASR_BF16) return VMASR_SL_T(bf16_t, bf16_t);
    if (xdt == VMASR_F32 && ydt == VMASR_BF16) return VMASR_SL_T(float, bf16_t);
    if (xdt == VMASR_F16 && ydt == VMASR_F16) return VMASR_SL_T(f16_t, f16_t);
    if (xdt == VMASR_F32 && ydt == VMASR_F16) return VMASR_SL_T(float, f16_t);
#undef VMASR_SL_T
    set_error("small_linear: unsupported dtype pair (%d -> %d)", xdt, ydt);
    return VMASR_EINVAL;
}

}  // namespace
}  // namespace vmasr

using namespace vmasr;

VMASR_EXPORT int vmasr_small_linear_supported(int32_t in_features, int32_t out_features) {
    return supported(in_features, out_features) ? 1 : 0;
}

VMASR_EXPORT int vmasr_small_linear_fwd(const void *x, const float *w, const float *bias, void *y, int64_t rows,
                                        int32_t in_features, int32_t out_features, int32_t x_dtype, int32_t y_dtype,
                                        vmasr_stream_t stream) {
    VMASR_REQUIRE(x && w && y, VMASR_EINVAL, "small_linear_fwd: null tensor");
    VMASR_REQUIRE(rows > 0 && supported(in_features, out_features), VMASR_EINVAL,
                  "small_linear_fwd: need rows > 0, in/out in {1,2,4,8}, in*out <= 32");
    VMASR_REQUIRE(aligned_to(x, 16) && aligned_to(y, 16), VMASR_EALIGN, "small_linear_fwd: 16-byte alignment required");
    const double es = x_dtype == VMASR_F32 ? 4 : 2, ey = y_dtype == VMASR_F32 ? 4 : 2;
    if (int e = dispatch<0>(x_dtype, y_dtype, in_features, out_features, static_cast<hipStream_t>(stream),
                            rows * (in_features * es + out_features * ey), x, w, bias, nullptr, y, nullptr, rows))
        return e;
    return check_launch("small_linear_fwd");
}

VMASR_EXPORT size_t vmasr_small_linear_bwd_workspace(int64_t rows, int32_t in_features, int32_t out_features) {
    if (rows <= 0 || !supported(in_features, out_features)) return 0;
    return (size_t)grid_for(rows) * (in_features * out_features + out_features) * sizeof(float);
}

VMASR_EXPORT int vmasr_small_linear_bwd(const void *x, const float *w, const void *gy, void *dx, float *dw, float *db,
                                        float *ws, int64_t rows, int32_t in_features, int32_t out_features,
                                        int32_t x_dtype, int32_t gy_dtype, vmasr_stream_t stream) {
    VMASR_REQUIRE(x && w && gy, VMASR_EINVAL, "small_linear_bwd: null tensor");
    VMASR_REQUIRE(rows > 0 && supported(in_features, out_features), VMASR_EINVAL,
                  "small_linear_bwd: need rows > 0, in/out in {1,2,4,8}, in*out <= 32");
    VMASR_REQUIRE(aligned_to(x, 16) && aligned_to(gy, 16) && (!dx || aligned_to(dx, 16)), VMASR_EALIGN,
                  "small_linear_bwd: 16-byte alignment required");
    const bool wants = dw || db;
    VMASR_REQUIRE(!wants || ws, VMASR_ENOSPACE, "small_linear_bwd: workspace required for dw/db");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const double es = x_dtype == VMASR_F32 ? 4 : 2, ey = gy_dtype == VMASR_F32 ? 4 : 2;
    const double bytes = rows * (in_features * es * (dx ? 2 : 1) + out_features * ey);
    if (int e = dispatch<1>(x_dtype, gy_dtype, in_features, out_features, st, bytes, x, w, nullptr, gy, dx,
                            wants ? ws : nullptr, rows))
        return e;
    if (wants) {
        const int ncol = in_features * out_features + out_features, nblk = grid_for(rows);
        VMASR_LAUNCH(VMASR_K_SMALL_LINEAR_REDUCE, (double)nblk * ncol * 4, colsum_kernel, dim3((ncol + 3) / 4), dim3(256), 0,
                     st, ws, nblk, ncol, dw, in_features * out_features, db);
    }
    return check_launch("small_linear_bwd");
}

// ---- fp32 Linear with float64 accumulation (the fp32 parity path) -----------------------------------------------------------
// y[r][n] = sum_k x[r][k] W[n][k] + b[n], fp32 operands and result, the dot product accumulated in float64 and rounded ONCE.
// Why: hipBLASLt's fp32 GEMMs accumulate K >= 256 in an order that leaves the result 1.4 - 1.7x further from the exact value than
// torch's CPU fp32 evaluation of the same reference line (tools/linear_accuracy.py, profiles/r03_linear_accuracy.log), and the
// Linear layers were the largest single-family term of the full-size forward's distance from float64.  With the
// accumulation in float64 the family's error is the final rounding alone (<= 0.5 ulp), below any fp32 summation order — the
// reference's included.  Used by vm_asr_amd/linear.py outside autocast only (model/vmamba.py:855,881,498-500, model/model.py:
// 57-116); the bf16 benchmark path never reaches it.  64 x 64 output tile, K step 16, 4 x 4 outputs per thread.
namespace vmasr {
namespace {

__global__ __launch_bounds__(256) void linear_f64acc_kernel(const float *__restrict__ x, const float *__restrict__ w,
                                                            const float *__restrict__ bias, float *__restrict__ y, const long M, const int N,
                                                            const int K) {
    __shared__ float xs[16][68], ws[16][68];
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const long m0 = (long)blockIdx.x * 64;
    const int n0 = blockIdx.y * 64;
    double acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = 0.0;
    const int lr = threadIdx.x >> 2, lk = (threadIdx.x & 3) * 4;     // loader: row 0..63, 4 consecutive k
    for (int k0 = 0; k0 < K; k0 += 16) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int k = k0 + lk + e;
            const long m = m0 + lr;
            const int n = n0 + lr;
            xs[lk + e][lr] = (m < M && k < K) ? x[m * K + k] : 0.f;
            ws[lk + e][lr] = (n < N && k < K) ? w[(size_t)n * K + k] : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) {
            double a[4], b[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) a[i] = (double)xs[kk][ty * 4 + i];
#pragma unroll
            for (int j = 0; j < 4; ++j) b[j] = (double)ws[kk][tx * 4 + j];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = fma(a[i], b[j], acc[i][j]);
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const long m = m0 + ty * 4 + i;
        if (m >= M) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = n0 + tx * 4 + j;
            if (n < N) y[m * N + n] = (float)(acc[i][j] + (bias ? (double)bias[n] : 0.0));
        }
    }
}

}  // namespace
}  // namespace vmasr

VMASR_EXPORT int vmasr_linear_f64acc(const float *x, const float *w, const float *bias, float *y, int64_t rows, int32_t out_features,
                                     int32_t in_features, vmasr_stream_t stream) {
    VMASR_REQUIRE(x && w && y, VMASR_EINVAL, "linear_f64acc: null tensor");
    VMASR_REQUIRE(rows >= 0 && out_features > 0 && in_features > 0, VMASR_EINVAL, "linear_f64acc: bad sizes");
    if (rows == 0) return 0;
    const dim3 grid((unsigned)((rows + 63) / 64), (unsigned)((out_features + 63) / 64));
    VMASR_REQUIRE(grid.y <= 65535, VMASR_EINVAL, "linear_f64acc: out_features too large");
    hipStream_t st = static_cast<hipStream_t>(stream);
    VMASR_LAUNCH(VMASR_K_SMALL_LINEAR_FWD, 4.0 * ((double)rows * (in_features + out_features) + (double)out_features * in_features),
                 vmasr::linear_f64acc_kernel, grid, dim3(256), 0, st, x, w, bias, y, (long)rows, out_features, in_features);
    return vmasr::check_launch("linear_f64acc");
}
