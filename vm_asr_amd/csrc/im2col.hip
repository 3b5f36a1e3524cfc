// im2col.hip — column / scatter kernels of the period discriminator's (k,1) convolutions, gfx950.
//
// The reference's MultiPeriodDiscriminator (model/discriminator.py:21-147) is five stacks of
// Conv2d((5,1), stride (3,1)) / ((5,1), 1) / ((3,1), 1) over the signal folded to (T/p, p).  MIOpen has no
// tuned solver for them on gfx950 (naive_conv_* fallbacks, 40+ ms per call), so the host side runs them as
// GEMMs on a channel-last (N = B*p sequences, H, C) layout; these kernels build the GEMM operand and
// scatter its gradient back — pure HBM-bound data movement, previously ATen strided copies
// (`x.unfold(...).reshape` at ~1 TB/s) and `_unfold_backward` (2.7 ms per step):
//
//   im2col : cols[n, h1, j, c] = x[n, h1*s + j - pad, c]   (0 outside [0,H))     -> (N*H1, k*C) row-major
//   col2im : dx[n, h, c] = sum_{j : (h + pad - j) % s == 0, h1 = (h + pad - j)/s in [0,H1)} dcols[n, h1, j, c]
//
// col2im is a gather (<= ceil(k/s) terms per element): no atomics, deterministic.
// Algorithmic bytes: im2col reads N*H*C, writes N*H1*k*C; col2im reads N*H1*k*C, writes N*H*C.
#include <algorithm>
#include <type_traits>

#include "common.h"

namespace vmasr {
namespace {

struct ColGeom {
    int N, H, C, k, stride, pad, H1;
    long rows_out;  // im2col: rows written (>= N*H1; the surplus is zero-filled: padding of a stacked operand)
};

// V = elements per 16-byte vector (4 fp32 / 8 bf16); C % V == 0 on the vector path
template <typename T, int V>
__global__ __launch_bounds__(256) void im2col_kernel(const T *__restrict__ x, T *__restrict__ cols, const ColGeom g) {
    const int cv = g.C / V;                                   // vectors per tap
    const long total = g.rows_out * g.k * cv;
    using Vec = typename std::conditional<V == 1, T, uint4>::type;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % cv);
        long r = i / cv;
        const int j = (int)(r % g.k);
        r /= g.k;
        const int h1 = (int)(r % g.H1);
        const int n = (int)(r / g.H1);
        const int h = h1 * g.stride + j - g.pad;
        Vec v{};
        if (n < g.N && h >= 0 && h < g.H) v = reinterpret_cast<const Vec *>(x + ((size_t)n * g.H + h) * g.C)[c];
        reinterpret_cast<Vec *>(cols)[i] = v;
    }
}

// im2col of fp32 input with the error-compensated bf16 split of split.hip fused into the store: hi = bf16(v),
// lo = bf16(v - hi) go to two (rows_out, k*C) bf16 operands (4 B written per element instead of 4 B written + 4 B
// re-read + 4 B written by im2col followed by vmasr_split_bf16).
__global__ __launch_bounds__(256) void im2col_split_kernel(const float *__restrict__ x, bf16_t *__restrict__ hi,
                                                           bf16_t *__restrict__ lo, const ColGeom g) {
    const int cv = g.C / 4;
    const long total = g.rows_out * g.k * cv;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % cv);
        long r = i / cv;
        const int j = (int)(r % g.k);
        r /= g.k;
        const int h1 = (int)(r % g.H1);
        const int n = (int)(r / g.H1);
        const int h = h1 * g.stride + j - g.pad;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (n < g.N && h >= 0 && h < g.H) v = reinterpret_cast<const float4 *>(x + ((size_t)n * g.H + h) * g.C)[c];
        const float e[4] = {v.x, v.y, v.z, v.w};
        union { uint2 raw; bf16_t b[4]; } H, L;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            H.b[q] = (bf16_t)e[q];
            L.b[q] = (bf16_t)(e[q] - (float)H.b[q]);
        }
        reinterpret_cast<uint2 *>(hi)[i] = H.raw;
        reinterpret_cast<uint2 *>(lo)[i] = L.raw;
    }
}

template <typename T, int V>
__global__ __launch_bounds__(256) void col2im_kernel(const T *__restrict__ dcols, T *__restrict__ dx, const ColGeom g) {
    const int cv = g.C / V;
    const long total = (long)g.N * g.H * cv;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % cv);
        const long r = i / cv;
        const int h = (int)(r % g.H);
        const int n = (int)(r / g.H);
        float acc[V];
#pragma unroll
        for (int e = 0; e < V; ++e) acc[e] = 0.f;
        for (int j = (h + g.pad) % g.stride; j < g.k; j += g.stride) {
            const int q = h + g.pad - j;
            if (q < 0) break;
            const int h1 = q / g.stride;
            if (h1 >= g.H1) continue;
            const T *src = dcols + ((((size_t)n * g.H1 + h1) * g.k + j) * g.C) + (size_t)c * V;
            if constexpr (V == 1) {
                acc[0] += to_f32(src[0]);
            } else {
                union { uint4 raw; T e[V]; } q4;
                q4.raw = *reinterpret_cast<const uint4 *>(src);
#pragma unroll
                for (int e = 0; e < V; ++e) acc[e] += to_f32(q4.e[e]);
            }
        }
        T *dst = dx + ((size_t)n * g.H + h) * g.C + (size_t)c * V;
        if constexpr (V == 1) {
            dst[0] = from_f32<T>(acc[0]);
        } else {
            union { uint4 raw; T e[V]; } o;
#pragma unroll
            for (int e = 0; e < V; ++e) o.e[e] = from_f32<T>(acc[e]);
            *reinterpret_cast<uint4 *>(dst) = o.raw;
        }
    }
}

template <typename T>
int launch(bool fwd, const void *src, void *dst, const ColGeom &g, hipStream_t st) {
    constexpr int V = 16 / sizeof(T);
    const bool vec = (g.C % V == 0) && aligned_to(src, 16) && aligned_to(dst, 16);
    const long cvn = vec ? g.C / V : g.C;
    const long total = fwd ? g.rows_out * g.k * cvn : (long)g.N * g.H * cvn;
    const int blocks = (int)std::min<long>((total + 255) / 256, 256L * 64);   // grid-stride: ~64 workgroups per CU at most
    const double bytes = ((double)g.N * g.H * g.C + (double)g.N * g.H1 * g.k * g.C) * sizeof(T);
    if (fwd) {
        if (vec) VMASR_LAUNCH(VMASR_K_IM2COL, bytes, (im2col_kernel<T, V>), dim3(blocks), dim3(256), 0, st, static_cast<const T *>(src), static_cast<T *>(dst), g);
        else VMASR_LAUNCH(VMASR_K_IM2COL, bytes, (im2col_kernel<T, 1>), dim3(blocks), dim3(256), 0, st, static_cast<const T *>(src), static_cast<T *>(dst), g);
    } else {
        if (vec) VMASR_LAUNCH(VMASR_K_COL2IM, bytes, (col2im_kernel<T, V>), dim3(blocks), dim3(256), 0, st, static_cast<const T *>(src), static_cast<T *>(dst), g);
        else VMASR_LAUNCH(VMASR_K_COL2IM, bytes, (col2im_kernel<T, 1>), dim3(blocks), dim3(256), 0, st, static_cast<const T *>(src), static_cast<T *>(dst), g);
    }
    return check_launch(fwd ? "im2col_kx1" : "col2im_kx1");
}

int run(bool fwd, const void *src, void *dst, int64_t N, int32_t H, int32_t C, int32_t k, int32_t stride, int32_t pad,
        int64_t rows_out, int32_t dtype, hipStream_t st) {
    VMASR_REQUIRE(src && dst, VMASR_EINVAL, "im2col_kx1: null tensor");
    VMASR_REQUIRE(N > 0 && H > 0 && C > 0 && k > 0 && stride > 0 && pad >= 0 && H + 2 * pad >= k, VMASR_EINVAL,
                  "im2col_kx1: bad geometry (N=%ld H=%d C=%d k=%d stride=%d pad=%d)", (long)N, H, C, k, stride, pad);
    VMASR_REQUIRE(N <= 0x7fffffff, VMASR_EINVAL, "im2col_kx1: too many sequences");
    const int H1 = (H + 2 * pad - k) / stride + 1;
    VMASR_REQUIRE(rows_out == 0 || rows_out >= N * H1, VMASR_EINVAL, "im2col_kx1: rows_out smaller than N*H1");
    const ColGeom g{(int)N, H, C, k, stride, pad, H1, rows_out ? (long)rows_out : (long)N * H1};
    switch (dtype) {
        case VMASR_F32: return launch<float>(fwd, src, dst, g, st);
        case VMASR_F16: return launch<f16_t>(fwd, src, dst, g, st);
        case VMASR_BF16: return launch<bf16_t>(fwd, src, dst, g, st);
    }
    set_error("im2col_kx1: unsupported dtype %d", dtype);
    return VMASR_EINVAL;
}

}  // namespace
}  // namespace vmasr

using namespace vmasr;

VMASR_EXPORT int vmasr_im2col_kx1(const void *x, void *cols, int64_t N, int32_t H, int32_t C, int32_t k, int32_t stride,
                                  int32_t pad, int64_t rows_out, int32_t dtype, vmasr_stream_t stream) {
    return run(true, x, cols, N, H, C, k, stride, pad, rows_out, dtype, static_cast<hipStream_t>(stream));
}

VMASR_EXPORT int vmasr_im2col_kx1_split(const float *x, void *hi, void *lo, int64_t N, int32_t H, int32_t C, int32_t k,
                                        int32_t stride, int32_t pad, int64_t rows_out, vmasr_stream_t stream) {
    VMASR_REQUIRE(x && hi && lo, VMASR_EINVAL, "im2col_kx1_split: null tensor");
    VMASR_REQUIRE(N > 0 && H > 0 && C > 0 && k > 0 && stride > 0 && pad >= 0 && H + 2 * pad >= k, VMASR_EINVAL,
                  "im2col_kx1_split: bad geometry (N=%ld H=%d C=%d k=%d stride=%d pad=%d)", (long)N, H, C, k, stride, pad);
    VMASR_REQUIRE(N <= 0x7fffffff, VMASR_EINVAL, "im2col_kx1_split: too many sequences");
    VMASR_REQUIRE(C % 4 == 0 && aligned_to(x, 16) && aligned_to(hi, 8) && aligned_to(lo, 8), VMASR_EINVAL,
                  "im2col_kx1_split: needs C %% 4 == 0 and aligned operands");
    const int H1 = (H + 2 * pad - k) / stride + 1;
    VMASR_REQUIRE(rows_out == 0 || rows_out >= N * H1, VMASR_EINVAL, "im2col_kx1_split: rows_out smaller than N*H1");
    const ColGeom g{(int)N, H, C, k, stride, pad, H1, rows_out ? (long)rows_out : (long)N * H1};
    const long total = g.rows_out * k * (C / 4);
    const int blocks = (int)std::min<long>((total + 255) / 256, 256L * 64);
    hipStream_t st = static_cast<hipStream_t>(stream);
    const double bytes = (double)N * H * C * 4 + (double)g.rows_out * k * C * 4;
    VMASR_LAUNCH(VMASR_K_IM2COL, bytes, im2col_split_kernel, dim3(blocks), dim3(256), 0, st, x, static_cast<bf16_t *>(hi),
                 static_cast<bf16_t *>(lo), g);
    return check_launch("im2col_kx1_split");
}

VMASR_EXPORT int vmasr_col2im_kx1(const void *dcols, void *dx, int64_t N, int32_t H, int32_t C, int32_t k, int32_t stride,
                                  int32_t pad, int32_t dtype, vmasr_stream_t stream) {
    return run(false, dcols, dx, N, H, C, k, stride, pad, 0, dtype, static_cast<hipStream_t>(stream));
}
