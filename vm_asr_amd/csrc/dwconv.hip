// dwconv.hip — SS2D depthwise 3x3 conv (pad 1) + bias + SiLU, forward and backward.
//
// Replaces `self.conv2d` (nn.Conv2d(groups=d_inner, k=3, pad=1)) followed by `self.act`
// (SiLU) in SS2D.forwardv2 — model/vmamba.py:859-868,1543-1545.  HBM-bound stencil:
// forward reads x once and writes y once (2 D L s bytes); neighbour taps come from L1/L2.
// Backward is two passes: (A) recompute the pre-activation, form gp = gy * silu'(pre),
// store it and reduce dw (9 taps) / db per workgroup -> a handful of float atomics;
// (B) dx = transposed stencil of gp.
#include "common.h"

namespace vmasr {
namespace {

constexpr int kChunk = 4096;  // elements of one (b,c) plane per workgroup

__device__ __forceinline__ float sigmoid_f(float v) { return 1.f / (1.f + __expf(-v)); }

template <typename T>
__device__ __forceinline__ float conv_at(const T *__restrict__ xp, const float (&wr)[9], float bias, int h,
                                         int w, int H, int W, float (&taps)[9]) {
    float acc = bias;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int hh = h + i - 1, ww = w + j - 1;
            const float v = (hh >= 0 && hh < H && ww >= 0 && ww < W) ? to_f32(xp[(size_t)hh * W + ww]) : 0.f;
            taps[i * 3 + j] = v;
            acc = fmaf(wr[i * 3 + j], v, acc);
        }
    return acc;
}

template <typename T>
__global__ __launch_bounds__(256) void dwconv_silu_fwd_kernel(const T *__restrict__ x, const float *__restrict__ w,
                                                              const float *__restrict__ bias, T *__restrict__ y,
                                                              const int C, const int H, const int W) {
    const int c = blockIdx.y, b = blockIdx.z;
    const int HW = H * W;
    const T *xp = x + ((size_t)b * C + c) * HW;
    T *yp = y + ((size_t)b * C + c) * HW;
    float wr[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) wr[k] = w[c * 9 + k];
    const float bv = bias ? bias[c] : 0.f;
    const int l_end = min(HW, (int)(blockIdx.x + 1) * kChunk);
    for (int l = blockIdx.x * kChunk + threadIdx.x; l < l_end; l += 256) {
        const int h = l / W, ww = l - h * W;
        float taps[9];
        const float pre = conv_at(xp, wr, bv, h, ww, H, W, taps);
        yp[l] = from_f32<T>(pre * sigmoid_f(pre));
    }
}

template <typename T>
__global__ __launch_bounds__(256) void dwconv_silu_bwd_a_kernel(const T *__restrict__ x, const float *__restrict__ w,
                                                                const float *__restrict__ bias,
                                                                const T *__restrict__ gy, float *__restrict__ gp,
                                                                float *__restrict__ dw, float *__restrict__ db,
                                                                const int C, const int H, const int W) {
    __shared__ float s_part[4][10];
    const int c = blockIdx.y, b = blockIdx.z;
    const int HW = H * W;
    const T *xp = x + ((size_t)b * C + c) * HW;
    const T *gyp = gy + ((size_t)b * C + c) * HW;
    float *gpp = gp + ((size_t)b * C + c) * HW;
    float wr[9], acc[10];
#pragma unroll
    for (int k = 0; k < 9; ++k) wr[k] = w[c * 9 + k];
#pragma unroll
    for (int k = 0; k < 10; ++k) acc[k] = 0.f;
    const float bv = bias ? bias[c] : 0.f;
    const int l_end = min(HW, (int)(blockIdx.x + 1) * kChunk);
    for (int l = blockIdx.x * kChunk + threadIdx.x; l < l_end; l += 256) {
        const int h = l / W, ww = l - h * W;
        float taps[9];
        const float pre = conv_at(xp, wr, bv, h, ww, H, W, taps);
        const float s = sigmoid_f(pre);
        const float g = to_f32(gyp[l]) * (s * (1.f + pre * (1.f - s)));
        gpp[l] = g;
#pragma unroll
        for (int k = 0; k < 9; ++k) acc[k] = fmaf(g, taps[k], acc[k]);
        acc[9] += g;
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < 10; ++k) {
        const float s = wave_sum(acc[k]);
        if (lane == 0) s_part[wave][k] = s;
    }
    __syncthreads();
    if (threadIdx.x < 10) {
        const float s = s_part[0][threadIdx.x] + s_part[1][threadIdx.x] + s_part[2][threadIdx.x] + s_part[3][threadIdx.x];
        if (threadIdx.x < 9) atomicAdd(dw + c * 9 + threadIdx.x, s);
        else if (db) atomicAdd(db + c, s);
    }
}

template <typename T>
__global__ __launch_bounds__(256) void dwconv_silu_bwd_b_kernel(const float *__restrict__ gp, const float *__restrict__ w,
                                                                T *__restrict__ dx, const int C, const int H,
                                                                const int W) {
    const int c = blockIdx.y, b = blockIdx.z;
    const int HW = H * W;
    const float *gpp = gp + ((size_t)b * C + c) * HW;
    T *dxp = dx + ((size_t)b * C + c) * HW;
    float wr[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) wr[k] = w[c * 9 + k];
    const int l_end = min(HW, (int)(blockIdx.x + 1) * kChunk);
    for (int l = blockIdx.x * kChunk + threadIdx.x; l < l_end; l += 256) {
        const int h = l / W, ww = l - h * W;
        float acc = 0.f;
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const int h2 = h - (i - 1), w2 = ww - (j - 1);
                if (h2 >= 0 && h2 < H && w2 >= 0 && w2 < W) acc = fmaf(wr[i * 3 + j], gpp[(size_t)h2 * W + w2], acc);
            }
        dxp[l] = from_f32<T>(acc);
    }
}

int check_shape(int B, int C, int H, int W, int dtype, const char *what) {
    VMASR_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0, VMASR_EINVAL, "%s: non-positive size", what);
    VMASR_REQUIRE(B <= 65535 && C <= 65535, VMASR_EINVAL, "%s: B and C must be <= 65535", what);
    VMASR_REQUIRE((long)H * W < (1L << 31), VMASR_EINVAL, "%s: plane too large", what);
    VMASR_REQUIRE(dtype == VMASR_F32 || dtype == VMASR_F16 || dtype == VMASR_BF16, VMASR_EINVAL,
                  "%s: dtype must be fp32/fp16/bf16", what);
    return 0;
}

}  // namespace
}  // namespace vmasr

using namespace vmasr;

VMASR_EXPORT int vmasr_dwconv_silu_fwd(const void *x, const float *w, const float *bias, void *y, int32_t B,
                                       int32_t C, int32_t H, int32_t W, int32_t dtype, vmasr_stream_t stream) {
    if (int e = check_shape(B, C, H, W, dtype, "dwconv_silu_fwd")) return e;
    VMASR_REQUIRE(x && w && y, VMASR_EINVAL, "dwconv_silu_fwd: null tensor");
    const dim3 grid((H * W + kChunk - 1) / kChunk, C, B);
    hipStream_t st = static_cast<hipStream_t>(stream);
    const double bytes = 2.0 * B * C * H * W * (dtype == VMASR_F32 ? 4 : 2);  // read x, write y
    switch (dtype) {
        case VMASR_F32: VMASR_LAUNCH(VMASR_K_DWCONV_FWD, bytes, dwconv_silu_fwd_kernel<float>, grid, dim3(256), 0, st, (const float *)x, w, bias, (float *)y, C, H, W); break;
        case VMASR_F16: VMASR_LAUNCH(VMASR_K_DWCONV_FWD, bytes, dwconv_silu_fwd_kernel<f16_t>, grid, dim3(256), 0, st, (const f16_t *)x, w, bias, (f16_t *)y, C, H, W); break;
        default: VMASR_LAUNCH(VMASR_K_DWCONV_FWD, bytes, dwconv_silu_fwd_kernel<bf16_t>, grid, dim3(256), 0, st, (const bf16_t *)x, w, bias, (bf16_t *)y, C, H, W);
    }
    return check_launch("dwconv_silu_fwd");
}

VMASR_EXPORT int vmasr_dwconv_silu_bwd(const void *x, const float *w, const float *bias, const void *gy, void *dx,
                                       float *dw, float *db, float *ws, int32_t B, int32_t C, int32_t H, int32_t W,
                                       int32_t dtype, vmasr_stream_t stream) {
    if (int e = check_shape(B, C, H, W, dtype, "dwconv_silu_bwd")) return e;
    VMASR_REQUIRE(x && w && gy && dx && dw && ws, VMASR_EINVAL, "dwconv_silu_bwd: null tensor");
    const dim3 grid((H * W + kChunk - 1) / kChunk, C, B);
    hipStream_t st = static_cast<hipStream_t>(stream);
    const double n = (double)B * C * H * W, es = dtype == VMASR_F32 ? 4 : 2;
    const double bytes_a = n * (2 * es + 4), bytes_b = n * (4 + es);  // a: read x, gy, write gp;  b: read gp, write dx
    switch (dtype) {
        case VMASR_F32:
            VMASR_LAUNCH(VMASR_K_DWCONV_BWD_A, bytes_a, dwconv_silu_bwd_a_kernel<float>, grid, dim3(256), 0, st, (const float *)x, w, bias, (const float *)gy, ws, dw, db, C, H, W);
            VMASR_LAUNCH(VMASR_K_DWCONV_BWD_B, bytes_b, dwconv_silu_bwd_b_kernel<float>, grid, dim3(256), 0, st, ws, w, (float *)dx, C, H, W);
            break;
        case VMASR_F16:
            VMASR_LAUNCH(VMASR_K_DWCONV_BWD_A, bytes_a, dwconv_silu_bwd_a_kernel<f16_t>, grid, dim3(256), 0, st, (const f16_t *)x, w, bias, (const f16_t *)gy, ws, dw, db, C, H, W);
            VMASR_LAUNCH(VMASR_K_DWCONV_BWD_B, bytes_b, dwconv_silu_bwd_b_kernel<f16_t>, grid, dim3(256), 0, st, ws, w, (f16_t *)dx, C, H, W);
            break;
        default:
            VMASR_LAUNCH(VMASR_K_DWCONV_BWD_A, bytes_a, dwconv_silu_bwd_a_kernel<bf16_t>, grid, dim3(256), 0, st, (const bf16_t *)x, w, bias, (const bf16_t *)gy, ws, dw, db, C, H, W);
            VMASR_LAUNCH(VMASR_K_DWCONV_BWD_B, bytes_b, dwconv_silu_bwd_b_kernel<bf16_t>, grid, dim3(256), 0, st, ws, w, (bf16_t *)dx, C, H, W);
    }
    return check_launch("dwconv_silu_bwd");
}
