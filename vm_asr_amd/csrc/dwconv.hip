// dwconv.hip — SS2D depthwise 3x3 conv (pad 1) + bias + SiLU, forward and backward.
//
// Replaces `self.conv2d` (nn.Conv2d(groups=d_inner, k=3, pad=1)) followed by `self.act`
// (SiLU) in SS2D.forwardv2 — model/vmamba.py:859-868,1543-1545.  HBM-bound stencil:
// forward reads x once and writes y once (2 D L s bytes); neighbour taps come from L1/L2.
// Backward is two passes: (A) recompute the pre-activation, form gp = gy * silu'(pre),
// store it and reduce dw (9 taps) / db per workgroup -> a handful of float atomics;
// (B) dx = transposed stencil of gp.
#include <algorithm>

#include "common.h"

namespace vmasr {
namespace {

// elements of one (b,c) plane per workgroup.  Measured in the step (make VARIANT=... DEFS=-DVMASR_DW_CHUNK=...): 4096 -> forward 11.1 us,
// backward A 17.2, B 8.9 us per launch; 2048 -> 9.2 / 16.9 / 8.3; 1024 -> 8.3 / 21.1 (four times the atomics per channel) / 8.2
#ifndef VMASR_DW_CHUNK
#define VMASR_DW_CHUNK 2048
#endif
constexpr int kChunk = VMASR_DW_CHUNK;

// v_rcp_f32 (1 ulp) instead of the ~12-instruction IEEE division sequence
__device__ __forceinline__ float sigmoid_f(float v) { return __builtin_amdgcn_rcpf(1.f + __expf(-v)); }

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
                                                              const int C, const int H, const int W, const int chunk) {
    const int c = blockIdx.y, b = blockIdx.z;
    const int HW = H * W;
    const T *xp = x + ((size_t)b * C + c) * HW;
    T *yp = y + ((size_t)b * C + c) * HW;
    float wr[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) wr[k] = w[c * 9 + k];
    const float bv = bias ? bias[c] : 0.f;
    const int l_end = min(HW, (int)(blockIdx.x + 1) * chunk);
    for (int l = blockIdx.x * chunk + threadIdx.x; l < l_end; l += 256) {
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
                                                                const int C, const int H, const int W, const int chunk, unsigned *det) {
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
    const int l_end = min(HW, (int)(blockIdx.x + 1) * chunk);
    for (int l = blockIdx.x * chunk + threadIdx.x; l < l_end; l += 256) {
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
    det_enter(det);                          // deterministic mode: the workgroups' atomics in workgroup order (common.h)
    if (threadIdx.x < 10) {
        const float s = s_part[0][threadIdx.x] + s_part[1][threadIdx.x] + s_part[2][threadIdx.x] + s_part[3][threadIdx.x];
        if (threadIdx.x < 9) atomicAdd(dw + c * 9 + threadIdx.x, s);
        else if (db) atomicAdd(db + c, s);
    }
    det_leave(det);
}

template <typename T>
__global__ __launch_bounds__(256) void dwconv_silu_bwd_b_kernel(const float *__restrict__ gp, const float *__restrict__ w,
                                                                T *__restrict__ dx, const int C, const int H,
                                                                const int W, const int chunk) {
    const int c = blockIdx.y, b = blockIdx.z;
    const int HW = H * W;
    const float *gpp = gp + ((size_t)b * C + c) * HW;
    T *dxp = dx + ((size_t)b * C + c) * HW;
    float wr[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) wr[k] = w[c * 9 + k];
    const int l_end = min(HW, (int)(blockIdx.x + 1) * chunk);
    for (int l = blockIdx.x * chunk + threadIdx.x; l < l_end; l += 256) {
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


// ---- vectorised variants: 4 consecutive outputs of one image row per thread (W % 4 == 0) -------
// Each thread loads the three input rows as one 16-B / 8-B vector plus the two halo scalars (L1
// hits: the neighbouring thread fetched them as part of its vector), i.e. 3 vector + 6 scalar
// loads per 4 outputs instead of 36 scalar loads.
template <typename T>
__device__ __forceinline__ void load_row6(const T *__restrict__ row, bool row_ok, int w0, int W, float (&v)[6]) {
    if (!row_ok) {
#pragma unroll
        for (int i = 0; i < 6; ++i) v[i] = 0.f;
        return;
    }
    float q[4];
    load4<T, true>(row, w0, W, q);
    v[0] = w0 > 0 ? to_f32(row[w0 - 1]) : 0.f;
    v[1] = q[0]; v[2] = q[1]; v[3] = q[2]; v[4] = q[3];
    v[5] = w0 + 4 < W ? to_f32(row[w0 + 4]) : 0.f;
}

// MODE 0: y = silu(conv + bias).  MODE 1 (backward a): gp = gy * silu'(conv + bias) -> gp (fp32),
// and the 9 + 1 weight/bias gradient sums of the block -> atomics.
template <typename T, int MODE>
__global__ __launch_bounds__(256) void dwconv_silu_vec_kernel(const T *__restrict__ x, const float *__restrict__ w,
                                                              const float *__restrict__ bias, const T *__restrict__ gy,
                                                              T *__restrict__ y, float *__restrict__ gp,
                                                              float *__restrict__ dw, float *__restrict__ db,
                                                              const int C, const int H, const int W, const int chunk, unsigned *det) {
    __shared__ float s_part[4][10];
    const int c = blockIdx.y, b = blockIdx.z;
    const int HW = H * W;
    const size_t plane = ((size_t)b * C + c) * HW;
    const T *xp = x + plane;
    float wr[9], acc[10];
#pragma unroll
    for (int k = 0; k < 9; ++k) wr[k] = w[c * 9 + k];
#pragma unroll
    for (int k = 0; k < 10; ++k) acc[k] = 0.f;
    const float bv = bias ? bias[c] : 0.f;
    const int l_end = min(HW, (int)(blockIdx.x + 1) * chunk);
    for (int l = blockIdx.x * chunk + threadIdx.x * 4; l < l_end; l += 1024) {
        const int h = l / W, w0 = l - h * W;
        float r0[6], r1[6], r2[6];
        load_row6(xp + (size_t)(h - 1) * W, h > 0, w0, W, r0);
        load_row6(xp + (size_t)h * W, true, w0, W, r1);
        load_row6(xp + (size_t)(h + 1) * W, h + 1 < H, w0, W, r2);
        float o[4], g4[4];
        if (MODE == 1) load4<T, true>(gy + plane, l, HW, g4);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float pre = bv;
            pre = fmaf(wr[0], r0[i], pre); pre = fmaf(wr[1], r0[i + 1], pre); pre = fmaf(wr[2], r0[i + 2], pre);
            pre = fmaf(wr[3], r1[i], pre); pre = fmaf(wr[4], r1[i + 1], pre); pre = fmaf(wr[5], r1[i + 2], pre);
            pre = fmaf(wr[6], r2[i], pre); pre = fmaf(wr[7], r2[i + 1], pre); pre = fmaf(wr[8], r2[i + 2], pre);
            const float sg = sigmoid_f(pre);
            if (MODE == 0) {
                o[i] = pre * sg;
            } else {
                const float g = g4[i] * (sg * (1.f + pre * (1.f - sg)));
                o[i] = g;
                acc[0] = fmaf(g, r0[i], acc[0]); acc[1] = fmaf(g, r0[i + 1], acc[1]); acc[2] = fmaf(g, r0[i + 2], acc[2]);
                acc[3] = fmaf(g, r1[i], acc[3]); acc[4] = fmaf(g, r1[i + 1], acc[4]); acc[5] = fmaf(g, r1[i + 2], acc[5]);
                acc[6] = fmaf(g, r2[i], acc[6]); acc[7] = fmaf(g, r2[i + 1], acc[7]); acc[8] = fmaf(g, r2[i + 2], acc[8]);
                acc[9] += g;
            }
        }
        if (MODE == 0) store4<T, true>(y + plane, l, HW, o);
        else store4<float, true>(gp + plane, l, HW, o);
    }
    if (MODE == 1) {
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
        for (int k = 0; k < 10; ++k) {
            const float sres = wave_sum(acc[k]);
            if (lane == 0) s_part[wave][k] = sres;
        }
        __syncthreads();
        det_enter(det);
        if (threadIdx.x < 10) {
            const float sres = s_part[0][threadIdx.x] + s_part[1][threadIdx.x] + s_part[2][threadIdx.x] + s_part[3][threadIdx.x];
            if (threadIdx.x < 9) atomicAdd(dw + c * 9 + threadIdx.x, sres);
            else if (db) atomicAdd(db + c, sres);
        }
        det_leave(det);
    }
}

// backward b, vectorised: dx[h,w] = sum_{i,j} w[i,j] * gp[h-i+1, w-j+1]
template <typename T>
__global__ __launch_bounds__(256) void dwconv_silu_bwd_b_vec_kernel(const float *__restrict__ gp, const float *__restrict__ w,
                                                                    T *__restrict__ dx, const int C, const int H,
                                                                    const int W, const int chunk) {
    const int c = blockIdx.y, b = blockIdx.z;
    const int HW = H * W;
    const size_t plane = ((size_t)b * C + c) * HW;
    const float *gpp = gp + plane;
    float wr[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) wr[k] = w[c * 9 + k];
    const int l_end = min(HW, (int)(blockIdx.x + 1) * chunk);
    for (int l = blockIdx.x * chunk + threadIdx.x * 4; l < l_end; l += 1024) {
        const int h = l / W, w0 = l - h * W;
        float r0[6], r1[6], r2[6];  // gp rows h+1, h, h-1 pair with kernel rows 0, 1, 2
        load_row6(gpp + (size_t)(h + 1) * W, h + 1 < H, w0, W, r0);
        load_row6(gpp + (size_t)h * W, true, w0, W, r1);
        load_row6(gpp + (size_t)(h - 1) * W, h > 0, w0, W, r2);
        float o[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            // column offset: kernel column j pairs with gp column w - (j-1): j=0 -> w+1, j=2 -> w-1
            float a = 0.f;
            a = fmaf(wr[0], r0[i + 2], a); a = fmaf(wr[1], r0[i + 1], a); a = fmaf(wr[2], r0[i], a);
            a = fmaf(wr[3], r1[i + 2], a); a = fmaf(wr[4], r1[i + 1], a); a = fmaf(wr[5], r1[i], a);
            a = fmaf(wr[6], r2[i + 2], a); a = fmaf(wr[7], r2[i + 1], a); a = fmaf(wr[8], r2[i], a);
            o[i] = a;
        }
        store4<T, true>(dx + plane, l, HW, o);
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
    const int chunk = kChunk;
    const dim3 grid((H * W + chunk - 1) / chunk, C, B);
    hipStream_t st = static_cast<hipStream_t>(stream);
    const double bytes = 2.0 * B * C * H * W * (dtype == VMASR_F32 ? 4 : 2);  // read x, write y
    const size_t al = dtype == VMASR_F32 ? 16 : 8;
    const bool vec = W % 4 == 0 && aligned_to(x, al) && aligned_to(y, al);
#define VMASR_DW_FWD(TT)                                                                                             \
    do {                                                                                                             \
        if (vec) VMASR_LAUNCH(VMASR_K_DWCONV_FWD, bytes, (dwconv_silu_vec_kernel<TT, 0>), grid, dim3(256), 0, st,     \
                              (const TT *)x, w, bias, (const TT *)nullptr, (TT *)y, (float *)nullptr, (float *)nullptr, \
                              (float *)nullptr, C, H, W, chunk, (unsigned *)nullptr);                                                     \
        else VMASR_LAUNCH(VMASR_K_DWCONV_FWD, bytes, dwconv_silu_fwd_kernel<TT>, grid, dim3(256), 0, st, (const TT *)x, w, \
                          bias, (TT *)y, C, H, W, chunk);                                                            \
    } while (0)
    switch (dtype) {
        case VMASR_F32: VMASR_DW_FWD(float); break;
        case VMASR_F16: VMASR_DW_FWD(f16_t); break;
        default: VMASR_DW_FWD(bf16_t);
    }
#undef VMASR_DW_FWD
    return check_launch("dwconv_silu_fwd");
}

VMASR_EXPORT int vmasr_dwconv_silu_bwd(const void *x, const float *w, const float *bias, const void *gy, void *dx,
                                       float *dw, float *db, float *ws, int32_t B, int32_t C, int32_t H, int32_t W,
                                       int32_t dtype, vmasr_stream_t stream) {
    if (int e = check_shape(B, C, H, W, dtype, "dwconv_silu_bwd")) return e;
    VMASR_REQUIRE(x && w && gy && dx && dw && ws, VMASR_EINVAL, "dwconv_silu_bwd: null tensor");
    // pass A ends in 10 atomics per workgroup on the channel's dw / db: at most 64 workgroups per plane (same-address atomics
    // serialise at ~13 ns each); pass B has no reduction and keeps the small chunk
    const int chunk = kChunk, chunk_a = std::max(kChunk, ((H * W / 64 + 1023) / 1024) * 1024);
    const dim3 grid((H * W + chunk - 1) / chunk, C, B), grid_a((H * W + chunk_a - 1) / chunk_a, C, B);
    hipStream_t st = static_cast<hipStream_t>(stream);
    const double n = (double)B * C * H * W, es = dtype == VMASR_F32 ? 4 : 2;
    const double bytes_a = n * (2 * es + 4), bytes_b = n * (4 + es);  // a: read x, gy, write gp;  b: read gp, write dx
    const size_t al = dtype == VMASR_F32 ? 16 : 8;
    const bool vec = W % 4 == 0 && aligned_to(x, al) && aligned_to(gy, al) && aligned_to(dx, al) && aligned_to(ws, 16);
    unsigned *det = det_ticket(VMASR_K_DWCONV_BWD_A);
#define VMASR_DW_BWD(TT)                                                                                              \
    do {                                                                                                              \
        if (vec) {                                                                                                    \
            VMASR_LAUNCH(VMASR_K_DWCONV_BWD_A, bytes_a, (dwconv_silu_vec_kernel<TT, 1>), grid_a, dim3(256), 0, st,      \
                         (const TT *)x, w, bias, (const TT *)gy, (TT *)nullptr, ws, dw, db, C, H, W, chunk_a, det); \
            VMASR_LAUNCH(VMASR_K_DWCONV_BWD_B, bytes_b, dwconv_silu_bwd_b_vec_kernel<TT>, grid, dim3(256), 0, st, ws, w, \
                         (TT *)dx, C, H, W, chunk);                                                                   \
        } else {                                                                                                      \
            VMASR_LAUNCH(VMASR_K_DWCONV_BWD_A, bytes_a, dwconv_silu_bwd_a_kernel<TT>, grid_a, dim3(256), 0, st, (const TT *)x, \
                         w, bias, (const TT *)gy, ws, dw, db, C, H, W, chunk_a, det);                                 \
            VMASR_LAUNCH(VMASR_K_DWCONV_BWD_B, bytes_b, dwconv_silu_bwd_b_kernel<TT>, grid, dim3(256), 0, st, ws, w,    \
                         (TT *)dx, C, H, W, chunk);                                                                   \
        }                                                                                                             \
    } while (0)
    switch (dtype) {
        case VMASR_F32: VMASR_DW_BWD(float); break;
        case VMASR_F16: VMASR_DW_BWD(f16_t); break;
        default: VMASR_DW_BWD(bf16_t);
    }
#undef VMASR_DW_BWD
    return check_launch("dwconv_silu_bwd");
}
