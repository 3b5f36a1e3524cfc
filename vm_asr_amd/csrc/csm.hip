// csm.hip — SS2D cross-scan / cross-merge for gfx950.
//
// Replaces the Triton kernels of the reference (model/csm_triton.py:7-79 scan, :82-154
// merge; PyTorch semantics model/vmamba.py:27-73):
//   scan :  xs[b,0,c,h*W+w] = x[b,c,h,w]      xs[b,1,c,w*H+h] = x[b,c,h,w]
//           xs[b,2] = flip(xs[b,0])            xs[b,3] = flip(xs[b,1])
//   merge:  y[b,c,h*W+w] = (ys0[l0] + ys2[L-1-l0]) + (ys1[l1] + ys3[L-1-l1]),  l1 = w*H+h
// Each is the other's backward.  HBM-bound pure data movement (5 D L s bytes per call):
// one 64x64 tile per workgroup, transposed through a padded LDS tile so that all four
// direction streams are read/written as contiguous 256-B wave accesses.
#include "common.h"

namespace vmasr {
namespace {

constexpr int kT = 64;

// raw copy of an element when the types agree (16-bit payloads travel as uint16_t), else convert
template <typename TO, typename TI>
__device__ __forceinline__ TO cvt(TI v) {
    if constexpr (sizeof(TI) == sizeof(TO)) return *reinterpret_cast<TO *>(&v);
    else return from_f32<TO>(to_f32(v));
}

template <typename T, typename TO = T>
__global__ __launch_bounds__(256) void cross_scan_kernel(const T *__restrict__ x, TO *__restrict__ xs,
                                                         const int C, const int H, const int W) {
    __shared__ TO tile[kT][kT + 1];
    const int ntw = (W + kT - 1) / kT;
    const int tw = blockIdx.x % ntw, th = blockIdx.x / ntw;
    const int c = blockIdx.y, b = blockIdx.z;
    const int h0 = th * kT, w0 = tw * kT;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const size_t L = (size_t)H * W;
    const T *xp = x + ((size_t)b * C + c) * L;
    TO *y0 = xs + (((size_t)b * 4 + 0) * C + c) * L;
    TO *y1 = xs + (((size_t)b * 4 + 1) * C + c) * L;
    TO *y2 = xs + (((size_t)b * 4 + 2) * C + c) * L;
    TO *y3 = xs + (((size_t)b * 4 + 3) * C + c) * L;
#pragma unroll 4
    for (int j = 0; j < kT / 4; ++j) {
        const int hl = ty + 4 * j, h = h0 + hl, w = w0 + tx;
        if (h < H && w < W) {
            const size_t l0 = (size_t)h * W + w;
            const TO v = cvt<TO>(xp[l0]);
            tile[hl][tx] = v;
            y0[l0] = v;
            y2[L - 1 - l0] = v;
        }
    }
    __syncthreads();
#pragma unroll 4
    for (int j = 0; j < kT / 4; ++j) {
        const int wl = ty + 4 * j, w = w0 + wl, h = h0 + tx;
        if (h < H && w < W) {
            const size_t l1 = (size_t)w * H + h;
            const TO v = tile[tx][wl];
            y1[l1] = v;
            y3[L - 1 - l1] = v;
        }
    }
}

template <typename T, typename TO = T>
__global__ __launch_bounds__(256) void cross_merge_kernel(const T *__restrict__ ys, TO *__restrict__ y,
                                                          const int C, const int H, const int W) {
    __shared__ float tile[kT][kT + 1];
    const int ntw = (W + kT - 1) / kT;
    const int tw = blockIdx.x % ntw, th = blockIdx.x / ntw;
    const int c = blockIdx.y, b = blockIdx.z;
    const int h0 = th * kT, w0 = tw * kT;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const size_t L = (size_t)H * W;
    const T *y0 = ys + (((size_t)b * 4 + 0) * C + c) * L;
    const T *y1 = ys + (((size_t)b * 4 + 1) * C + c) * L;
    const T *y2 = ys + (((size_t)b * 4 + 2) * C + c) * L;
    const T *y3 = ys + (((size_t)b * 4 + 3) * C + c) * L;
    TO *yp = y + ((size_t)b * C + c) * L;
#pragma unroll 4
    for (int j = 0; j < kT / 4; ++j) {
        const int wl = ty + 4 * j, w = w0 + wl, h = h0 + tx;
        if (h < H && w < W) {
            const size_t l1 = (size_t)w * H + h;
            tile[tx][wl] = to_f32(y1[l1]) + to_f32(y3[L - 1 - l1]);
        }
    }
    __syncthreads();
#pragma unroll 4
    for (int j = 0; j < kT / 4; ++j) {
        const int hl = ty + 4 * j, h = h0 + hl, w = w0 + tx;
        if (h < H && w < W) {
            const size_t l0 = (size_t)h * W + w;
            yp[l0] = from_f32<TO>((to_f32(y0[l0]) + to_f32(y2[L - 1 - l0])) + tile[hl][tx]);
        }
    }
}

int check_shape(const void *a, const void *b, int B, int C, int H, int W, int dtype, const char *what) {
    VMASR_REQUIRE(a && b, VMASR_EINVAL, "%s: null tensor", what);
    VMASR_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0, VMASR_EINVAL, "%s: non-positive size", what);
    VMASR_REQUIRE(B <= 65535 && C <= 65535, VMASR_EINVAL, "%s: B and C must be <= 65535", what);
    VMASR_REQUIRE(dtype == VMASR_F32 || dtype == VMASR_F16 || dtype == VMASR_BF16, VMASR_EINVAL,
                  "%s: dtype must be fp32/fp16/bf16", what);
    return 0;
}

}  // namespace
}  // namespace vmasr

using namespace vmasr;

VMASR_EXPORT int vmasr_cross_scan(const void *x, void *xs, int32_t B, int32_t C, int32_t H, int32_t W,
                                  int32_t dtype, vmasr_stream_t stream) {
    if (int e = check_shape(x, xs, B, C, H, W, dtype, "cross_scan")) return e;
    const dim3 grid(((W + kT - 1) / kT) * ((H + kT - 1) / kT), C, B);
    hipStream_t st = static_cast<hipStream_t>(stream);
    const double bytes = 5.0 * B * C * H * W * (dtype == VMASR_F32 ? 4 : 2);  // 1 read + 4 writes
    if (dtype == VMASR_F32)
        VMASR_LAUNCH(VMASR_K_CROSS_SCAN, bytes, cross_scan_kernel<float>, grid, dim3(256), 0, st, (const float *)x, (float *)xs, C, H, W);
    else  // 16-bit payloads are moved as raw bits
        VMASR_LAUNCH(VMASR_K_CROSS_SCAN, bytes, cross_scan_kernel<uint16_t>, grid, dim3(256), 0, st, (const uint16_t *)x, (uint16_t *)xs, C, H, W);
    return check_launch("cross_scan");
}

VMASR_EXPORT int vmasr_cross_merge(const void *ys, void *y, int32_t B, int32_t C, int32_t H, int32_t W,
                                   int32_t dtype, vmasr_stream_t stream) {
    if (int e = check_shape(ys, y, B, C, H, W, dtype, "cross_merge")) return e;
    const dim3 grid(((W + kT - 1) / kT) * ((H + kT - 1) / kT), C, B);
    hipStream_t st = static_cast<hipStream_t>(stream);
    const double bytes = 5.0 * B * C * H * W * (dtype == VMASR_F32 ? 4 : 2);  // 4 reads + 1 write
    switch (dtype) {
        case VMASR_F32:
            VMASR_LAUNCH(VMASR_K_CROSS_MERGE, bytes, cross_merge_kernel<float>, grid, dim3(256), 0, st, (const float *)ys, (float *)y, C, H, W);
            break;
        case VMASR_F16:
            VMASR_LAUNCH(VMASR_K_CROSS_MERGE, bytes, cross_merge_kernel<f16_t>, grid, dim3(256), 0, st, (const f16_t *)ys, (f16_t *)y, C, H, W);
            break;
        default:
            VMASR_LAUNCH(VMASR_K_CROSS_MERGE, bytes, cross_merge_kernel<bf16_t>, grid, dim3(256), 0, st, (const bf16_t *)ys, (bf16_t *)y, C, H, W);
    }
    return check_launch("cross_merge");
}

// Converting variants: 16-bit activations scanned straight into the fp32 streams the selective scan
// consumes (SS2D casts xs to float right after the scan, model/vmamba.py:1487-1491), and fp32 stream
// gradients merged straight into the 16-bit activation gradient.  Supported: 16-bit -> fp32 (scan),
// fp32 -> 16-bit (merge); equal dtypes are the plain entry points.
VMASR_EXPORT int vmasr_cross_scan_cvt(const void *x, void *xs, int32_t B, int32_t C, int32_t H, int32_t W,
                                      int32_t in_dtype, int32_t out_dtype, vmasr_stream_t stream) {
    if (in_dtype == out_dtype) return vmasr_cross_scan(x, xs, B, C, H, W, in_dtype, stream);
    if (int e = check_shape(x, xs, B, C, H, W, in_dtype, "cross_scan_cvt")) return e;
    VMASR_REQUIRE(out_dtype == VMASR_F32 && in_dtype != VMASR_F32, VMASR_EINVAL, "cross_scan_cvt: only 16-bit -> fp32");
    const dim3 grid(((W + kT - 1) / kT) * ((H + kT - 1) / kT), C, B);
    hipStream_t st = static_cast<hipStream_t>(stream);
    const double bytes = (double)B * C * H * W * (2 + 4 * 4);
    if (in_dtype == VMASR_BF16)
        VMASR_LAUNCH(VMASR_K_CROSS_SCAN, bytes, (cross_scan_kernel<bf16_t, float>), grid, dim3(256), 0, st, (const bf16_t *)x, (float *)xs, C, H, W);
    else
        VMASR_LAUNCH(VMASR_K_CROSS_SCAN, bytes, (cross_scan_kernel<f16_t, float>), grid, dim3(256), 0, st, (const f16_t *)x, (float *)xs, C, H, W);
    return check_launch("cross_scan_cvt");
}

VMASR_EXPORT int vmasr_cross_merge_cvt(const void *ys, void *y, int32_t B, int32_t C, int32_t H, int32_t W,
                                       int32_t in_dtype, int32_t out_dtype, vmasr_stream_t stream) {
    if (in_dtype == out_dtype) return vmasr_cross_merge(ys, y, B, C, H, W, in_dtype, stream);
    if (int e = check_shape(ys, y, B, C, H, W, in_dtype, "cross_merge_cvt")) return e;
    VMASR_REQUIRE(in_dtype == VMASR_F32 && out_dtype != VMASR_F32, VMASR_EINVAL, "cross_merge_cvt: only fp32 -> 16-bit");
    const dim3 grid(((W + kT - 1) / kT) * ((H + kT - 1) / kT), C, B);
    hipStream_t st = static_cast<hipStream_t>(stream);
    const double bytes = (double)B * C * H * W * (4 * 4 + 2);
    if (out_dtype == VMASR_BF16)
        VMASR_LAUNCH(VMASR_K_CROSS_MERGE, bytes, (cross_merge_kernel<float, bf16_t>), grid, dim3(256), 0, st, (const float *)ys, (bf16_t *)y, C, H, W);
    else
        VMASR_LAUNCH(VMASR_K_CROSS_MERGE, bytes, (cross_merge_kernel<float, f16_t>), grid, dim3(256), 0, st, (const float *)ys, (f16_t *)y, C, H, W);
    return check_launch("cross_merge_cvt");
}
