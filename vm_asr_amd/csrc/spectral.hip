// spectral.hip — the matrix-vector products of spectral-norm power iteration, for gfx950.
//
// torch.nn.utils.parametrizations.spectral_norm (what the reference's MPD gets through the
// inverted ternary at model/discriminator.py:37) runs u <- normalize(W v), v <- normalize(W^T u)
// on every training forward, for 30 conv weights of up to 1024 x 5120 fp32.  As hipBLASLt GEMMs
// with one column these take 34 us each (0.6 TB/s) and 8 launches per iteration; `torch.mv`
// costs 4 ms of HOST time per call on ROCm 7.2.  Here: HBM-bound kernels.
//
//   gemv_rows : t[r] = sum_c W[r,c] v[c]      one wave per row, 16-B loads
//   gemv_cols : s[c] += sum_{r in chunk} W[r,c] u[r]   thread per 4 columns, 32-row chunks, atomics
//   l2_normalize : y = x / max(||x||, eps)    one workgroup
#include "common.h"

namespace vmasr {
namespace {

__global__ __launch_bounds__(256) void gemv_rows_kernel(const float *__restrict__ W, const float *__restrict__ v,
                                                        float *__restrict__ t, const int R, const int C) {
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= R) return;
    const float *row = W + (size_t)r * C;
    float acc = 0.f;
    if ((C & 3) == 0) {
        for (int c = lane * 4; c < C; c += 256) {
            const float4 a = *reinterpret_cast<const float4 *>(row + c);
            const float4 b = *reinterpret_cast<const float4 *>(v + c);
            acc += (a.x * b.x + a.y * b.y) + (a.z * b.z + a.w * b.w);
        }
    } else {
        for (int c = lane; c < C; c += 64) acc = fmaf(row[c], v[c], acc);
    }
    acc = wave_sum(acc);
    if (lane == 0) t[r] = acc;
}

constexpr int kRowChunk = 32;

__global__ __launch_bounds__(256) void gemv_cols_kernel(const float *__restrict__ W, const float *__restrict__ u,
                                                        float *__restrict__ s, const int R, const int C) {
    const int c0 = (blockIdx.x * 256 + threadIdx.x) * 4;
    const int r0 = blockIdx.y * kRowChunk, r1 = min(R, r0 + kRowChunk);
    if (c0 >= C) return;
    float a[4] = {0.f, 0.f, 0.f, 0.f};
    if ((C & 3) == 0) {
        for (int r = r0; r < r1; ++r) {
            const float4 w = *reinterpret_cast<const float4 *>(W + (size_t)r * C + c0);
            const float ur = u[r];
            a[0] = fmaf(w.x, ur, a[0]); a[1] = fmaf(w.y, ur, a[1]); a[2] = fmaf(w.z, ur, a[2]); a[3] = fmaf(w.w, ur, a[3]);
        }
    } else {
        for (int r = r0; r < r1; ++r) {
            const float ur = u[r];
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (c0 + i < C) a[i] = fmaf(W[(size_t)r * C + c0 + i], ur, a[i]);
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
        if (c0 + i < C) atomicAdd(s + c0 + i, a[i]);
}

// y = x / max(||x||_2, eps); also zeroes `clear` (n_clear floats) for the next gemv_cols accumulation
__global__ __launch_bounds__(1024) void l2_normalize_kernel(const float *__restrict__ x, float *__restrict__ y, const int n,
                                                            const float eps, float *__restrict__ clear, const int n_clear) {
    __shared__ float s_part[16];
    float acc = 0.f;
    for (int i = threadIdx.x; i < n; i += blockDim.x) acc = fmaf(x[i], x[i], acc);
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = acc;
    __syncthreads();
    float tot = 0.f;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) tot += s_part[w];
    const float inv = 1.f / fmaxf(sqrtf(tot), eps);
    for (int i = threadIdx.x; i < n; i += blockDim.x) y[i] = x[i] * inv;
    for (int i = threadIdx.x; i < n_clear; i += blockDim.x) clear[i] = 0.f;
}

}  // namespace
}  // namespace vmasr

using namespace vmasr;

// n_iter rounds of  u <- normalize(W v);  v <- normalize(W^T u)  in place.  W (R,C) fp32 row-major,
// u (R), v (C); ws: (R + C) floats of scratch.
VMASR_EXPORT int vmasr_spectral_power_iter(const float *W, float *u, float *v, float *ws, int32_t R, int32_t C,
                                           int32_t n_iter, float eps, vmasr_stream_t stream) {
    VMASR_REQUIRE(W && u && v && ws, VMASR_EINVAL, "spectral_power_iter: null tensor");
    VMASR_REQUIRE(R > 0 && C > 0 && n_iter >= 0, VMASR_EINVAL, "spectral_power_iter: bad size");
    VMASR_REQUIRE(aligned_to(W, 16) && aligned_to(v, 16), VMASR_EALIGN, "spectral_power_iter: 16-byte alignment required");
    hipStream_t st = static_cast<hipStream_t>(stream);
    float *t = ws, *s = ws + R;
    const double wb = (double)R * C * 4;
    if (n_iter > 0) (void)hipMemsetAsync(s, 0, (size_t)C * sizeof(float), st);
    for (int it = 0; it < n_iter; ++it) {
        VMASR_LAUNCH(VMASR_K_SPECTRAL, wb, gemv_rows_kernel, dim3((R + 3) / 4), dim3(256), 0, st, W, v, t, R, C);
        VMASR_LAUNCH(VMASR_K_SPECTRAL, R * 8.0, l2_normalize_kernel, dim3(1), dim3(1024), 0, st, t, u, R, eps, (float *)nullptr, 0);
        VMASR_LAUNCH(VMASR_K_SPECTRAL, wb, gemv_cols_kernel, dim3((C + 1023) / 1024, (R + kRowChunk - 1) / kRowChunk), dim3(256), 0,
                     st, W, u, s, R, C);
        // normalise into v and clear the accumulator for the next round
        VMASR_LAUNCH(VMASR_K_SPECTRAL, C * 8.0, l2_normalize_kernel, dim3(1), dim3(1024), 0, st, s, v, C, eps, s, C);
    }
    return check_launch("spectral_power_iter");
}
