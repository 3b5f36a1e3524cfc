// split.hip — error-compensated bf16 operands for fp32-precision GEMMs on the bf16 matrix cores (gfx950).
//
// The period discriminator (model/discriminator.py:21-147) runs in fp32 in the reference (its losses sit
// outside autocast, trainer/trainer.py:138-142).  gfx950 has no TF32/xf32 path and its f32-input MFMA runs at
// the vector rate (157 TFLOP/s, 1/16 of bf16), so the 3.8 TFLOP of (k,1)-convolution GEMMs per training step
// are bound by that peak.  An fp32 value splits exactly into  x = hi + lo + r  with hi = bf16(x),
// lo = bf16(x - hi), |r| <= 2^-17 |x|;  a product of two such sums is
//     a b = a_hi b_hi + a_hi b_lo + a_lo b_hi + O(2^-16 |a b|)
// i.e. THREE bf16 MFMA GEMMs accumulated in fp32 reproduce the fp32 GEMM to ~1e-5 relative per product (random
// sign: ~1e-6 on a K = 5120 dot product) — fp32 parity (1e-4) at up to 16/3 of the fp32 matrix rate.
// This file holds the splitter (memory-bound: 4 B read, 2 + 2 B written per element); the GEMMs are hipBLASLt's.
#include <algorithm>

#include "common.h"

namespace vmasr {
namespace {

__device__ __forceinline__ void split1(float x, bf16_t &hi, bf16_t &lo) {
    hi = (bf16_t)x;                    // round to nearest even (v_cvt_pk_bf16_f32)
    lo = (bf16_t)(x - (float)hi);      // exact difference, rounded once
}

__global__ __launch_bounds__(256) void split_bf16_kernel(const float *__restrict__ x, bf16_t *__restrict__ hi,
                                                         bf16_t *__restrict__ lo, const size_t n8, const size_t n) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += stride) {
        const float4 a = reinterpret_cast<const float4 *>(x)[2 * i], b = reinterpret_cast<const float4 *>(x)[2 * i + 1];
        const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
        union { uint4 raw; bf16_t e[8]; } h, l;
#pragma unroll
        for (int j = 0; j < 8; ++j) split1(v[j], h.e[j], l.e[j]);
        reinterpret_cast<uint4 *>(hi)[i] = h.raw;
        reinterpret_cast<uint4 *>(lo)[i] = l.raw;
    }
    // tail (n % 8 elements)
    for (size_t i = n8 * 8 + (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) split1(x[i], hi[i], lo[i]);
}

}  // namespace
}  // namespace vmasr

using namespace vmasr;

VMASR_EXPORT int vmasr_split_bf16(const float *x, void *hi, void *lo, int64_t n, vmasr_stream_t stream) {
    VMASR_REQUIRE(x && hi && lo, VMASR_EINVAL, "split_bf16: null tensor");
    VMASR_REQUIRE(n >= 0, VMASR_EINVAL, "split_bf16: negative size");
    if (n == 0) return 0;
    const bool vec = aligned_to(x, 16) && aligned_to(hi, 16) && aligned_to(lo, 16);
    const size_t n8 = vec ? (size_t)n / 8 : 0;
    const long want = (long)((n8 ? n8 : (size_t)n) + 255) / 256;
    const int blocks = (int)std::min<long>(std::max<long>(want, 1), 256L * 16);
    hipStream_t st = static_cast<hipStream_t>(stream);
    VMASR_LAUNCH(VMASR_K_SPLIT_BF16, 8.0 * (double)n, split_bf16_kernel, dim3(blocks), dim3(256), 0, st, x,
                 static_cast<bf16_t *>(hi), static_cast<bf16_t *>(lo), n8, (size_t)n);
    return check_launch("split_bf16");
}
