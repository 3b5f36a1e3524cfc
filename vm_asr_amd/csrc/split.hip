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

// ---- bias + GELU epilogue of the discriminator's GEMMs, and its backward fused with the bf16 split -------------------
// forward : pre = acc + bias[col] (in place), act = GELU(pre)  (exact erf form, torch's default)     r 4, w 8 B/elt
// backward: gx = g * GELU'(pre) -> (hi, lo) bf16 split of gx (the operands of the dcols / dW GEMM triples) and
//           db[col] += sum_rows gx.  gx itself is never written: r 8, w 4 B/elt instead of GELU-backward (r 8, w 4)
//           + bias-gradient reduction (r 4) + split (r 4, w 4).
namespace vmasr {
namespace {

__device__ __forceinline__ float gelu_f(float x) { return 0.5f * x * (1.f + erff(x * 0.70710678118654752f)); }
__device__ __forceinline__ float gelu_grad_f(float x) {
    return 0.5f * (1.f + erff(x * 0.70710678118654752f)) + x * 0.3989422804014327f * __expf(-0.5f * x * x);
}

// grid: (row chunks, slots); block 256 threads; each thread owns 4 consecutive columns of one or more rows per pass.
// acc is (parts, slots, M, N): the pre-activation is the SUM of the parts (the three products of an error-compensated
// GEMM triple, written side by side instead of accumulated by two read-modify-write passes) + bias, stored into part 0.
__global__ __launch_bounds__(256) void bias_gelu_fwd_kernel(float *__restrict__ acc, const float *__restrict__ bias,
                                                            float *__restrict__ act, const long M, const int N, const int parts,
                                                            const size_t part_stride) {
    const int slot = blockIdx.y;
    const int nv = N / 4;                                   // float4 per row
    const long total = M * nv;
    float *a = acc + (size_t)slot * M * N, *o = act + (size_t)slot * M * N;
    const float *b = bias + (size_t)slot * N;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % nv);
        float4 v = reinterpret_cast<float4 *>(a)[i];
        for (int p = 1; p < parts; ++p) {
            const float4 q = reinterpret_cast<const float4 *>(a + (size_t)p * part_stride)[i];
            v.x += q.x; v.y += q.y; v.z += q.z; v.w += q.w;
        }
        const float4 bb = reinterpret_cast<const float4 *>(b)[c];
        v.x += bb.x; v.y += bb.y; v.z += bb.z; v.w += bb.w;
        reinterpret_cast<float4 *>(a)[i] = v;
        reinterpret_cast<float4 *>(o)[i] = make_float4(gelu_f(v.x), gelu_f(v.y), gelu_f(v.z), gelu_f(v.w));
    }
}

// Weight operand of the stacked convolutions' GEMM triples: w (n, N, K) fp32 -> out (n, K, 3N) bf16 = [hi^T | hi^T | lo^T].
// Columns [0, N) and [2N, 3N) are the forward GEMM's B operands (y = ch hi^T + cl hi^T + ch lo^T), the whole thing is
// the B^T of the concatenated-contraction column-gradient GEMM.  One pass (r 4, w 6 B/elt) instead of transpose copy +
// split + cat (r 14, w 14).  64 x 64 tiles through LDS; grid (K tiles, N tiles, n).
__global__ __launch_bounds__(256) void weight_prep_kernel(const float *__restrict__ w, bf16_t *__restrict__ out, const int N, const int K) {
    __shared__ float tile[64][65];
    const int slot = blockIdx.z, k0 = blockIdx.x * 64, o0 = blockIdx.y * 64;
    const float *src = w + (size_t)slot * N * K;
    bf16_t *dst = out + (size_t)slot * K * 3 * N;
    const int tx = threadIdx.x % 64, ty = threadIdx.x / 64;          // 4 rows of 64 per pass
    for (int r = ty; r < 64; r += 4) {
        const int o = o0 + r, k = k0 + tx;
        tile[r][tx] = (o < N && k < K) ? src[(size_t)o * K + k] : 0.f;
    }
    __syncthreads();
    for (int r = ty; r < 64; r += 4) {                               // r: k within the tile, tx: o within the tile
        const int k = k0 + r, o = o0 + tx;
        if (k < K && o < N) {
            bf16_t h, l;
            split1(tile[tx][r], h, l);
            bf16_t *row = dst + (size_t)k * 3 * N;
            row[o] = h; row[N + o] = h; row[2 * N + o] = l;
        }
    }
}

constexpr int kGbRows = 256;   // rows per workgroup of the backward (column sums leave as one atomic per column and workgroup)
constexpr int kGbLanes = 64;   // float4 lanes (256 columns) per workgroup: a (256 rows x 256 columns) tile, so that
                               // (M/256) x (N/256) x slots workgroups fill the chip (120 workgroups at 256 rows x N before)

__global__ __launch_bounds__(256) void gelu_bwd_split_kernel(const float *__restrict__ pre, const float *__restrict__ g,
                                                             bf16_t *__restrict__ hi, bf16_t *__restrict__ lo,
                                                             bf16_t *__restrict__ cat3, float *__restrict__ gx,
                                                             float *__restrict__ db, const long M, const int N, const int has_act,
                                                             unsigned *det) {
    const int slot = blockIdx.z;
    const int nv = N / 4;
    const int lanes = nv < kGbLanes ? nv : kGbLanes;          // float4 columns of this workgroup's chunk
    const int rows_per_pass = 256 / lanes;
    const int c = blockIdx.y * kGbLanes + threadIdx.x % lanes, rr = threadIdx.x / lanes;
    const size_t base = (size_t)slot * M * N;
    const long r0 = (long)blockIdx.x * kGbRows;
    double s[4] = {0.0, 0.0, 0.0, 0.0};   // the bias gradient sums 10^4..10^5 rows: fp64 in-thread, fp32 only across workgroups
    if (rr < rows_per_pass && c < nv) {
        for (long r = r0 + rr; r < r0 + kGbRows && r < M; r += rows_per_pass) {
            const size_t i = (base + (size_t)r * N) / 4 + c;
            float4 gv = reinterpret_cast<const float4 *>(g)[i];
            if (has_act) {
                const float4 p = reinterpret_cast<const float4 *>(pre)[i];
                gv.x *= gelu_grad_f(p.x); gv.y *= gelu_grad_f(p.y); gv.z *= gelu_grad_f(p.z); gv.w *= gelu_grad_f(p.w);
            }
            s[0] += gv.x; s[1] += gv.y; s[2] += gv.z; s[3] += gv.w;
            if (gx) {   // fp32 gradient for the plain-fp32 GEMM layers (no split operands)
                reinterpret_cast<float4 *>(gx)[i] = gv;
                continue;
            }
            const float e[4] = {gv.x, gv.y, gv.z, gv.w};
            union { uint2 raw; bf16_t b[4]; } H, L;
#pragma unroll
            for (int q = 0; q < 4; ++q) split1(e[q], H.b[q], L.b[q]);
            if (hi) {
                reinterpret_cast<uint2 *>(hi)[i] = H.raw;
                reinterpret_cast<uint2 *>(lo)[i] = L.raw;
            }
            if (cat3) {   // [hi | lo | hi] rows of width 3N: the A operand of the concatenated-contraction column-gradient GEMM
                uint2 *row = reinterpret_cast<uint2 *>(cat3 + ((size_t)slot * M + r) * 3 * N);
                row[c] = H.raw; row[nv + c] = L.raw; row[2 * nv + c] = H.raw;
            }
        }
    }
    if (db) {   // fold the row groups of the workgroup in LDS, then ONE atomic per column and workgroup
        __shared__ float red[256][4];
        red[threadIdx.x][0] = (float)s[0]; red[threadIdx.x][1] = (float)s[1]; red[threadIdx.x][2] = (float)s[2]; red[threadIdx.x][3] = (float)s[3];
        __syncthreads();
        det_enter(det);                      // deterministic mode: the workgroups' atomics in workgroup order (common.h)
        if (rr == 0 && c < nv) {
            float t[4] = {0.f, 0.f, 0.f, 0.f};
            for (int q = 0; q < rows_per_pass; ++q) {
#pragma unroll
                for (int e = 0; e < 4; ++e) t[e] += red[q * lanes + threadIdx.x][e];
            }
            float *d = db + (size_t)slot * N + c * 4;
            atomicAdd(d + 0, t[0]); atomicAdd(d + 1, t[1]); atomicAdd(d + 2, t[2]); atomicAdd(d + 3, t[3]);
        }
        det_leave(det);
    }
}

}  // namespace
}  // namespace vmasr

namespace vmasr {
namespace {
// out[i, e] = sum_{p < P} sum_{s < S} parts[((p * n + i) * S + s) * NK + e]: the three products x S contraction slabs of a
// weight-gradient GEMM triple (discriminator.py: _dw3) in one streaming pass (ATen's strided two-axis reduction runs
// this shape at 2.3 TB/s).  grid (blocks, n); NK % 4 == 0.
__global__ __launch_bounds__(256) void sum_parts_kernel(const float *__restrict__ parts, float *__restrict__ out, const int P, const int n,
                                                        const int S, const size_t NK4) {
    const int i = blockIdx.y;
    float4 *o = reinterpret_cast<float4 *>(out) + (size_t)i * NK4;
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < NK4; e += (size_t)gridDim.x * 256) {
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int p = 0; p < P; ++p)
            for (int s = 0; s < S; ++s) {
                const float4 v = reinterpret_cast<const float4 *>(parts)[(((size_t)p * n + i) * S + s) * NK4 + e];
                acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
            }
        o[e] = acc;
    }
}
}  // namespace
}  // namespace vmasr

VMASR_EXPORT int vmasr_sum_parts(const float *parts, float *out, int32_t P, int32_t n, int32_t S, int64_t NK, vmasr_stream_t stream) {
    using namespace vmasr;
    VMASR_REQUIRE(parts && out, VMASR_EINVAL, "sum_parts: null tensor");
    VMASR_REQUIRE(P > 0 && n > 0 && n <= 65535 && S > 0 && NK > 0 && NK % 4 == 0 && aligned_to(parts, 16) && aligned_to(out, 16), VMASR_EINVAL,
                  "sum_parts: bad shape / alignment (NK must be a multiple of 4)");
    const size_t NK4 = (size_t)NK / 4;
    const int blocks = (int)std::min<size_t>((NK4 + 255) / 256, 256 * 8);
    VMASR_LAUNCH(VMASR_K_SPLIT_BF16, 4.0 * ((double)P * S + 1.0) * n * (double)NK, sum_parts_kernel, dim3(blocks, n), dim3(256), 0,
                 static_cast<hipStream_t>(stream), parts, out, P, n, S, NK4);
    return check_launch("sum_parts");
}

VMASR_EXPORT int vmasr_weight_prep_split(const float *w, void *out, int32_t n, int32_t N, int32_t K, vmasr_stream_t stream) {
    VMASR_REQUIRE(w && out, VMASR_EINVAL, "weight_prep_split: null tensor");
    VMASR_REQUIRE(n > 0 && n <= 65535 && N > 0 && K > 0, VMASR_EINVAL, "weight_prep_split: bad shape (n=%d N=%d K=%d)", n, N, K);
    const dim3 grid((K + 63) / 64, (N + 63) / 64, n);
    VMASR_REQUIRE(grid.y <= 65535, VMASR_EINVAL, "weight_prep_split: N too large");
    hipStream_t st = static_cast<hipStream_t>(stream);
    VMASR_LAUNCH(VMASR_K_SPLIT_BF16, 10.0 * n * (double)N * K, weight_prep_kernel, grid, dim3(256), 0, st, w, static_cast<bf16_t *>(out), N, K);
    return check_launch("weight_prep_split");
}

VMASR_EXPORT int vmasr_bias_gelu_fwd(float *acc, const float *bias, float *act, int32_t slots, int64_t M, int32_t N, int32_t parts,
                                     vmasr_stream_t stream) {
    VMASR_REQUIRE(acc && bias && act, VMASR_EINVAL, "bias_gelu_fwd: null tensor");
    VMASR_REQUIRE(slots > 0 && slots <= 65535 && M > 0 && N > 0 && N % 4 == 0 && parts >= 1 && parts <= 8, VMASR_EINVAL,
                  "bias_gelu_fwd: bad shape");
    VMASR_REQUIRE(aligned_to(acc, 16) && aligned_to(bias, 16) && aligned_to(act, 16), VMASR_EINVAL, "bias_gelu_fwd: unaligned");
    const long total = M * (N / 4);
    const int blocks = (int)std::min<long>((total + 255) / 256, 256L * 16);
    hipStream_t st = static_cast<hipStream_t>(stream);
    VMASR_LAUNCH(VMASR_K_BIAS_GELU, (8.0 + 4.0 * parts) * slots * (double)M * N, bias_gelu_fwd_kernel, dim3(blocks, slots), dim3(256), 0,
                 st, acc, bias, act, (long)M, N, parts, (size_t)slots * M * N);
    return check_launch("bias_gelu_fwd");
}

VMASR_EXPORT int vmasr_gelu_bwd_split(const float *pre, const float *g, void *hi, void *lo, void *cat3, float *db, int32_t slots,
                                      int64_t M, int32_t N, vmasr_stream_t stream) {
    VMASR_REQUIRE(g && ((hi && lo) || (!hi && !lo && cat3)), VMASR_EINVAL,
                  "gelu_bwd_split: null tensor (g, and hi + lo or cat3 alone, are required)");
    VMASR_REQUIRE(slots > 0 && slots <= 65535 && M > 0 && N > 0 && N % 4 == 0 && N <= 1024, VMASR_EINVAL,
                  "gelu_bwd_split: bad shape (N must be a multiple of 4, <= 1024)");
    VMASR_REQUIRE(aligned_to(g, 16) && (!pre || aligned_to(pre, 16)) && (!hi || (aligned_to(hi, 8) && aligned_to(lo, 8))) && (!db || aligned_to(db, 16)),
                  VMASR_EINVAL, "gelu_bwd_split: unaligned");
    VMASR_REQUIRE(!cat3 || aligned_to(cat3, 8), VMASR_EINVAL, "gelu_bwd_split: unaligned cat3");
    const int blocks = (int)((M + kGbRows - 1) / kGbRows);
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int chunks = (N / 4 + kGbLanes - 1) / kGbLanes;
    const double per_elem = (pre ? 8.0 : 4.0) + (hi ? 4.0 : 0.0) + (cat3 ? 6.0 : 0.0);
    VMASR_LAUNCH(VMASR_K_BIAS_GELU, per_elem * slots * (double)M * N, gelu_bwd_split_kernel, dim3(blocks, chunks, slots), dim3(256), 0,
                 st, pre, g, static_cast<bf16_t *>(hi), static_cast<bf16_t *>(lo), static_cast<bf16_t *>(cat3), static_cast<float *>(nullptr), db, (long)M, N,
                 pre ? 1 : 0, det_ticket(VMASR_K_BIAS_GELU));
    return check_launch("gelu_bwd_split");
}

VMASR_EXPORT int vmasr_gelu_bwd(const float *pre, const float *g, float *gx, float *db, int32_t slots, int64_t M, int32_t N,
                                vmasr_stream_t stream) {
    VMASR_REQUIRE(pre && g && gx, VMASR_EINVAL, "gelu_bwd: null tensor");
    VMASR_REQUIRE(slots > 0 && slots <= 65535 && M > 0 && N > 0 && N % 4 == 0 && N <= 1024, VMASR_EINVAL,
                  "gelu_bwd: bad shape (N must be a multiple of 4, <= 1024)");
    VMASR_REQUIRE(aligned_to(g, 16) && aligned_to(pre, 16) && aligned_to(gx, 16) && (!db || aligned_to(db, 16)), VMASR_EINVAL,
                  "gelu_bwd: unaligned");
    const int blocks = (int)((M + kGbRows - 1) / kGbRows);
    const int chunks = (N / 4 + kGbLanes - 1) / kGbLanes;
    hipStream_t st = static_cast<hipStream_t>(stream);
    VMASR_LAUNCH(VMASR_K_BIAS_GELU, 12.0 * slots * (double)M * N, gelu_bwd_split_kernel, dim3(blocks, chunks, slots), dim3(256), 0, st, pre, g,
                 static_cast<bf16_t *>(nullptr), static_cast<bf16_t *>(nullptr), static_cast<bf16_t *>(nullptr), gx, db, (long)M, N, 1, det_ticket(VMASR_K_BIAS_GELU));
    return check_launch("gelu_bwd");
}
