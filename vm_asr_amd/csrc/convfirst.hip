// convfirst.hip — the period discriminator's FIRST convolution (1 -> 32 channels, kernel (5,1), stride (3,1)) + GELU, stacked.
//
// Reference: model/discriminator.py:26-40,100-104 — the signal folded to (T/p, p) enters Conv2d(1, 32, (5,1), (3,1),
// padding (2,0)) followed by GELU.  As im2col + GEMM the contraction is 5 long: the GEMM kernels hipBLASLt has for
// (rows x 5)(5 x 32) run at a few % of anything, and the (rows, 32) pre-activation makes two extra round trips
// (GEMM out -> epilogue).  Here, for all n period discriminators in one launch:
//   fwd : pre[s, r, o] = b[s, o] + sum_j x_s[seq, 3 h + j - 2] w[s, o, j],  act = GELU(pre)       (r = seq * H1 + h)
//   bwd : gx = g * GELU'(pre);  db[s, o] += sum_r gx;  dw[s, o, j] += sum_r gx[r, o] x_s[seq, 3 h + j - 2];
//         dcols[s, r, j] = sum_o gx[r, o] w[s, o, j]   (the (rows, 5) column gradient, scattered to the signal by col2im)
// Thread = (row, group of 4 channels): the 8 threads of a row are adjacent lanes, so every store instruction of a wave
// covers 8 rows x 128 B = 1 KB contiguous.  Writes dominate: 2 x 4 B per (row, channel) forward.
#include <algorithm>

#include "common.h"

namespace vmasr {
namespace {

constexpr int kCfSlots = 8;
constexpr int kCfN = 32;      // output channels
constexpr int kCfK = 5;       // taps
constexpr int kCfStride = 3, kCfPad = 2;

struct CfSlots {
    const float *x[kCfSlots];   // (N_s, H_s) fp32
    long M[kCfSlots];           // valid output rows N_s * H1_s
    int H[kCfSlots], H1[kCfSlots];
};

__device__ __forceinline__ float cf_gelu(float v) { return 0.5f * v * (1.f + erff(v * 0.70710678118654752f)); }
__device__ __forceinline__ float cf_gelu_grad(float v) {
    return 0.5f * (1.f + erff(v * 0.70710678118654752f)) + v * 0.3989422804014327f * __expf(-0.5f * v * v);
}

__device__ __forceinline__ void cf_taps(const CfSlots &t, int s, long r, float (&xv)[kCfK]) {
    const int H1 = t.H1[s], H = t.H[s];
    const long seq = r / H1;
    const int h = (int)(r % H1);
    const float *xs = t.x[s] + seq * H;
#pragma unroll
    for (int j = 0; j < kCfK; ++j) {
        const int p = h * kCfStride + j - kCfPad;
        xv[j] = (p >= 0 && p < H) ? xs[p] : 0.f;
    }
}

// grid (blocks, n); w (n, 32, 5), b (n, 32); pre / act (n, rows, 32)
__global__ __launch_bounds__(256) void conv_first_fwd_kernel(const CfSlots t, const float *__restrict__ w, const float *__restrict__ b,
                                                             float *__restrict__ pre, float *__restrict__ act, const long rows) {
    const int s = blockIdx.y, cg = threadIdx.x & 7;
    float wv[4][kCfK], bv[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        bv[c] = b[s * kCfN + cg * 4 + c];
#pragma unroll
        for (int j = 0; j < kCfK; ++j) wv[c][j] = w[((size_t)s * kCfN + cg * 4 + c) * kCfK + j];
    }
    const long M = t.M[s];
    for (long r = (long)blockIdx.x * 32 + (threadIdx.x >> 3); r < rows; r += (long)gridDim.x * 32) {
        float4 p = make_float4(0.f, 0.f, 0.f, 0.f), a = p;
        if (r < M) {
            float xv[kCfK];
            cf_taps(t, s, r, xv);
            float o[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                o[c] = bv[c];
#pragma unroll
                for (int j = 0; j < kCfK; ++j) o[c] = fmaf(xv[j], wv[c][j], o[c]);
            }
            p = make_float4(o[0], o[1], o[2], o[3]);
            a = make_float4(cf_gelu(o[0]), cf_gelu(o[1]), cf_gelu(o[2]), cf_gelu(o[3]));
        }
        const size_t off = (((size_t)s * rows + r) * kCfN) / 4 + cg;
        reinterpret_cast<float4 *>(pre)[off] = p;
        reinterpret_cast<float4 *>(act)[off] = a;
    }
}

// grid (blocks, n).  dcols (n, rows, 5) / dw (n, 32, 5) / db (n, 32) may be NULL; dw, db are accumulated (zero-initialised).
__global__ __launch_bounds__(256) void conv_first_bwd_kernel(const CfSlots t, const float *__restrict__ w, const float *__restrict__ pre,
                                                             const float *__restrict__ g, float *__restrict__ dcols, float *__restrict__ dw,
                                                             float *__restrict__ db, const long rows, unsigned *det) {
    const int s = blockIdx.y, cg = threadIdx.x & 7, lane = threadIdx.x & 63;
    float wv[4][kCfK], aw[4][kCfK], ab[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        ab[c] = 0.f;
#pragma unroll
        for (int j = 0; j < kCfK; ++j) { wv[c][j] = w[((size_t)s * kCfN + cg * 4 + c) * kCfK + j]; aw[c][j] = 0.f; }
    }
    const long M = t.M[s];
    for (long r = (long)blockIdx.x * 32 + (threadIdx.x >> 3); r - (threadIdx.x >> 3) < rows; r += (long)gridDim.x * 32) {
        float sj[kCfK] = {0.f, 0.f, 0.f, 0.f, 0.f};
        if (r < M) {
            const size_t off = (((size_t)s * rows + r) * kCfN) / 4 + cg;
            const float4 pv = reinterpret_cast<const float4 *>(pre)[off], gv = reinterpret_cast<const float4 *>(g)[off];
            const float gx[4] = {gv.x * cf_gelu_grad(pv.x), gv.y * cf_gelu_grad(pv.y), gv.z * cf_gelu_grad(pv.z), gv.w * cf_gelu_grad(pv.w)};
            float xv[kCfK];
            cf_taps(t, s, r, xv);
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                ab[c] += gx[c];
#pragma unroll
                for (int j = 0; j < kCfK; ++j) {
                    aw[c][j] = fmaf(gx[c], xv[j], aw[c][j]);
                    sj[j] = fmaf(gx[c], wv[c][j], sj[j]);
                }
            }
        }
        if (dcols) {   // sum over the 8 channel groups of the row (adjacent lanes): xor 1, 2, 4
#pragma unroll
            for (int j = 0; j < kCfK; ++j) {
                float v = sj[j];
                v += xor1(v); v += xor2(v); v += xor4(v);
                sj[j] = v;
            }
            if (cg == 0 && r < rows) {
                float *dst = dcols + ((size_t)s * rows + r) * kCfK;
#pragma unroll
                for (int j = 0; j < kCfK; ++j) dst[j] = sj[j];     // zeros on the padding rows
            }
        }
    }
    if (dw || db) {
        // threads with equal cg hold the same (channel, tap) slots: fold the 8 row positions of a wave (xor 8, 16, 32), then
        // the block's 4 waves through LDS, then one atomic per value and workgroup
        __shared__ float fold[4][8][24];
        float vals[24];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
#pragma unroll
            for (int j = 0; j < kCfK; ++j) vals[c * kCfK + j] = aw[c][j];
            vals[20 + c] = ab[c];
        }
#pragma unroll
        for (int e = 0; e < 24; ++e) {
            float v = vals[e];
            v += xor8(v);
            v += __shfl_xor(v, 16);
            v += __shfl_xor(v, 32);
            vals[e] = v;
        }
        if (lane < 8) {
#pragma unroll
            for (int e = 0; e < 24; ++e) fold[threadIdx.x >> 6][lane][e] = vals[e];
        }
        __syncthreads();
        det_enter(det);
        if (threadIdx.x < 8 * 24) {
            const int g8 = threadIdx.x / 24, e = threadIdx.x % 24;
            const float v = (fold[0][g8][e] + fold[1][g8][e]) + (fold[2][g8][e] + fold[3][g8][e]);
            if (e < 20) { if (dw) atomicAdd(dw + ((size_t)s * kCfN + g8 * 4 + e / kCfK) * kCfK + e % kCfK, v); }
            else if (db) atomicAdd(db + s * kCfN + g8 * 4 + (e - 20), v);
        }
        det_leave(det);
    }
}

int cf_fill(CfSlots &t, const void *const *xs, const int64_t *Ns, const int32_t *Hs, int n, int64_t rows, const char *what) {
    VMASR_REQUIRE(xs && Ns && Hs, VMASR_EINVAL, "%s: null argument", what);
    VMASR_REQUIRE(n > 0 && n <= kCfSlots && rows > 0, VMASR_EINVAL, "%s: 1..%d slots (got %d)", what, kCfSlots, n);
    for (int s = 0; s < n; ++s) {
        VMASR_REQUIRE(xs[s] && Ns[s] > 0 && Hs[s] > 0 && Hs[s] + 2 * kCfPad >= kCfK, VMASR_EINVAL, "%s: slot %d bad geometry", what, s);
        const int H1 = (Hs[s] + 2 * kCfPad - kCfK) / kCfStride + 1;
        VMASR_REQUIRE(Ns[s] * H1 <= rows, VMASR_EINVAL, "%s: rows smaller than N*H1 of slot %d", what, s);
        t.x[s] = static_cast<const float *>(xs[s]);
        t.M[s] = Ns[s] * H1;
        t.H[s] = Hs[s];
        t.H1[s] = H1;
    }
    return 0;
}

}  // namespace
}  // namespace vmasr

using namespace vmasr;

VMASR_EXPORT int vmasr_conv_first_fwd(const void *const *xs, const int64_t *Ns, const int32_t *Hs, int32_t n, const float *w, const float *b,
                                      float *pre, float *act, int64_t rows, vmasr_stream_t stream) {
    VMASR_REQUIRE(w && b && pre && act && aligned_to(pre, 16) && aligned_to(act, 16), VMASR_EINVAL, "conv_first_fwd: null / unaligned tensor");
    CfSlots t{};
    if (int e = cf_fill(t, xs, Ns, Hs, n, rows, "conv_first_fwd")) return e;
    const int blocks = (int)std::min<long>((rows + 31) / 32, 256L * 16);
    VMASR_LAUNCH(VMASR_K_CONV_POST, 8.0 * n * (double)rows * kCfN, conv_first_fwd_kernel, dim3(blocks, n), dim3(256), 0,
                 static_cast<hipStream_t>(stream), t, w, b, pre, act, (long)rows);
    return check_launch("conv_first_fwd");
}

VMASR_EXPORT int vmasr_conv_first_bwd(const void *const *xs, const int64_t *Ns, const int32_t *Hs, int32_t n, const float *w, const float *pre,
                                      const float *g, float *dcols, float *dw, float *db, int64_t rows, vmasr_stream_t stream) {
    VMASR_REQUIRE(w && pre && g && aligned_to(pre, 16) && aligned_to(g, 16), VMASR_EINVAL, "conv_first_bwd: null / unaligned tensor");
    CfSlots t{};
    if (int e = cf_fill(t, xs, Ns, Hs, n, rows, "conv_first_bwd")) return e;
    if (!dcols && !dw && !db) return VMASR_OK;
    // every workgroup ends in 192 atomics on the SAME 160 + 32 addresses of its slot: they serialise (~13 ns each per address), so few,
    // long-running workgroups win — measured in the step at B = 4 (average launch of the conv_first / conv_post kernel family, 10 per
    // step, only this cap varied): 2048 workgroups per slot 82.5 us, 1024 -> 65.8, 512 -> 56.2, 384 -> 57.6, 256 -> 53.5, 128 -> 61.0
#ifndef VMASR_CF_BWD_BLOCKS
#define VMASR_CF_BWD_BLOCKS 256L
#endif
    const int blocks = (int)std::min<long>((rows + 31) / 32, VMASR_CF_BWD_BLOCKS);
    VMASR_LAUNCH(VMASR_K_CONV_POST, 8.0 * n * (double)rows * kCfN, conv_first_bwd_kernel, dim3(blocks, n), dim3(256), 0,
                 static_cast<hipStream_t>(stream), t, w, pre, g, dcols, dw, db, (long)rows, det_ticket(VMASR_K_CONV_POST));
    return check_launch("conv_first_bwd");
}
