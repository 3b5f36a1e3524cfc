// convpost.hip — the period discriminator's last convolution (1024 -> 1 channels, kernel (3,1)) on the stacked feature maps.
//
// Reference: model/discriminator.py:45,106-109 — conv_post = Conv2d(1024, 1, (3, 1), 1, padding=(1, 0)) after the five
// strided convolutions; its flattened output is the discriminator's score.  As im2col + GEMM (the form of the other layers)
// a ONE-output-channel convolution is all data movement: the (rows, 3*1024) column operand (377 MB per pass at batch 4) is
// written, then read by a GEMV, its gradient written and scattered back — 1.45 ms per training step for 0.4 GFLOP.
// Here the convolution runs directly on the previous layer's stacked output x (n, rows, C) (slot s: M_s = N_s * H_s valid
// rows = N_s sequences of H_s positions, zero padding at the sequence ends, rows beyond M_s are padding):
//   fwd : y[s, r] = b_s + sum_j sum_c x[s, r + j - 1, c] w[s, j, c]       one pass over x  (r 4C B per row, w 4 B)
//   bwd : dx[s, q, c] = sum_j gy[s, q - j + 1] w[s, j, c]                  one pass: reads x (for dw) and gy, writes dx
//         dw[s, j, c] += sum_r gy[s, r] x[s, r + j - 1, c],  db[s] += sum_r gy[s, r]
// A wave owns a run of consecutive rows; lane l holds channels {4l..4l+3} + 256 i of the three taps in registers
// (C = 256 VPL, VPL <= 4), walks its rows (8 .. 32 of them) once with three rotating per-lane partial sums (the row just read completes the
// output one row up) and reduces each finished output over the wave with DPP.  HBM-bound streaming, no LDS.
#include <algorithm>

#include "common.h"

namespace vmasr {
namespace {

constexpr int kCpSlots = 8;
// rows per wave (a launch argument): the maps are only ~10^4 rows per slot, so long runs leave the chip with < 6 waves per CU
// and one row load in flight per wave — measured 2.0 TB/s at 32 / 64 rows per wave.  Short runs re-read two neighbour rows per
// run (L2 hits); the weight-gradient variant folds 3 C sums per workgroup into atomics, so it keeps longer runs.
constexpr int kCpRunFwd = 8;
#ifndef VMASR_CP_RUN_DW
#define VMASR_CP_RUN_DW 32
#endif
constexpr int kCpRunBwdDx = 8, kCpRunBwdDw = VMASR_CP_RUN_DW;

struct CpSlots {
    long M[kCpSlots];
    int H[kCpSlots];
};

template <int VPL>
__device__ __forceinline__ void load_row(const float *__restrict__ p, int lane, float4 (&v)[VPL]) {
#pragma unroll
    for (int i = 0; i < VPL; ++i) v[i] = reinterpret_cast<const float4 *>(p)[i * 64 + lane];
}

__device__ __forceinline__ float dot4(const float4 a, const float4 b) { return (a.x * b.x + a.y * b.y) + (a.z * b.z + a.w * b.w); }

// grid (ceil(rows / (4 * RUN)), n), 4 waves per block
template <int VPL>
__global__ __launch_bounds__(256) void conv_post_fwd_kernel(const float *__restrict__ x, const float *__restrict__ w, const float *__restrict__ b,
                                                            float *__restrict__ y, const CpSlots t, const long rows, const int run) {
    constexpr int C = VPL * 256;
    const int s = blockIdx.y, lane = threadIdx.x & 63;
    const long r0 = ((long)blockIdx.x * 4 + (threadIdx.x >> 6)) * run;
    if (r0 >= rows) return;
    const long M = t.M[s], r1 = std::min(rows, r0 + run);
    const int H = t.H[s];
    const float *xs = x + (size_t)s * rows * C;
    float *ys = y + (size_t)s * rows;
    float4 w0[VPL], w1[VPL], w2[VPL];
    load_row<VPL>(w + ((size_t)s * 3 + 0) * C, lane, w0);
    load_row<VPL>(w + ((size_t)s * 3 + 1) * C, lane, w1);
    load_row<VPL>(w + ((size_t)s * 3 + 2) * C, lane, w2);
    const float bias = b[s];
    // rows q = r0-1 .. r1: x[q] adds tap 2 to output q-1, tap 1 to output q, tap 0 to output q+1 (same sequence only)
    float a_prev = 0.f, a_cur = 0.f, a_next = 0.f;      // per-lane partials of outputs q-1, q, q+1
    for (long q = r0 - 1; q <= r1; ++q) {
        if (q >= 0 && q < M) {
            float4 v[VPL];
            load_row<VPL>(xs + (size_t)q * C, lane, v);
            const int h = (int)(q % H);
            float p2 = 0.f, p1 = 0.f, p0 = 0.f;
#pragma unroll
            for (int i = 0; i < VPL; ++i) { p2 += dot4(v[i], w2[i]); p1 += dot4(v[i], w1[i]); p0 += dot4(v[i], w0[i]); }
            if (h > 0) a_prev += p2;          // output q-1 exists in this sequence
            a_cur += p1;
            if (h < H - 1) a_next += p0;      // output q+1 exists in this sequence
        }
        const long r = q - 1;                 // finished output
        if (r >= r0 && r < r1) {
            const float tot = wave_sum(a_prev);
            if (lane == 0) ys[r] = r < M ? tot + bias : 0.f;
        }
        a_prev = a_cur; a_cur = a_next; a_next = 0.f;
    }
}

// grid (ceil(rows / (4 * RUN)), n).  dx, dw, db may each be NULL.
template <int VPL>
__global__ __launch_bounds__(256) void conv_post_bwd_kernel(const float *__restrict__ x, const float *__restrict__ w, const float *__restrict__ gy,
                                                            float *__restrict__ dx, float *__restrict__ dw, float *__restrict__ db,
                                                            const CpSlots t, const long rows, const int run, unsigned *det) {
    constexpr int C = VPL * 256;
    const int s = blockIdx.y, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long r0 = ((long)blockIdx.x * 4 + wave) * run;
    const long M = t.M[s], r1 = std::min(rows, r0 + run);
    const int H = t.H[s];
    const float *xs = x + (size_t)s * rows * C, *gs = gy + (size_t)s * rows;
    float4 w0[VPL], w1[VPL], w2[VPL], d0[VPL], d1[VPL], d2[VPL];
    load_row<VPL>(w + ((size_t)s * 3 + 0) * C, lane, w0);
    load_row<VPL>(w + ((size_t)s * 3 + 1) * C, lane, w1);
    load_row<VPL>(w + ((size_t)s * 3 + 2) * C, lane, w2);
#pragma unroll
    for (int i = 0; i < VPL; ++i) d0[i] = d1[i] = d2[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    float bsum = 0.f;
    for (long q = r0; q < r1; ++q) {
        float4 o[VPL];
#pragma unroll
        for (int i = 0; i < VPL; ++i) o[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (q < M) {
            const int h = (int)(q % H);
            // outputs that read x[q]: r = q+1 (tap 0), q (tap 1), q-1 (tap 2), within the sequence
            const float g0 = h < H - 1 ? gs[q + 1] : 0.f, g1 = gs[q], g2 = h > 0 ? gs[q - 1] : 0.f;
            bsum += g1;
            if (dx) {
#pragma unroll
                for (int i = 0; i < VPL; ++i) {
                    o[i].x = g0 * w0[i].x + g1 * w1[i].x + g2 * w2[i].x; o[i].y = g0 * w0[i].y + g1 * w1[i].y + g2 * w2[i].y;
                    o[i].z = g0 * w0[i].z + g1 * w1[i].z + g2 * w2[i].z; o[i].w = g0 * w0[i].w + g1 * w1[i].w + g2 * w2[i].w;
                }
            }
            if (dw) {
                float4 v[VPL];
                load_row<VPL>(xs + (size_t)q * C, lane, v);
#pragma unroll
                for (int i = 0; i < VPL; ++i) {
                    d0[i].x = fmaf(g0, v[i].x, d0[i].x); d0[i].y = fmaf(g0, v[i].y, d0[i].y); d0[i].z = fmaf(g0, v[i].z, d0[i].z); d0[i].w = fmaf(g0, v[i].w, d0[i].w);
                    d1[i].x = fmaf(g1, v[i].x, d1[i].x); d1[i].y = fmaf(g1, v[i].y, d1[i].y); d1[i].z = fmaf(g1, v[i].z, d1[i].z); d1[i].w = fmaf(g1, v[i].w, d1[i].w);
                    d2[i].x = fmaf(g2, v[i].x, d2[i].x); d2[i].y = fmaf(g2, v[i].y, d2[i].y); d2[i].z = fmaf(g2, v[i].z, d2[i].z); d2[i].w = fmaf(g2, v[i].w, d2[i].w);
                }
            }
        }
        if (dx) {
            float *dst = dx + ((size_t)s * rows + q) * C;
#pragma unroll
            for (int i = 0; i < VPL; ++i) reinterpret_cast<float4 *>(dst)[i * 64 + lane] = o[i];
        }
    }
    // fold the block's four waves in LDS, then one atomic per weight and workgroup
    det_enter(det);                      // deterministic mode: the workgroups' atomics in workgroup order (common.h)
    if (dw) {
        __shared__ float4 fold[4][3 * VPL][64];
#pragma unroll
        for (int i = 0; i < VPL; ++i) { fold[wave][i][lane] = d0[i]; fold[wave][VPL + i][lane] = d1[i]; fold[wave][2 * VPL + i][lane] = d2[i]; }
        __syncthreads();
        if (wave == 0) {
#pragma unroll
            for (int e = 0; e < 3 * VPL; ++e) {
                const float4 a = fold[0][e][lane], bb = fold[1][e][lane], c = fold[2][e][lane], d = fold[3][e][lane];
                const int j = e / VPL, i = e % VPL;
                float *dst = dw + ((size_t)s * 3 + j) * C + (i * 64 + lane) * 4;
                atomicAdd(dst + 0, (a.x + bb.x) + (c.x + d.x)); atomicAdd(dst + 1, (a.y + bb.y) + (c.y + d.y));
                atomicAdd(dst + 2, (a.z + bb.z) + (c.z + d.z)); atomicAdd(dst + 3, (a.w + bb.w) + (c.w + d.w));
            }
        }
    }
    if (det == nullptr) {
        if (db && lane == 0 && bsum != 0.f) atomicAdd(db + s, bsum);   // gy is wave-uniform: every lane holds the run's sum
    } else if (db) {                     // deterministic mode: the four waves' sums in a fixed order, one atomic per workgroup
        __shared__ float bs[4];
        if (lane == 0) bs[wave] = bsum;
        __syncthreads();
        if (threadIdx.x == 0) atomicAdd(db + s, (bs[0] + bs[1]) + (bs[2] + bs[3]));
    }
    det_leave(det);
}

int cp_check(const int64_t *Ms, const int32_t *Hs, int n, int64_t rows, int C, int k, CpSlots &t, const char *what) {
    VMASR_REQUIRE(Ms && Hs, VMASR_EINVAL, "%s: null argument", what);
    VMASR_REQUIRE(n > 0 && n <= kCpSlots && rows > 0 && k == 3 && C % 256 == 0 && C >= 256 && C <= 1024, VMASR_EINVAL,
                  "%s: needs 1..%d slots, kernel 3 and C in {256, 512, 768, 1024} (n=%d k=%d C=%d)", what, kCpSlots, n, k, C);
    for (int s = 0; s < n; ++s) {
        VMASR_REQUIRE(Hs[s] > 0 && Ms[s] >= 0 && Ms[s] <= rows && Ms[s] % Hs[s] == 0, VMASR_EINVAL,
                      "%s: slot %d: %ld valid rows of %ld must be whole sequences of %d", what, s, (long)Ms[s], (long)rows, Hs[s]);
        t.M[s] = Ms[s];
        t.H[s] = Hs[s];
    }
    return 0;
}

}  // namespace
}  // namespace vmasr

using namespace vmasr;

VMASR_EXPORT int vmasr_conv_post_supported(int32_t C, int32_t k) { return (k == 3 && C % 256 == 0 && C >= 256 && C <= 1024) ? 1 : 0; }

VMASR_EXPORT int vmasr_conv_post_fwd(const float *x, const float *w, const float *b, float *y, const int64_t *Ms, const int32_t *Hs, int32_t n,
                                     int64_t rows, int32_t C, int32_t k, vmasr_stream_t stream) {
    VMASR_REQUIRE(x && w && b && y, VMASR_EINVAL, "conv_post_fwd: null tensor");
    CpSlots t{};
    if (int e = cp_check(Ms, Hs, n, rows, C, k, t, "conv_post_fwd")) return e;
    VMASR_REQUIRE(aligned_to(x, 16) && aligned_to(w, 16), VMASR_EINVAL, "conv_post_fwd: unaligned");
    const dim3 grid((unsigned)((rows + 4 * kCpRunFwd - 1) / (4 * kCpRunFwd)), n);
    hipStream_t st = static_cast<hipStream_t>(stream);
    const double bytes = (double)n * rows * (C + 1) * 4.0;
    switch (C / 256) {
        case 1: VMASR_LAUNCH(VMASR_K_CONV_POST, bytes, conv_post_fwd_kernel<1>, grid, dim3(256), 0, st, x, w, b, y, t, (long)rows, kCpRunFwd); break;
        case 2: VMASR_LAUNCH(VMASR_K_CONV_POST, bytes, conv_post_fwd_kernel<2>, grid, dim3(256), 0, st, x, w, b, y, t, (long)rows, kCpRunFwd); break;
        case 3: VMASR_LAUNCH(VMASR_K_CONV_POST, bytes, conv_post_fwd_kernel<3>, grid, dim3(256), 0, st, x, w, b, y, t, (long)rows, kCpRunFwd); break;
        default: VMASR_LAUNCH(VMASR_K_CONV_POST, bytes, conv_post_fwd_kernel<4>, grid, dim3(256), 0, st, x, w, b, y, t, (long)rows, kCpRunFwd); break;
    }
    return check_launch("conv_post_fwd");
}

VMASR_EXPORT int vmasr_conv_post_bwd(const float *x, const float *w, const float *gy, float *dx, float *dw, float *db, const int64_t *Ms,
                                     const int32_t *Hs, int32_t n, int64_t rows, int32_t C, int32_t k, vmasr_stream_t stream) {
    VMASR_REQUIRE(x && w && gy, VMASR_EINVAL, "conv_post_bwd: null tensor");
    CpSlots t{};
    if (int e = cp_check(Ms, Hs, n, rows, C, k, t, "conv_post_bwd")) return e;
    VMASR_REQUIRE(aligned_to(x, 16) && aligned_to(w, 16) && (!dx || aligned_to(dx, 16)) && (!dw || aligned_to(dw, 16)), VMASR_EINVAL,
                  "conv_post_bwd: unaligned");
    if (!dx && !dw && !db) return VMASR_OK;
    const int run = dw ? kCpRunBwdDw : kCpRunBwdDx;
    const dim3 grid((unsigned)((rows + 4 * run - 1) / (4 * run)), n);
    hipStream_t st = static_cast<hipStream_t>(stream);
    const double bytes = (double)n * rows * ((dw ? C : 0) + (dx ? C : 0) + 1.0) * 4.0;
    switch (C / 256) {
        case 1: VMASR_LAUNCH(VMASR_K_CONV_POST, bytes, conv_post_bwd_kernel<1>, grid, dim3(256), 0, st, x, w, gy, dx, dw, db, t, (long)rows, run, det_ticket(VMASR_K_CONV_POST)); break;
        case 2: VMASR_LAUNCH(VMASR_K_CONV_POST, bytes, conv_post_bwd_kernel<2>, grid, dim3(256), 0, st, x, w, gy, dx, dw, db, t, (long)rows, run, det_ticket(VMASR_K_CONV_POST)); break;
        case 3: VMASR_LAUNCH(VMASR_K_CONV_POST, bytes, conv_post_bwd_kernel<3>, grid, dim3(256), 0, st, x, w, gy, dx, dw, db, t, (long)rows, run, det_ticket(VMASR_K_CONV_POST)); break;
        default: VMASR_LAUNCH(VMASR_K_CONV_POST, bytes, conv_post_bwd_kernel<4>, grid, dim3(256), 0, st, x, w, gy, dx, dw, db, t, (long)rows, run, det_ticket(VMASR_K_CONV_POST)); break;
    }
    return check_launch("conv_post_bwd");
}
