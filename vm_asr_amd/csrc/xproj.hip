// xproj.hip — SS2D's per-direction input projections (x_proj and dt_proj) for gfx950.
//
// Replaces the two einsums of SS2D.forward_corev2 and the copies/casts around them
// (model/vmamba.py:1473-1491):
//     x_dbl = einsum('b k d l, k c d -> b k c l', xs, x_proj_weight)      c = R + 2N
//     dts, Bs, Cs = split(x_dbl, [R, N, N]);  dts = einsum('b k r l, k d r -> b k d l', dts, dt_projs_weight)
//     .contiguous() x3, .to(float) x4
// As batched GEMMs these are degenerate (inner dimensions D and R = 1..8, output width 3..10;
// the weight gradient of the 512x512 block is a (2 x 10^6) @ (10^6 x 3) product that hipBLASLt
// runs in 1.6 ms).  They are memory-bound maps over (b, k, l):
//
//   forward : one thread owns 4 consecutive positions l of one (b,k): it streams the D rows of xs
//             once, keeps the c accumulators in registers, writes Bs/Cs (and the low-rank dt rows,
//             kept for the backward) and expands dts = W_dt * dtr for the D rows.
//             reads K D L, writes K D L (+ small)
//   backward A (per position): d_dtr = W_dt^T ddts, dx_dbl = [d_dtr, dBs, dCs] (kept for B),
//             dxs = W_x^T dx_dbl (+ du, the scan's own gradient wrt xs: fused add)
//   backward B (per row d): dW_x[k,:,d] = sum_{b,l} dx_dbl * xs[d],  dW_dt[k,d,:] = sum ddts[d] * dtr
//             one wave per (4 rows, 4096-position chunk), wave-reduced sums leave as float atomics
// fp32 accumulation and fp32 weights; xs may be fp32 or 16-bit.  c <= 16 (d_state <= 4 at R <= 8).
#include "common.h"

#include <cstdlib>

namespace vmasr {
namespace {

constexpr int kMaxC = 16;  // R + 2N
constexpr int kMaxR = 8;

struct XpGeom {
    int B, K, D, N, R, L;
};

template <typename T, bool VEC>
__global__ __launch_bounds__(256) void xproj_fwd_kernel(const T *__restrict__ xs, const float *__restrict__ Wx,
                                                        const float *__restrict__ Wdt, float *__restrict__ dts,
                                                        float *__restrict__ Bs, float *__restrict__ Cs,
                                                        float *__restrict__ dtr, const XpGeom g) {
    extern __shared__ float s_w[];  // Wx[k]: C*D, then Wdt[k]: D*R
    const int C = g.R + 2 * g.N;
    const int k = blockIdx.y, b = blockIdx.z;
    float *s_wx = s_w, *s_wdt = s_w + C * g.D;
    for (int i = threadIdx.x; i < C * g.D; i += blockDim.x) s_wx[i] = Wx[(size_t)k * C * g.D + i];
    for (int i = threadIdx.x; i < g.D * g.R; i += blockDim.x) s_wdt[i] = Wdt[(size_t)k * g.D * g.R + i];
    __syncthreads();
    const int l0 = (blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (l0 >= g.L) return;
    const T *xp = xs + ((size_t)b * g.K + k) * g.D * g.L;
    float acc[kMaxC][4];
#pragma unroll
    for (int c = 0; c < kMaxC; ++c) acc[c][0] = acc[c][1] = acc[c][2] = acc[c][3] = 0.f;
    for (int d = 0; d < g.D; ++d) {
        float v[4];
        load4<T, VEC>(xp + (size_t)d * g.L, l0, g.L, v);
#pragma unroll
        for (int c = 0; c < kMaxC; ++c)
            if (c < C) {
                const float w = s_wx[c * g.D + d];
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[c][i] = fmaf(w, v[i], acc[c][i]);
            }
    }
    // dt rows (kept for the backward), then B and C
#pragma unroll
    for (int c = 0; c < kMaxC; ++c)
        if (c < C) {
            float *dst;
            if (c < g.R) dst = dtr + (((size_t)b * g.K + k) * g.R + c) * g.L;
            else if (c < g.R + g.N) dst = Bs + (((size_t)b * g.K + k) * g.N + (c - g.R)) * g.L;
            else dst = Cs + (((size_t)b * g.K + k) * g.N + (c - g.R - g.N)) * g.L;
            store4<float, VEC>(dst, l0, g.L, acc[c]);
        }
    float *dp = dts + ((size_t)b * g.K + k) * g.D * g.L;
    for (int d = 0; d < g.D; ++d) {
        float o[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int r = 0; r < kMaxR; ++r)
            if (r < g.R) {
                const float w = s_wdt[d * g.R + r];
#pragma unroll
                for (int i = 0; i < 4; ++i) o[i] = fmaf(w, acc[r][i], o[i]);
            }
        store4<float, VEC>(dp + (size_t)d * g.L, l0, g.L, o);
    }
}

template <typename T, bool VEC>
__global__ __launch_bounds__(256) void xproj_bwd_a_kernel(const float *__restrict__ ddts, const float *__restrict__ dBs,
                                                          const float *__restrict__ dCs, const float *__restrict__ du,
                                                          const float *__restrict__ Wx, const float *__restrict__ Wdt,
                                                          T *__restrict__ dxs, float *__restrict__ dxdbl,
                                                          const XpGeom g) {
    extern __shared__ float s_w[];
    const int C = g.R + 2 * g.N;
    const int k = blockIdx.y, b = blockIdx.z;
    float *s_wx = s_w, *s_wdt = s_w + C * g.D;
    for (int i = threadIdx.x; i < C * g.D; i += blockDim.x) s_wx[i] = Wx[(size_t)k * C * g.D + i];
    for (int i = threadIdx.x; i < g.D * g.R; i += blockDim.x) s_wdt[i] = Wdt[(size_t)k * g.D * g.R + i];
    __syncthreads();
    const int l0 = (blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (l0 >= g.L) return;
    const size_t row0 = ((size_t)b * g.K + k) * g.D;
    float acc[kMaxC][4];
#pragma unroll
    for (int c = 0; c < kMaxC; ++c) acc[c][0] = acc[c][1] = acc[c][2] = acc[c][3] = 0.f;
    // d_dtr[r] = sum_d Wdt[d,r] * ddts[d]
    for (int d = 0; d < g.D; ++d) {
        float v[4];
        load4<float, VEC>(ddts + (row0 + d) * g.L, l0, g.L, v);
#pragma unroll
        for (int r = 0; r < kMaxR; ++r)
            if (r < g.R) {
                const float w = s_wdt[d * g.R + r];
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[r][i] = fmaf(w, v[i], acc[r][i]);
            }
    }
#pragma unroll
    for (int c = 0; c < kMaxC; ++c)
        if (c >= g.R && c < C) {
            const float *src = c < g.R + g.N ? dBs + (((size_t)b * g.K + k) * g.N + (c - g.R)) * g.L
                                             : dCs + (((size_t)b * g.K + k) * g.N + (c - g.R - g.N)) * g.L;
            load4<float, VEC>(src, l0, g.L, acc[c]);
        }
#pragma unroll
    for (int c = 0; c < kMaxC; ++c)
        if (c < C) store4<float, VEC>(dxdbl + (((size_t)b * g.K + k) * C + c) * g.L, l0, g.L, acc[c]);
    // dxs[d] = sum_c Wx[c,d] * dx_dbl[c] (+ du[d])
    for (int d = 0; d < g.D; ++d) {
        float o[4] = {0.f, 0.f, 0.f, 0.f};
        if (du) load4<float, VEC>(du + (row0 + d) * g.L, l0, g.L, o);
#pragma unroll
        for (int c = 0; c < kMaxC; ++c)
            if (c < C) {
                const float w = s_wx[c * g.D + d];
#pragma unroll
                for (int i = 0; i < 4; ++i) o[i] = fmaf(w, acc[c][i], o[i]);
            }
        store4<T, VEC>(dxs + (row0 + d) * g.L, l0, g.L, o);
    }
}

// ---- D-parallel variants for the deep stages (d_inner >= 64, a few thousand positions) ------------------------------
// The position-parallel kernels above walk the D rows serially per thread: at D = 128, L = 1024 that is 16 workgroups
// and 128 dependent loads per thread — 107 us forward / 119 us backward for 8 MB of traffic, independent of the batch.
// Here a workgroup owns 64 positions (16 quads) x 16 row groups: thread (q, dg) reduces rows dg, dg+16, ... of its quad,
// the four row groups of a wave fold by DPP-free lane shuffles (xor 16, 32), the four waves through LDS; the expansion
// back over D (dts = W_dt dtr, dxs = W_x^T dx_dbl) is again split by row group.  L/64 x K x B workgroups.
constexpr int kDpTL = 64;
constexpr int kDparMinD = 64;

template <int NACC>
__device__ __forceinline__ void dpar_reduce(float (&acc)[NACC][4], int n, float *red, int w, int q, bool writer) {
#pragma unroll
    for (int c = 0; c < NACC; ++c)
        if (c < n) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float v = acc[c][i];
                v += __shfl_xor(v, 16, 64);
                v += __shfl_xor(v, 32, 64);
                acc[c][i] = v;
            }
            if (writer) *reinterpret_cast<float4 *>(red + (c * 4 + w) * kDpTL + q * 4) = make_float4(acc[c][0], acc[c][1], acc[c][2], acc[c][3]);
        }
}

__device__ __forceinline__ void dpar_fold(const float *red, int c, int q, float (&o)[4]) {
    o[0] = o[1] = o[2] = o[3] = 0.f;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        const float4 t = *reinterpret_cast<const float4 *>(red + (c * 4 + w) * kDpTL + q * 4);
        o[0] += t.x; o[1] += t.y; o[2] += t.z; o[3] += t.w;
    }
}

template <typename T>
__global__ __launch_bounds__(256) void xproj_fwd_dpar_kernel(const T *__restrict__ xs, const float *__restrict__ Wx,
                                                             const float *__restrict__ Wdt, float *__restrict__ dts,
                                                             float *__restrict__ Bs, float *__restrict__ Cs,
                                                             float *__restrict__ dtr, const XpGeom g) {
    extern __shared__ float s_w[];  // Wx[k]: C*D, Wdt[k]: D*R, red: C*4*64
    const int C = g.R + 2 * g.N;
    const int k = blockIdx.y, b = blockIdx.z;
    float *s_wx = s_w, *s_wdt = s_w + C * g.D, *red = s_wdt + g.D * g.R;
    for (int i = threadIdx.x; i < C * g.D; i += blockDim.x) s_wx[i] = Wx[(size_t)k * C * g.D + i];
    for (int i = threadIdx.x; i < g.D * g.R; i += blockDim.x) s_wdt[i] = Wdt[(size_t)k * g.D * g.R + i];
    __syncthreads();
    const int q = threadIdx.x & 15, dg = threadIdx.x >> 4, w = threadIdx.x >> 6;
    const int l0 = blockIdx.x * kDpTL + q * 4;
    const bool in = l0 < g.L;                               // L % 4 == 0: a quad is all in or all out
    const size_t bk = (size_t)b * g.K + k;
    const T *xp = xs + bk * g.D * g.L;
    float acc[kMaxC][4];
#pragma unroll
    for (int c = 0; c < kMaxC; ++c) acc[c][0] = acc[c][1] = acc[c][2] = acc[c][3] = 0.f;
    if (in)
        for (int d = dg; d < g.D; d += 16) {
            float v[4];
            load4<T, true>(xp + (size_t)d * g.L, l0, g.L, v);
#pragma unroll
            for (int c = 0; c < kMaxC; ++c)
                if (c < C) {
                    const float wv = s_wx[c * g.D + d];
#pragma unroll
                    for (int i = 0; i < 4; ++i) acc[c][i] = fmaf(wv, v[i], acc[c][i]);
                }
        }
    dpar_reduce<kMaxC>(acc, C, red, w, q, (dg & 3) == 0);
    __syncthreads();
    if (!in) return;
    if (dg < C) {   // row dg of x_dbl: dt rows (kept for the backward), then B and C
        float o[4];
        dpar_fold(red, dg, q, o);
        float *dst;
        if (dg < g.R) dst = dtr + (bk * g.R + dg) * g.L;
        else if (dg < g.R + g.N) dst = Bs + (bk * g.N + (dg - g.R)) * g.L;
        else dst = Cs + (bk * g.N + (dg - g.R - g.N)) * g.L;
        store4<float, true>(dst, l0, g.L, o);
    }
    float xr[kMaxR][4];
#pragma unroll
    for (int r = 0; r < kMaxR; ++r)
        if (r < g.R) dpar_fold(red, r, q, xr[r]);
    float *dp = dts + bk * g.D * g.L;
    for (int d = dg; d < g.D; d += 16) {
        float o[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int r = 0; r < kMaxR; ++r)
            if (r < g.R) {
                const float wv = s_wdt[d * g.R + r];
#pragma unroll
                for (int i = 0; i < 4; ++i) o[i] = fmaf(wv, xr[r][i], o[i]);
            }
        store4<float, true>(dp + (size_t)d * g.L, l0, g.L, o);
    }
}

template <typename T>
__global__ __launch_bounds__(256) void xproj_bwd_a_dpar_kernel(const float *__restrict__ ddts, const float *__restrict__ dBs,
                                                               const float *__restrict__ dCs, const float *__restrict__ du,
                                                               const float *__restrict__ Wx, const float *__restrict__ Wdt,
                                                               T *__restrict__ dxs, float *__restrict__ dxdbl, const XpGeom g) {
    extern __shared__ float s_w[];
    const int C = g.R + 2 * g.N;
    const int k = blockIdx.y, b = blockIdx.z;
    float *s_wx = s_w, *s_wdt = s_w + C * g.D, *red = s_wdt + g.D * g.R;
    for (int i = threadIdx.x; i < C * g.D; i += blockDim.x) s_wx[i] = Wx[(size_t)k * C * g.D + i];
    for (int i = threadIdx.x; i < g.D * g.R; i += blockDim.x) s_wdt[i] = Wdt[(size_t)k * g.D * g.R + i];
    __syncthreads();
    const int q = threadIdx.x & 15, dg = threadIdx.x >> 4, w = threadIdx.x >> 6;
    const int l0 = blockIdx.x * kDpTL + q * 4;
    const bool in = l0 < g.L;
    const size_t bk = (size_t)b * g.K + k, row0 = bk * g.D;
    float ar[kMaxR][4];
#pragma unroll
    for (int r = 0; r < kMaxR; ++r) ar[r][0] = ar[r][1] = ar[r][2] = ar[r][3] = 0.f;
    // d_dtr[r] = sum_d Wdt[d,r] * ddts[d]
    if (in)
        for (int d = dg; d < g.D; d += 16) {
            float v[4];
            load4<float, true>(ddts + (row0 + d) * g.L, l0, g.L, v);
#pragma unroll
            for (int r = 0; r < kMaxR; ++r)
                if (r < g.R) {
                    const float wv = s_wdt[d * g.R + r];
#pragma unroll
                    for (int i = 0; i < 4; ++i) ar[r][i] = fmaf(wv, v[i], ar[r][i]);
                }
        }
    dpar_reduce<kMaxR>(ar, g.R, red, w, q, (dg & 3) == 0);
    __syncthreads();
    if (!in) return;
    // dx_dbl = [d_dtr, dBs, dCs] for this quad (every row group needs all C rows for the expansion)
    float xd[kMaxC][4];
#pragma unroll
    for (int c = 0; c < kMaxC; ++c)
        if (c < C) {
            if (c < g.R) dpar_fold(red, c, q, xd[c]);
            else {
                const float *src = c < g.R + g.N ? dBs + (bk * g.N + (c - g.R)) * g.L : dCs + (bk * g.N + (c - g.R - g.N)) * g.L;
                load4<float, true>(src, l0, g.L, xd[c]);
            }
        }
    if (dg == 0) {
#pragma unroll
        for (int c = 0; c < kMaxC; ++c)
            if (c < C) store4<float, true>(dxdbl + (bk * C + c) * g.L, l0, g.L, xd[c]);
    }
    // dxs[d] = sum_c Wx[c,d] * dx_dbl[c] (+ du[d])
    for (int d = dg; d < g.D; d += 16) {
        float o[4] = {0.f, 0.f, 0.f, 0.f};
        if (du) load4<float, true>(du + (row0 + d) * g.L, l0, g.L, o);
#pragma unroll
        for (int c = 0; c < kMaxC; ++c)
            if (c < C) {
                const float wv = s_wx[c * g.D + d];
#pragma unroll
                for (int i = 0; i < 4; ++i) o[i] = fmaf(wv, xd[c][i], o[i]);
            }
        store4<T, true>(dxs + (row0 + d) * g.L, l0, g.L, o);
    }
}

// ---- deep stages on the matrix cores (d_state 1, d_inner a multiple of 32, L a multiple of 32) -----------------------
// Both projections are skinny GEMMs over the rows of one direction: x_dbl (C x L) = Wx[k] (C x D) . xs (D x L) and
// dts (D x L) = Wdt[k] (D x R) . x_dbl[:R].  The row-parallel kernels above need the whole Wx / Wdt in LDS per workgroup,
// two barriers and run one workgroup per CU (22-27 us per call on <= 4 MB).  Here a WAVE owns 32 positions and nothing is
// shared: v_mfma_f32_32x32x2_f32 (exact fp32 products, fp32 accumulation — the numerics of the FMA kernels) with the
// contraction index on the two lane halves, so every operand is ONE element per lane in its natural layout: A = a weight
// row element, B = xs[d + (lane >> 5)][pos + (lane & 31)] (two coalesced 128-byte rows per instruction).  The C x 32
// result comes back with the position on the lane; one exchange between the lane halves gives every lane all C rows of its
// position, which are at once the B operand of the second product (k = dt component on the lane half).
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NMAX>
__device__ __forceinline__ void rows_to_lanes(const f32x16 &acc, const int n, const int half, float (&v)[NMAX]) {
    // accumulator row c lives in lane half (c >> 2) & 1, register (c & 3) + 4 (c >> 3)
#pragma unroll
    for (int c = 0; c < NMAX; ++c)
        if (c < n) {
            const float mine = acc[(c & 3) + 4 * (c >> 3)];
            const float other = __shfl_xor(mine, 32, 64);
            v[c] = (half == ((c >> 2) & 1)) ? mine : other;
        }
}

template <typename T>
__global__ __launch_bounds__(256) void xproj_fwd_mfma_kernel(const T *__restrict__ xs, const float *__restrict__ Wx,
                                                             const float *__restrict__ Wdt, float *__restrict__ dts,
                                                             float *__restrict__ Bs, float *__restrict__ Cs,
                                                             float *__restrict__ dtr, const XpGeom g) {
    const int C = g.R + 2;
    const int lane = threadIdx.x & 63, j = lane & 31, kq = lane >> 5;
    const int k = blockIdx.y, b = blockIdx.z;
    const int pos = (blockIdx.x * 4 + (threadIdx.x >> 6)) * 32 + j;
    if (pos - j >= g.L) return;                                     // wave-uniform (L is a multiple of 32)
    const size_t bk = (size_t)b * g.K + k;
    const T *xp = xs + bk * g.D * g.L + pos;
    const float *wx = Wx + (size_t)k * C * g.D + (j < C ? j : 0) * g.D;
    f32x16 acc, acc1;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = acc1[i] = 0.f;
    // 32 rows of D per step: all 32 operand loads in flight before the first product; two accumulators halve the
    // dependent-MFMA chain (64 cycles per f32 32x32x2)
    for (int d0 = 0; d0 < g.D; d0 += 32) {
        float av[16], bv[16];
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            av[t] = j < C ? wx[d0 + 2 * t + kq] : 0.f;
            bv[t] = to_f32(xp[(size_t)(d0 + 2 * t + kq) * g.L]);
        }
#pragma unroll
        for (int t = 0; t < 16; t += 2) {
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[t], bv[t], acc, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(av[t + 1], bv[t + 1], acc1, 0, 0, 0);
        }
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] += acc1[i];
    float xd[kMaxC];
    rows_to_lanes<kMaxC>(acc, C, kq, xd);
#pragma unroll
    for (int c = 0; c < kMaxC; ++c)
        if (c < C && (c & 1) == kq) {   // the two halves hold the same values: each stores every other row
            float *dst = c < g.R ? dtr + (bk * g.R + c) * g.L : (c == g.R ? Bs + bk * g.L : Cs + bk * g.L);
            dst[pos] = xd[c];
        }
    // dts = Wdt[k] . x_dbl[:R]: 32 rows of D per product, dt component on the lane half
    const float *wd = Wdt + (size_t)k * g.D * g.R;
    float *dp = dts + bk * g.D * g.L + pos;
    for (int dt = 0; dt < g.D; dt += 32) {
        f32x16 o;
#pragma unroll
        for (int i = 0; i < 16; ++i) o[i] = 0.f;
#pragma unroll
        for (int q0 = 0; q0 < kMaxR; q0 += 2)
            if (q0 < g.R) {
                const float a = q0 + kq < g.R ? wd[(size_t)(dt + j) * g.R + q0 + kq] : 0.f;
                o = __builtin_amdgcn_mfma_f32_32x32x2f32(a, kq ? xd[q0 + 1 < kMaxC ? q0 + 1 : q0] : xd[q0], o, 0, 0, 0);
            }
#pragma unroll
        for (int i = 0; i < 16; ++i) dp[(size_t)(dt + (i & 3) + 8 * (i >> 2) + 4 * kq) * g.L] = o[i];
    }
}

// backward of the two projections wrt their input: d(x_dbl)[:R] = Wdt[k]^T . ddts, rows R, R+1 = dBs, dCs;
// dxs = Wx[k]^T . d(x_dbl) (+ du); d(x_dbl) is kept (ws) for the weight-gradient kernel
template <typename T>
__global__ __launch_bounds__(256) void xproj_bwd_a_mfma_kernel(const float *__restrict__ ddts, const float *__restrict__ dBs,
                                                               const float *__restrict__ dCs, const float *__restrict__ du,
                                                               const float *__restrict__ Wx, const float *__restrict__ Wdt,
                                                               T *__restrict__ dxs, float *__restrict__ dxdbl, const XpGeom g) {
    const int C = g.R + 2;
    const int lane = threadIdx.x & 63, j = lane & 31, kq = lane >> 5;
    const int k = blockIdx.y, b = blockIdx.z;
    const int pos = (blockIdx.x * 4 + (threadIdx.x >> 6)) * 32 + j;
    if (pos - j >= g.L) return;
    const size_t bk = (size_t)b * g.K + k, row0 = bk * g.D;
    const float *gp = ddts + row0 * g.L + pos;
    const float *wd = Wdt + (size_t)k * g.D * g.R + (j < g.R ? j : 0);
    f32x16 acc, acc1;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = acc1[i] = 0.f;
    for (int d0 = 0; d0 < g.D; d0 += 32) {
        float av[16], bv[16];
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            av[t] = j < g.R ? wd[(size_t)(d0 + 2 * t + kq) * g.R] : 0.f;
            bv[t] = gp[(size_t)(d0 + 2 * t + kq) * g.L];
        }
#pragma unroll
        for (int t = 0; t < 16; t += 2) {
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[t], bv[t], acc, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(av[t + 1], bv[t + 1], acc1, 0, 0, 0);
        }
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] += acc1[i];
    float gd[kMaxC];
    rows_to_lanes<kMaxC>(acc, g.R, kq, gd);
#pragma unroll
    for (int c = 0; c < kMaxC; ++c) {
        if (c == g.R) gd[c] = dBs[bk * g.L + pos];
        if (c == g.R + 1) gd[c] = dCs[bk * g.L + pos];
    }
#pragma unroll
    for (int c = 0; c < kMaxC; ++c)
        if (c < C && (c & 1) == kq) dxdbl[(bk * C + c) * g.L + pos] = gd[c];
    const float *wx = Wx + (size_t)k * C * g.D;
    for (int dt = 0; dt < g.D; dt += 32) {
        f32x16 o;
#pragma unroll
        for (int i = 0; i < 16; ++i) o[i] = 0.f;
#pragma unroll
        for (int c0 = 0; c0 < kMaxC; c0 += 2)
            if (c0 < C) {
                const float a = c0 + kq < C ? wx[(size_t)(c0 + kq) * g.D + dt + j] : 0.f;
                o = __builtin_amdgcn_mfma_f32_32x32x2f32(a, kq ? gd[c0 + 1 < kMaxC ? c0 + 1 : c0] : gd[c0], o, 0, 0, 0);
            }
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const size_t off = (row0 + dt + (i & 3) + 8 * (i >> 2) + 4 * kq) * g.L + pos;
            dxs[off] = from_f32<T>(o[i] + (du ? du[off] : 0.f));
        }
    }
}

// one wave per (k, 4 rows d, 4096-position chunk of one batch element): partial dWx[k, c, d] and
// dWdt[k, d, r] -> float atomics (a few hundred adds per address at most; outputs zero-initialised)
constexpr int kXpChunk = 4096;   // positions per wave; 1024 when that leaves the chip under-filled (deep stages)

template <typename T, bool VEC>
__global__ __launch_bounds__(256) void xproj_bwd_b_kernel(const T *__restrict__ xs, const float *__restrict__ ddts,
                                                          const float *__restrict__ dxdbl, const float *__restrict__ dtr,
                                                          float *__restrict__ dWx, float *__restrict__ dWdt,
                                                          const XpGeom g, const int chunk, unsigned *det) {
    constexpr int RW = 4;  // rows per wave
    const int C = g.R + 2 * g.N;
    const int lane = threadIdx.x & 63;
    const int nrb = (g.D + RW - 1) / RW;
    const int wid_raw = blockIdx.x * 4 + (threadIdx.x >> 6);  // (k, row-block)
    const bool active = wid_raw < g.K * nrb;
    if (!active && det == nullptr) return;                // (deterministic mode: every wave reaches the ordered tail)
    const int wid = active ? wid_raw : 0;
    const int k = wid / nrb, d0 = (wid % nrb) * RW;
    const int b = blockIdx.z;
    const int l_begin = blockIdx.y * chunk, l_end = active ? min(g.L, l_begin + chunk) : l_begin;
    float ax[RW][kMaxC], at[RW][kMaxR];
#pragma unroll
    for (int j = 0; j < RW; ++j) {
#pragma unroll
        for (int c = 0; c < kMaxC; ++c) ax[j][c] = 0.f;
#pragma unroll
        for (int r = 0; r < kMaxR; ++r) at[j][r] = 0.f;
    }
    const size_t bk = (size_t)b * g.K + k;
    for (int l0 = l_begin + lane * 4; l0 < l_end; l0 += 256) {
        float xv[RW][4], gv[RW][4];
#pragma unroll
        for (int j = 0; j < RW; ++j) {
            if (d0 + j < g.D) {
                load4<T, VEC>(xs + (bk * g.D + d0 + j) * g.L, l0, g.L, xv[j]);
                load4<float, VEC>(ddts + (bk * g.D + d0 + j) * g.L, l0, g.L, gv[j]);
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i) { xv[j][i] = 0.f; gv[j][i] = 0.f; }
            }
        }
#pragma unroll
        for (int c = 0; c < kMaxC; ++c)
            if (c < C) {
                float q[4];
                load4<float, VEC>(dxdbl + (bk * C + c) * g.L, l0, g.L, q);
#pragma unroll
                for (int j = 0; j < RW; ++j)
                    ax[j][c] += (q[0] * xv[j][0] + q[1] * xv[j][1]) + (q[2] * xv[j][2] + q[3] * xv[j][3]);
            }
#pragma unroll
        for (int r = 0; r < kMaxR; ++r)
            if (r < g.R) {
                float q[4];
                load4<float, VEC>(dtr + (bk * g.R + r) * g.L, l0, g.L, q);
#pragma unroll
                for (int j = 0; j < RW; ++j)
                    at[j][r] += (q[0] * gv[j][0] + q[1] * gv[j][1]) + (q[2] * gv[j][2] + q[3] * gv[j][3]);
            }
    }
    auto tail = [&]() {
        if (!active) return;
#pragma unroll
        for (int j = 0; j < RW; ++j) {
            if (d0 + j >= g.D) break;
#pragma unroll
            for (int c = 0; c < kMaxC; ++c)
                if (c < C) {
                    const float s = wave_sum(ax[j][c]);
                    if (lane == 0) atomicAdd(dWx + ((size_t)k * C + c) * g.D + d0 + j, s);
                }
#pragma unroll
            for (int r = 0; r < kMaxR; ++r)
                if (r < g.R) {
                    const float s = wave_sum(at[j][r]);
                    if (lane == 0) atomicAdd(dWdt + ((size_t)k * g.D + d0 + j) * g.R + r, s);
                }
        }
    };
    // deterministic mode (common.h): workgroups in workgroup order, their four waves in wave order
    det_enter(det);
    VMASR_DET_WAVE_ORDER(det, 4, tail());
    det_leave(det);
}

// the MFMA kernels: d_state 1, whole 32-row / 32-position tiles, even dt_rank (the contraction runs two at a time)
bool mfma_ok(const XpGeom &g, bool vec) {
    static const bool on = [] { const char *e = getenv("VMASR_XPROJ_MFMA"); return e && e[0] == '1'; }();   // opt-in: measured slower
    return on && vec && g.N == 1 && g.D >= 64 && g.D % 32 == 0 && g.L % 32 == 0 && g.R % 2 == 0 && g.R + 2 <= kMaxC;
}

int check(const XpGeom &g, int dtype, const char *what) {
    VMASR_REQUIRE(g.B > 0 && g.K > 0 && g.D > 0 && g.N > 0 && g.R > 0 && g.L > 0, VMASR_EINVAL, "%s: non-positive size", what);
    VMASR_REQUIRE(g.R <= kMaxR && g.R + 2 * g.N <= kMaxC, VMASR_EINVAL, "%s: need dt_rank <= %d and dt_rank + 2*d_state <= %d",
                  what, kMaxR, kMaxC);
    VMASR_REQUIRE(g.B <= 65535 && g.K <= 65535, VMASR_EINVAL, "%s: batch / directions too large", what);
    VMASR_REQUIRE((size_t)(g.R + 2 * g.N + g.R) * g.D * 4 <= 64 * 1024, VMASR_EINVAL, "%s: weights do not fit LDS", what);
    VMASR_REQUIRE(dtype == VMASR_F32 || dtype == VMASR_F16 || dtype == VMASR_BF16, VMASR_EINVAL, "%s: bad dtype", what);
    return 0;
}

}  // namespace
}  // namespace vmasr

using namespace vmasr;

VMASR_EXPORT int vmasr_xproj_supported(int32_t d_state, int32_t dt_rank, int32_t d_inner) {
    return (dt_rank >= 1 && dt_rank <= kMaxR && d_state >= 1 && dt_rank + 2 * d_state <= kMaxC &&
            (size_t)(2 * dt_rank + 2 * d_state) * d_inner * 4 <= 64 * 1024) ? 1 : 0;
}

VMASR_EXPORT int vmasr_xproj_fwd(const void *xs, const float *Wx, const float *Wdt, float *dts, float *Bs, float *Cs,
                                 float *dtr, int32_t B, int32_t K, int32_t D, int32_t N, int32_t R, int32_t L,
                                 int32_t dtype, vmasr_stream_t stream) {
    const XpGeom g{B, K, D, N, R, L};
    if (int e = check(g, dtype, "xproj_fwd")) return e;
    VMASR_REQUIRE(xs && Wx && Wdt && dts && Bs && Cs && dtr, VMASR_EINVAL, "xproj_fwd: null tensor");
    const bool vec = L % 4 == 0 && aligned_to(xs, dtype == VMASR_F32 ? 16 : 8) && aligned_to(dts, 16) && aligned_to(Bs, 16) &&
                     aligned_to(Cs, 16) && aligned_to(dtr, 16);
    const int C = R + 2 * N;
    const size_t sm = (size_t)(C + R) * D * sizeof(float);
    const dim3 grid((L + 1023) / 1024, K, B);
    hipStream_t st = static_cast<hipStream_t>(stream);
    const double es = dtype == VMASR_F32 ? 4 : 2;
    const double bytes = (double)B * K * L * (D * (es + 4.0) + (C + 0.0) * 4.0);
    const size_t smd = sm + (size_t)C * 4 * kDpTL * sizeof(float);
    if (mfma_ok(g, vec)) {                             // deep stages on the matrix cores (xproj_fwd_mfma_kernel)
        const dim3 gm((L / 32 + 3) / 4, K, B);
#define VMASR_XPM(TT) VMASR_LAUNCH(VMASR_K_XPROJ_FWD, bytes, (xproj_fwd_mfma_kernel<TT>), gm, dim3(256), 0, st, (const TT *)xs, Wx, Wdt, dts, Bs, Cs, dtr, g)
        if (dtype == VMASR_F32) VMASR_XPM(float);
        else if (dtype == VMASR_F16) VMASR_XPM(f16_t);
        else VMASR_XPM(bf16_t);
#undef VMASR_XPM
        return check_launch("xproj_fwd");
    }
    if (vec && D >= kDparMinD && smd <= 64 * 1024) {   // deep stages: row-parallel workgroups (see xproj_fwd_dpar_kernel)
        const dim3 gd((L + kDpTL - 1) / kDpTL, K, B);
#define VMASR_XPD(TT) VMASR_LAUNCH(VMASR_K_XPROJ_FWD, bytes, (xproj_fwd_dpar_kernel<TT>), gd, dim3(256), smd, st, (const TT *)xs, Wx, Wdt, dts, Bs, Cs, dtr, g)
        if (dtype == VMASR_F32) VMASR_XPD(float);
        else if (dtype == VMASR_F16) VMASR_XPD(f16_t);
        else VMASR_XPD(bf16_t);
#undef VMASR_XPD
        return check_launch("xproj_fwd");
    }
#define VMASR_XP(TT, V) VMASR_LAUNCH(VMASR_K_XPROJ_FWD, bytes, (xproj_fwd_kernel<TT, V>), grid, dim3(256), sm, st, (const TT *)xs, Wx, Wdt, dts, Bs, Cs, dtr, g)
    if (dtype == VMASR_F32) { if (vec) VMASR_XP(float, true); else VMASR_XP(float, false); }
    else if (dtype == VMASR_F16) { if (vec) VMASR_XP(f16_t, true); else VMASR_XP(f16_t, false); }
    else { if (vec) VMASR_XP(bf16_t, true); else VMASR_XP(bf16_t, false); }
#undef VMASR_XP
    return check_launch("xproj_fwd");
}

VMASR_EXPORT int vmasr_xproj_bwd(const void *xs, const float *Wx, const float *Wdt, const float *dtr, const float *ddts,
                                 const float *dBs, const float *dCs, const float *du, void *dxs, float *dWx, float *dWdt,
                                 float *ws, int32_t B, int32_t K, int32_t D, int32_t N, int32_t R, int32_t L,
                                 int32_t dtype, vmasr_stream_t stream) {
    const XpGeom g{B, K, D, N, R, L};
    if (int e = check(g, dtype, "xproj_bwd")) return e;
    VMASR_REQUIRE(xs && Wx && Wdt && dtr && ddts && dBs && dCs && dxs && dWx && dWdt && ws, VMASR_EINVAL,
                  "xproj_bwd: null tensor");
    const bool vec = L % 4 == 0 && aligned_to(xs, dtype == VMASR_F32 ? 16 : 8) && aligned_to(dxs, dtype == VMASR_F32 ? 16 : 8) &&
                     aligned_to(ddts, 16) && aligned_to(dBs, 16) && aligned_to(dCs, 16) && aligned_to(dtr, 16) &&
                     aligned_to(ws, 16) && (!du || aligned_to(du, 16));
    const int C = R + 2 * N;
    const size_t sm = (size_t)(C + R) * D * sizeof(float);
    const dim3 grid((L + 1023) / 1024, K, B);
    hipStream_t st = static_cast<hipStream_t>(stream);
    const double es = dtype == VMASR_F32 ? 4 : 2;
    const double bytes_a = (double)B * K * L * (D * (4.0 + es + (du ? 4.0 : 0.0)) + 2.0 * C * 4.0);
    const double bytes_b = (double)B * K * L * (D * (4.0 + es) + (double)(C + R) * 4.0);
    const int nwaves = K * ((D + 3) / 4);
    const size_t smd = sm + (size_t)C * 4 * kDpTL * sizeof(float);
    const bool dpar = vec && D >= kDparMinD && smd <= 64 * 1024;
    const dim3 gd((L + kDpTL - 1) / kDpTL, K, B);
    // weight gradients: one wave per (4 rows, chunk of positions); shorter chunks when 4096 leaves < 4 waves per CU
    const int chunk = ((long)((nwaves + 3) / 4) * ((L + kXpChunk - 1) / kXpChunk) * B < 1024 && L > 1024) ? 1024 : kXpChunk;
#define VMASR_XPB(TT, V)                                                                                                 \
    do {                                                                                                                 \
        if (mfma_ok(g, vec)) VMASR_LAUNCH(VMASR_K_XPROJ_BWD_A, bytes_a, (xproj_bwd_a_mfma_kernel<TT>), dim3((L / 32 + 3) / 4, K, B), \
                                          dim3(256), 0, st, ddts, dBs, dCs, du, Wx, Wdt, (TT *)dxs, ws, g);               \
        else if (dpar) VMASR_LAUNCH(VMASR_K_XPROJ_BWD_A, bytes_a, (xproj_bwd_a_dpar_kernel<TT>), gd, dim3(256), smd, st, ddts, dBs, dCs, du, \
                               Wx, Wdt, (TT *)dxs, ws, g);                                                               \
        else VMASR_LAUNCH(VMASR_K_XPROJ_BWD_A, bytes_a, (xproj_bwd_a_kernel<TT, V>), grid, dim3(256), sm, st, ddts, dBs, dCs, du, \
                          Wx, Wdt, (TT *)dxs, ws, g);                                                                    \
        VMASR_LAUNCH(VMASR_K_XPROJ_BWD_B, bytes_b, (xproj_bwd_b_kernel<TT, V>),                                           \
                     dim3((nwaves + 3) / 4, (L + chunk - 1) / chunk, B), dim3(256), 0, st,                                \
                     (const TT *)xs, ddts, ws, dtr, dWx, dWdt, g, chunk, det_ticket(VMASR_K_XPROJ_BWD_B));               \
    } while (0)
    if (dtype == VMASR_F32) { if (vec) VMASR_XPB(float, true); else VMASR_XPB(float, false); }
    else if (dtype == VMASR_F16) { if (vec) VMASR_XPB(f16_t, true); else VMASR_XPB(f16_t, false); }
    else { if (vec) VMASR_XPB(bf16_t, true); else VMASR_XPB(bf16_t, false); }
#undef VMASR_XPB
    return check_launch("xproj_bwd");
}
