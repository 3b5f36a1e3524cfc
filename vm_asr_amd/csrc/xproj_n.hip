// xproj_n.hip — SS2D's per-direction input projections (x_proj and dt_proj) for a general state dimension (d_state N > 1), gfx950.
//
// Replaces, for N > 1 (BASELINE configs[4]: `MODEL.VSSM.SSM_D_STATE 32`), the two einsums of SS2D.forward_corev2 and the copies /
// casts around them (model/vmamba.py:1473-1491):
//     x_dbl = einsum('b k d l, k c d -> b k c l', xs, x_proj_weight)      c = R + 2N  (66 .. 80 rows at N = 32)
//     dts, Bs, Cs = split(x_dbl, [R, N, N]);  dts = einsum('b k r l, k d r -> b k d l', dts, dt_projs_weight)
//     .contiguous() x3, .to(float) x4
// With C = R + 2N rows these ARE matrix products (unlike the N = 1 maps of xproj.hip: C = 3 .. 10), but skinny ones: per
// direction (C x D) . (D x L) with D = 2 .. 512 — 11-16 flop per byte of activation traffic: bound by HBM, not by the matrix cores.
// The reference ran them as batched hipBLASLt GEMMs plus 7 copy / cast passes (115 ms of the 372 ms step at configs[4]).  Here:
//
//   * v_mfma_f32_32x32x2_f32 (exact fp32 products, fp32 accumulation: the numerics of an fp32 FMA loop) with the contraction index
//     on the two lane halves, so every operand is ONE element per lane in its natural layout — no LDS, no transposes:
//     forward   A = a weight element, B = xs[d + (lane >> 5)][pos + (lane & 31)]: a WAVE owns 32 positions, walks the C rows in
//               blocks of 32 and writes dt rows / Bs / Cs straight from the accumulator (two 128-byte runs per store instruction),
//               then expands dts = Wdt . dt from the dt rows it still holds;
//     backward  d(dt) = Wdt^T . ddts, dxs = Wx^T . [d(dt); dBs; dCs] (+ the scan's own gradient wrt xs), same layout;
//     weights   dWx = [d(dt); dBs; dCs] . xs^T and dWdt = ddts . dt^T contract over POSITIONS: each lane streams its own row
//               (16 bytes per load, the two lane halves take alternate quads) — 32 x 32 tiles over chunks of positions, float atomics.
// fp32 weights and accumulation; xs may be fp32 or 16-bit.  d_inner even, dt_rank <= 16.
#include "common.h"

#include <algorithm>

namespace vmasr {
namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int kNMaxR = 16;

struct XnGeom {
    int B, K, D, N, R, L, C;
};

// accumulator register i of a lane in half kq holds row (i & 3) + 8 (i >> 2) + 4 kq of the 32 x 32 tile, column = lane & 31
__device__ __forceinline__ int acc_row(const int i, const int kq) { return (i & 3) + 8 * (i >> 2) + 4 * kq; }

__device__ __forceinline__ f32x16 zero16() {
    f32x16 z;
#pragma unroll
    for (int i = 0; i < 16; ++i) z[i] = 0.f;
    return z;
}

// rows 0 .. n-1 (n <= 16) of the tile for THIS lane's column: row c lives in half (c >> 2) & 1, register (c & 3) + 4 (c >> 3)
__device__ __forceinline__ void rows_to_lanes16(const f32x16 &acc, const int n, const int half, float (&v)[kNMaxR]) {
#pragma unroll
    for (int c = 0; c < kNMaxR; ++c) {
        v[c] = 0.f;
        if (c < n) {
            const float mine = acc[(c & 3) + 4 * (c >> 3)];
            const float other = __shfl_xor(mine, 32, 64);
            v[c] = (half == ((c >> 2) & 1)) ? mine : other;
        }
    }
}

// ---- forward ------------------------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void xproj_nfwd_kernel(const T *__restrict__ xs, const float *__restrict__ Wx,
                                                         const float *__restrict__ Wdt, float *__restrict__ dts,
                                                         float *__restrict__ Bs, float *__restrict__ Cs, float *__restrict__ dtr,
                                                         const XnGeom g) {
    const int lane = threadIdx.x & 63, j = lane & 31, kq = lane >> 5;
    const int k = blockIdx.y, b = blockIdx.z;
    const int p0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * 32;
    if (p0 >= g.L) return;                                         // wave-uniform
    const int pos = p0 + j;
    const bool pv = pos < g.L;
    const size_t bk = (size_t)b * g.K + k;
    const T *xp = xs + bk * g.D * g.L + (pv ? pos : 0);
    float xd[kNMaxR];
#pragma unroll
    for (int c = 0; c < kNMaxR; ++c) xd[c] = 0.f;
    const int ncb = (g.C + 31) / 32;
    for (int cb = 0; cb < ncb; ++cb) {
        const int crow = cb * 32 + j;                              // the weight row this lane feeds
        const bool cv = crow < g.C;
        const float *wx = Wx + ((size_t)k * g.C + (cv ? crow : 0)) * g.D;
        f32x16 acc = zero16(), acc1 = zero16();
        // 32 rows of D per step: all operand loads in flight before the first product; two accumulators halve the dependent chain
        for (int d0 = 0; d0 < g.D; d0 += 32) {
            float av[16], bv[16];
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                const int d = d0 + 2 * t + kq;
                const bool dv = d < g.D;
                av[t] = (dv && cv) ? wx[d] : 0.f;
                bv[t] = (dv && pv) ? to_f32(xp[(size_t)d * g.L]) : 0.f;
            }
#pragma unroll
            for (int t = 0; t < 16; t += 2) {
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[t], bv[t], acc, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(av[t + 1], bv[t + 1], acc1, 0, 0, 0);
            }
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] += acc1[i];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int c = cb * 32 + acc_row(i, kq);
            if (c < g.C && pv) {
                float *dst = c < g.R ? dtr + (bk * g.R + c) * g.L
                                     : (c < g.R + g.N ? Bs + (bk * g.N + (c - g.R)) * g.L : Cs + (bk * g.N + (c - g.R - g.N)) * g.L);
                dst[pos] = acc[i];
            }
        }
        if (cb == 0) rows_to_lanes16(acc, g.R, kq, xd);            // dt_rank <= 16: the dt rows are in the first block
    }
    // dts = Wdt[k] . dt: 32 rows of D per product, dt component on the lane half
    const float *wd = Wdt + (size_t)k * g.D * g.R;
    for (int dt0 = 0; dt0 < g.D; dt0 += 32) {
        f32x16 o = zero16();
        const bool rv = dt0 + j < g.D;
#pragma unroll
        for (int q0 = 0; q0 < kNMaxR; q0 += 2)
            if (q0 < g.R) {
                const float a = (rv && q0 + kq < g.R) ? wd[(size_t)(dt0 + j) * g.R + q0 + kq] : 0.f;
                o = __builtin_amdgcn_mfma_f32_32x32x2f32(a, kq ? xd[q0 + 1 < kNMaxR ? q0 + 1 : q0] : xd[q0], o, 0, 0, 0);
            }
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int d = dt0 + acc_row(i, kq);
            if (d < g.D && pv) dts[(bk * g.D + d) * g.L + pos] = o[i];
        }
    }
}

// ---- backward wrt the input: d(dt) = Wdt[k]^T . ddts (kept: ddtr, for the weight-gradient kernel), dxs = Wx[k]^T . [d(dt); dBs; dCs] (+ du)
template <typename T>
__global__ __launch_bounds__(256) void xproj_nbwd_a_kernel(const float *__restrict__ ddts, const float *__restrict__ dBs,
                                                           const float *__restrict__ dCs, const float *__restrict__ du,
                                                           const float *__restrict__ Wx, const float *__restrict__ Wdt,
                                                           T *__restrict__ dxs, float *__restrict__ ddtr, const XnGeom g) {
    const int lane = threadIdx.x & 63, j = lane & 31, kq = lane >> 5;
    const int k = blockIdx.y, b = blockIdx.z;
    const int p0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * 32;
    if (p0 >= g.L) return;
    const int pos = p0 + j;
    const bool pv = pos < g.L;
    const size_t bk = (size_t)b * g.K + k, row0 = bk * g.D;
    const float *gp = ddts + row0 * g.L + (pv ? pos : 0);
    const float *wd = Wdt + (size_t)k * g.D * g.R + (j < g.R ? j : 0);
    f32x16 acc = zero16(), acc1 = zero16();
    for (int d0 = 0; d0 < g.D; d0 += 32) {
        float av[16], bv[16];
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            const int d = d0 + 2 * t + kq;
            const bool dv = d < g.D;
            av[t] = (dv && j < g.R) ? wd[(size_t)d * g.R] : 0.f;
            bv[t] = (dv && pv) ? gp[(size_t)d * g.L] : 0.f;
        }
#pragma unroll
        for (int t = 0; t < 16; t += 2) {
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[t], bv[t], acc, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(av[t + 1], bv[t + 1], acc1, 0, 0, 0);
        }
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] += acc1[i];
    float gd[kNMaxR];
    rows_to_lanes16(acc, g.R, kq, gd);
#pragma unroll
    for (int c = 0; c < kNMaxR; ++c)
        if (c < g.R && (c & 1) == kq && pv) ddtr[(bk * g.R + c) * g.L + pos] = gd[c];   // the halves hold the same values: each stores every other row
    // rows R .. C-1 of d(x_dbl) are dBs / dCs as they lie in memory
    auto gload = [&](const int c) -> float {
        if (c >= g.C || !pv) return 0.f;
        return c < g.R + g.N ? dBs[(bk * g.N + (c - g.R)) * g.L + pos] : dCs[(bk * g.N + (c - g.R - g.N)) * g.L + pos];
    };
    const float *wx = Wx + (size_t)k * g.C * g.D;
    const int c_even = (g.R + 1) & ~1;                               // first pair that lies wholly behind the dt rows
    for (int dt0 = 0; dt0 < g.D; dt0 += 32) {
        f32x16 o = zero16(), o1 = zero16();
        const bool rv = dt0 + j < g.D;
#pragma unroll
        for (int q0 = 0; q0 < kNMaxR; q0 += 2)
            if (q0 < g.R) {
                const int c = q0 + kq;
                float bval = kq ? gd[q0 + 1 < kNMaxR ? q0 + 1 : q0] : gd[q0];
                if (c >= g.R) bval = gload(c);                          // odd dt_rank: the partner of the last dt row is the first B row
                const float a = (rv && c < g.C) ? wx[(size_t)c * g.D + dt0 + j] : 0.f;
                o = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bval, o, 0, 0, 0);
            }
        for (int c0 = c_even; c0 < g.C; c0 += 16) {
            float av[8], bv[8];
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                const int c = c0 + 2 * t + kq;
                av[t] = (rv && c < g.C) ? wx[(size_t)c * g.D + dt0 + j] : 0.f;
                bv[t] = gload(c);
            }
#pragma unroll
            for (int t = 0; t < 8; t += 2) {
                o = __builtin_amdgcn_mfma_f32_32x32x2f32(av[t], bv[t], o, 0, 0, 0);
                o1 = __builtin_amdgcn_mfma_f32_32x32x2f32(av[t + 1], bv[t + 1], o1, 0, 0, 0);
            }
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int d = dt0 + acc_row(i, kq);
            if (d < g.D && pv) {
                const size_t off = (row0 + d) * g.L + pos;
                dxs[off] = from_f32<T>(o[i] + o1[i] + (du ? du[off] : 0.f));
            }
        }
    }
}

// ---- weight gradients: out[m][n] += sum_l A_m[l] B_n[l] over this block's chunk of positions, one 32 x 32 tile per block ---------
// A rows: up to three row segments (pointer, row count) of fp32 rows of length L; B rows: one segment of TB rows.
struct Seg3 {
    const float *p[3];
    int n[3];
};

template <typename TB>
__global__ __launch_bounds__(256) void xproj_nbwd_b_kernel(const Seg3 A, const long a_bk_stride0, const long a_bk_stride1, const long a_bk_stride2,
                                                           const TB *__restrict__ Bp, const long b_bk_stride, const int M, const int Nc,
                                                           float *__restrict__ out, const long out_k_stride, const int ldo, const int K,
                                                           const int L, const int chunk, const bool vec, unsigned *det) {
    const int lane = threadIdx.x & 63, j = lane & 31, kq = lane >> 5, wave = threadIdx.x >> 6;
    const int ntn = (Nc + 31) / 32;
    const int mb = blockIdx.y / ntn, nb = blockIdx.y % ntn;
    const long bk = blockIdx.z;
    const int k = (int)(bk % K);
    // the lane's A row (m) and B row (n)
    const int m = mb * 32 + j, n = nb * 32 + j;
    const float *arow = nullptr;
    if (m < M) {
        if (m < A.n[0]) arow = A.p[0] + bk * a_bk_stride0 + (long)m * L;
        else if (m < A.n[0] + A.n[1]) arow = A.p[1] + bk * a_bk_stride1 + (long)(m - A.n[0]) * L;
        else arow = A.p[2] + bk * a_bk_stride2 + (long)(m - A.n[0] - A.n[1]) * L;
    }
    const TB *brow = n < Nc ? Bp + bk * b_bk_stride + (long)n * L : nullptr;
    const int per = chunk / 4;                                   // positions per wave (a multiple of 8)
    const int l_begin = blockIdx.x * chunk + wave * per, l_end = min(L, l_begin + per);
    f32x16 acc = zero16(), acc1 = zero16();
    for (int lb = l_begin; lb < l_end; lb += 8) {                // uniform trip count (MFMA ignores EXEC): the two lane halves take
        const int l = lb + 4 * kq;                               // alternate quads of positions, masked past the end
        float a4[4], b4[4];
        if (arow) load4u<float, true>(arow, l, l_end, a4, vec && l + 3 < l_end);
        else { a4[0] = a4[1] = a4[2] = a4[3] = 0.f; }
        if (brow) load4u<TB, true>(brow, l, l_end, b4, vec && l + 3 < l_end);
        else { b4[0] = b4[1] = b4[2] = b4[3] = 0.f; }
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[0], b4[0], acc, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[1], b4[1], acc1, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[2], b4[2], acc, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[3], b4[3], acc1, 0, 0, 0);
    }
    float *o = out + (long)k * out_k_stride;
    // (deterministic mode, common.h: the workgroups' atomics in workgroup order, the four waves of a workgroup in wave order — each
    //  wave's atomics performed, `s_waitcnt vmcnt(0)`, before the next wave issues its own)
    auto add_tile = [&]() {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int mm = mb * 32 + acc_row(i, kq);
            if (mm < M && n < Nc) atomicAdd(o + (long)mm * ldo + n, acc[i] + acc1[i]);
        }
        if (det) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    };
    det_enter(det);
    VMASR_DET_WAVE_ORDER(det, 4, add_tile());
    det_leave(det);
}

int check(const XnGeom &g, int dtype, const char *what) {
    VMASR_REQUIRE(g.B > 0 && g.K > 0 && g.D > 0 && g.N > 1 && g.R > 0 && g.L > 0, VMASR_EINVAL, "%s: sizes (d_state > 1 expected here)", what);
    VMASR_REQUIRE(g.R <= kNMaxR && g.D % 2 == 0 && g.N <= 256, VMASR_EINVAL, "%s: need dt_rank <= %d, even d_inner, d_state <= 256", what, kNMaxR);
    VMASR_REQUIRE(g.K <= 65535 && g.B <= 65535 && (long)g.B * g.K <= 65535, VMASR_EINVAL, "%s: batch x directions too large", what);
    VMASR_REQUIRE(dtype == VMASR_F32 || dtype == VMASR_F16 || dtype == VMASR_BF16, VMASR_EINVAL, "%s: bad dtype", what);
    return 0;
}

}  // namespace
}  // namespace vmasr

using namespace vmasr;

VMASR_EXPORT int vmasr_xproj_n_supported(int32_t d_state, int32_t dt_rank, int32_t d_inner) {
    return (d_state > 1 && d_state <= 256 && dt_rank >= 1 && dt_rank <= kNMaxR && d_inner >= 2 && d_inner % 2 == 0) ? 1 : 0;
}

// floats of the backward's workspace: the gradient of the low-rank dt rows, (B, K, R, L)
VMASR_EXPORT size_t vmasr_xproj_n_ws_floats(int32_t B, int32_t K, int32_t R, int32_t L) { return (size_t)B * K * R * L; }

VMASR_EXPORT int vmasr_xproj_n_fwd(const void *xs, const float *Wx, const float *Wdt, float *dts, float *Bs, float *Cs, float *dtr,
                                   int32_t B, int32_t K, int32_t D, int32_t N, int32_t R, int32_t L, int32_t dtype, vmasr_stream_t stream) {
    const XnGeom g{B, K, D, N, R, L, R + 2 * N};
    if (int e = check(g, dtype, "xproj_n_fwd")) return e;
    VMASR_REQUIRE(xs && Wx && Wdt && dts && Bs && Cs && dtr, VMASR_EINVAL, "xproj_n_fwd: null tensor");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const double es = dtype == VMASR_F32 ? 4 : 2;
    const double bytes = (double)B * K * L * (D * (es + 4.0) + g.C * 4.0);
    const dim3 grid(((L + 31) / 32 + 3) / 4, K, B);
#define VMASR_XN(TT) VMASR_LAUNCH(VMASR_K_XPROJ_FWD, bytes, (xproj_nfwd_kernel<TT>), grid, dim3(256), 0, st, (const TT *)xs, Wx, Wdt, dts, Bs, Cs, dtr, g)
    if (dtype == VMASR_F32) VMASR_XN(float);
    else if (dtype == VMASR_F16) VMASR_XN(f16_t);
    else VMASR_XN(bf16_t);
#undef VMASR_XN
    return check_launch("xproj_n_fwd");
}

VMASR_EXPORT int vmasr_xproj_n_bwd(const void *xs, const float *Wx, const float *Wdt, const float *dtr, const float *ddts,
                                   const float *dBs, const float *dCs, const float *du, void *dxs, float *dWx, float *dWdt, float *ws,
                                   int32_t B, int32_t K, int32_t D, int32_t N, int32_t R, int32_t L, int32_t dtype, vmasr_stream_t stream) {
    const XnGeom g{B, K, D, N, R, L, R + 2 * N};
    if (int e = check(g, dtype, "xproj_n_bwd")) return e;
    VMASR_REQUIRE(xs && Wx && Wdt && dtr && ddts && dBs && dCs && dxs && dWx && dWdt && ws, VMASR_EINVAL, "xproj_n_bwd: null tensor");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const double es = dtype == VMASR_F32 ? 4 : 2;
    const double bytes_a = (double)B * K * L * (D * (4.0 + es + (du ? 4.0 : 0.0)) + (2.0 * N + R) * 4.0);
    const double bytes_b = (double)B * K * L * (D * (4.0 + es) + (double)(g.C + R) * 4.0);
    const dim3 grid(((L + 31) / 32 + 3) / 4, K, B);
#define VMASR_XNA(TT) VMASR_LAUNCH(VMASR_K_XPROJ_BWD_A, bytes_a, (xproj_nbwd_a_kernel<TT>), grid, dim3(256), 0, st, ddts, dBs, dCs, du, Wx, Wdt, (TT *)dxs, ws, g)
    if (dtype == VMASR_F32) VMASR_XNA(float);
    else if (dtype == VMASR_F16) VMASR_XNA(f16_t);
    else VMASR_XNA(bf16_t);
#undef VMASR_XNA
    // weight gradients: 32 x 32 tiles x chunks of positions (enough blocks to fill the chip, at least 512 positions per block)
    const bool vec = L % 4 == 0 && aligned_to(xs, 16) && aligned_to(ddts, 16) && aligned_to(dBs, 16) && aligned_to(dCs, 16) && aligned_to(dtr, 16) && aligned_to(ws, 16);
    const int tiles_x = ((g.C + 31) / 32) * ((D + 31) / 32), tiles_t = ((D + 31) / 32) * ((R + 31) / 32);
    int chunk = 4096;
    while (chunk > 512 && (long)((L + chunk - 1) / chunk) * tiles_x * B * K < 2048) chunk >>= 1;
    unsigned *det = det_ticket(VMASR_K_XPROJ_BWD_B);       // deterministic mode: ordered accumulation (null otherwise)
    const Seg3 ax{{ws, dBs, dCs}, {R, N, N}};
    const dim3 gx((L + chunk - 1) / chunk, tiles_x, B * K), gt((L + chunk - 1) / chunk, tiles_t, B * K);
#define VMASR_XNB(TT)                                                                                                                \
    VMASR_LAUNCH(VMASR_K_XPROJ_BWD_B, bytes_b, (xproj_nbwd_b_kernel<TT>), gx, dim3(256), 0, st, ax, (long)R * L, (long)N * L, (long)N * L, \
                 (const TT *)xs, (long)D * L, g.C, D, dWx, (long)g.C * D, D, K, L, chunk, vec, det)
    if (dtype == VMASR_F32) VMASR_XNB(float);
    else if (dtype == VMASR_F16) VMASR_XNB(f16_t);
    else VMASR_XNB(bf16_t);
#undef VMASR_XNB
    const Seg3 at{{ddts, nullptr, nullptr}, {D, 0, 0}};
    VMASR_LAUNCH(VMASR_K_XPROJ_BWD_B, 0.0, (xproj_nbwd_b_kernel<float>), gt, dim3(256), 0, st, at, (long)D * L, 0L, 0L, dtr, (long)R * L, D, R,
                 dWdt, (long)D * R, R, K, L, chunk, vec, det);
    return check_launch("xproj_n_bwd");
}
