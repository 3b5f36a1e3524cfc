// sscan_n.hip — selective scan for a general state dimension (d_state N > 1), forward and backward, gfx950 (wave64).
//
// Same operator as sscan.hip (selective_scan_cuda_core.{fwd,bwd}: cus/selective_scan.cpp:157-349; kernels
// cus/selective_scan_fwd_kernel.cuh:101-158 — the loop over states at :124 — and cus/selective_scan_bwd_kernel.cuh:125-241),
// for the calls where every row carries N states (BASELINE configs[4]: `MODEL.VSSM.SSM_D_STATE 32`).  The work is
// N * L * rows state-steps of ~10 (forward) / ~25 (backward) arithmetic instructions each against 12-20 bytes per ROW-step:
// VALU-issue bound by a factor of N, not HBM bound.  The mapping is therefore chosen for instructions per state-step:
//
//   * lanes = time (64 lanes x 4 consecutive steps = the 256-step tile of sscan.hip: coalesced 16-byte loads, the
//     sums over states (y, du, ddelta) stay inside a lane), states are walked in PAIRS held in one 64-bit register
//     pair: decay, recurrence and every gradient product run as v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32 — two
//     states per instruction;
//   * ONE decay exp(delta A) per (row, step, state), shared by the forward recompute and the adjoint recurrence of
//     the backward: 2^(delta A log2 e) with the integer part split off by the 1.5*2^23 trick, a degree-6 polynomial
//     on [-1/2, 1/2] in packed FMAs and the exponent added with v_lshl_add_u32 — 5.5 instructions per state-step
//     (sscan.hip's scalar form: 7-13);
//   * the 64 lane aggregates of the two states of a pair are scanned TOGETHER: the DPP stages of the two states are
//     interleaved, which fills the two wait states a DPP read needs after the VALU write of its source (no s_nop);
//   * the running state of a row (h per state, the adjoint carry G per state) lives in a per-wave LDS slice, read as
//     one ds_read_b64 per pair; A (pre-multiplied by log2 e) is staged there once per task;
//   * backward: a wave owns R rows of one (batch, group) and walks them INSIDE the pair loop, so B / C are loaded once
//     per R rows and dB / dC are summed over those rows in registers; the W waves of a workgroup (other rows of the
//     same group, same tile) reduce them through LDS once per block of pairs (one barrier pair per block, not per
//     state) and leave as 256-byte runs of float atomics;
//   * the adjoint uses G_t = a_t g_t (ss2d.hip): G_t = a_t (dout_t C_t + G_{t+1}) needs only the step's own decay, so a
//     tile needs one scalar per state from its right neighbour.
// Numerics: fp32, same recurrence, softplus and association as sscan.hip; the decay is within 1.3 * 2^-24 relative of
// exp (rms 0.45), unbiased (tests/test_gpu_kernels.py: goldens n8 / n32, oracle, L = 524 288 stress).
#include "sscan_n.h"

#include "scan_prims.h"

#include <stdlib.h>

#include <algorithm>

namespace vmasr {
namespace {

typedef float v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ v2f splat(const float x) { return (v2f){x, x}; }
__device__ __forceinline__ v2f fma2(const v2f a, const v2f b, const v2f c) { return __builtin_elementwise_fma(a, b, c); }

constexpr float kMagic = 12582912.f;   // 1.5 * 2^23: adding it rounds to an integer and leaves that integer in the low mantissa bits
constexpr float kZmax = 125.f;         // |delta A log2 e| up to here: the exponent add cannot leave the normal range

// 2^f on |f| <= 1/2, near-minimax fit of (2^f - 1) / f (relative error 2e-9 before rounding)
__device__ __forceinline__ v2f exp2_poly(const v2f f) {
    v2f p = splat(1.5353427443187684e-4f);
    p = fma2(p, f, splat(1.339887734502554e-3f));
    p = fma2(p, f, splat(9.61843691766262e-3f));
    p = fma2(p, f, splat(5.5503323674201965e-2f));
    p = fma2(p, f, splat(2.4022647738456726e-1f));
    p = fma2(p, f, splat(6.931471824645996e-1f));
    return fma2(p, f, splat(1.f));
}

// a = exp(dl A) for a pair of states, A2 = A log2 e.  ROBUST false: the caller has checked |dl A2| <= kZmax for the whole
// wave (a scalar branch), so the exponent is added to the bits directly.  z = dl A2 is never rounded: f = fma(dl, A2, -n).
template <bool ROBUST>
__device__ __forceinline__ v2f decay2(const float dl, const v2f A2) {
    if constexpr (!ROBUST) {
        const v2f t = fma2(splat(dl), A2, splat(kMagic));
        const v2f nf = t - splat(kMagic);
        const v2f p = exp2_poly(fma2(splat(dl), A2, -nf));
        v2f r;
        r.x = __int_as_float(__float_as_int(p.x) + (__float_as_int(t.x) << 23));
        r.y = __int_as_float(__float_as_int(p.y) + (__float_as_int(t.y) << 23));
        return r;
    } else {   // any finite argument: clamp, ldexp (underflows to 0, overflows to inf as exp does)
        v2f z = splat(dl) * A2;
        z.x = __builtin_amdgcn_fmed3f(z.x, -160.f, 160.f);
        z.y = __builtin_amdgcn_fmed3f(z.y, -160.f, 160.f);
        const v2f nf = (z + splat(kMagic)) - splat(kMagic);
        const v2f p = exp2_poly(z - nf);
        return (v2f){ldexpf(p.x, (int)nf.x), ldexpf(p.y, (int)nf.y)};
    }
}

// ---- scans of the lane aggregates of TWO independent recurrences (the two states of a pair) ---------------------------
// One Hillis-Steele stage for both: b <- a b_src + b, a <- a a_src (lanes without a source lane keep their value: the
// identity the scan needs).  Order b0 b1 a0 a1: every DPP read is at least two instructions behind the write of its source.
#define VMASR_SCAN2_STAGE(CTRL)                                     \
    "v_fmac_f32_dpp %0, %0, %1 " CTRL "\n\t"                        \
    "v_fmac_f32_dpp %2, %2, %3 " CTRL "\n\t"                        \
    "v_mul_f32_dpp %1, %1, %1 " CTRL "\n\t"                         \
    "v_mul_f32_dpp %3, %3, %3 " CTRL "\n\t"

struct Pair2 {
    v2f a, b;   // h -> a h + b, two states
};

__device__ __forceinline__ Pair2 then2(const Pair2 first, const Pair2 second) {
    return {second.a * first.a, fma2(second.a, first.b, second.b)};
}

// forward: excl = composition of lanes [0, lane), tot = all lanes (wave-uniform)
__device__ __forceinline__ void wave_scan_fwd2(const Pair2 v, Pair2 &excl, Pair2 &tot) {
    float a0 = v.a.x, b0 = v.b.x, a1 = v.a.y, b1 = v.b.y;
    asm volatile("s_nop 1\n\t"
                 VMASR_SCAN2_STAGE("row_shr:1 row_mask:0xf bank_mask:0xf")
                 VMASR_SCAN2_STAGE("row_shr:2 row_mask:0xf bank_mask:0xf")
                 VMASR_SCAN2_STAGE("row_shr:4 row_mask:0xf bank_mask:0xf")
                 VMASR_SCAN2_STAGE("row_shr:8 row_mask:0xf bank_mask:0xf")
                 VMASR_SCAN2_STAGE("row_bcast:15 row_mask:0xa bank_mask:0xf")
                 VMASR_SCAN2_STAGE("row_bcast:31 row_mask:0xc bank_mask:0xf")
                 : "+v"(b0), "+v"(a0), "+v"(b1), "+v"(a1));
    tot.a = (v2f){readlane_f(a0, 63), readlane_f(a1, 63)};
    tot.b = (v2f){readlane_f(b0, 63), readlane_f(b1, 63)};
    float ea0 = 1.f, eb0 = 0.f, ea1 = 1.f, eb1 = 0.f;   // exclusive = inclusive one lane down; lane 0 keeps the identity
    asm volatile("s_nop 1\n\t"
                 "v_mov_b32_dpp %0, %4 wave_shr:1 row_mask:0xf bank_mask:0xf\n\t"
                 "v_mov_b32_dpp %1, %5 wave_shr:1 row_mask:0xf bank_mask:0xf\n\t"
                 "v_mov_b32_dpp %2, %6 wave_shr:1 row_mask:0xf bank_mask:0xf\n\t"
                 "v_mov_b32_dpp %3, %7 wave_shr:1 row_mask:0xf bank_mask:0xf"
                 : "+v"(eb0), "+v"(eb1), "+v"(ea0), "+v"(ea1) : "v"(b0), "v"(b1), "v"(a0), "v"(a1));
    excl.a = (v2f){ea0, ea1};
    excl.b = (v2f){eb0, eb1};
}

// reverse (g_l = b_l + a_l g_{l+1}): excl = composition of lanes (lane, 63] applied from the right, tot = all lanes.
// Inside the 16-lane rows by DPP; across the rows (row_bcast only goes upwards) from the row totals read with v_readlane.
__device__ __forceinline__ void wave_scan_rev2(const Pair2 v, const int lane, Pair2 &excl, Pair2 &tot) {
    float a0 = v.a.x, b0 = v.b.x, a1 = v.a.y, b1 = v.b.y;
    asm volatile("s_nop 1\n\t"
                 VMASR_SCAN2_STAGE("row_shl:1 row_mask:0xf bank_mask:0xf")
                 VMASR_SCAN2_STAGE("row_shl:2 row_mask:0xf bank_mask:0xf")
                 VMASR_SCAN2_STAGE("row_shl:4 row_mask:0xf bank_mask:0xf")
                 VMASR_SCAN2_STAGE("row_shl:8 row_mask:0xf bank_mask:0xf")
                 : "+v"(b0), "+v"(a0), "+v"(b1), "+v"(a1));
    const Pair2 t0{(v2f){readlane_f(a0, 0), readlane_f(a1, 0)}, (v2f){readlane_f(b0, 0), readlane_f(b1, 0)}};
    const Pair2 t1{(v2f){readlane_f(a0, 16), readlane_f(a1, 16)}, (v2f){readlane_f(b0, 16), readlane_f(b1, 16)}};
    const Pair2 t2{(v2f){readlane_f(a0, 32), readlane_f(a1, 32)}, (v2f){readlane_f(b0, 32), readlane_f(b1, 32)}};
    const Pair2 t3{(v2f){readlane_f(a0, 48), readlane_f(a1, 48)}, (v2f){readlane_f(b0, 48), readlane_f(b1, 48)}};
    const Pair2 s1 = then2(t3, t2), s0 = then2(s1, t1);   // rows to the right of row 1 / row 0
    tot = then2(s0, t0);
    const int row = lane >> 4;
    Pair2 suf;
    suf.a = row == 3 ? splat(1.f) : (row == 2 ? t3.a : (row == 1 ? s1.a : s0.a));
    suf.b = row == 3 ? splat(0.f) : (row == 2 ? t3.b : (row == 1 ? s1.b : s0.b));
    // in-row exclusive suffix: the value one lane up (identity at the end of a row)
    float ea0 = 1.f, eb0 = 0.f, ea1 = 1.f, eb1 = 0.f;
    asm volatile("s_nop 1\n\t"
                 "v_mov_b32_dpp %0, %4 row_shl:1 row_mask:0xf bank_mask:0xf\n\t"
                 "v_mov_b32_dpp %1, %5 row_shl:1 row_mask:0xf bank_mask:0xf\n\t"
                 "v_mov_b32_dpp %2, %6 row_shl:1 row_mask:0xf bank_mask:0xf\n\t"
                 "v_mov_b32_dpp %3, %7 row_shl:1 row_mask:0xf bank_mask:0xf"
                 : "+v"(eb0), "+v"(eb1), "+v"(ea0), "+v"(ea1) : "v"(b0), "v"(b1), "v"(a0), "v"(a1));
    excl = then2(suf, Pair2{(v2f){ea0, ea1}, (v2f){eb0, eb1}});
}

// ---- geometry ------------------------------------------------------------------------------------------------------------
struct NFwdGeom {
    int tiles_per_task, nseg, np;   // np = state pairs = ceil(N / 2)
};

template <typename T, bool VEC>
__device__ __forceinline__ void load_pair4(const T *__restrict__ base, const int64_t dstate_stride, const int n0, const bool has1,
                                           const int t0, const int L, const bool full, v2f (&v)[kItems]) {
    float x0[kItems], x1[kItems];
    load4u<T, VEC>(base + n0 * dstate_stride, t0, L, x0, full);
    if (has1) {
        load4u<T, VEC>(base + (n0 + 1) * dstate_stride, t0, L, x1, full);
    } else {
#pragma unroll
        for (int i = 0; i < kItems; ++i) x1[i] = 0.f;
    }
#pragma unroll
    for (int i = 0; i < kItems; ++i) v[i] = (v2f){x0[i], x1[i]};
}

// =====================================================================================================================
// forward.  MODE 0 walk (carry-in zero at tile 0, sequential over [tile0, tile1), writes x per tile), 1 apply (carry-in
// from x[tile0 - 1], already scanned by the carry kernel), 2 aggregates only (tile-local (prod a, h_end | h_in = 0) -> x).
// One wave = one row.  Per-wave LDS slice: A log2e [2 np], h [2 np], p [2 np].
// =====================================================================================================================
template <typename T, bool VEC, int MODE>
__global__ __launch_bounds__(256) void sscan_nfwd_kernel(const vmasr_sscan_params p, const NFwdGeom geo) {
    extern __shared__ __attribute__((aligned(16))) float s_dyn[];
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int L = p.seqlen, N = p.dstate, NP2 = 2 * geo.np;
    const int ntiles = (L + kTile - 1) / kTile;
    // tasks: row fastest (the waves of a workgroup share B / C lines), then segment, then batch
    const int task = xcd_remap(blockIdx.x, gridDim.x) * 4 + wave;
    const int d = task % p.dim, rest = task / p.dim, seg = rest % geo.nseg, b = rest / geo.nseg;
    if (b >= p.batch) return;
    const int g = d / (p.dim / p.n_groups);
    const int tile0 = seg * geo.tiles_per_task, tile1 = min(ntiles, tile0 + geo.tiles_per_task);

    float *sA = s_dyn + (size_t)wave * 3 * NP2, *sH = sA + NP2, *sP = sH + NP2;
    const T *__restrict__ Bg = static_cast<const T *>(p.B_ptr) + b * p.B_batch_stride + g * p.B_group_stride;
    const T *__restrict__ Cg = static_cast<const T *>(p.C_ptr) + b * p.C_batch_stride + g * p.C_group_stride;
    const float *__restrict__ Ap = static_cast<const float *>(p.A_ptr) + d * p.A_d_stride;
    float *__restrict__ xp = static_cast<float *>(p.x_ptr);
    const T *__restrict__ u_row = static_cast<const T *>(p.u_ptr) + b * p.u_batch_stride + d * p.u_d_stride;
    const T *__restrict__ dl_row = static_cast<const T *>(p.delta_ptr) + b * p.delta_batch_stride + d * p.delta_d_stride;
    T *__restrict__ out_row = static_cast<T *>(p.out_ptr) + b * p.out_batch_stride + d * p.out_d_stride;
    const float Dv = p.D_ptr ? static_cast<const float *>(p.D_ptr)[d] : 0.f;
    const float bias = p.delta_bias_ptr ? static_cast<const float *>(p.delta_bias_ptr)[d] : 0.f;
    const size_t xrow = ((size_t)b * p.dim + d) * p.n_chunks;   // in chunks

    float amax = 0.f;
    for (int n = lane; n < NP2; n += kWave) {
        const float a2 = n < N ? Ap[n * p.A_dstate_stride] * kLog2e : 0.f;   // the pad state of an odd N: a = 1, B = C = 0
        float h = 0.f, pr = 1.f;
        if (MODE == 1 && tile0 > 0 && n < N) {
            pr = xp[((xrow + tile0 - 1) * N + n) * 2 + 0];
            h = xp[((xrow + tile0 - 1) * N + n) * 2 + 1];
        }
        sA[n] = a2; sH[n] = h; sP[n] = pr;
        amax = fmaxf(amax, fabsf(a2));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) amax = fmaxf(amax, __shfl_xor(amax, o));
    const bool xvec = (N & 1) == 0;   // (…) * N + n0 is even: 16-byte aligned pairs of (p, h)

    for (int tile = tile0; tile < tile1; ++tile) {
        const int t0 = tile * kTile + lane * kItems;
        const bool full = (tile + 1) * kTile <= L;   // wave-uniform
        float uv[kItems], dl[kItems], du[kItems];
        load4u<T, VEC>(u_row, t0, L, uv, full);
        // steps past the end of a ragged tile are identity steps (delta = 0: a = 1, b = 0)
        load4u<T, VEC>(dl_row, t0, L, dl, full, p.delta_softplus ? -INFINITY : -bias);
        float dmax = 0.f;
#pragma unroll
        for (int i = 0; i < kItems; ++i) {
            const float v = dl[i] + bias;
            dl[i] = p.delta_softplus ? softplus_f(v) : v;
            du[i] = dl[i] * uv[i];
            dmax = fmaxf(dmax, fabsf(dl[i]));
        }
        const bool robust = __builtin_amdgcn_ballot_w64(!(dmax * amax <= kZmax)) != 0;   // scalar branch
        v2f y2[kItems];
#pragma unroll
        for (int i = 0; i < kItems; ++i) y2[i] = splat(0.f);

        for (int n0 = 0; n0 < N; n0 += 2) {
            const bool has1 = n0 + 1 < N;
            v2f Bv[kItems], Cv[kItems], a[kItems], bb[kItems];
            load_pair4<T, VEC>(Bg, p.B_dstate_stride, n0, has1, t0, L, full, Bv);
            if (MODE != 2) load_pair4<T, VEC>(Cg, p.C_dstate_stride, n0, has1, t0, L, full, Cv);
            const v2f A2 = *reinterpret_cast<const v2f *>(sA + n0);
            if (robust) {
#pragma unroll
                for (int i = 0; i < kItems; ++i) a[i] = decay2<true>(dl[i], A2);
            } else {
#pragma unroll
                for (int i = 0; i < kItems; ++i) a[i] = decay2<false>(dl[i], A2);
            }
#pragma unroll
            for (int i = 0; i < kItems; ++i) bb[i] = splat(du[i]) * Bv[i];
            Pair2 agg{a[0], bb[0]};
#pragma unroll
            for (int i = 1; i < kItems; ++i) agg = then2(agg, Pair2{a[i], bb[i]});
            Pair2 excl, tot;
            wave_scan_fwd2(agg, excl, tot);
            float *xi = xp + ((xrow + tile) * N + n0) * 2;
            if constexpr (MODE == 2) {
                if (lane == 0) {
                    if (xvec) *reinterpret_cast<float4 *>(xi) = make_float4(tot.a.x, tot.b.x, tot.a.y, tot.b.y);
                    else {
                        *reinterpret_cast<float2 *>(xi) = make_float2(tot.a.x, tot.b.x);
                        if (has1) *reinterpret_cast<float2 *>(xi + 2) = make_float2(tot.a.y, tot.b.y);
                    }
                }
            } else {
                const v2f hin = *reinterpret_cast<const v2f *>(sH + n0);
                v2f h = fma2(excl.a, hin, excl.b);
#pragma unroll
                for (int i = 0; i < kItems; ++i) {
                    h = fma2(a[i], h, bb[i]);
                    y2[i] = fma2(h, Cv[i], y2[i]);
                }
                const v2f hout = fma2(tot.a, hin, tot.b);
                if (lane == 0) {
                    *reinterpret_cast<v2f *>(sH + n0) = hout;
                    if constexpr (MODE == 0) {
                        const v2f pout = tot.a * *reinterpret_cast<const v2f *>(sP + n0);
                        *reinterpret_cast<v2f *>(sP + n0) = pout;
                        if (xvec) *reinterpret_cast<float4 *>(xi) = make_float4(pout.x, hout.x, pout.y, hout.y);
                        else {
                            *reinterpret_cast<float2 *>(xi) = make_float2(pout.x, hout.x);
                            if (has1) *reinterpret_cast<float2 *>(xi + 2) = make_float2(pout.y, hout.y);
                        }
                    }
                }
            }
        }
        if constexpr (MODE != 2) {
            float outv[kItems];
#pragma unroll
            for (int i = 0; i < kItems; ++i) outv[i] = fmaf(Dv, uv[i], y2[i].x + y2[i].y);
            store4u<T, VEC>(out_row, t0, L, outv, full);
        }
    }
}

// =====================================================================================================================
// backward.  MODE 0: the workgroup walks [tile0, tile1) from the right, adjoint carries G in LDS; 1: carry-in per task
// read from ws (scanned by the reverse carry kernel); 2: per-tile reverse aggregates -> ws.
// Workgroup = W waves; wave w owns rows d0 .. d0 + R - 1 of ONE group; the waves walk the same tiles in lockstep.
// Dynamic LDS: per wave 4 R NP2 floats (A log2e | A | G | dA accumulators) + (W > 1) the dB / dC reduction buffer
// [W][PB][4][256] floats.
// =====================================================================================================================
struct NBwdGeom {
    int tiles_per_task, nseg, W, wg_per_group, np, PB;
    unsigned *det;   // deterministic mode: the workgroups run one after the other (common.h), else null
};

template <typename T, int R, bool VEC, int MODE>
__global__ __launch_bounds__(512) void sscan_nbwd_kernel(const vmasr_sscan_bwd_params q, const NBwdGeom geo) {
    const vmasr_sscan_params &p = q.f;
    extern __shared__ __attribute__((aligned(16))) float s_dyn[];
    det_enter(geo.det);
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int W = geo.W, L = p.seqlen, N = p.dstate, NP2 = 2 * geo.np;
    const int ntiles = (L + kTile - 1) / kTile;
    const int rpg = p.dim / p.n_groups;
    // block -> (workgroup-in-group fastest, group, segment, batch)
    int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int wgi = bid % geo.wg_per_group; bid /= geo.wg_per_group;
    const int g = bid % p.n_groups; bid /= p.n_groups;
    const int seg = bid % geo.nseg;
    const int b = bid / geo.nseg;
    const int d0 = g * rpg + (wgi * W + wave) * R;
    const int tile0 = seg * geo.tiles_per_task, tile1 = min(ntiles, tile0 + geo.tiles_per_task);

    float *sA = s_dyn + (size_t)wave * 4 * R * NP2, *sAr = sA + R * NP2, *sG = sAr + R * NP2, *sdA = sG + R * NP2;
    float *red = s_dyn + (size_t)W * 4 * R * NP2;   // [W][PB][4][kTile]

    const T *__restrict__ Bg = static_cast<const T *>(p.B_ptr) + b * p.B_batch_stride + g * p.B_group_stride;
    const T *__restrict__ Cg = static_cast<const T *>(p.C_ptr) + b * p.C_batch_stride + g * p.C_group_stride;
    const float *__restrict__ xp = static_cast<const float *>(p.x_ptr);
    float *__restrict__ ws = static_cast<float *>(q.ws_ptr);
    float *__restrict__ dBg = static_cast<float *>(q.dB_ptr) + ((size_t)b * p.n_groups + g) * N * L;
    float *__restrict__ dCg = static_cast<float *>(q.dC_ptr) + ((size_t)b * p.n_groups + g) * N * L;

    const T *u_row[R], *dl_row[R], *do_row[R];
    T *du_row[R], *dd_row[R];
    float Dv[R], bias[R], amax[R], accD[R], accBias[R];
    const size_t xrow0 = ((size_t)b * p.dim + d0) * p.n_chunks;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int d = d0 + r;
        u_row[r] = static_cast<const T *>(p.u_ptr) + b * p.u_batch_stride + d * p.u_d_stride;
        dl_row[r] = static_cast<const T *>(p.delta_ptr) + b * p.delta_batch_stride + d * p.delta_d_stride;
        do_row[r] = static_cast<const T *>(q.dout_ptr) + b * q.dout_batch_stride + d * q.dout_d_stride;
        du_row[r] = static_cast<T *>(q.du_ptr) + b * q.du_batch_stride + d * q.du_d_stride;
        dd_row[r] = static_cast<T *>(q.ddelta_ptr) + b * q.ddelta_batch_stride + d * q.ddelta_d_stride;
        Dv[r] = p.D_ptr ? static_cast<const float *>(p.D_ptr)[d] : 0.f;
        bias[r] = p.delta_bias_ptr ? static_cast<const float *>(p.delta_bias_ptr)[d] : 0.f;
        accD[r] = 0.f; accBias[r] = 0.f;
        const float *Ap = static_cast<const float *>(p.A_ptr) + d * p.A_d_stride;
        const bool carry = MODE == 1 && tile1 * kTile < L;
        float am = 0.f;
        for (int n = lane; n < NP2; n += kWave) {
            const float A = n < N ? Ap[n * p.A_dstate_stride] : 0.f;
            sA[r * NP2 + n] = A * kLog2e;
            sAr[r * NP2 + n] = A;
            sG[r * NP2 + n] = (carry && n < N) ? ws[((xrow0 + (size_t)r * p.n_chunks + tile1 - 1) * N + n) * 2 + 1] : 0.f;
            sdA[r * NP2 + n] = 0.f;
            am = fmaxf(am, fabsf(A * kLog2e));
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) am = fmaxf(am, __shfl_xor(am, o));
        amax[r] = am;
    }
    const bool xvec = (N & 1) == 0;

    for (int tile = tile1 - 1; tile >= tile0; --tile) {
        const int t0 = tile * kTile + lane * kItems;
        const bool full = (tile + 1) * kTile <= L;   // wave-uniform
        float uv[R][kItems], dl[R][kItems], dov[R][kItems], sig[R][kItems], du_[R][kItems];
        v2f du2[R][kItems], dd2[R][kItems];
        float zmax = 0.f;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            load4u<T, VEC>(dl_row[r], t0, L, dl[r], full, p.delta_softplus ? -INFINITY : -bias[r]);   // identity steps past the end
            load4u<T, VEC>(do_row[r], t0, L, dov[r], full);
            if (MODE != 2) load4u<T, VEC>(u_row[r], t0, L, uv[r], full);
        }
#pragma unroll
        for (int r = 0; r < R; ++r) {
            float dmax = 0.f;
#pragma unroll
            for (int i = 0; i < kItems; ++i) {
                const float v = dl[r][i] + bias[r];
                if (MODE == 2) {
                    dl[r][i] = p.delta_softplus ? softplus_f(v) : v;
                } else if (p.delta_softplus) {
                    softplus_sigmoid_f(v, dl[r][i], sig[r][i]);
                } else {
                    dl[r][i] = v;
                    sig[r][i] = 1.f;
                }
                dmax = fmaxf(dmax, fabsf(dl[r][i]));
                if (MODE != 2) {
                    du_[r][i] = dl[r][i] * uv[r][i];
                    accD[r] = fmaf(dov[r][i], uv[r][i], accD[r]);
                    du2[r][i] = splat(0.f);
                    dd2[r][i] = splat(0.f);
                }
            }
            zmax = fmaxf(zmax, dmax * amax[r]);
        }
        const bool robust = __builtin_amdgcn_ballot_w64(!(zmax <= kZmax)) != 0;   // scalar branch

        for (int pb0 = 0; pb0 < geo.np; pb0 += geo.PB) {
            const int npb = min(geo.PB, geo.np - pb0);
            for (int pp = 0; pp < npb; ++pp) {
                const int n0 = 2 * (pb0 + pp);
                const bool has1 = n0 + 1 < N;
                v2f Bv[kItems], Cv[kItems], dBv[kItems], dCv[kItems];
                load_pair4<T, VEC>(Cg, p.C_dstate_stride, n0, has1, t0, L, full, Cv);
                if (MODE != 2) load_pair4<T, VEC>(Bg, p.B_dstate_stride, n0, has1, t0, L, full, Bv);
#pragma unroll
                for (int i = 0; i < kItems; ++i) { dBv[i] = splat(0.f); dCv[i] = splat(0.f); }
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const v2f A2 = *reinterpret_cast<const v2f *>(sA + r * NP2 + n0);
                    v2f a[kItems], e[kItems], be[kItems];
                    if (robust) {
#pragma unroll
                        for (int i = 0; i < kItems; ++i) a[i] = decay2<true>(dl[r][i], A2);
                    } else {
#pragma unroll
                        for (int i = 0; i < kItems; ++i) a[i] = decay2<false>(dl[r][i], A2);
                    }
                    // adjoint elements (a_t, a_t dout_t C_t), composed against the scan order
#pragma unroll
                    for (int i = 0; i < kItems; ++i) {
                        be[i] = splat(dov[r][i]) * Cv[i];
                        e[i] = a[i] * be[i];
                    }
                    Pair2 ragg{a[kItems - 1], e[kItems - 1]};
#pragma unroll
                    for (int i = kItems - 2; i >= 0; --i) ragg = then2(ragg, Pair2{a[i], e[i]});
                    Pair2 rexcl, rtot;
                    wave_scan_rev2(ragg, lane, rexcl, rtot);
                    const size_t ci = ((xrow0 + (size_t)r * p.n_chunks + tile) * N + n0) * 2;
                    if constexpr (MODE == 2) {
                        if (lane == 0) {
                            if (xvec) *reinterpret_cast<float4 *>(ws + ci) = make_float4(rtot.a.x, rtot.b.x, rtot.a.y, rtot.b.y);
                            else {
                                *reinterpret_cast<float2 *>(ws + ci) = make_float2(rtot.a.x, rtot.b.x);
                                if (has1) *reinterpret_cast<float2 *>(ws + ci + 2) = make_float2(rtot.a.y, rtot.b.y);
                            }
                        }
                        continue;
                    }
                    // forward recurrence of this tile restarted from the saved state
                    v2f hin = splat(0.f);
                    if (tile > 0) {
                        const float *xi = xp + ci - (size_t)N * 2;
                        hin.x = xi[1];
                        if (has1) hin.y = xi[3];
                    }
                    v2f bb[kItems], hv[kItems];
#pragma unroll
                    for (int i = 0; i < kItems; ++i) bb[i] = splat(du_[r][i]) * Bv[i];
                    Pair2 agg{a[0], bb[0]};
#pragma unroll
                    for (int i = 1; i < kItems; ++i) agg = then2(agg, Pair2{a[i], bb[i]});
                    Pair2 excl, tot;
                    wave_scan_fwd2(agg, excl, tot);
                    v2f h = fma2(excl.a, hin, excl.b);
#pragma unroll
                    for (int i = 0; i < kItems; ++i) { h = fma2(a[i], h, bb[i]); hv[i] = h; }
                    // adjoint recurrence inside the lane, against the scan order
                    const v2f Gin = *reinterpret_cast<const v2f *>(sG + r * NP2 + n0);
                    const v2f Ar = *reinterpret_cast<const v2f *>(sAr + r * NP2 + n0);
                    v2f Gn = fma2(rexcl.a, Gin, rexcl.b);   // G of the step right after this lane's last one
                    v2f accA = splat(0.f);
#pragma unroll
                    for (int i = kItems - 1; i >= 0; --i) {
                        const v2f gcur = be[i] + Gn;          // adjoint of h at this step
                        Gn = a[i] * gcur;
                        const v2f gB = gcur * Bv[i];
                        const v2f ax = hv[i] - bb[i];         // a_t h_{t-1}
                        du2[r][i] = fma2(gB, splat(dl[r][i]), du2[r][i]);
                        dd2[r][i] = fma2(gB, splat(uv[r][i]), dd2[r][i]);
                        dd2[r][i] = fma2(gcur * Ar, ax, dd2[r][i]);
                        accA = fma2(gcur * splat(dl[r][i]), ax, accA);
                        dBv[i] = fma2(gcur, splat(du_[r][i]), dBv[i]);
                        dCv[i] = fma2(splat(dov[r][i]), hv[i], dCv[i]);
                    }
                    const float sA0 = wave_sum(accA.x), sA1 = wave_sum(accA.y);
                    const v2f Gout = fma2(rtot.a, Gin, rtot.b);
                    if (lane == 0) {
                        *reinterpret_cast<v2f *>(sG + r * NP2 + n0) = Gout;
                        v2f *acc = reinterpret_cast<v2f *>(sdA + r * NP2 + n0);
                        *acc = *acc + (v2f){sA0, sA1};
                    }
                    __builtin_amdgcn_sched_barrier(0);   // one row at a time: interleaving the rows only multiplies the live registers
                }
                if constexpr (MODE != 2) {
                    if (W == 1) {
                        float *dst[4] = {dBg + (size_t)n0 * L, dBg + (size_t)(n0 + 1) * L, dCg + (size_t)n0 * L, dCg + (size_t)(n0 + 1) * L};
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            if ((k & 1) && !has1) continue;
                            float v4[kItems];
#pragma unroll
                            for (int i = 0; i < kItems; ++i) v4[i] = k == 0 ? dBv[i].x : (k == 1 ? dBv[i].y : (k == 2 ? dCv[i].x : dCv[i].y));
                            if (geo.wg_per_group == 1) {
                                store4u<float, VEC>(dst[k], t0, L, v4, full);
                            } else {
#pragma unroll
                                for (int i = 0; i < kItems; ++i)
                                    if (full || t0 + i < L) atomicAdd(dst[k] + t0 + i, v4[i]);
                            }
                        }
                    } else {
                        float *mine = red + ((size_t)(wave * geo.PB + pp) * 4) * kTile + lane * kItems;
                        *reinterpret_cast<float4 *>(mine) = make_float4(dBv[0].x, dBv[1].x, dBv[2].x, dBv[3].x);
                        *reinterpret_cast<float4 *>(mine + kTile) = make_float4(dBv[0].y, dBv[1].y, dBv[2].y, dBv[3].y);
                        *reinterpret_cast<float4 *>(mine + 2 * kTile) = make_float4(dCv[0].x, dCv[1].x, dCv[2].x, dCv[3].x);
                        *reinterpret_cast<float4 *>(mine + 3 * kTile) = make_float4(dCv[0].y, dCv[1].y, dCv[2].y, dCv[3].y);
                    }
                }
            }
            if constexpr (MODE != 2) {
                if (W > 1) {
                    // sum the W partial tiles of this block of pairs; leave as contiguous runs along the sequence
                    lds_barrier();
                    const int tbase = tile * kTile;
                    for (int eidx = threadIdx.x; eidx < npb * 4 * kTile; eidx += blockDim.x) {
                        const int idx = eidx % kTile, k = (eidx / kTile) % 4, pp = eidx / (4 * kTile);
                        const int n = 2 * (pb0 + pp) + (k & 1);
                        const int t = tbase + idx;
                        if (n >= N || t >= L) continue;
                        float s = 0.f;
                        for (int w = 0; w < W; ++w) s += red[((size_t)(w * geo.PB + pp) * 4 + k) * kTile + idx];
                        float *dst = ((k & 2) ? dCg : dBg) + (size_t)n * L + t;
                        if (geo.wg_per_group == 1) *dst = s; else atomicAdd(dst, s);
                    }
                    lds_barrier();
                }
            }
        }
        if constexpr (MODE != 2) {
#pragma unroll
            for (int r = 0; r < R; ++r) {
                float duv[kItems], ddv[kItems];
#pragma unroll
                for (int i = 0; i < kItems; ++i) {
                    duv[i] = fmaf(Dv[r], dov[r][i], du2[r][i].x + du2[r][i].y);
                    ddv[i] = (dd2[r][i].x + dd2[r][i].y) * sig[r][i];
                    accBias[r] += (full || t0 + i < L) ? ddv[i] : 0.f;
                }
                store4u<T, VEC>(du_row[r], t0, L, duv, full);
                store4u<T, VEC>(dd_row[r], t0, L, ddv, full);
            }
        }
    }
    if constexpr (MODE != 2) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int d = d0 + r;
            const float sD = wave_sum(accD[r]), sB = wave_sum(accBias[r]);
            if (lane == 0) {
                if (q.dD_ptr) atomicAdd(static_cast<float *>(q.dD_ptr) + d, sD);
                if (q.ddelta_bias_ptr) atomicAdd(static_cast<float *>(q.ddelta_bias_ptr) + d, sB);
            }
            for (int n = lane; n < N; n += kWave)
                atomicAdd(static_cast<float *>(q.dA_ptr) + d * q.dA_d_stride + n * q.dA_dstate_stride, sdA[r * NP2 + n]);
        }
    }
    det_leave(geo.det);
}

}  // namespace

// ---- host side --------------------------------------------------------------------------------------------------------
int sscan_n_fwd(const vmasr_sscan_params &p, int split, int tiles_per_task, int nseg, bool vec, hipStream_t st) {
    const int ntiles = (p.seqlen + kTile - 1) / kTile;
    const int np = (p.dstate + 1) / 2;
    const NFwdGeom geo{tiles_per_task, nseg, np};
    const long ntasks = (long)p.batch * p.dim * nseg;
    const int nblocks = (int)((ntasks + 3) / 4);
    const size_t smem = (size_t)4 * 3 * 2 * np * sizeof(float);
    // algorithmic bytes: read u, delta, B, C; write out
    const double es = p.dtype == VMASR_F32 ? 4 : 2, KD = p.dim, KN = (double)p.n_groups * p.dstate, BL = (double)p.batch * p.seqlen;
    const double full = (3 * KD + 2 * KN) * BL * es, agg = (2 * KD + KN) * BL * es;
    const double xb = (double)p.batch * p.dim * ntiles * p.dstate * 2 * 4 * 2;
#define VMASR_NFWD(MODE, KID, BYTES)                                                                                             \
    do {                                                                                                                          \
        if (p.dtype == VMASR_F32) {                                                                                               \
            if (vec) VMASR_LAUNCH(KID, BYTES, (sscan_nfwd_kernel<float, true, MODE>), dim3(nblocks), dim3(256), smem, st, p, geo); \
            else VMASR_LAUNCH(KID, BYTES, (sscan_nfwd_kernel<float, false, MODE>), dim3(nblocks), dim3(256), smem, st, p, geo);    \
        } else if (p.dtype == VMASR_F16) {                                                                                        \
            if (vec) VMASR_LAUNCH(KID, BYTES, (sscan_nfwd_kernel<f16_t, true, MODE>), dim3(nblocks), dim3(256), smem, st, p, geo); \
            else VMASR_LAUNCH(KID, BYTES, (sscan_nfwd_kernel<f16_t, false, MODE>), dim3(nblocks), dim3(256), smem, st, p, geo);    \
        } else {                                                                                                                  \
            if (vec) VMASR_LAUNCH(KID, BYTES, (sscan_nfwd_kernel<bf16_t, true, MODE>), dim3(nblocks), dim3(256), smem, st, p, geo); \
            else VMASR_LAUNCH(KID, BYTES, (sscan_nfwd_kernel<bf16_t, false, MODE>), dim3(nblocks), dim3(256), smem, st, p, geo);   \
        }                                                                                                                         \
    } while (0)
    if (split != 1) {
        VMASR_NFWD(0, VMASR_K_SSCAN_FWD, full);
        return check_launch("sscan_fwd(N)");
    }
    VMASR_NFWD(2, VMASR_K_SSCAN_FWD_AGG, agg);
    sscan_launch_carry(false, static_cast<float *>(p.x_ptr), p.batch * p.dim * p.dstate, ntiles, p.dstate, xb, st);
    VMASR_NFWD(1, VMASR_K_SSCAN_FWD_APPLY, full);
#undef VMASR_NFWD
    return check_launch("sscan_fwd(N, split)");
}

namespace {

template <typename T, int R>
int launch_nbwd(const vmasr_sscan_bwd_params &q, int split, int tiles_per_task, int nseg, bool vec, hipStream_t st) {
    const vmasr_sscan_params &p = q.f;
    const int ntiles = (p.seqlen + kTile - 1) / kTile;
    const int np = (p.dstate + 1) / 2;
    const int rbg = (p.dim / p.n_groups) / R;   // row-blocks per group
    static const int env_w = [] { const char *e = getenv("VMASR_NBWD_WAVES"); return e ? atoi(e) : 0; }();
    static const int env_pb = [] { const char *e = getenv("VMASR_NBWD_PB"); return e ? atoi(e) : 0; }();
    int W = env_w > 0 ? env_w : 4;
    if (W > 8) W = 8;
    while (rbg % W) W >>= 1;
    const int PB = W > 1 ? std::min(np, env_pb > 0 ? env_pb : 2) : 1;
    NBwdGeom geo{tiles_per_task, nseg, W, rbg / W, np, PB, nullptr};
    const long nblocks = (long)p.batch * nseg * p.n_groups * geo.wg_per_group;
    const size_t smem_state = (size_t)W * 4 * R * 2 * np * sizeof(float);
    const size_t smem = smem_state + (W > 1 ? (size_t)W * PB * 4 * kTile * sizeof(float) : 0);
    VMASR_REQUIRE(smem <= 160 * 1024, VMASR_EINVAL, "sscan_bwd: d_state %d needs %zu bytes of LDS", p.dstate, smem);
    // algorithmic bytes: read u, delta, dout, B, C; write du, ddelta, dB, dC (dB / dC fp32)
    const double es = sizeof(T), KD = p.dim, KN = (double)p.n_groups * p.dstate, BL = (double)p.batch * p.seqlen;
    const double full = (5 * KD * es + 2 * KN * es + 2 * KN * 4) * BL, agg = (2 * KD + KN) * BL * es;
    const double xb = (double)p.batch * p.dim * ntiles * p.dstate * 2 * 4 * 2;
#define VMASR_NBWD(MODE, KID, BYTES, SM)                                                                                          \
    do {                                                                                                                          \
        if (vec) VMASR_LAUNCH(KID, BYTES, (sscan_nbwd_kernel<T, R, true, MODE>), dim3((int)nblocks), dim3(64 * W), SM, st, q, geo); \
        else VMASR_LAUNCH(KID, BYTES, (sscan_nbwd_kernel<T, R, false, MODE>), dim3((int)nblocks), dim3(64 * W), SM, st, q, geo);    \
    } while (0)
    if (smem > 64 * 1024) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&sscan_nbwd_kernel<T, R, true, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&sscan_nbwd_kernel<T, R, false, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&sscan_nbwd_kernel<T, R, true, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&sscan_nbwd_kernel<T, R, false, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    }
    if (split != 1) {
        geo.det = det_ticket(VMASR_K_SSCAN_BWD);
        VMASR_NBWD(0, VMASR_K_SSCAN_BWD, full, smem);
        return check_launch("sscan_bwd(N)");
    }
    VMASR_NBWD(2, VMASR_K_SSCAN_BWD_AGG, agg, smem_state);
    sscan_launch_carry(true, static_cast<float *>(q.ws_ptr), p.batch * p.dim * p.dstate, ntiles, p.dstate, xb, st);
    geo.det = det_ticket(VMASR_K_SSCAN_BWD_APPLY);
    VMASR_NBWD(1, VMASR_K_SSCAN_BWD_APPLY, full, smem);
#undef VMASR_NBWD
    return check_launch("sscan_bwd(N, split)");
}

template <typename T>
int dispatch_nbwd(const vmasr_sscan_bwd_params &q, int split, int tiles_per_task, int nseg, int rows, bool vec, hipStream_t st) {
    const int rpg = q.f.dim / q.f.n_groups;
    int R = rows > 0 ? rows : 2;
    while (R > 1 && rpg % R) R >>= 1;
    if (R >= 4) return launch_nbwd<T, 4>(q, split, tiles_per_task, nseg, vec, st);
    if (R == 2) return launch_nbwd<T, 2>(q, split, tiles_per_task, nseg, vec, st);
    return launch_nbwd<T, 1>(q, split, tiles_per_task, nseg, vec, st);
}

}  // namespace

int sscan_n_bwd(const vmasr_sscan_bwd_params &q, int split, int tiles_per_task, int nseg, int rows, bool vec, hipStream_t st) {
    switch (q.f.dtype) {
        case VMASR_F32: return dispatch_nbwd<float>(q, split, tiles_per_task, nseg, rows, vec, st);
        case VMASR_F16: return dispatch_nbwd<f16_t>(q, split, tiles_per_task, nseg, rows, vec, st);
        default: return dispatch_nbwd<bf16_t>(q, split, tiles_per_task, nseg, rows, vec, st);
    }
}

}  // namespace vmasr
