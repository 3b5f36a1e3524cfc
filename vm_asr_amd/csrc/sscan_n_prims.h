// sscan_n_prims.h — device primitives of the general-d_state selective scan (sscan_n.hip): packed state-pair arithmetic, the
// lock-step decay, the scans of two recurrences at once.  (Separate so that tools/scan_prims_probe can time them in isolation.)
#pragma once
#include "scan_prims.h"

namespace vmasr {
namespace {

typedef float v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ v2f splat(const float x) { return (v2f){x, x}; }
__device__ __forceinline__ v2f fma2(const v2f a, const v2f b, const v2f c) { return __builtin_elementwise_fma(a, b, c); }

constexpr float kMagic = 12582912.f;   // 1.5 * 2^23: adding it rounds to an integer and leaves that integer in the low mantissa bits
constexpr float kZmax = 125.f;         // |delta A log2 e| up to here: the exponent add cannot leave the normal range

// a_i = exp(dl_i A) for a pair of states and the kItems steps of a lane, A2 = A log2 e.  The four evaluations run in
// LOCK-STEP (coefficient by coefficient): a v_pk_fma_f32 that reads the result of the previous instruction costs a wait
// state, four independent Horner chains side by side cost none.
// ROBUST false: the caller has checked |dl A2| <= kZmax for the whole workgroup (a scalar branch), so the exponent is added to
// the bits directly.  z = dl A2 is never rounded: f = fma(dl, A2, -n).
template <bool ROBUST>
__device__ __forceinline__ void decay2x4(const float (&dl)[kItems], const v2f A2, v2f (&a)[kItems]) {
    v2f t[kItems], f[kItems], p[kItems];
    if constexpr (!ROBUST) {
#pragma unroll
        for (int i = 0; i < kItems; ++i) t[i] = fma2(splat(dl[i]), A2, splat(kMagic));
#pragma unroll
        for (int i = 0; i < kItems; ++i) f[i] = fma2(splat(dl[i]), A2, splat(kMagic) - t[i]);
    } else {   // any finite argument: clamp, ldexp (underflows to 0, overflows to inf as exp does)
#pragma unroll
        for (int i = 0; i < kItems; ++i) {
            v2f z = splat(dl[i]) * A2;
            z.x = __builtin_amdgcn_fmed3f(z.x, -160.f, 160.f);
            z.y = __builtin_amdgcn_fmed3f(z.y, -160.f, 160.f);
            t[i] = z + splat(kMagic);
            f[i] = z - (t[i] - splat(kMagic));
        }
    }
    // 2^f on |f| <= 1/2, near-minimax fit of (2^f - 1) / f (relative error 2e-9 before rounding)
#pragma unroll
    for (int i = 0; i < kItems; ++i) p[i] = fma2(splat(1.5353427443187684e-4f), f[i], splat(1.339887734502554e-3f));
#pragma unroll
    for (int i = 0; i < kItems; ++i) p[i] = fma2(p[i], f[i], splat(9.61843691766262e-3f));
#pragma unroll
    for (int i = 0; i < kItems; ++i) p[i] = fma2(p[i], f[i], splat(5.5503323674201965e-2f));
#pragma unroll
    for (int i = 0; i < kItems; ++i) p[i] = fma2(p[i], f[i], splat(2.4022647738456726e-1f));
#pragma unroll
    for (int i = 0; i < kItems; ++i) p[i] = fma2(p[i], f[i], splat(6.931471824645996e-1f));
#pragma unroll
    for (int i = 0; i < kItems; ++i) p[i] = fma2(p[i], f[i], splat(1.f));
#pragma unroll
    for (int i = 0; i < kItems; ++i) {
        if constexpr (!ROBUST) {
            a[i].x = __int_as_float(__float_as_int(p[i].x) + (__float_as_int(t[i].x) << 23));
            a[i].y = __int_as_float(__float_as_int(p[i].y) + (__float_as_int(t[i].y) << 23));
        } else {
            a[i].x = ldexpf(p[i].x, __float_as_int(t[i].x) - __float_as_int(kMagic));
            a[i].y = ldexpf(p[i].y, __float_as_int(t[i].y) - __float_as_int(kMagic));
        }
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
template <bool TOT = true>
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
    if constexpr (TOT) {
        tot.a = (v2f){readlane_f(a0, 63), readlane_f(a1, 63)};
        tot.b = (v2f){readlane_f(b0, 63), readlane_f(b1, 63)};
    }
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
// Inside the 16-lane rows by DPP.  Across the rows row_bcast only goes upwards, so the rows are closed from the right in three
// steps: row 2 composes with the total of row 3 (v_readlane of lane 48), row 1 with the suffix at lane 32, row 0 with the one at
// lane 16 — each step two FMAs and two multiplies under the row's exec mask, with the totals as scalar operands.
__device__ __forceinline__ void wave_scan_rev2(const Pair2 v, const int lane, Pair2 &excl, Pair2 &tot) {
    float a0 = v.a.x, b0 = v.b.x, a1 = v.a.y, b1 = v.b.y;
    asm volatile("s_nop 1\n\t"
                 VMASR_SCAN2_STAGE("row_shl:1 row_mask:0xf bank_mask:0xf")
                 VMASR_SCAN2_STAGE("row_shl:2 row_mask:0xf bank_mask:0xf")
                 VMASR_SCAN2_STAGE("row_shl:4 row_mask:0xf bank_mask:0xf")
                 VMASR_SCAN2_STAGE("row_shl:8 row_mask:0xf bank_mask:0xf")
                 : "+v"(b0), "+v"(a0), "+v"(b1), "+v"(a1));
    const int row = lane >> 4;
#pragma unroll
    for (int src = 48; src >= 16; src -= 16) {   // the suffix right of row (src / 16 - 1) sits at lane src
        const float ta0 = readlane_f(a0, src), tb0 = readlane_f(b0, src), ta1 = readlane_f(a1, src), tb1 = readlane_f(b1, src);
        if (row == src / 16 - 1) {
            b0 = fmaf(a0, tb0, b0); a0 *= ta0;
            b1 = fmaf(a1, tb1, b1); a1 *= ta1;
        }
    }
    tot.a = (v2f){readlane_f(a0, 0), readlane_f(a1, 0)};
    tot.b = (v2f){readlane_f(b0, 0), readlane_f(b1, 0)};
    float ea0 = 1.f, eb0 = 0.f, ea1 = 1.f, eb1 = 0.f;   // exclusive = inclusive one lane up; lane 63 keeps the identity
    asm volatile("s_nop 1\n\t"
                 "v_mov_b32_dpp %0, %4 wave_shl:1 row_mask:0xf bank_mask:0xf\n\t"
                 "v_mov_b32_dpp %1, %5 wave_shl:1 row_mask:0xf bank_mask:0xf\n\t"
                 "v_mov_b32_dpp %2, %6 wave_shl:1 row_mask:0xf bank_mask:0xf\n\t"
                 "v_mov_b32_dpp %3, %7 wave_shl:1 row_mask:0xf bank_mask:0xf"
                 : "+v"(eb0), "+v"(eb1), "+v"(ea0), "+v"(ea1) : "v"(b0), "v"(b1), "v"(a0), "v"(a1));
    excl.a = (v2f){ea0, ea1};
    excl.b = (v2f){eb0, eb1};
}

}  // namespace
}  // namespace vmasr
