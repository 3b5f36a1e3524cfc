// scan_prims.h — wave64 primitives shared by the selective-scan kernels (sscan.hip, ss2d.hip), gfx950.
//
// The recurrence h_t = a_t h_{t-1} + b_t is a scan over the monoid (a,b); 64 lane aggregates are scanned inside
// each 16-lane DPP row with row_shr / row_shl moves and across the four rows with row_bcast / v_readlane
// (no LDS traffic, no ds_bpermute).
#pragma once
#include "common.h"

namespace vmasr {
namespace {

constexpr int kItems = 4;
constexpr int kTile = kWave * kItems;  // 256 == VMASR_SSCAN_CHUNK
constexpr float kLog2e = 1.4426950408889634f;
constexpr int kMaxDState = 256;
constexpr int kMaxBlockWaves = 16;
static_assert(kTile == VMASR_SSCAN_CHUNK, "tile must equal the saved-state chunk");

// Workgroup barrier for LDS hand-offs only: waits for this wave's LDS traffic, not for its
// outstanding global loads and stores (__syncthreads() carries a full fence = s_waitcnt vmcnt(0)).
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

struct Pair {
    float a, b;  // h -> a*h + b
};

// apply `first`, then `second`
__device__ __forceinline__ Pair then(Pair first, Pair second) {
    return {second.a * first.a, fmaf(second.a, first.b, second.b)};
}

template <int CTRL>
__device__ __forceinline__ float dpp_mov(float old, float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(old), __float_as_int(v), CTRL, 0xf, 0xf, false));
}
constexpr int kRowShr = 0x110, kRowShl = 0x100;

template <int CTRL>
__device__ __forceinline__ Pair dpp_pair(Pair v) {  // lanes without a source in their row get the identity
    return {dpp_mov<CTRL>(1.f, v.a), dpp_mov<CTRL>(0.f, v.b)};
}

__device__ __forceinline__ Pair lane_pair(Pair v, int lane) { return {readlane_f(v.a, lane), readlane_f(v.b, lane)}; }

// One Hillis-Steele stage inside the 16-lane rows, v <- then(v shifted by SH lanes, v), as TWO DPP-modified
// VALU instructions: lanes without a source lane in their row are disabled by the DPP control (bound_ctrl 0),
// i.e. keep v — exactly the identity the scan needs.  The compiler's own form of this (update_dpp + mul + fma)
// is 7 instructions per stage: two identity moves, a nop, two DPP moves, mul, fmac; the kernels are
// VALU-issue bound.  `s_nop 1`: the two wait states a DPP read needs after a VALU write of its source.
#define VMASR_SCAN_STAGE(CTRL, SH)                                                                              \
    asm volatile("s_nop 1\n\t"                                                                                  \
                 "v_fmac_f32_dpp %0, %0, %1 " CTRL ":" #SH " row_mask:0xf bank_mask:0xf\n\t"                      \
                 "v_mul_f32_dpp %1, %1, %1 " CTRL ":" #SH " row_mask:0xf bank_mask:0xf"                           \
                 : "+v"(v.b), "+v"(v.a))

// Forward scan of the 64 lane aggregates: `excl` = composition of lanes [0, lane), `total` =
// composition of all lanes (wave-uniform).
__device__ __forceinline__ void wave_scan_fwd(Pair v, int lane, Pair &excl, Pair &total) {
    VMASR_SCAN_STAGE("row_shr", 1);   // b <- a*b_prev + b first (uses the old a), then a <- a*a_prev
    VMASR_SCAN_STAGE("row_shr", 2);
    VMASR_SCAN_STAGE("row_shr", 4);
    VMASR_SCAN_STAGE("row_shr", 8);   // inclusive inside each 16-lane row
    // across the rows with the GFX9 broadcast controls: lane 15 of each row into rows 1 and 3, then lane 31
    // into rows 2 and 3 (row_mask selects the receiving rows) -> inclusive scan of the whole wave in 12 VALU
    // instructions, instead of 8 readlanes + 3 compositions + 6 selects for the row prefixes
    asm volatile("s_nop 1\n\t"
                 "v_fmac_f32_dpp %0, %0, %1 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
                 "v_mul_f32_dpp %1, %1, %1 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
                 "s_nop 1\n\t"
                 "v_fmac_f32_dpp %0, %0, %1 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
                 "v_mul_f32_dpp %1, %1, %1 row_bcast:31 row_mask:0xc bank_mask:0xf"
                 : "+v"(v.b), "+v"(v.a));
    total = lane_pair(v, 63);
    // exclusive = inclusive shifted by one lane across the wave; lane 0 keeps the identity
    Pair e{1.f, 0.f};
    asm volatile("s_nop 1\n\t"
                 "v_mov_b32_dpp %0, %2 wave_shr:1 row_mask:0xf bank_mask:0xf\n\t"
                 "v_mov_b32_dpp %1, %3 wave_shr:1 row_mask:0xf bank_mask:0xf"
                 : "+v"(e.a), "+v"(e.b) : "v"(v.a), "v"(v.b));
    excl = e;
    (void)lane;
}

// Reverse scan (g_i = b_i + a_i g_{i+1}): `excl` = composition of lanes (lane, 63] applied from the
// right, `total` = all lanes.
__device__ __forceinline__ void wave_scan_rev(Pair v, int lane, Pair &excl, Pair &total) {
    VMASR_SCAN_STAGE("row_shl", 1);
    VMASR_SCAN_STAGE("row_shl", 2);
    VMASR_SCAN_STAGE("row_shl", 4);
    VMASR_SCAN_STAGE("row_shl", 8);   // suffix-inclusive inside each row
    const Pair t0 = lane_pair(v, 0), t1 = lane_pair(v, 16), t2 = lane_pair(v, 32), t3 = lane_pair(v, 48);
    const Pair s1 = then(t3, t2), s0 = then(s1, t1);  // rows to the right of row 1 / row 0
    total = then(s0, t0);
    const int row = lane >> 4;
    const Pair suf = row == 3 ? Pair{1.f, 0.f} : (row == 2 ? t3 : (row == 1 ? s1 : s0));
    excl = then(suf, dpp_pair<kRowShl + 1>(v));
}

// value of lane+1 (lane 63 gets `last`)
__device__ __forceinline__ float shift_from_next_lane(float v, int lane, float last) {
    float r = dpp_mov<kRowShl + 1>(0.f, v);
    const float f16 = readlane_f(v, 16), f32 = readlane_f(v, 32), f48 = readlane_f(v, 48);
    r = lane == 15 ? f16 : r;
    r = lane == 31 ? f32 : r;
    r = lane == 47 ? f48 : r;
    return lane == 63 ? last : r;
}

// softplus with the reference's threshold (cus/selective_scan_fwd_kernel.cuh:115-118).
// log1p(e) without the libm call: a 4-term series below 2^-6, Kahan's log(u)*e/(u-1) above.
// ln(u) for u in [1, 2^29]: v_log_f32 (log2, 1 ulp) times ln 2.  HIP's __logf expands to the denormal-safe
// OCML sequence (range check, ldexp, v_log_f32, a 4-instruction split multiply by ln 2, inf fix-up: 12
// instructions) — none of which this argument range needs.
__device__ __forceinline__ float ln_fast(float u) { return __builtin_amdgcn_logf(u) * 0.6931471805599453f; }

// Decay of one step, a = exp(delta A)  (cus/selective_scan_fwd_kernel.cuh:125-127 uses exp2f(delta A log2e) under
// --use_fast_math; the CPU reference torch.exp).  NOT v_exp_f32: the hardware exp2 is a 1-ulp approximation, and for
// the a = 0.9 .. 0.999 of this model an ulp of a is 1e-5 .. 1e-3 of (1 - a), the quantity the recurrence actually
// depends on.  Measured on MI355X against float64 (tools/accuracy_probe.py, profiles/r03_accuracy_probe.log): with
// v_exp_f32 the fused core's output is 1.7x and its d(A), d(dt_bias) gradients 10x further from float64 than the
// sequential fp32 recurrence of selective_scan_ref; with a correctly rounded exp they are equal.  This is a standard
// Cody-Waite reduction + degree-6 polynomial in FMAs: <= 0.93 ulp, rms 0.29 ulp (correct rounding: 0.29), unbiased;
// 12 full-rate VALU instructions instead of v_mul + the quarter-rate v_exp_f32.
#if defined(VMASR_PRECISE_DECAY)     // accuracy experiments: correctly rounded through float64
__device__ __forceinline__ float decay_f(float dl, float A) { return (float)exp((double)dl * (double)A); }
#elif defined(VMASR_FAST_DECAY)      // round 1-2 behaviour, for A/B measurements
__device__ __forceinline__ float decay_f(float dl, float A) { return __builtin_amdgcn_exp2f(dl * (A * kLog2e)); }
#else
__device__ __forceinline__ float decay_poly(const float r) {   // e^r on |r| <= ln2 / 2: near-minimax fit of (e^r - 1 - r) / r^2
    float p = 0.0013933643931522965f;
    p = fmaf(p, r, 0.008363175205886364f);
    p = fmaf(p, r, 0.04166646674275398f);
    p = fmaf(p, r, 0.16666576266288757f);
    p = fmaf(p, r, 0.5f);
    p = fmaf(p, r, 1.f);
    return fmaf(p, r, 1.f);
}
__device__ __forceinline__ float decay_f(float dl, float A) {
    const float z = fmaxf(dl * A, -104.f);             // exp(-104) < the smallest denormal; also keeps n finite
    // wave-uniform fast path: with the model's delta = 1e-3 .. 1e-1 and A ~ -1 every lane has |z| < ln2 / 2, i.e. n = 0 —
    // no range reduction, no ldexp (7 instead of 13 VALU instructions; the branch is scalar)
    if (__builtin_amdgcn_ballot_w64(z < -0.34657359f) == 0) return decay_poly(z);
    const float n = __builtin_rintf(z * kLog2e);
    float r = fmaf(n, -0.693145751953125f, z);          // ln 2 = hi + lo, hi exact in 12 bits: n * hi is exact
    r = fmaf(n, -1.428606765330187e-06f, r);
    return ldexpf(decay_poly(r), (int)n);
}
#endif

#ifdef VMASR_PRECISE_SOFTPLUS
__device__ __forceinline__ float softplus_f(float x) { return x <= 20.f ? (float)log1p(exp((double)x)) : x; }
__device__ __forceinline__ void softplus_sigmoid_f(float x, float &sp, float &sig) {
    const double e = exp((double)fminf(x, 20.f));
    sp = x <= 20.f ? (float)log1p(e) : x;
    sig = x <= 20.f ? (float)(e / (1.0 + e)) : 1.f;
}
#else
__device__ __forceinline__ float softplus_f(float x) {
    const float e = __expf(fminf(x, 20.f));
    const float u = 1.f + e;
    // e / (u - 1) as e * rcp(u - 1): one v_rcp_f32 instead of the ~12-instruction IEEE division sequence
    // (the kernels are VALU-issue bound); u == 1 gives inf/NaN here, discarded by the select below
    const float big = ln_fast(u) * (e * __builtin_amdgcn_rcpf(u - 1.f));
    const float small = e * fmaf(e, fmaf(e, fmaf(e, -0.25f, 0.33333334f), -0.5f), 1.f);
    const float sp = e < 0.015625f ? small : big;
    return x <= 20.f ? sp : x;
}

// softplus and its derivative sigmoid(x) = e / (1 + e) from ONE exponential (backward pass)
__device__ __forceinline__ void softplus_sigmoid_f(float x, float &sp, float &sig) {
    const float e = __expf(fminf(x, 20.f));
    const float u = 1.f + e;
    const float big = ln_fast(u) * (e * __builtin_amdgcn_rcpf(u - 1.f));
    const float small = e * fmaf(e, fmaf(e, fmaf(e, -0.25f, 0.33333334f), -0.5f), 1.f);
    const float s = e < 0.015625f ? small : big;
    sp = x <= 20.f ? s : x;
    sig = x <= 20.f ? e * __builtin_amdgcn_rcpf(u) : 1.f;
}
#endif

}  // namespace
}  // namespace vmasr
