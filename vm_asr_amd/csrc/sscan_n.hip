// sscan_n.hip — selective scan for a general state dimension (d_state N > 1), forward and backward, gfx950 (wave64).
//
// Same operator as sscan.hip (selective_scan_cuda_core.{fwd,bwd}: cus/selective_scan.cpp:157-349; kernels
// cus/selective_scan_fwd_kernel.cuh:101-158 — the loop over states at :124 — and cus/selective_scan_bwd_kernel.cuh:125-241),
// for the calls where every row carries N states (BASELINE configs[4]: `MODEL.VSSM.SSM_D_STATE 32`).  The work is
// N * L * rows state-steps of ~14 (forward) / ~35 (backward) arithmetic instructions each against 12-20 bytes per ROW-step:
// VALU-issue bound by a factor of N, not HBM bound.  The mapping is chosen for instructions per state-step and for enough
// independent waves without extra passes:
//
//   * lanes = time (64 lanes x 4 consecutive steps = the 256-step tile of sscan.hip; the sums over states (y, du, ddelta)
//     stay inside a lane), states are walked in PAIRS held in one 64-bit register pair: decay, recurrence and every
//     gradient product run as v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32 — two states per instruction;
//   * ONE decay exp(delta A) per (row, step, state), shared by the forward recompute and the adjoint recurrence of
//     the backward: 2^(delta A log2 e) with the integer part split off by the 1.5*2^23 trick, a degree-6 polynomial
//     on [-1/2, 1/2] in packed FMAs and the exponent added with v_lshl_add_u32 — 5.5 instructions per state-step
//     (sscan.hip's scalar form: 7-13);
//   * the 64 lane aggregates of the two states of a pair are scanned TOGETHER: the DPP stages of the two states are
//     interleaved, which fills the two wait states a DPP read needs after the VALU write of its source (no s_nop);
//   * PAIR-OWNER WAVES: the W waves of a workgroup own different state pairs of the SAME rows and tile.  A wave loads
//     the B / C tile of its pairs once per tile into registers and keeps it while the workgroup walks a block of RB rows
//     of the group (B / C traffic / RB, no LDS staging), and in the backward it sums dB / dC of its pairs over those rows
//     in registers (no cross-wave reduction for them at all).  What is summed over STATES (y; du, ddelta) is reduced
//     across the waves through LDS once per batch of rows, by the threads that also did that batch's softplus: the
//     per-row prologue (loads, softplus / sigmoid, delta u) runs once per position, not once per wave;
//   * the workgroup walks its tiles sequentially (forward upwards, backward downwards) with the running state of every
//     (row, state) in LDS: rows x pairs x W waves give the parallelism, so no aggregate pass is needed unless a call has
//     very few rows (the 1024x512 output block: 2 rows per group), which takes the 3-phase split (aggregates -> carry
//     kernel -> apply) of sscan.hip with the same kernels;
//   * the adjoint uses G_t = a_t g_t (ss2d.hip): G_t = a_t (dout_t C_t + G_{t+1}) needs only the step's own decay, so a
//     tile needs one scalar per state from its right neighbour.
// Numerics: fp32, same recurrence, softplus and association as sscan.hip; the decay is within 1.3 * 2^-24 relative of
// exp (rms 0.45), unbiased (tests/test_gpu_kernels.py: goldens n8 / n32, oracle, L = 524 288 stress).
#include "sscan_n.h"

#include "sscan_n_prims.h"

#include <stdlib.h>

#include <algorithm>

namespace vmasr {
namespace {

// ---- geometry ------------------------------------------------------------------------------------------------------------
struct PGeom {
    int tiles_per_task, nseg;   // segments along L (split plan), tiles per segment
    int np;                     // state pairs = ceil(N / 2)
    int W;                      // waves per workgroup = ceil(np / PP): wave w owns pairs w PP .. w PP + PP - 1
    int RB, RS, nrb;            // rows per workgroup, rows per batch (prologue / reduction granule), row blocks per group
    int dA_lanes;               // backward: dA partial sums per LANE in LDS ([W][RB][64] pairs), summed once at the end
    unsigned *det;              // deterministic mode: the workgroups run one after the other (common.h), else null
};


__host__ __device__ inline int align4(int v) { return (v + 3) & ~3; }

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

__device__ __forceinline__ void store_pair(float *xi, const bool xvec, const bool has1, const v2f a, const v2f b) {
    if (xvec) {
        *reinterpret_cast<float4 *>(xi) = make_float4(a.x, b.x, a.y, b.y);
    } else {
        *reinterpret_cast<float2 *>(xi) = make_float2(a.x, b.x);
        if (has1) *reinterpret_cast<float2 *>(xi + 2) = make_float2(a.y, b.y);
    }
}

// =====================================================================================================================
// forward.  MODE 0 walk (carry-in zero at tile 0, sequential over [tile0, tile1), writes x per tile), 1 apply (carry-in
// from x[tile0 - 1], already scanned by the carry kernel), 2 aggregates only (tile-local (prod a, h_end | h_in = 0) -> x).
// Workgroup = W pair-owner waves x RB rows of one (batch, group) x a segment of tiles.
// LDS (floats): A log2e [RB][2 np] | h [RB][2 np] | p [RB][2 np] | bias, D, max|A log2e| [RB] | stage [RS][2][256] (delta,
// delta u) | ypart [RS][W][256].
// =====================================================================================================================
// Q: positions per thread in the prologue / epilogue of a batch = ceil(RS 256 / threads): 1 for W >= 4 waves, 2 for W = 2..3, 4 for W = 1
template <typename T, bool VEC, int PP, int Q, int RSV, int MODE>
__global__ __launch_bounds__(PP == 1 ? 1024 : 512) void sscan_pfwd_kernel(const vmasr_sscan_params p, const PGeom geo) {
    extern __shared__ __attribute__((aligned(16))) float s_dyn[];
    const int tid = threadIdx.x, lane = tid & (kWave - 1), nthr = blockDim.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int L = p.seqlen, N = p.dstate, NP2 = 2 * geo.np, RB = geo.RB, W = geo.W;
    constexpr int RS = RSV;    // rows per batch: compile-time, so that the rows of a batch are unrolled side by side (two independent
                               // dependency chains per wave: the scan and the recurrences are serial inside a row)
    const int ntiles = (L + kTile - 1) / kTile;
    const int D = p.dim / p.n_groups;
    // block -> (row block fastest: the workgroups that read the same B / C tiles are neighbours, group, segment, batch)
    int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int rb = bid % geo.nrb; bid /= geo.nrb;
    const int g = bid % p.n_groups; bid /= p.n_groups;
    const int seg = bid % geo.nseg;
    const int b = bid / geo.nseg;
    const int d0 = g * D + rb * RB;
    const int tile0 = seg * geo.tiles_per_task, tile1 = min(ntiles, tile0 + geo.tiles_per_task);

    float *sA = s_dyn, *sH = sA + RB * NP2, *sP = sH + RB * NP2, *sbias = sP + RB * NP2, *sDk = sbias + RB, *samax = sDk + RB;
    float *stage = s_dyn + align4(3 * RB * NP2 + 3 * RB), *ypart = stage + RS * 2 * kTile;
    int *sflag = reinterpret_cast<int *>(ypart + (size_t)RS * W * kTile);   // [2]

    const T *__restrict__ Bg = static_cast<const T *>(p.B_ptr) + b * p.B_batch_stride + g * p.B_group_stride;
    const T *__restrict__ Cg = static_cast<const T *>(p.C_ptr) + b * p.C_batch_stride + g * p.C_group_stride;
    const T *__restrict__ ub = static_cast<const T *>(p.u_ptr) + b * p.u_batch_stride;
    const T *__restrict__ dlb = static_cast<const T *>(p.delta_ptr) + b * p.delta_batch_stride;
    T *__restrict__ ob = static_cast<T *>(p.out_ptr) + b * p.out_batch_stride;
    float *__restrict__ xp = static_cast<float *>(p.x_ptr);
    const size_t xrow0 = ((size_t)b * p.dim + d0) * p.n_chunks;   // in chunks; row r: + r * n_chunks

    for (int e = tid; e < RB * NP2; e += nthr) {
        const int r = e / NP2, n = e % NP2;
        const float A = n < N ? static_cast<const float *>(p.A_ptr)[(d0 + r) * p.A_d_stride + n * p.A_dstate_stride] : 0.f;
        float h = 0.f, pr = 1.f;   // (the pad state of an odd N: a = 1, B = C = 0)
        if (MODE == 1 && tile0 > 0 && n < N) {
            const float *xi = xp + ((xrow0 + (size_t)r * p.n_chunks + tile0 - 1) * N + n) * 2;
            pr = xi[0];
            h = xi[1];
        }
        sA[e] = A * kLog2e; sH[e] = h; sP[e] = pr;
    }
    for (int r = tid; r < RB; r += nthr) {
        sbias[r] = p.delta_bias_ptr ? static_cast<const float *>(p.delta_bias_ptr)[d0 + r] : 0.f;
        sDk[r] = p.D_ptr ? static_cast<const float *>(p.D_ptr)[d0 + r] : 0.f;
        float am = 0.f;
        for (int n = 0; n < N; ++n) am = fmaxf(am, fabsf(static_cast<const float *>(p.A_ptr)[(d0 + r) * p.A_d_stride + n * p.A_dstate_stride]));
        samax[r] = am * kLog2e;
    }
    if (tid < 2) sflag[tid] = 0;
    __syncthreads();
    const bool xvec = (N & 1) == 0;   // (...) * N + n0 is even: 16-byte aligned pairs of (p, h)

    float pre_u[Q], pre_d[Q];
    auto prefetch = [&](const int tile, const int kb) {
#pragma unroll
        for (int qi = 0; qi < Q; ++qi) {
            const int pos = qi * nthr + tid;
            pre_u[qi] = 0.f; pre_d[qi] = 0.f;
            if (pos < RS * kTile) {
                const int r = kb + pos / kTile, t = tile * kTile + pos % kTile;
                if (t < L) {
                    pre_u[qi] = to_f32(ub[(d0 + r) * p.u_d_stride + t]);
                    pre_d[qi] = to_f32(dlb[(d0 + r) * p.delta_d_stride + t]);
                }
            }
        }
    };
    if (tile0 < tile1) prefetch(tile0, 0);
    int batch = 0;

    // B / C of my pairs: this tile's in (Bv, Cv), the next tile's loading into (Bn, Cn) while this one is worked on
    v2f Bv[PP][kItems], Cv[PP][kItems], Bn[PP][kItems], Cn[PP][kItems];
    auto load_bc = [&](const int tile, v2f (&Bd)[PP][kItems], v2f (&Cd)[PP][kItems]) {
        const int t0 = tile * kTile + lane * kItems;
        const bool full = (tile + 1) * kTile <= L;
#pragma unroll
        for (int j = 0; j < PP; ++j) {
            const int n0 = 2 * (wave * PP + j);
            if (n0 < N) {
                load_pair4<T, VEC>(Bg, p.B_dstate_stride, n0, n0 + 1 < N, t0, L, full, Bd[j]);
                if (MODE != 2) load_pair4<T, VEC>(Cg, p.C_dstate_stride, n0, n0 + 1 < N, t0, L, full, Cd[j]);
            }
        }
    };
    if (tile0 < tile1) load_bc(tile0, Bn, Cn);

    for (int tile = tile0; tile < tile1; ++tile) {
        const int t0 = tile * kTile + lane * kItems;
        const bool full = (tile + 1) * kTile <= L;   // wave-uniform
#pragma unroll
        for (int j = 0; j < PP; ++j)
#pragma unroll
            for (int i = 0; i < kItems; ++i) { Bv[j][i] = Bn[j][i]; Cv[j][i] = Cn[j][i]; }
        if (tile + 1 < tile1) load_bc(tile + 1, Bn, Cn);
        for (int kb = 0; kb < RB; kb += RS) {
            // ---- prologue of the batch: delta = softplus(.), delta u, D u — once per position, by whichever thread holds it
            // (its loads were issued one batch ahead: pre_u / pre_d)
            float Du[Q];
            bool flag = false;
#pragma unroll
            for (int qi = 0; qi < Q; ++qi) {
                const int pos = qi * nthr + tid;
                if (pos < RS * kTile) {
                    const int rr = pos / kTile, idx = pos % kTile, r = kb + rr, t = tile * kTile + idx;
                    float u = 0.f, dlv = 0.f;   // steps past the end of a ragged tile are identity steps (delta = 0: a = 1, b = 0)
                    if (t < L) {
                        u = pre_u[qi];
                        const float v = pre_d[qi] + sbias[r];
                        dlv = p.delta_softplus ? softplus_f(v) : v;
                    }
                    stage[(rr * 2 + 0) * kTile + idx] = dlv;
                    stage[(rr * 2 + 1) * kTile + idx] = dlv * u;
                    Du[qi] = sDk[r] * u;
                    flag |= !(fabsf(dlv) * samax[r] <= kZmax);
                }
            }
            {   // loads of the next batch (this tile's next rows, else the next tile's first rows): in flight during the pair work
                int nkb = kb + RS, ntile = tile;
                if (nkb >= RB) { nkb = 0; ntile = tile + 1; }
                if (ntile < tile1) prefetch(ntile, nkb);
            }
            if (__builtin_amdgcn_ballot_w64(flag) != 0 && lane == 0) sflag[batch & 1] = 1;
            lds_barrier();
            const bool robust = sflag[batch & 1] != 0;
            if (tid == 0) sflag[(batch + 1) & 1] = 0;   // the other slot: next set after the second barrier of this batch
            ++batch;
            // ---- my pairs of every row of the batch
#pragma unroll
            for (int rr = 0; rr < RS; ++rr) {
                const int r = kb + rr;
                const float4 dl4 = *reinterpret_cast<const float4 *>(stage + (rr * 2 + 0) * kTile + lane * kItems);
                const float4 du4 = *reinterpret_cast<const float4 *>(stage + (rr * 2 + 1) * kTile + lane * kItems);
                const float dl[kItems] = {dl4.x, dl4.y, dl4.z, dl4.w}, du[kItems] = {du4.x, du4.y, du4.z, du4.w};
                v2f y2[kItems];
#pragma unroll
                for (int i = 0; i < kItems; ++i) y2[i] = splat(0.f);
#pragma unroll
                for (int j = 0; j < PP; ++j) {
                    const int n0 = 2 * (wave * PP + j);
                    if (n0 >= N) continue;
                    const bool has1 = n0 + 1 < N;
                    const v2f A2 = *reinterpret_cast<const v2f *>(sA + r * NP2 + n0);
                    v2f a[kItems], bb[kItems];
                    if (robust) decay2x4<true>(dl, A2, a);
                    else decay2x4<false>(dl, A2, a);
#pragma unroll
                    for (int i = 0; i < kItems; ++i) bb[i] = splat(du[i]) * Bv[j][i];
                    Pair2 agg{a[0], bb[0]};
#pragma unroll
                    for (int i = 1; i < kItems; ++i) agg = then2(agg, Pair2{a[i], bb[i]});
                    Pair2 excl, tot;
                    wave_scan_fwd2(agg, excl, tot);
                    float *xi = xp + ((xrow0 + (size_t)r * p.n_chunks + tile) * N + n0) * 2;
                    if constexpr (MODE == 2) {
                        if (lane == 0) store_pair(xi, xvec, has1, tot.a, tot.b);
                    } else {
                        const v2f hin = *reinterpret_cast<const v2f *>(sH + r * NP2 + n0);
                        v2f h = fma2(excl.a, hin, excl.b);
#pragma unroll
                        for (int i = 0; i < kItems; ++i) {
                            h = fma2(a[i], h, bb[i]);
                            y2[i] = fma2(h, Cv[j][i], y2[i]);
                        }
                        const v2f hout = fma2(tot.a, hin, tot.b);
                        if (lane == 0) {
                            *reinterpret_cast<v2f *>(sH + r * NP2 + n0) = hout;
                            if constexpr (MODE == 0) {
                                const v2f pout = tot.a * *reinterpret_cast<const v2f *>(sP + r * NP2 + n0);
                                *reinterpret_cast<v2f *>(sP + r * NP2 + n0) = pout;
                                store_pair(xi, xvec, has1, pout, hout);
                            }
                        }
                    }
                }
                if constexpr (MODE != 2)
                    *reinterpret_cast<float4 *>(ypart + (size_t)(rr * W + wave) * kTile + lane * kItems) =
                        make_float4(y2[0].x + y2[0].y, y2[1].x + y2[1].y, y2[2].x + y2[2].y, y2[3].x + y2[3].y);
            }
            lds_barrier();
            // ---- epilogue: y = D u + sum over the pair-owner waves
            if constexpr (MODE != 2) {
#pragma unroll
                for (int qi = 0; qi < Q; ++qi) {
                    const int pos = qi * nthr + tid;
                    if (pos < RS * kTile) {
                        const int rr = pos / kTile, idx = pos % kTile, r = kb + rr, t = tile * kTile + idx;
                        float y = Du[qi];
                        const float *yp = ypart + (size_t)rr * W * kTile + idx;
                        int w = 0;
                        for (; w + 4 <= W; w += 4)   // four independent LDS reads in flight
                            y += (yp[(w + 0) * kTile] + yp[(w + 1) * kTile]) + (yp[(w + 2) * kTile] + yp[(w + 3) * kTile]);
                        for (; w < W; ++w) y += yp[w * kTile];
                        if (t < L) ob[(d0 + r) * p.out_d_stride + t] = from_f32<T>(y);
                    }
                }
            }
        }
    }
}

// =====================================================================================================================
// backward.  MODE 0: the workgroup walks [tile0, tile1) from the right, adjoint carries G in LDS; 1: carry-in per task
// read from ws (scanned by the reverse carry kernel); 2: per-tile reverse aggregates -> ws.
// Same workgroup shape as the forward.  LDS (floats): A log2e | A | G | dA accumulators [RB][2 np] each | bias, D,
// max|A log2e| [RB] | dD, ddelta_bias partial sums [RB][4] each | stage [RS][4][256] (delta, u, dout, delta u) |
// part [RS][W][2][256] (du, ddelta summed over the wave's states).
// =====================================================================================================================
template <typename T, bool VEC, int PP, int Q, int MODE>
__global__ __launch_bounds__(PP == 1 ? 1024 : 512) void sscan_pbwd_kernel(const vmasr_sscan_bwd_params q, const PGeom geo) {
    const vmasr_sscan_params &p = q.f;
    extern __shared__ __attribute__((aligned(16))) float s_dyn[];
    if (MODE != 2) det_enter(geo.det);
    const int tid = threadIdx.x, lane = tid & (kWave - 1), nthr = blockDim.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int L = p.seqlen, N = p.dstate, NP2 = 2 * geo.np, RB = geo.RB, RS = geo.RS, W = geo.W;
    const int ntiles = (L + kTile - 1) / kTile;
    const int D = p.dim / p.n_groups;
    int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int rb = bid % geo.nrb; bid /= geo.nrb;
    const int g = bid % p.n_groups; bid /= p.n_groups;
    const int seg = bid % geo.nseg;
    const int b = bid / geo.nseg;
    const int d0 = g * D + rb * RB;
    const int tile0 = seg * geo.tiles_per_task, tile1 = min(ntiles, tile0 + geo.tiles_per_task);

    float *sA = s_dyn, *sAr = sA + RB * NP2, *sG = sAr + RB * NP2, *sdA = sG + RB * NP2;
    float *sbias = sdA + RB * NP2, *sDk = sbias + RB, *samax = sDk + RB, *saccD = samax + RB, *saccB = saccD + 4 * RB;
    float *stage = s_dyn + align4(4 * RB * NP2 + 11 * RB), *part = stage + RS * 4 * kTile;
    int *sflag = reinterpret_cast<int *>(part + (size_t)RS * W * 2 * kTile);   // [2] (+ 2 pad)
    float *xpose = part + (size_t)RS * W * 2 * kTile + 4;                        // [W][256]: transpose buffer of the dB / dC atomics
    float *sdAl = xpose + (geo.nrb > 1 ? (size_t)W * kTile : 0);                 // [W][RB][64][2]: per-lane dA partial sums (geo.dA_lanes)

    const T *__restrict__ Bg = static_cast<const T *>(p.B_ptr) + b * p.B_batch_stride + g * p.B_group_stride;
    const T *__restrict__ Cg = static_cast<const T *>(p.C_ptr) + b * p.C_batch_stride + g * p.C_group_stride;
    const T *__restrict__ ub = static_cast<const T *>(p.u_ptr) + b * p.u_batch_stride;
    const T *__restrict__ dlb = static_cast<const T *>(p.delta_ptr) + b * p.delta_batch_stride;
    const T *__restrict__ dob = static_cast<const T *>(q.dout_ptr) + b * q.dout_batch_stride;
    T *__restrict__ dub = static_cast<T *>(q.du_ptr) + b * q.du_batch_stride;
    T *__restrict__ ddb = static_cast<T *>(q.ddelta_ptr) + b * q.ddelta_batch_stride;
    const float *__restrict__ xp = static_cast<const float *>(p.x_ptr);
    float *__restrict__ ws = static_cast<float *>(q.ws_ptr);
    float *__restrict__ dBg = static_cast<float *>(q.dB_ptr) + ((size_t)b * p.n_groups + g) * N * L;
    float *__restrict__ dCg = static_cast<float *>(q.dC_ptr) + ((size_t)b * p.n_groups + g) * N * L;
    const size_t xrow0 = ((size_t)b * p.dim + d0) * p.n_chunks;

    const bool carry = MODE == 1 && tile1 * kTile < L;
    for (int e = tid; e < RB * NP2; e += nthr) {
        const int r = e / NP2, n = e % NP2;
        const float A = n < N ? static_cast<const float *>(p.A_ptr)[(d0 + r) * p.A_d_stride + n * p.A_dstate_stride] : 0.f;
        sA[e] = A * kLog2e;
        sAr[e] = A;
        sG[e] = (carry && n < N) ? ws[((xrow0 + (size_t)r * p.n_chunks + tile1 - 1) * N + n) * 2 + 1] : 0.f;
        sdA[e] = 0.f;
    }
    for (int r = tid; r < RB; r += nthr) {
        sbias[r] = p.delta_bias_ptr ? static_cast<const float *>(p.delta_bias_ptr)[d0 + r] : 0.f;
        sDk[r] = p.D_ptr ? static_cast<const float *>(p.D_ptr)[d0 + r] : 0.f;
        float am = 0.f;
        for (int n = 0; n < N; ++n) am = fmaxf(am, fabsf(static_cast<const float *>(p.A_ptr)[(d0 + r) * p.A_d_stride + n * p.A_dstate_stride]));
        samax[r] = am * kLog2e;
    }
    for (int e = tid; e < 8 * RB; e += nthr) saccD[e] = 0.f;   // saccD | saccB
    if (geo.dA_lanes)
        for (int e = tid; e < W * RB * 128; e += nthr) sdAl[e] = 0.f;
    if (tid < 2) sflag[tid] = 0;
    __syncthreads();
    const bool xvec = (N & 1) == 0;

    float pre_u[Q], pre_d[Q], pre_y[Q];
    auto prefetch = [&](const int tile, const int kb) {
#pragma unroll
        for (int qi = 0; qi < Q; ++qi) {
            const int pos = qi * nthr + tid;
            pre_u[qi] = 0.f; pre_d[qi] = 0.f; pre_y[qi] = 0.f;
            if (pos < RS * kTile) {
                const int r = kb + pos / kTile, t = tile * kTile + pos % kTile;
                if (t < L) {
                    pre_d[qi] = to_f32(dlb[(d0 + r) * p.delta_d_stride + t]);
                    pre_y[qi] = to_f32(dob[(d0 + r) * q.dout_d_stride + t]);
                    if (MODE != 2) pre_u[qi] = to_f32(ub[(d0 + r) * p.u_d_stride + t]);
                }
            }
        }
    };
    if (tile0 < tile1) prefetch(tile1 - 1, 0);
    int batch = 0;

    // B / C of my pairs for a tile: loaded once per tile, into the SAME registers right after the tile's last unit — in flight
    // behind the dB / dC write-out, the epilogue and the next tile's first prologue (no second buffer: the backward has no registers to spare)
    v2f Bv[PP][kItems], Cv[PP][kItems];
    auto load_bc = [&](const int tile) {
        const int t0 = tile * kTile + lane * kItems;
        const bool full = (tile + 1) * kTile <= L;
#pragma unroll
        for (int j = 0; j < PP; ++j) {
            const int n0 = 2 * (wave * PP + j);
            if (n0 < N) {
                load_pair4<T, VEC>(Cg, p.C_dstate_stride, n0, n0 + 1 < N, t0, L, full, Cv[j]);
                if (MODE != 2) load_pair4<T, VEC>(Bg, p.B_dstate_stride, n0, n0 + 1 < N, t0, L, full, Bv[j]);
            }
        }
    };
    if (tile0 < tile1) load_bc(tile1 - 1);

    for (int tile = tile1 - 1; tile >= tile0; --tile) {
        const int t0 = tile * kTile + lane * kItems;
        const bool full = (tile + 1) * kTile <= L;   // wave-uniform
        v2f dBv[PP][kItems], dCv[PP][kItems];
#pragma unroll
        for (int j = 0; j < PP; ++j)
#pragma unroll
            for (int i = 0; i < kItems; ++i) { dBv[j][i] = splat(0.f); dCv[j][i] = splat(0.f); }
        for (int kb = 0; kb < RB; kb += RS) {
            // ---- prologue: delta = softplus(.), its derivative, delta u — once per position (loads issued one batch ahead)
            float sg[Q], dyq[Q], pdq[Q];
            bool flag = false;
#pragma unroll
            for (int qi = 0; qi < Q; ++qi) {
                const int pos = qi * nthr + tid;
                sg[qi] = 0.f; dyq[qi] = 0.f; pdq[qi] = 0.f;
                if (pos < RS * kTile) {
                    const int rr = pos / kTile, idx = pos % kTile, r = kb + rr, t = tile * kTile + idx;
                    float u = 0.f, dlv = 0.f, dy = 0.f, s = 0.f;   // identity steps past the end of a ragged tile
                    if (t < L) {
                        const float v = pre_d[qi] + sbias[r];
                        dy = pre_y[qi];
                        if (MODE == 2) {
                            dlv = p.delta_softplus ? softplus_f(v) : v;
                        } else {
                            u = pre_u[qi];
                            if (p.delta_softplus) softplus_sigmoid_f(v, dlv, s);
                            else { dlv = v; s = 1.f; }
                        }
                    }
                    float *st = stage + (size_t)rr * 4 * kTile + idx;
                    st[0] = dlv; st[kTile] = u; st[2 * kTile] = dy; st[3 * kTile] = dlv * u;
                    sg[qi] = s; dyq[qi] = dy; pdq[qi] = dy * u;
                    flag |= !(fabsf(dlv) * samax[r] <= kZmax);
                }
            }
            {   // loads of the next batch (this tile's next rows, else the first rows of the tile to the left)
                int nkb = kb + RS, ntile = tile;
                if (nkb >= RB) { nkb = 0; ntile = tile - 1; }
                if (ntile >= tile0) prefetch(ntile, nkb);
            }
            // saved states entering this tile, for my pairs of the rows of this batch (uniform loads, used after the forward scan)
            v2f hsave[2][PP];
            if constexpr (MODE != 2) {
#pragma unroll
                for (int rr = 0; rr < 2; ++rr)
#pragma unroll
                    for (int j = 0; j < PP; ++j) {
                        const int n0 = 2 * (wave * PP + j);
                        hsave[rr][j] = splat(0.f);
                        if (rr < RS && n0 < N && tile > 0) {
                            const float *xi = xp + ((xrow0 + (size_t)(kb + rr) * p.n_chunks + tile - 1) * N + n0) * 2;
                            hsave[rr][j].x = xi[1];
                            if (n0 + 1 < N) hsave[rr][j].y = xi[3];
                        }
                    }
            }
            if (__builtin_amdgcn_ballot_w64(flag) != 0 && lane == 0) sflag[batch & 1] = 1;
            lds_barrier();
            const bool robust = sflag[batch & 1] != 0;
            if (tid == 0) sflag[(batch + 1) & 1] = 0;
            ++batch;
#pragma unroll
            for (int rr = 0; rr < 2; ++rr) {
                if (rr >= RS) break;
                const int r = kb + rr;
                const float *st = stage + (size_t)rr * 4 * kTile + lane * kItems;
                const float4 dl4 = *reinterpret_cast<const float4 *>(st), u4 = *reinterpret_cast<const float4 *>(st + kTile);
                const float4 dy4 = *reinterpret_cast<const float4 *>(st + 2 * kTile), du4 = *reinterpret_cast<const float4 *>(st + 3 * kTile);
                const float dl[kItems] = {dl4.x, dl4.y, dl4.z, dl4.w}, uv[kItems] = {u4.x, u4.y, u4.z, u4.w};
                const float dov[kItems] = {dy4.x, dy4.y, dy4.z, dy4.w}, du_[kItems] = {du4.x, du4.y, du4.z, du4.w};
                v2f du2[kItems], dd2[kItems];
#pragma unroll
                for (int i = 0; i < kItems; ++i) { du2[i] = splat(0.f); dd2[i] = splat(0.f); }
#pragma unroll
                for (int j = 0; j < PP; ++j) {
                    const int n0 = 2 * (wave * PP + j);
                    if (n0 >= N) continue;
                    const bool has1 = n0 + 1 < N;
                    const v2f A2 = *reinterpret_cast<const v2f *>(sA + r * NP2 + n0);
                    v2f a[kItems], e[kItems], be[kItems];
                    if (robust) decay2x4<true>(dl, A2, a);
                    else decay2x4<false>(dl, A2, a);
                    // adjoint elements (a_t, a_t dout_t C_t), composed against the scan order
#pragma unroll
                    for (int i = 0; i < kItems; ++i) {
                        be[i] = splat(dov[i]) * Cv[j][i];
                        e[i] = a[i] * be[i];
                    }
                    Pair2 ragg{a[kItems - 1], e[kItems - 1]};
#pragma unroll
                    for (int i = kItems - 2; i >= 0; --i) ragg = then2(ragg, Pair2{a[i], e[i]});
                    Pair2 rexcl, rtot;
                    wave_scan_rev2(ragg, lane, rexcl, rtot);
                    const size_t ci = ((xrow0 + (size_t)r * p.n_chunks + tile) * N + n0) * 2;
                    if constexpr (MODE == 2) {
                        if (lane == 0) store_pair(ws + ci, xvec, has1, rtot.a, rtot.b);
                        continue;
                    }
                    // forward recurrence of this tile restarted from the saved state
                    const v2f hin = hsave[rr][j];
                    v2f bb[kItems], hv[kItems];
#pragma unroll
                    for (int i = 0; i < kItems; ++i) bb[i] = splat(du_[i]) * Bv[j][i];
                    Pair2 agg{a[0], bb[0]};
#pragma unroll
                    for (int i = 1; i < kItems; ++i) agg = then2(agg, Pair2{a[i], bb[i]});
                    Pair2 excl, tot;
                    wave_scan_fwd2<false>(agg, excl, tot);   // (the end-of-tile state is not needed: x holds it)
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
                        const v2f gB = gcur * Bv[j][i];
                        const v2f ax = hv[i] - bb[i];         // a_t h_{t-1}
                        du2[i] = fma2(gB, splat(dl[i]), du2[i]);
                        dd2[i] = fma2(gB, splat(uv[i]), dd2[i]);
                        dd2[i] = fma2(gcur * Ar, ax, dd2[i]);
                        accA = fma2(gcur * splat(dl[i]), ax, accA);
                        dBv[j][i] = fma2(gcur, splat(du_[i]), dBv[j][i]);
                        dCv[j][i] = fma2(splat(dov[i]), hv[i], dCv[j][i]);
                    }
                    const v2f Gout = fma2(rtot.a, Gin, rtot.b);
                    if (PP == 1 && geo.dA_lanes) {            // my lane's slot of (wave, row): summed over the lanes at the end
                        v2f *slot = reinterpret_cast<v2f *>(sdAl + ((size_t)(wave * RB + r) * kWave + lane) * 2);
                        *slot = *slot + accA;
                        if (lane == 0) *reinterpret_cast<v2f *>(sG + r * NP2 + n0) = Gout;
                    } else {
                        const float sA0 = wave_sum(accA.x), sA1 = wave_sum(accA.y);
                        if (lane == 0) {
                            *reinterpret_cast<v2f *>(sG + r * NP2 + n0) = Gout;
                            v2f *acc = reinterpret_cast<v2f *>(sdA + r * NP2 + n0);
                            *acc = *acc + (v2f){sA0, sA1};
                        }
                    }
                }
                if constexpr (MODE != 2) {
                    float *pt = part + (size_t)(rr * W + wave) * 2 * kTile + lane * kItems;
                    *reinterpret_cast<float4 *>(pt) = make_float4(du2[0].x + du2[0].y, du2[1].x + du2[1].y, du2[2].x + du2[2].y, du2[3].x + du2[3].y);
                    *reinterpret_cast<float4 *>(pt + kTile) = make_float4(dd2[0].x + dd2[0].y, dd2[1].x + dd2[1].y, dd2[2].x + dd2[2].y, dd2[3].x + dd2[3].y);
                }
            }
            lds_barrier();
            // ---- epilogue: du = D dout + sum over states, ddelta = (sum over states) * softplus'; dD, ddelta_bias partials
            if constexpr (MODE != 2) {
#pragma unroll
                for (int qi = 0; qi < Q; ++qi) {
                    const int pos = qi * nthr + tid;
                    if (qi * nthr + wave * kWave < RS * kTile) {   // wave-uniform: the 64 positions of a wave lie in one row
                        const int rr = pos / kTile, idx = pos % kTile, r = kb + rr, t = tile * kTile + idx;
                        float du = sDk[r] * dyq[qi], dd = 0.f;
                        const float *pt = part + (size_t)rr * W * 2 * kTile + idx;
                        int w = 0;
                        for (; w + 4 <= W; w += 4) {   // eight independent LDS reads in flight
                            du += (pt[(w + 0) * 2 * kTile] + pt[(w + 1) * 2 * kTile]) + (pt[(w + 2) * 2 * kTile] + pt[(w + 3) * 2 * kTile]);
                            dd += (pt[(w + 0) * 2 * kTile + kTile] + pt[(w + 1) * 2 * kTile + kTile]) +
                                  (pt[(w + 2) * 2 * kTile + kTile] + pt[(w + 3) * 2 * kTile + kTile]);
                        }
                        for (; w < W; ++w) {
                            du += pt[w * 2 * kTile];
                            dd += pt[w * 2 * kTile + kTile];
                        }
                        dd *= sg[qi];
                        if (t < L) {
                            dub[(d0 + r) * q.du_d_stride + t] = from_f32<T>(du);
                            ddb[(d0 + r) * q.ddelta_d_stride + t] = from_f32<T>(dd);
                        } else {
                            dd = 0.f;
                        }
                        const float sD = wave_sum(pdq[qi]), sB = wave_sum(dd);
                        if (lane == 0) {   // slot (row, wave & 3): written by this wave only
                            saccD[r * 4 + (wave & 3)] += sD;
                            saccB[r * 4 + (wave & 3)] += sB;
                        }
                    }
                }
            }
        }
        if (tile > tile0) load_bc(tile - 1);
        if constexpr (MODE != 2) {
            // dB / dC of my pairs, summed over the RB rows in registers.  One workgroup per group: 16-byte stores.  Otherwise float
            // atomics, transposed through a per-wave LDS kilobyte so that every atomic instruction covers 64 CONSECUTIVE floats
            // (two 128-byte lines) instead of one float out of every four (eight lines): the L2 performs one pass per line.
#pragma unroll
            for (int j = 0; j < PP; ++j) {
                const int n0 = 2 * (wave * PP + j);
                if (n0 >= N) continue;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int n = n0 + (k & 1);
                    if (n >= N) continue;
                    float v4[kItems];
#pragma unroll
                    for (int i = 0; i < kItems; ++i) v4[i] = k == 0 ? dBv[j][i].x : (k == 1 ? dBv[j][i].y : (k == 2 ? dCv[j][i].x : dCv[j][i].y));
                    float *dst = ((k & 2) ? dCg : dBg) + (size_t)n * L;
                    if (geo.nrb == 1) {
                        store4u<float, VEC>(dst, t0, L, v4, full);
                    } else {
                        float *tr = xpose + wave * kTile;
                        *reinterpret_cast<float4 *>(tr + lane * kItems) = make_float4(v4[0], v4[1], v4[2], v4[3]);
                        // (same wave writes and reads: LDS operations of a wave complete in order)
#pragma unroll
                        for (int i = 0; i < kItems; ++i) {
                            const int t = tile * kTile + i * kWave + lane;
                            const float v = tr[i * kWave + lane];
                            if (full || t < L) atomicAdd(dst + t, v);
                        }
                    }
                }
            }
        }
    }
    if constexpr (MODE != 2) {
        if (PP == 1 && geo.dA_lanes) {       // fold the per-lane partial sums of my pair, row by row
            const int n0 = 2 * wave;
            for (int r = 0; r < RB; ++r) {
                const v2f v = *reinterpret_cast<const v2f *>(sdAl + ((size_t)(wave * RB + r) * kWave + lane) * 2);
                const float s0 = wave_sum(v.x), s1 = wave_sum(v.y);
                if (lane == 0 && n0 < N) {
                    sdA[r * NP2 + n0] = s0;
                    sdA[r * NP2 + n0 + 1] = s1;       // (the pad state of an odd N: never read)
                }
            }
        }
        __syncthreads();
        for (int e = tid; e < RB * N; e += nthr) {
            const int r = e / N, n = e % N;
            atomicAdd(static_cast<float *>(q.dA_ptr) + (d0 + r) * q.dA_d_stride + n * q.dA_dstate_stride, sdA[r * NP2 + n]);
        }
        for (int r = tid; r < RB; r += nthr) {
            if (q.dD_ptr) atomicAdd(static_cast<float *>(q.dD_ptr) + d0 + r, (saccD[r * 4] + saccD[r * 4 + 1]) + (saccD[r * 4 + 2] + saccD[r * 4 + 3]));
            if (q.ddelta_bias_ptr)
                atomicAdd(static_cast<float *>(q.ddelta_bias_ptr) + d0 + r, (saccB[r * 4] + saccB[r * 4 + 1]) + (saccB[r * 4 + 2] + saccB[r * 4 + 3]));
        }
        det_leave(geo.det);
    }
}

// ---- launch plan -------------------------------------------------------------------------------------------------------
struct PPlan {
    int PP, W, RB, RS, nrb, nseg, tiles_per_task, split;
};

int g_plan_rb = 0;   // tuning override (VMASR_SSCAN_N_RB): rows per workgroup

PPlan make_pplan(const vmasr_sscan_params &p, int split_req) {
    PPlan pl{};
    const int np = (p.dstate + 1) / 2, D = p.dim / p.n_groups;
    const int ntiles = (p.seqlen + kTile - 1) / kTile;
    const long rows = (long)p.batch * p.dim;
    static const int env_rb = [] { const char *e = getenv("VMASR_SSCAN_N_RB"); return e ? atoi(e) : 0; }();
    static const int env_pp = [] { const char *e = getenv("VMASR_SSCAN_N_PP"); return e ? atoi(e) : 0; }();
    // pairs per wave: one while there are at most 16 pairs (d_state <= 32: up to 16 pair-owner waves), else as many as keep <= 16 waves.
    // (Giving the waves of shallow calls — the output blocks' 2 .. 32 rows per group — 2 or 4 pairs each, to amortise the per-tile costs
    // over more units, was measured 2.5-4x SLOWER: with 4-8 waves per workgroup at ~250 VGPRs the SIMDs hold 2 waves, and this kernel's
    // serial scan / recurrence chains need >= 4 to hide their latency: profiles/r05_scan_n_microbench_pp.log.  VMASR_SSCAN_N_PP forces it.)
    int PP = np <= 16 ? 1 : (np <= 32 ? 2 : 4);      // (d_state <= 128 here; above that sscan.hip's one-state-at-a-time kernels)
    auto plan_rows = [&](int pp) {
        pl.PP = pp;
        pl.W = (np + pp - 1) / pp;
        pl.RS = (pl.W >= 8 && D % 2 == 0) ? 2 : 1;
        const long target = std::max(256L, 4096L / pl.W);          // one 16-wave workgroup per CU, or enough smaller ones for ~4096 waves
        long rb = env_rb > 0 ? env_rb : std::max(1L, rows / target);
        rb = std::min<long>(rb, 64);
        int RB = pl.RS;
        for (int c = pl.RS; c <= D && c <= rb; c += pl.RS)
            if (D % c == 0) RB = c;
        pl.RB = RB;
        pl.nrb = D / RB;
        return target;
    };
    long target = plan_rows(PP);
    if (env_pp > 0 && (env_pp == 1 || env_pp == 2 || env_pp == 4) && (np + env_pp - 1) / env_pp <= 16 && (env_pp == 1 || (np + env_pp - 1) / env_pp >= 4))
        target = plan_rows(env_pp);
    const long wgs = (long)p.batch * p.n_groups * pl.nrb;
    int nseg = 1;
    if (split_req == 1 || (split_req < 0 && wgs * 2 <= target)) nseg = (int)std::min<long>(ntiles, std::max(2L, (target + wgs - 1) / wgs));
    if (ntiles < 2) nseg = 1;
    pl.tiles_per_task = (ntiles + nseg - 1) / nseg;
    pl.nseg = (ntiles + pl.tiles_per_task - 1) / pl.tiles_per_task;
    pl.split = pl.nseg > 1 ? 1 : 0;
    if (split_req == 1 && ntiles >= 2) pl.split = 1;
    return pl;
}

size_t fwd_lds_floats(const PPlan &pl, int np) { return align4(3 * pl.RB * 2 * np + 3 * pl.RB) + (size_t)pl.RS * 2 * kTile + (size_t)pl.RS * pl.W * kTile + 4; }
// dA: every (row, pair) unit ends in two 64-lane sums.  With one pair per wave and a small row block the lanes keep their partial
// sums in LDS instead ([W][RB][64] float pairs, each slot owned by one lane: plain read-modify-write) and sum them ONCE per task.
bool bwd_dA_lanes(const PPlan &pl) { return pl.PP == 1 && (size_t)pl.W * pl.RB * 128 * sizeof(float) <= 64 * 1024; }
size_t bwd_lds_floats(const PPlan &pl, int np) {
    return align4(4 * pl.RB * 2 * np + 11 * pl.RB) + (size_t)pl.RS * 4 * kTile + (size_t)pl.RS * pl.W * 2 * kTile + 4 + (pl.nrb > 1 ? (size_t)pl.W * kTile : 0) +
           (bwd_dA_lanes(pl) ? (size_t)pl.W * pl.RB * 128 : 0);
}

template <typename T, bool VEC, int MODE>
void launch_pfwd(int kid, double bytes, const vmasr_sscan_params &p, const PPlan &pl, hipStream_t st) {
    const int np = (p.dstate + 1) / 2;
    const PGeom geo{pl.tiles_per_task, pl.nseg, np, pl.W, pl.RB, pl.RS, pl.nrb, 0, nullptr};
    const dim3 grid((unsigned)((long)p.batch * pl.nseg * p.n_groups * pl.nrb)), block(64 * pl.W);
    const size_t smem = fwd_lds_floats(pl, np) * sizeof(float);
#define VMASR_PF_(PPV, QV, RSV_)                                                                                                        \
    do {                                                                                                                           \
        if (smem > 64 * 1024)                                                                                                      \
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&sscan_pfwd_kernel<T, VEC, PPV, QV, RSV_, MODE>),             \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);                                      \
        VMASR_LAUNCH(kid, bytes, (sscan_pfwd_kernel<T, VEC, PPV, QV, RSV_, MODE>), grid, block, smem, st, p, geo);                 \
    } while (0)
#define VMASR_PF(PPV, QV)                      \
    do {                                       \
        if (pl.RS == 2) VMASR_PF_(PPV, QV, 2); \
        else VMASR_PF_(PPV, QV, 1);            \
    } while (0)
    // (several pairs per wave only with >= 4 waves: one position per thread in the prologue)
    if (pl.PP == 1) {
        if (pl.W >= 4) VMASR_PF(1, 1);
        else if (pl.W >= 2) VMASR_PF_(1, 2, 1);
        else VMASR_PF_(1, 4, 1);
    } else if (pl.PP == 2) VMASR_PF(2, 1);
    else VMASR_PF(4, 1);
#undef VMASR_PF
#undef VMASR_PF_
}

template <typename T, bool VEC, int MODE>
void launch_pbwd(int kid, double bytes, const vmasr_sscan_bwd_params &q, const PPlan &pl, unsigned *det, hipStream_t st) {
    const vmasr_sscan_params &p = q.f;
    const int np = (p.dstate + 1) / 2;
    const PGeom geo{pl.tiles_per_task, pl.nseg, np, pl.W, pl.RB, pl.RS, pl.nrb, bwd_dA_lanes(pl) ? 1 : 0, det};
    const dim3 grid((unsigned)((long)p.batch * pl.nseg * p.n_groups * pl.nrb)), block(64 * pl.W);
    const size_t smem = bwd_lds_floats(pl, np) * sizeof(float);
#define VMASR_PB(PPV, QV)                                                                                                          \
    do {                                                                                                                           \
        if (smem > 64 * 1024)                                                                                                      \
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&sscan_pbwd_kernel<T, VEC, PPV, QV, MODE>),                   \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);                                      \
        VMASR_LAUNCH(kid, bytes, (sscan_pbwd_kernel<T, VEC, PPV, QV, MODE>), grid, block, smem, st, q, geo);                       \
    } while (0)
    if (pl.PP == 1) {
        if (pl.W >= 4) VMASR_PB(1, 1); else if (pl.W >= 2) VMASR_PB(1, 2); else VMASR_PB(1, 4);
    } else if (pl.PP == 2) VMASR_PB(2, 1);
    else VMASR_PB(4, 1);
#undef VMASR_PB
}

template <typename T, bool VEC>
int run_pfwd(const vmasr_sscan_params &p, const PPlan &pl, hipStream_t st) {
    const int ntiles = (p.seqlen + kTile - 1) / kTile;
    // algorithmic bytes: read u, delta, B, C; write out
    const double es = sizeof(T), KD = p.dim, KN = (double)p.n_groups * p.dstate, BL = (double)p.batch * p.seqlen;
    const double full = (3 * KD + 2 * KN) * BL * es, agg = (2 * KD + KN) * BL * es;
    const double xb = (double)p.batch * p.dim * ntiles * p.dstate * 2 * 4 * 2;
    if (!pl.split) {
        launch_pfwd<T, VEC, 0>(VMASR_K_SSCAN_FWD, full, p, pl, st);
        return check_launch("sscan_fwd(N)");
    }
    launch_pfwd<T, VEC, 2>(VMASR_K_SSCAN_FWD_AGG, agg, p, pl, st);
    sscan_launch_carry(false, static_cast<float *>(p.x_ptr), p.batch * p.dim * p.dstate, ntiles, p.dstate, xb, st);
    launch_pfwd<T, VEC, 1>(VMASR_K_SSCAN_FWD_APPLY, full, p, pl, st);
    return check_launch("sscan_fwd(N, split)");
}

template <typename T, bool VEC>
int run_pbwd(const vmasr_sscan_bwd_params &q, const PPlan &pl, hipStream_t st) {
    const vmasr_sscan_params &p = q.f;
    const int ntiles = (p.seqlen + kTile - 1) / kTile;
    // algorithmic bytes: read u, delta, dout, B, C; write du, ddelta, dB, dC (dB / dC fp32)
    const double es = sizeof(T), KD = p.dim, KN = (double)p.n_groups * p.dstate, BL = (double)p.batch * p.seqlen;
    const double full = (5 * KD * es + 2 * KN * es + 2 * KN * 4) * BL, agg = (2 * KD + KN) * BL * es;
    const double xb = (double)p.batch * p.dim * ntiles * p.dstate * 2 * 4 * 2;
    if (!pl.split) {
        launch_pbwd<T, VEC, 0>(VMASR_K_SSCAN_BWD, full, q, pl, det_ticket(VMASR_K_SSCAN_BWD), st);
        return check_launch("sscan_bwd(N)");
    }
    launch_pbwd<T, VEC, 2>(VMASR_K_SSCAN_BWD_AGG, agg, q, pl, nullptr, st);
    sscan_launch_carry(true, static_cast<float *>(q.ws_ptr), p.batch * p.dim * p.dstate, ntiles, p.dstate, xb, st);
    launch_pbwd<T, VEC, 1>(VMASR_K_SSCAN_BWD_APPLY, full, q, pl, det_ticket(VMASR_K_SSCAN_BWD_APPLY), st);
    return check_launch("sscan_bwd(N, split)");
}

}  // namespace

// ---- entry points (called from sscan.hip) ------------------------------------------------------------------------------
size_t sscan_n_bwd_ws_floats(const vmasr_sscan_params &p, int split_req) {
    const PPlan pl = make_pplan(p, split_req);
    if (!pl.split) return 0;
    const size_t ntiles = (p.seqlen + kTile - 1) / kTile;
    return (size_t)p.batch * p.dim * ntiles * p.dstate * 2;   // reverse aggregates -> adjoint carries
}

int sscan_n_fwd(const vmasr_sscan_params &p, int split_req, bool vec, hipStream_t st) {
    const PPlan pl = make_pplan(p, split_req);
    VMASR_REQUIRE(fwd_lds_floats(pl, (p.dstate + 1) / 2) * sizeof(float) <= 160 * 1024, VMASR_EINVAL, "sscan_fwd: d_state %d needs too much LDS", p.dstate);
    switch (p.dtype) {
        case VMASR_F32: return vec ? run_pfwd<float, true>(p, pl, st) : run_pfwd<float, false>(p, pl, st);
        case VMASR_F16: return run_pfwd<f16_t, false>(p, pl, st);      // (16-bit I/O: element-wise loads only — every shipped config scans fp32)
        default: return run_pfwd<bf16_t, false>(p, pl, st);
    }
}

int sscan_n_bwd(const vmasr_sscan_bwd_params &q, int split_req, bool vec, hipStream_t st) {
    const vmasr_sscan_params &p = q.f;
    const PPlan pl = make_pplan(p, split_req);
    VMASR_REQUIRE(bwd_lds_floats(pl, (p.dstate + 1) / 2) * sizeof(float) <= 160 * 1024, VMASR_EINVAL, "sscan_bwd: d_state %d needs too much LDS", p.dstate);
    switch (p.dtype) {
        case VMASR_F32: return vec ? run_pbwd<float, true>(q, pl, st) : run_pbwd<float, false>(q, pl, st);
        case VMASR_F16: return run_pbwd<f16_t, false>(q, pl, st);      // (16-bit I/O: element-wise loads only — every shipped config scans fp32)
        default: return run_pbwd<bf16_t, false>(q, pl, st);
    }
}

}  // namespace vmasr
