// sscan.hip — SS2D selective scan, forward and backward, for gfx950 (wave64).
//
// Replaces selective_scan_cuda_core.{fwd,bwd} of the reference
// (kernels/selective_scan/csrc/selective_scan/cus/selective_scan.cpp:157-349, kernels
// cus/selective_scan_fwd_kernel.cuh:61-172 and cus/selective_scan_bwd_kernel.cuh:66-273).
// Same mathematics, different machine mapping:
//
//   * unit of work = one WAVE-TILE: 64 lanes x 4 consecutive steps = 256 steps of R rows
//     that share one (batch, group) and therefore one B/C tile (loaded once per wave);
//   * the recurrence h_t = a_t h_{t-1} + b_t is a scan over the monoid (a,b): 4 steps are
//     composed per lane, the 64 lane aggregates are scanned inside each 16-lane DPP row with
//     row_shr / row_shl moves and across the four rows with v_readlane (no LDS traffic, no
//     ds_bpermute), and the running state is a wave-uniform register (N==1) or an LDS slot
//     (general N);
//   * the sequence axis is parallelised three ways, chosen per call shape:
//       walk   one wave walks a whole row tile by tile            (rows*batch fills the chip)
//       block  the W<=16 waves of a workgroup take consecutive tiles of one row and exchange
//              tile totals through LDS: one pass, W-fold parallelism along L   (forward)
//       split  tile-parallel 3-phase scan: per-tile aggregates -> scan of aggregates -> apply
//              (the longest sequences with the fewest rows; also the backward's adjoint carry);
//   * x holds the state at the END of every 256-step tile, so the backward restarts the
//     forward recurrence of every tile independently; only the reverse (adjoint) scan
//     carries across tiles;
//   * dB/dC (summed over the rows of a group) are reduced across the waves of a workgroup
//     in LDS and leave as 256-B contiguous float atomics (or plain stores when one
//     workgroup owns the whole group); dA/dD/ddelta_bias leave the split backward as
//     per-task partials that a small kernel reduces (no same-address atomics).
//
// HBM roofline: algorithmic bytes are (3 KD + 2 K N) L s forward and (5 KD + 4 K N) L s
// backward per clip.
#include "common.h"
#include "scan_prims.h"
#include "sscan_n.h"

#include <stdlib.h>

namespace vmasr {
namespace {

// ---- task geometry ----------------------------------------------------------------------
struct FwdGeom {
    int tiles_per_task, nseg;
};

struct TaskMap {
    int b, d0, g, tile0, tile1;
    bool valid;
};

// tasks: row-block fastest (the waves of a workgroup share B/C lines), then segment, then batch.
template <int R>
__device__ __forceinline__ TaskMap map_task(const vmasr_sscan_params &p, int task, int tiles_per_task,
                                            int nseg, int ntiles) {
    TaskMap m;
    const int nrb = p.dim / R;
    const int rb = task % nrb;
    const int rest = task / nrb;
    const int seg = rest % nseg;
    m.b = rest / nseg;
    m.valid = m.b < p.batch;
    m.d0 = rb * R;
    m.g = m.d0 / (p.dim / p.n_groups);
    m.tile0 = seg * tiles_per_task;
    m.tile1 = min(ntiles, m.tile0 + tiles_per_task);
    return m;
}

// =====================================================================================
// forward
//   MODE 0 walk : carry-in zero at tile 0, sequential over [tile0,tile1), writes x per tile
//   MODE 1 apply: carry-in from x[tile0-1] (already scanned by the carry kernel), no x write
//   MODE 2 agg  : aggregates only: writes the tile-local pair (prod a, h_end | h_in = 0) to x
//   MODE 3 block: the waves of the workgroup own consecutive tiles of one row-block, exchange
//                 tile totals through LDS (one barrier per W tiles), write x per tile
// DYN: general d_state (runtime loop over states, running state in LDS, R must be 1).
// x[...,1] is the state at the end of a tile; x[...,0] is diagnostic (a product of decays).
// =====================================================================================
template <typename T, int R, bool DYN, bool VEC, int MODE>
__global__ __launch_bounds__(MODE == 3 ? 1024 : 256) void sscan_fwd_kernel(const vmasr_sscan_params p,
                                                                           const FwdGeom geo) {
    static_assert(!DYN || R == 1, "general-N path handles one row per wave");
    static_assert(!(DYN && MODE == 3), "block mode is the d_state == 1 path");
    __shared__ float s_h[DYN ? 4 * kMaxDState : 1];
    __shared__ float s_p[DYN ? 4 * kMaxDState : 1];
    __shared__ float2 s_tot[MODE == 3 ? 2 * kMaxBlockWaves * R : 1];

    const int lane = threadIdx.x & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // wave-uniform: row pointers live in SGPRs
    const int ntiles = (p.seqlen + kTile - 1) / kTile;
    TaskMap m;
    int W = 1;
    if constexpr (MODE == 3) {
        W = blockDim.x >> 6;
        const int nrb = p.dim / R;
        const int bid = xcd_remap(blockIdx.x, gridDim.x);
        m.b = bid / nrb;
        m.d0 = (bid % nrb) * R;
        m.g = m.d0 / (p.dim / p.n_groups);
        m.tile0 = 0;
        m.tile1 = ntiles;
        m.valid = true;
    } else {
        const int bid = xcd_remap(blockIdx.x, gridDim.x);
        m = map_task<R>(p, bid * 4 + wave, geo.tiles_per_task, geo.nseg, ntiles);
        if (!m.valid) return;
    }
    const int L = p.seqlen, N = DYN ? p.dstate : 1;

    const T *__restrict__ Bg = static_cast<const T *>(p.B_ptr) + m.b * p.B_batch_stride + m.g * p.B_group_stride;
    const T *__restrict__ Cg = static_cast<const T *>(p.C_ptr) + m.b * p.C_batch_stride + m.g * p.C_group_stride;
    const float *__restrict__ Ap = static_cast<const float *>(p.A_ptr);
    float *__restrict__ xp = static_cast<float *>(p.x_ptr);

    const T *u_row[R], *dl_row[R];
    T *out_row[R];
    float Dv[R], bias[R], h_in[R], p_in[R], An1[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int d = m.d0 + r;
        u_row[r] = static_cast<const T *>(p.u_ptr) + m.b * p.u_batch_stride + d * p.u_d_stride;
        dl_row[r] = static_cast<const T *>(p.delta_ptr) + m.b * p.delta_batch_stride + d * p.delta_d_stride;
        out_row[r] = static_cast<T *>(p.out_ptr) + m.b * p.out_batch_stride + d * p.out_d_stride;
        Dv[r] = p.D_ptr ? static_cast<const float *>(p.D_ptr)[d] : 0.f;
        bias[r] = p.delta_bias_ptr ? static_cast<const float *>(p.delta_bias_ptr)[d] : 0.f;
        An1[r] = DYN ? 0.f : Ap[d * p.A_d_stride];
        h_in[r] = 0.f;
        p_in[r] = 1.f;
    }
    // running state at entry of the first tile
    const size_t xrow0 = ((size_t)m.b * p.dim + m.d0) * p.n_chunks;  // in chunks
    if constexpr (DYN) {
        for (int n = lane; n < N; n += kWave) {
            float h = 0.f, pr = 1.f;
            if (MODE == 1 && m.tile0 > 0) {
                h = xp[((xrow0 + m.tile0 - 1) * N + n) * 2 + 1];
                pr = xp[((xrow0 + m.tile0 - 1) * N + n) * 2 + 0];
            }
            s_h[wave * kMaxDState + n] = h;
            s_p[wave * kMaxDState + n] = pr;
        }
    } else if (MODE == 1 && m.tile0 > 0) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const size_t xi = ((xrow0 + (size_t)r * p.n_chunks) + m.tile0 - 1) * 2;
            p_in[r] = xp[xi];
            h_in[r] = xp[xi + 1];
        }
    }

    const int step = MODE == 3 ? W : 1;
    for (int tbase = m.tile0, it = 0; tbase < m.tile1; tbase += step, ++it) {
        const int tile = MODE == 3 ? tbase + wave : tbase;
        const int t0 = tile * kTile + lane * kItems;  // >= L for the idle waves of a ragged block
        const bool full = (tile + 1) * kTile <= L;    // wave-uniform: straight-line vector loads / stores
        float uv[R][kItems], dl[R][kItems], outv[R][kItems];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            load4u<T, VEC>(u_row[r], t0, L, uv[r], full);
            // steps past the end of a ragged tile must be identity steps (delta = 0: a = 1, b = 0): they load
            // the value that softplus / the bias add below maps to exactly 0, so no per-item masking is needed
            load4u<T, VEC>(dl_row[r], t0, L, dl[r], full, p.delta_softplus ? -INFINITY : -bias[r]);
        }
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
            for (int i = 0; i < kItems; ++i) {
                const float v = dl[r][i] + bias[r];
                dl[r][i] = p.delta_softplus ? softplus_f(v) : v;
                outv[r][i] = Dv[r] * uv[r][i];
            }

        for (int n = 0; n < N; ++n) {
            float Bv[kItems], Cv[kItems];
            load4u<T, VEC>(Bg + n * p.B_dstate_stride, t0, L, Bv, full);
            if (MODE != 2) load4u<T, VEC>(Cg + n * p.C_dstate_stride, t0, L, Cv, full);
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const float An = DYN ? Ap[(m.d0 + r) * p.A_d_stride + n * p.A_dstate_stride] : An1[r];
                float a[kItems], bb[kItems];
#pragma unroll
                for (int i = 0; i < kItems; ++i) {
                    a[i] = decay_f(dl[r][i], An);
                    bb[i] = dl[r][i] * uv[r][i] * Bv[i];
                }
                Pair agg{a[0], bb[0]};
#pragma unroll
                for (int i = 1; i < kItems; ++i) agg = then(agg, Pair{a[i], bb[i]});
                Pair excl, tot;
                wave_scan_fwd(agg, lane, excl, tot);
                if constexpr (MODE == 2) {
                    if (lane == 0) {
                        const size_t xi = (((xrow0 + (size_t)r * p.n_chunks) + tile) * N + n) * 2;
                        *reinterpret_cast<float2 *>(xp + xi) = make_float2(tot.a, tot.b);
                    }
                } else {
                    float hin, pin;
                    if constexpr (DYN) {
                        hin = s_h[wave * kMaxDState + n];
                        pin = s_p[wave * kMaxDState + n];
                    } else if constexpr (MODE == 3) {
                        // exchange tile totals; carry-in = running state composed with the waves before
                        float2 *slot = s_tot + ((it & 1) * kMaxBlockWaves) * R;
                        if (lane == 0) slot[wave * R + r] = make_float2(tot.a, tot.b);
                        lds_barrier();
                        float hrun = h_in[r];
                        hin = hrun;
                        for (int w = 0; w < W; ++w) {
                            const float2 tw = slot[w * R + r];
                            hrun = fmaf(tw.x, hrun, tw.y);
                            if (w + 1 == wave) hin = hrun;  // state after waves [0, wave)
                        }
                        h_in[r] = hrun;  // state after the whole super-tile (same in every wave)
                        pin = 1.f;
                    } else {
                        hin = h_in[r];
                        pin = p_in[r];
                    }
                    float h = fmaf(excl.a, hin, excl.b);
#pragma unroll
                    for (int i = 0; i < kItems; ++i) {
                        h = fmaf(a[i], h, bb[i]);
                        outv[r][i] = fmaf(h, Cv[i], outv[r][i]);
                    }
                    const float hout = fmaf(tot.a, hin, tot.b), pout = tot.a * pin;
                    if constexpr (DYN) {
                        if (lane == 0) {
                            s_h[wave * kMaxDState + n] = hout;
                            s_p[wave * kMaxDState + n] = pout;
                        }
                    } else if constexpr (MODE != 3) {
                        h_in[r] = hout;
                        p_in[r] = pout;
                    }
                    if ((MODE == 0 || MODE == 3) && lane == 0 && tile < ntiles) {
                        const size_t xi = (((xrow0 + (size_t)r * p.n_chunks) + tile) * N + n) * 2;
                        *reinterpret_cast<float2 *>(xp + xi) = make_float2(pout, hout);
                    }
                }
            }
        }
        if constexpr (MODE != 2) {
#pragma unroll
            for (int r = 0; r < R; ++r) store4u<T, VEC>(out_row[r], t0, L, outv[r], full);
        }
    }
}

// In-place inclusive scan of the per-tile aggregates of one (batch,row,state) sequence:
// x[c] <- x[0] then ... then x[c].  One wave per sequence; lanes own contiguous runs.
// REVERSE: exclusive scan from the right over (alpha, beta) pairs, used by the backward:
// ws[c].y <- adjoint entering tile c from tile c+1.
template <bool REVERSE>
__global__ __launch_bounds__(256) void sscan_carry_kernel(float *__restrict__ x, const int nseq,
                                                          const int n_chunks, const int N) {
    const int lane = threadIdx.x & (kWave - 1);
    const int seq = blockIdx.x * 4 + (threadIdx.x >> 6);  // (b*dim + d)*N + n
    if (seq >= nseq) return;
    const int row = seq / N, n = seq % N;
    float *base = x + ((size_t)row * n_chunks * N + n) * 2;
    const size_t cstride = (size_t)N * 2;
    const int per = (n_chunks + kWave - 1) / kWave;
    Pair excl, tot;
    if constexpr (!REVERSE) {
        const int c0 = lane * per, c1 = min(n_chunks, c0 + per);
        Pair agg{1.f, 0.f};
        for (int c = c0; c < c1; ++c) agg = then(agg, Pair{base[c * cstride], base[c * cstride + 1]});
        wave_scan_fwd(agg, lane, excl, tot);
        Pair run = excl;
        for (int c = c0; c < c1; ++c) {
            run = then(run, Pair{base[c * cstride], base[c * cstride + 1]});
            base[c * cstride] = run.a;
            base[c * cstride + 1] = run.b;
        }
    } else {
        // lanes own runs from the right end: lane 0 owns the LAST run
        const int c1 = n_chunks - lane * per, c0 = max(0, c1 - per);
        Pair agg{1.f, 0.f};  // composition "later tile first": g_out = b + a*g_in
        for (int c = c1 - 1; c >= c0; --c) agg = then(agg, Pair{base[c * cstride], base[c * cstride + 1]});
        wave_scan_fwd(agg, lane, excl, tot);  // lane order == right-to-left order
        Pair run = excl;
        for (int c = c1 - 1; c >= c0; --c) {
            const Pair mine{base[c * cstride], base[c * cstride + 1]};
            base[c * cstride + 1] = run.b;  // adjoint entering tile c from the right (0 at the end)
            run = then(run, mine);
        }
    }
}

// =====================================================================================
// backward
//   MODE 0: one workgroup walks [tile0,tile1) from the right, adjoint state in registers
//   MODE 1: adjoint carry-in per task read from ws (scanned by the reverse carry kernel)
//   MODE 2: per-tile reverse aggregates (prod alpha, beta-chain) written to ws
// Workgroup = W waves (blockDim.x/64) that own W consecutive row-blocks of ONE group and
// walk the same tiles in lockstep; dB/dC are reduced over the W*R rows in LDS.
// =====================================================================================
struct BwdGeom {
    int tiles_per_task, nseg, W, wg_per_group;  // wg_per_group = (rows_per_group / R) / W
    float *part;                                // MODE 1, d_state 1: (batch*dim, nseg, 3) partial sums or null
    int fused_carry;                            // MODE 1, d_state 1: fold the reverse aggregates in the prologue
    unsigned *det = nullptr;                    // deterministic mode (common.h): the workgroups of the launch run ONE AFTER THE OTHER in
                                                // workgroup order (this kernel's fp32 atomics sit inside its tile loop), else null
};

template <typename T, int R, bool DYN, bool VEC, int MODE>
__global__ __launch_bounds__(1024) void sscan_bwd_kernel(const vmasr_sscan_bwd_params q, const BwdGeom geo) {
    static_assert(!DYN || R == 1, "general-N path handles one row per wave");
    const vmasr_sscan_params &p = q.f;
    __shared__ float s_g[DYN ? 4 * kMaxDState : 1];   // adjoint carry per state (general N)
    __shared__ float s_an[DYN ? 4 * kMaxDState : 1];  // a of the first step of the tile to the right
    extern __shared__ __attribute__((aligned(16))) float s_red[];  // dB / dC partials: 2 x W x 256 floats (W > 1 only)

    det_enter(geo.det);
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // wave-uniform: row pointers live in SGPRs
    const int W = geo.W;
    const int L = p.seqlen, N = DYN ? p.dstate : 1;
    const int ntiles = (L + kTile - 1) / kTile;
    const int rpg = p.dim / p.n_groups;
    // block -> (wg-in-group fastest, group, segment, batch)
    int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int wgi = bid % geo.wg_per_group; bid /= geo.wg_per_group;
    const int g = bid % p.n_groups; bid /= p.n_groups;
    const int seg = bid % geo.nseg;
    const int b = bid / geo.nseg;
    const int d0 = g * rpg + (wgi * W + wave) * R;
    const int tile0 = seg * geo.tiles_per_task, tile1 = min(ntiles, tile0 + geo.tiles_per_task);

    const T *__restrict__ Bg = static_cast<const T *>(p.B_ptr) + b * p.B_batch_stride + g * p.B_group_stride;
    const T *__restrict__ Cg = static_cast<const T *>(p.C_ptr) + b * p.C_batch_stride + g * p.C_group_stride;
    const float *__restrict__ Ap = static_cast<const float *>(p.A_ptr);
    const float *__restrict__ xp = static_cast<const float *>(p.x_ptr);
    float *__restrict__ ws = static_cast<float *>(q.ws_ptr);
    float *__restrict__ dBg = static_cast<float *>(q.dB_ptr) + ((size_t)b * p.n_groups + g) * N * L;
    float *__restrict__ dCg = static_cast<float *>(q.dC_ptr) + ((size_t)b * p.n_groups + g) * N * L;

    const T *u_row[R], *dl_row[R], *do_row[R];
    T *du_row[R], *dd_row[R];
    float Dv[R], bias[R], g_in[R], a_nx[R], accA[R], accD[R], accBias[R], A1[R];
    const size_t xrow0 = ((size_t)b * p.dim + d0) * p.n_chunks;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int d = d0 + r;
        A1[r] = DYN ? 0.f : Ap[d * p.A_d_stride];  // d_state 1: loop-invariant (no load inside the tile loop)
        u_row[r] = static_cast<const T *>(p.u_ptr) + b * p.u_batch_stride + d * p.u_d_stride;
        dl_row[r] = static_cast<const T *>(p.delta_ptr) + b * p.delta_batch_stride + d * p.delta_d_stride;
        do_row[r] = static_cast<const T *>(q.dout_ptr) + b * q.dout_batch_stride + d * q.dout_d_stride;
        du_row[r] = static_cast<T *>(q.du_ptr) + b * q.du_batch_stride + d * q.du_d_stride;
        dd_row[r] = static_cast<T *>(q.ddelta_ptr) + b * q.ddelta_batch_stride + d * q.ddelta_d_stride;
        Dv[r] = p.D_ptr ? static_cast<const float *>(p.D_ptr)[d] : 0.f;
        bias[r] = p.delta_bias_ptr ? static_cast<const float *>(p.delta_bias_ptr)[d] : 0.f;
        g_in[r] = 0.f; a_nx[r] = 1.f; accA[r] = 0.f; accD[r] = 0.f; accBias[r] = 0.f;
    }
    // state entering from the right of the last tile of this task
    {
        const int tn = tile1 * kTile;  // first step of the tile to the right
#pragma unroll
        for (int r = 0; r < R; ++r) {
            float dnx = 0.f;
            if (tn < L) {
                const float v = to_f32(dl_row[r][tn]) + bias[r];
                dnx = p.delta_softplus ? softplus_f(v) : v;
            }
            if constexpr (DYN) {
                for (int n = lane; n < N; n += kWave) {
                    const float An = Ap[(d0 + r) * p.A_d_stride + n * p.A_dstate_stride];
                    s_an[wave * kMaxDState + n] = tn < L ? decay_f(dnx, An) : 1.f;
                    s_g[wave * kMaxDState + n] =
                        (MODE == 1 && tn < L) ? ws[(((xrow0 + tile1 - 1) * N) + n) * 2 + 1] : 0.f;
                }
            } else {
                const float An = Ap[(d0 + r) * p.A_d_stride];
                a_nx[r] = tn < L ? decay_f(dnx, An) : 1.f;
                if (MODE == 1 && tn < L) {
                    if (geo.fused_carry) {
                        // compose the reverse aggregates of every tile to the right of this task:
                        // lane l owns tiles tile1 + l*per .. (right-to-left order == descending tile)
                        const float *agg = ws + (xrow0 + (size_t)r * p.n_chunks) * 2;
                        const int nright = ntiles - tile1, per = (nright + kWave - 1) / kWave;
                        // adjoint entering from the right end is 0: fold tiles from the last one down
                        Pair mine{1.f, 0.f};
                        const int c_hi = ntiles - 1 - lane * per, c_lo = max(tile1, c_hi - per + 1);
                        for (int c = c_hi; c >= c_lo; --c) mine = then(mine, Pair{agg[c * 2], agg[c * 2 + 1]});
                        Pair ex, tt;
                        wave_scan_fwd(mine, lane, ex, tt);  // lane order == right-to-left order
                        g_in[r] = tt.b;                     // applied to an incoming adjoint of 0
                    } else {
                        g_in[r] = ws[((xrow0 + (size_t)r * p.n_chunks) + tile1 - 1) * 2 + 1];
                    }
                }
            }
        }
    }

    for (int tile = tile1 - 1; tile >= tile0; --tile) {
        const int t0 = tile * kTile + lane * kItems;
        const bool full = (tile + 1) * kTile <= L;  // wave-uniform
        float uv[R][kItems], dl[R][kItems], dov[R][kItems], duv[R][kItems], ddv[R][kItems], sig[R][kItems];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            load4u<T, VEC>(dl_row[r], t0, L, dl[r], full, p.delta_softplus ? -INFINITY : -bias[r]);   // identity steps past the end
            load4u<T, VEC>(do_row[r], t0, L, dov[r], full);
            if (MODE != 2) load4u<T, VEC>(u_row[r], t0, L, uv[r], full);
        }
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
            for (int i = 0; i < kItems; ++i) {
                const float v = dl[r][i] + bias[r];
                if (MODE == 2) {
                    dl[r][i] = p.delta_softplus ? softplus_f(v) : v;
                } else if (p.delta_softplus) {
                    softplus_sigmoid_f(v, dl[r][i], sig[r][i]);   // d softplus / dv = sigmoid(v) (1 above the threshold)
                } else {
                    dl[r][i] = v;
                    sig[r][i] = 1.f;
                }
                if (MODE != 2) {
                    duv[r][i] = Dv[r] * dov[r][i];
                    ddv[r][i] = 0.f;
                    accD[r] = fmaf(dov[r][i], uv[r][i], accD[r]);
                }
            }

        for (int n = 0; n < N; ++n) {
            float Bv[kItems], Cv[kItems], dBv[kItems], dCv[kItems];
            load4u<T, VEC>(Cg + n * p.C_dstate_stride, t0, L, Cv, full);
            if (MODE != 2) load4u<T, VEC>(Bg + n * p.B_dstate_stride, t0, L, Bv, full);
#pragma unroll
            for (int i = 0; i < kItems; ++i) { dBv[i] = 0.f; dCv[i] = 0.f; }
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const float Araw = DYN ? Ap[(d0 + r) * p.A_d_stride + n * p.A_dstate_stride] : A1[r];
                const float An = Araw;
                float a[kItems], al[kItems], be[kItems];
#pragma unroll
                for (int i = 0; i < kItems; ++i) a[i] = decay_f(dl[r][i], An);
                float anx, gin;
                if constexpr (DYN) { anx = s_an[wave * kMaxDState + n]; gin = s_g[wave * kMaxDState + n]; }
                else { anx = a_nx[r]; gin = g_in[r]; }
                // alpha_i = a_{t+1}: next item, next lane's first item, or the tile to the right
                const float a_up = shift_from_next_lane(a[0], lane, anx);
#pragma unroll
                for (int i = 0; i < kItems; ++i) {
                    al[i] = (i + 1 < kItems) ? a[i + 1] : a_up;
                    be[i] = dov[r][i] * Cv[i];
                }
                // lane aggregate from the right: g_first = qb + qa * g_after_lane
                Pair ragg{al[kItems - 1], be[kItems - 1]};
#pragma unroll
                for (int i = kItems - 2; i >= 0; --i) ragg = then(ragg, Pair{al[i], be[i]});
                Pair rexcl, rtot;
                wave_scan_rev(ragg, lane, rexcl, rtot);
                const float a_first = readlane_f(a[0], 0);
                if constexpr (MODE == 2) {
                    if (lane == 0) {
                        const size_t wi = (((xrow0 + (size_t)r * p.n_chunks) + tile) * N + n) * 2;
                        *reinterpret_cast<float2 *>(ws + wi) = make_float2(rtot.a, rtot.b);
                    }
                    continue;
                }
                // forward recurrence of this tile restarted from the saved state
                float hin = 0.f;
                if (tile > 0) hin = xp[(((xrow0 + (size_t)r * p.n_chunks) + tile - 1) * N + n) * 2 + 1];
                float bb[kItems];
#pragma unroll
                for (int i = 0; i < kItems; ++i) bb[i] = dl[r][i] * uv[r][i] * Bv[i];
                Pair agg{a[0], bb[0]};
#pragma unroll
                for (int i = 1; i < kItems; ++i) agg = then(agg, Pair{a[i], bb[i]});
                Pair excl, tot;
                wave_scan_fwd(agg, lane, excl, tot);
                float h = fmaf(excl.a, hin, excl.b);
                float hv[kItems];
#pragma unroll
                for (int i = 0; i < kItems; ++i) { h = fmaf(a[i], h, bb[i]); hv[i] = h; }
                // adjoint recurrence inside the lane
                float gcur = fmaf(rexcl.a, gin, rexcl.b);  // adjoint of the first step of the next lane
#pragma unroll
                for (int i = kItems - 1; i >= 0; --i) {
                    gcur = fmaf(al[i], gcur, be[i]);
                    const float gB = gcur * Bv[i];
                    const float ax = hv[i] - bb[i];  // a_t h_{t-1}
                    duv[r][i] = fmaf(gB, dl[r][i], duv[r][i]);
                    ddv[r][i] += fmaf(gB, uv[r][i], gcur * Araw * ax);
                    accA[r] = fmaf(gcur * dl[r][i], ax, accA[r]);
                    dBv[i] = fmaf(gcur * dl[r][i], uv[r][i], dBv[i]);
                    dCv[i] = fmaf(dov[r][i], hv[i], dCv[i]);
                }
                const float gout = fmaf(rtot.a, gin, rtot.b);
                if constexpr (DYN) {
                    // one (row, state) per iteration: flush dA now
                    const float s = wave_sum(accA[r]);
                    accA[r] = 0.f;
                    if (lane == 0) {
                        atomicAdd(static_cast<float *>(q.dA_ptr) + (d0 + r) * q.dA_d_stride + n * q.dA_dstate_stride, s);
                        s_g[wave * kMaxDState + n] = gout;
                        s_an[wave * kMaxDState + n] = a_first;
                    }
                } else {
                    g_in[r] = gout;
                    a_nx[r] = a_first;
                }
            }
            if constexpr (MODE != 2) {
                // reduce dB/dC over the W waves (rows) of the workgroup, then leave as 256-B runs
                const int tbase = tile * kTile;
                if (W == 1) {
                    if (VEC && t0 + 3 < L && geo.wg_per_group == 1) {
                        *reinterpret_cast<float4 *>(dBg + (size_t)n * L + t0) = make_float4(dBv[0], dBv[1], dBv[2], dBv[3]);
                        *reinterpret_cast<float4 *>(dCg + (size_t)n * L + t0) = make_float4(dCv[0], dCv[1], dCv[2], dCv[3]);
                    } else {
#pragma unroll
                        for (int i = 0; i < kItems; ++i) {
                            const int t = t0 + i;
                            if (t < L) {
                                if (geo.wg_per_group == 1) { dBg[(size_t)n * L + t] = dBv[i]; dCg[(size_t)n * L + t] = dCv[i]; }
                                else { atomicAdd(dBg + (size_t)n * L + t, dBv[i]); atomicAdd(dCg + (size_t)n * L + t, dCv[i]); }
                            }
                        }
                    }
                } else {
                    float *redB = s_red, *redC = s_red + W * kTile;
                    *reinterpret_cast<float4 *>(redB + wave * kTile + lane * kItems) = make_float4(dBv[0], dBv[1], dBv[2], dBv[3]);
                    *reinterpret_cast<float4 *>(redC + wave * kTile + lane * kItems) = make_float4(dCv[0], dCv[1], dCv[2], dCv[3]);
                    lds_barrier();
                    for (int e = threadIdx.x; e < 2 * kTile; e += blockDim.x) {
                        const int which = e / kTile, idx = e % kTile;
                        const float *src = which ? redC : redB;
                        float s = 0.f;
                        for (int w = 0; w < W; ++w) s += src[w * kTile + idx];
                        const int t = tbase + idx;
                        if (t < L) {
                            float *dst = (which ? dCg : dBg) + (size_t)n * L + t;
                            if (geo.wg_per_group == 1) *dst = s; else atomicAdd(dst, s);
                        }
                    }
                    lds_barrier();
                }
            }
        }
        if constexpr (MODE != 2) {
#pragma unroll
            for (int r = 0; r < R; ++r) {
#pragma unroll
                for (int i = 0; i < kItems; ++i) {
                    ddv[r][i] *= sig[r][i];
                    accBias[r] += (full || t0 + i < L) ? ddv[r][i] : 0.f;
                }
                store4u<T, VEC>(du_row[r], t0, L, duv[r], full);
                store4u<T, VEC>(dd_row[r], t0, L, ddv[r], full);
            }
        }
    }
    if constexpr (MODE != 2) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int d = d0 + r;
            const float sA = DYN ? 0.f : wave_sum(accA[r]);
            const float sD = wave_sum(accD[r]), sB = wave_sum(accBias[r]);
            if (lane == 0) {
                if (!DYN && geo.part) {  // per-task partials, reduced by sscan_bwd_reduce_kernel
                    float *pp = geo.part + (((size_t)b * p.dim + d) * geo.nseg + seg) * 3;
                    pp[0] = sA; pp[1] = sD; pp[2] = sB;
                } else {
                    if (!DYN) atomicAdd(static_cast<float *>(q.dA_ptr) + d * q.dA_d_stride, sA);
                    if (q.dD_ptr) atomicAdd(static_cast<float *>(q.dD_ptr) + d, sD);
                    if (q.ddelta_bias_ptr) atomicAdd(static_cast<float *>(q.ddelta_bias_ptr) + d, sB);
                }
            }
        }
    }
    det_leave(geo.det);
}

// one wave per row d: sum the (batch, segment) partials of dA / dD / ddelta_bias
__global__ __launch_bounds__(256) void sscan_bwd_reduce_kernel(const float *__restrict__ part, const int batch,
                                                               const int dim, const int nseg, float *__restrict__ dA,
                                                               const long dA_stride, float *__restrict__ dD,
                                                               float *__restrict__ dbias) {
    const int lane = threadIdx.x & (kWave - 1);
    const int d = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (d >= dim) return;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f;
    for (int b = 0; b < batch; ++b) {
        const float *pp = part + ((size_t)b * dim + d) * nseg * 3;
        for (int s = lane; s < nseg; s += kWave) {
            s0 += pp[s * 3 + 0];
            s1 += pp[s * 3 + 1];
            s2 += pp[s * 3 + 2];
        }
    }
    s0 = wave_sum(s0); s1 = wave_sum(s1); s2 = wave_sum(s2);
    if (lane == 0) {
        dA[d * dA_stride] += s0;  // caller zero-initialised; this launch is the only writer of row d
        if (dD) dD[d] += s1;
        if (dbias) dbias[d] += s2;
    }
}

// ---------------------------------------------------------------------------------------
int g_tune_rows = -1, g_tune_split = -1;

int validate(const vmasr_sscan_params &p) {
    VMASR_REQUIRE(p.batch > 0 && p.dim > 0 && p.seqlen > 0 && p.dstate > 0 && p.n_groups > 0, VMASR_EINVAL,
                  "sscan: non-positive size");
    VMASR_REQUIRE(p.dim % p.n_groups == 0, VMASR_EINVAL, "sscan: dims should be dividable by n_groups");
    VMASR_REQUIRE(p.dstate <= kMaxDState, VMASR_EINVAL, "sscan: only supports state dimension <= 256");
    VMASR_REQUIRE(p.dtype == VMASR_F32 || p.dtype == VMASR_F16 || p.dtype == VMASR_BF16, VMASR_EINVAL,
                  "sscan: dtype must be fp32/fp16/bf16");
    VMASR_REQUIRE(p.n_chunks == (p.seqlen + kTile - 1) / kTile, VMASR_EINVAL,
                  "sscan: n_chunks must be ceil(seqlen/%d)", kTile);
    VMASR_REQUIRE(p.u_ptr && p.delta_ptr && p.A_ptr && p.B_ptr && p.C_ptr, VMASR_EINVAL, "sscan: null tensor");
    VMASR_REQUIRE(p.x_ptr || p.n_chunks == 1, VMASR_EINVAL, "sscan: x is required when n_chunks > 1");
    return 0;
}

bool vec_ok(int esz, std::initializer_list<const void *> ptrs, std::initializer_list<int64_t> strides) {
    const size_t bytes = esz == 4 ? 16 : 8;
    for (const void *q : ptrs)
        if (q && !aligned_to(q, bytes)) return false;
    for (int64_t s : strides)
        if (s % 4 != 0) return false;
    return true;
}

// launch plan.  split: 0 walk, 1 tile-parallel 3-phase, 2 block (forward only; the backward of a
// block-planned call uses the split path).
struct Plan {
    int R, split, tiles_per_task, nseg, W;
};

constexpr long kEnoughWaves = 2048;  // 2 waves per SIMD over 1024 SIMDs

Plan make_plan(const vmasr_sscan_params &p, bool dyn, bool backward) {
    Plan pl{1, 0, 1, 1, 1};
    const int rpg = p.dim / p.n_groups;
    const int ntiles = (p.seqlen + kTile - 1) / kTile;
    int R = 1;
    // two-row groups (the 512x512 output block, KD = 8): one wave carries the whole group, so B/C are
    // loaded once; elsewhere one row per wave measured faster (more waves beat the shared loads)
    if (!dyn && !backward && rpg == 2) R = 2;
    if (!dyn && g_tune_rows > 0 && rpg % g_tune_rows == 0 && (g_tune_rows == 1 || g_tune_rows == 2 || g_tune_rows == 4))
        R = g_tune_rows;
    if (backward && R == 4) R = 2;  // the backward keeps ~25 live values per row and item: 4 rows only spill
    pl.R = R;
    const long row_tasks = (long)p.batch * (p.dim / R);
    int mode = 0;
    if (ntiles > 1 && row_tasks < kEnoughWaves) {
        const int W = ntiles < kMaxBlockWaves ? ntiles : kMaxBlockWaves;
        mode = (!dyn && !backward && row_tasks * W >= kEnoughWaves) ? 2 : 1;
    }
    if (g_tune_split >= 0 && ntiles > 1) mode = (g_tune_split == 2 && (dyn || backward)) ? 1 : g_tune_split;
    pl.split = mode;
    if (mode == 1) {
        // a few tiles per task once there are plenty of tasks (amortises the prologue)
        int tpt = 1;
        while (tpt < 8 && row_tasks * ((ntiles + 2 * tpt - 1) / (2 * tpt)) >= 4 * kEnoughWaves) tpt *= 2;
        pl.tiles_per_task = tpt;
        pl.nseg = (ntiles + tpt - 1) / tpt;
    } else {
        pl.tiles_per_task = ntiles;
        pl.nseg = 1;
        pl.W = mode == 2 ? (ntiles < kMaxBlockWaves ? ntiles : kMaxBlockWaves) : 1;
    }
    return pl;
}

template <typename T, int R, bool DYN, bool VEC>
int launch_fwd(const vmasr_sscan_params &p, const Plan &pl, hipStream_t st) {
    const int ntiles = (p.seqlen + kTile - 1) / kTile;
    const long ntasks = (long)p.batch * (p.dim / R) * pl.nseg;
    const int nblocks = (int)((ntasks + 3) / 4);
    const FwdGeom geo{pl.tiles_per_task, pl.nseg};
    // algorithmic bytes: read u, delta, B, C; write out
    const double es = sizeof(T), KD = p.dim, KN = (double)p.n_groups * p.dstate, BL = (double)p.batch * p.seqlen;
    const double full = (3 * KD + 2 * KN) * BL * es, agg = (2 * KD + KN) * BL * es;
    const double xb = (double)p.batch * p.dim * ntiles * p.dstate * 2 * 4 * 2;
    if (pl.split == 0) {
        VMASR_LAUNCH(VMASR_K_SSCAN_FWD, full, (sscan_fwd_kernel<T, R, DYN, VEC, 0>), dim3(nblocks), dim3(256), 0, st, p, geo);
        return check_launch("sscan_fwd");
    }
    if constexpr (!DYN) {
        if (pl.split == 2) {
            VMASR_LAUNCH(VMASR_K_SSCAN_FWD, full, (sscan_fwd_kernel<T, R, false, VEC, 3>), dim3(p.batch * (p.dim / R)),
                         dim3(64 * pl.W), 0, st, p, geo);
            return check_launch("sscan_fwd(block)");
        }
    }
    VMASR_LAUNCH(VMASR_K_SSCAN_FWD_AGG, agg, (sscan_fwd_kernel<T, R, DYN, VEC, 2>), dim3(nblocks), dim3(256), 0, st, p, geo);
    const int nseq = p.batch * p.dim * p.dstate;
    VMASR_LAUNCH(VMASR_K_SSCAN_FWD_CARRY, xb, (sscan_carry_kernel<false>), dim3((nseq + 3) / 4), dim3(256), 0, st,
                 static_cast<float *>(p.x_ptr), nseq, ntiles, p.dstate);
    VMASR_LAUNCH(VMASR_K_SSCAN_FWD_APPLY, full, (sscan_fwd_kernel<T, R, DYN, VEC, 1>), dim3(nblocks), dim3(256), 0, st, p, geo);
    return check_launch("sscan_fwd(split)");
}

size_t bwd_ws_floats(const vmasr_sscan_params &p, const Plan &pl) {
    if (pl.split != 1) return 0;
    const size_t ntiles = (p.seqlen + kTile - 1) / kTile;
    size_t n = (size_t)p.batch * p.dim * ntiles * p.dstate * 2;  // reverse aggregates / adjoint carries
    if (p.dstate == 1) n += (size_t)p.batch * p.dim * pl.nseg * 3;  // dA/dD/dbias partials
    return n;
}

template <typename T, int R, bool DYN, bool VEC>
int launch_bwd(const vmasr_sscan_bwd_params &q, const Plan &pl, hipStream_t st) {
    const vmasr_sscan_params &p = q.f;
    const int ntiles = (p.seqlen + kTile - 1) / kTile;
    const int rbg = (p.dim / p.n_groups) / R;  // row-blocks per group
    static const int env_w = [] { const char *e = getenv("VMASR_BWD_WAVES"); return e ? atoi(e) : 0; }();
    int W = DYN ? 4 : (env_w > 0 ? env_w : 8);  // 8 measured best (2..16); general N keeps per-wave LDS state: 4
    while (rbg % W) W >>= 1;
    BwdGeom geo{pl.tiles_per_task, pl.nseg, W, rbg / W, nullptr, 0};
    const long nblocks = (long)p.batch * pl.nseg * p.n_groups * geo.wg_per_group;
    const size_t smem = W > 1 ? (size_t)2 * W * kTile * sizeof(float) : 0;
    // algorithmic bytes: read u, delta, dout, B, C; write du, ddelta, dB, dC (dB/dC fp32)
    const double es = sizeof(T), KD = p.dim, KN = (double)p.n_groups * p.dstate, BL = (double)p.batch * p.seqlen;
    const double full = (5 * KD * es + 2 * KN * es + 2 * KN * 4) * BL, agg = (2 * KD + KN) * BL * es;
    const double xb = (double)p.batch * p.dim * ntiles * p.dstate * 2 * 4 * 2;
    if (pl.split != 1) {
        geo.det = det_ticket(VMASR_K_SSCAN_BWD);
        VMASR_LAUNCH(VMASR_K_SSCAN_BWD, full, (sscan_bwd_kernel<T, R, DYN, VEC, 0>), dim3((int)nblocks), dim3(64 * W), smem, st, q, geo);
        return check_launch("sscan_bwd");
    }
    VMASR_LAUNCH(VMASR_K_SSCAN_BWD_AGG, agg, (sscan_bwd_kernel<T, R, DYN, VEC, 2>), dim3((int)nblocks), dim3(64 * W), 0, st, q, geo);
    // moderate tile counts: every apply task folds the aggregates to its right itself (one launch less);
    // few segments: dA/dD/dbias leave as atomics (little contention) instead of partials + reduce
    const bool fuse = !DYN && ntiles <= 256;
    const bool partials = !DYN && (long)p.batch * pl.nseg > 64;
    if (!fuse) {
        const int nseq = p.batch * p.dim * p.dstate;
        VMASR_LAUNCH(VMASR_K_SSCAN_BWD_CARRY, xb, (sscan_carry_kernel<true>), dim3((nseq + 3) / 4), dim3(256), 0, st,
                     static_cast<float *>(q.ws_ptr), nseq, ntiles, p.dstate);
    }
    geo.fused_carry = fuse;
    if (partials) geo.part = static_cast<float *>(q.ws_ptr) + (size_t)p.batch * p.dim * ntiles * 2;
    geo.det = det_ticket(VMASR_K_SSCAN_BWD_APPLY);
    VMASR_LAUNCH(VMASR_K_SSCAN_BWD_APPLY, full, (sscan_bwd_kernel<T, R, DYN, VEC, 1>), dim3((int)nblocks), dim3(64 * W), smem, st, q, geo);
    if (partials)
        VMASR_LAUNCH(VMASR_K_SSCAN_BWD_CARRY, (double)p.batch * p.dim * pl.nseg * 12, sscan_bwd_reduce_kernel,
                     dim3((p.dim + 3) / 4), dim3(256), 0, st, geo.part, p.batch, p.dim, pl.nseg,
                     static_cast<float *>(q.dA_ptr), (long)q.dA_d_stride, static_cast<float *>(q.dD_ptr),
                     static_cast<float *>(q.ddelta_bias_ptr));
    return check_launch("sscan_bwd(split)");
}

#define VMASR_DISPATCH_R(FN, T, DYN, VEC, ...)                         \
    do {                                                               \
        if (DYN || pl.R == 1) return FN<T, 1, DYN, VEC>(__VA_ARGS__);  \
        if constexpr (!DYN) {                                          \
            if (pl.R == 2) return FN<T, 2, false, VEC>(__VA_ARGS__);   \
            return FN<T, 4, false, VEC>(__VA_ARGS__);                  \
        }                                                              \
    } while (0)

template <typename T>
int dispatch_fwd(const vmasr_sscan_params &p, const Plan &pl, bool dyn, bool vec, hipStream_t st) {
    if (dyn) { if (vec) VMASR_DISPATCH_R(launch_fwd, T, true, true, p, pl, st); else VMASR_DISPATCH_R(launch_fwd, T, true, false, p, pl, st); }
    else { if (vec) VMASR_DISPATCH_R(launch_fwd, T, false, true, p, pl, st); else VMASR_DISPATCH_R(launch_fwd, T, false, false, p, pl, st); }
    return VMASR_EINVAL;
}
template <typename T>
int dispatch_bwd(const vmasr_sscan_bwd_params &q, const Plan &pl, bool dyn, bool vec, hipStream_t st) {
    if (dyn) return vec ? launch_bwd<T, 1, true, true>(q, pl, st) : launch_bwd<T, 1, true, false>(q, pl, st);
    if (pl.R == 1) return vec ? launch_bwd<T, 1, false, true>(q, pl, st) : launch_bwd<T, 1, false, false>(q, pl, st);
    return vec ? launch_bwd<T, 2, false, true>(q, pl, st) : launch_bwd<T, 2, false, false>(q, pl, st);
}

}  // namespace
}  // namespace vmasr

namespace vmasr {
void sscan_launch_carry(bool reverse, float *x, int nseq, int n_chunks, int N, double bytes, hipStream_t st) {
    if (reverse)
        VMASR_LAUNCH(VMASR_K_SSCAN_BWD_CARRY, bytes, (sscan_carry_kernel<true>), dim3((nseq + 3) / 4), dim3(256), 0, st, x, nseq, n_chunks, N);
    else
        VMASR_LAUNCH(VMASR_K_SSCAN_FWD_CARRY, bytes, (sscan_carry_kernel<false>), dim3((nseq + 3) / 4), dim3(256), 0, st, x, nseq, n_chunks, N);
}
}  // namespace vmasr

using namespace vmasr;

// general d_state: the packed state-pair kernels of sscan_n.hip (VMASR_SSCAN_N_LEGACY=1: the one-state-at-a-time kernels of this
// file, kept for A/B measurements)
static bool n_legacy() {
    static const bool v = [] { const char *e = getenv("VMASR_SSCAN_N_LEGACY"); return e && atoi(e) != 0; }();
    return v;
}

static int n_split_req() { return g_tune_split < 0 ? -1 : (g_tune_split == 1 ? 1 : 0); }

VMASR_EXPORT int vmasr_sscan_chunk(void) { return kTile; }

VMASR_EXPORT void vmasr_sscan_tune(int rows, int split) {
    g_tune_rows = rows;
    g_tune_split = split;
}

VMASR_EXPORT int vmasr_sscan_fwd(const vmasr_sscan_params *pp, vmasr_stream_t stream) {
    VMASR_REQUIRE(pp, VMASR_EINVAL, "sscan_fwd: null params");
    const vmasr_sscan_params &p = *pp;
    if (int e = validate(p)) return e;
    VMASR_REQUIRE(p.out_ptr, VMASR_EINVAL, "sscan_fwd: null out");
    const bool dyn = p.dstate != 1;
    const int esz = p.dtype == VMASR_F32 ? 4 : 2;
    const bool vec = vec_ok(esz, {p.u_ptr, p.delta_ptr, p.B_ptr, p.C_ptr, p.out_ptr},
                            {p.u_batch_stride, p.u_d_stride, p.delta_batch_stride, p.delta_d_stride,
                             p.out_batch_stride, p.out_d_stride, p.B_batch_stride, p.B_group_stride,
                             p.B_dstate_stride, p.C_batch_stride, p.C_group_stride, p.C_dstate_stride});
    const Plan pl = make_plan(p, dyn, false);
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (dyn && p.dstate <= 128 && !n_legacy()) return sscan_n_fwd(p, n_split_req(), vec, st);
    switch (p.dtype) {
        case VMASR_F32: return dispatch_fwd<float>(p, pl, dyn, vec, st);
        case VMASR_F16: return dispatch_fwd<f16_t>(p, pl, dyn, vec, st);
        default: return dispatch_fwd<bf16_t>(p, pl, dyn, vec, st);
    }
}

VMASR_EXPORT size_t vmasr_sscan_bwd_workspace(const vmasr_sscan_bwd_params *q) {
    if (!q) return 0;
    const vmasr_sscan_params &p = q->f;
    if (p.batch <= 0 || p.dim <= 0 || p.seqlen <= 0 || p.dstate <= 0 || p.n_groups <= 0 || p.dim % p.n_groups) return 0;
    if (p.dstate != 1 && p.dstate <= 128 && !n_legacy()) return sscan_n_bwd_ws_floats(p, n_split_req()) * sizeof(float);
    return bwd_ws_floats(p, make_plan(p, p.dstate != 1, true)) * sizeof(float);
}

VMASR_EXPORT int vmasr_sscan_bwd(const vmasr_sscan_bwd_params *qq, vmasr_stream_t stream) {
    VMASR_REQUIRE(qq, VMASR_EINVAL, "sscan_bwd: null params");
    const vmasr_sscan_bwd_params &q = *qq;
    const vmasr_sscan_params &p = q.f;
    if (int e = validate(p)) return e;
    VMASR_REQUIRE(q.dout_ptr && q.du_ptr && q.ddelta_ptr && q.dA_ptr && q.dB_ptr && q.dC_ptr, VMASR_EINVAL,
                  "sscan_bwd: null tensor");
    const bool dyn = p.dstate != 1;
    const int esz = p.dtype == VMASR_F32 ? 4 : 2;
    const bool vec = vec_ok(esz, {p.u_ptr, p.delta_ptr, p.B_ptr, p.C_ptr, q.dout_ptr, q.du_ptr, q.ddelta_ptr, q.dB_ptr, q.dC_ptr},
                            {p.u_batch_stride, p.u_d_stride, p.delta_batch_stride, p.delta_d_stride,
                             q.dout_batch_stride, q.dout_d_stride, q.du_batch_stride, q.du_d_stride,
                             q.ddelta_batch_stride, q.ddelta_d_stride, p.B_batch_stride, p.B_group_stride,
                             p.B_dstate_stride, p.C_batch_stride, p.C_group_stride, p.C_dstate_stride, (int64_t)p.seqlen});
    const Plan pl = make_plan(p, dyn, true);
    const size_t need = ((dyn && p.dstate <= 128 && !n_legacy()) ? sscan_n_bwd_ws_floats(p, n_split_req()) : bwd_ws_floats(p, pl)) * sizeof(float);
    if (need)
        VMASR_REQUIRE(q.ws_ptr && q.ws_bytes >= need, VMASR_ENOSPACE, "sscan_bwd: workspace too small (%zu < %zu)",
                      q.ws_bytes, need);
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (dyn && p.dstate <= 128 && !n_legacy()) return sscan_n_bwd(q, n_split_req(), vec, st);
    switch (p.dtype) {
        case VMASR_F32: return dispatch_bwd<float>(q, pl, dyn, vec, st);
        case VMASR_F16: return dispatch_bwd<f16_t>(q, pl, dyn, vec, st);
        default: return dispatch_bwd<bf16_t>(q, pl, dyn, vec, st);
    }
}
