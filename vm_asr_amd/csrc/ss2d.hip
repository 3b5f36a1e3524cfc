// ss2d.hip — the SS2D core as ONE operator for gfx950: cross-scan, x_proj, dt_proj, the four directional
// selective scans and cross-merge fused around the scan (d_state 1, dt_rank 1, d_inner <= 32 — the three
// high-resolution stages of every shipped config, 63 % of the scan elements of a clip).
//
// Replaces, for those calls, the chain of SS2D.forward_corev2 (model/vmamba.py:1472-1497):
//     xs = CrossScan(x)                          model/csm_triton.py:7-79      (1 read + 4 writes of D L)
//     x_dbl = einsum(xs, x_proj_weight); dts = einsum(dts, dt_projs_weight)   (:1473-1477)
//     ys = selective_scan(xs, dts, A, Bs, Cs, Ds, dt_bias, softplus)          cus/selective_scan_fwd_kernel.cuh:61-172
//     y = CrossMerge(ys)                         model/csm_triton.py:82-154    (4 reads + 1 write)
// and their backward (cus/selective_scan_bwd_kernel.cuh:66-273 + the two data-movement kernels swapped).
//
// What makes the fusion possible (MI355X-first, not a translation):
//   * directions 0 and 2 visit the SAME memory positions of x in opposite orders, 1 and 3 those of x^T.  The scan
//     is tile-parallel (aggregates -> scan of aggregates -> apply, as sscan.hip's split mode), so ONE workgroup can
//     take one 256-position tile of x for BOTH directions: u is loaded once per pair, the two outputs are added in
//     registers and leave as one store in memory order (y0 + flip(y2) of CrossMerge, same association);
//   * the workgroup holds ALL d_inner rows of its tile (<= 8 waves x 4 rows), so the x_proj reduction over the
//     rows (dt, B, C per position and direction) happens on chip (two-level LDS reduce, 24 values per lane) and
//     delta = softplus(W_dt dt + bias) is computed in registers: xs, dts, Bs, Cs, ys never exist in HBM;
//   * what is left of cross-scan / cross-merge is one transpose of x and one transposing add of the two pair outputs.
// HBM traffic per (row, position), forward, bf16 activations: 2 (transpose r) + 2 (w) + 2 x [2 (agg) + 2 (apply)
// + 4 (out)] + 12 (merge) = 32 B against ~130 B of the unfused chain; the backward is analogous (see below).
//
// Numerics: fp32 everywhere (weights fp32, activations converted on load), same recurrence, softplus and
// exp2-based decay as sscan.hip; the x_proj / dt_proj sums are fp32 FMAs over d in wave order (the reference runs
// them as autocast einsums with a bf16-rounded intermediate: this path is closer to the fp32 oracle).
#include "common.h"
#include "scan_prims.h"

#include <algorithm>

namespace vmasr {
namespace {

constexpr int kXT = 64;  // transpose tile

// ---- x (B*D, H, W) -> xT (B*D, W, H), optional conversion ---------------------------------------------------
template <typename TI, typename TO>
__global__ __launch_bounds__(256) void transpose_hw_kernel(const TI *__restrict__ x, TO *__restrict__ xt, const int H,
                                                           const int W) {
    __shared__ float tile[kXT][kXT + 1];
    const int ntw = (W + kXT - 1) / kXT;
    const int tw = blockIdx.x % ntw, th = blockIdx.x / ntw;
    const size_t plane = (size_t)blockIdx.y * H * W;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
#pragma unroll 4
    for (int j = 0; j < kXT / 4; ++j) {
        const int h = th * kXT + ty + 4 * j, w = tw * kXT + tx;
        if (h < H && w < W) tile[ty + 4 * j][tx] = to_f32(x[plane + (size_t)h * W + w]);
    }
    __syncthreads();
#pragma unroll 4
    for (int j = 0; j < kXT / 4; ++j) {
        const int w = tw * kXT + ty + 4 * j, h = th * kXT + tx;
        if (h < H && w < W) xt[plane + (size_t)w * H + h] = from_f32<TO>(tile[tx][ty + 4 * j]);
    }
}

// ---- y (B*D, H, W) = a (B*D, H, W) + transpose(bT (B*D, W, H)) ----------------------------------------------
template <typename TO>
__global__ __launch_bounds__(256) void merge_pairs_kernel(const float *__restrict__ a, const float *__restrict__ bT,
                                                          TO *__restrict__ y, const int H, const int W) {
    __shared__ float tile[kXT][kXT + 1];
    const int ntw = (W + kXT - 1) / kXT;
    const int tw = blockIdx.x % ntw, th = blockIdx.x / ntw;
    const size_t plane = (size_t)blockIdx.y * H * W;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
#pragma unroll 4
    for (int j = 0; j < kXT / 4; ++j) {
        const int w = tw * kXT + ty + 4 * j, h = th * kXT + tx;
        if (h < H && w < W) tile[tx][ty + 4 * j] = bT[plane + (size_t)w * H + h];
    }
    __syncthreads();
#pragma unroll 4
    for (int j = 0; j < kXT / 4; ++j) {
        const int h = th * kXT + ty + 4 * j, w = tw * kXT + tx;
        if (h < H && w < W) {
            const size_t l = plane + (size_t)h * W + w;
            y[l] = from_f32<TO>(a[l] + tile[ty + 4 * j][tx]);
        }
    }
}

// ---- the fused pair kernels -----------------------------------------------------------------------------------
struct Geo {
    int B, D, L, ntiles, tile_groups;  // tile_groups = ntiles / TPG
};

template <typename T>
__device__ __forceinline__ void load_tile4(const T *__restrict__ p, float (&v)[4]) {
    if constexpr (sizeof(T) == 4) {
        const float4 q = *reinterpret_cast<const float4 *>(p);
        v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
    } else {
        union { uint2 raw; T e[4]; } q;
        q.raw = *reinterpret_cast<const uint2 *>(p);
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = to_f32(q.e[i]);
    }
}

// Sum `v[NV][4]` over the WPT waves that share a tile; every wave gets the sums.  Two levels: each wave writes its
// partials, wave `wr` adds the WPT partials of ceil(NV/WPT) of the values, everybody reads the NV sums back.
// red: [tiles][WPT][NV][64] float4, sums: [tiles][NV][64] float4 (LDS).
template <int NV, int WPT>
__device__ __forceinline__ void reduce_over_waves(float (&v)[NV][4], float4 *red, float4 *sums, const int tg, const int wr,
                                                  const int lane) {
    if constexpr (WPT == 1) return;
    float4 *mine = red + ((size_t)(tg * WPT + wr) * NV) * 64;
#pragma unroll
    for (int c = 0; c < NV; ++c) mine[c * 64 + lane] = make_float4(v[c][0], v[c][1], v[c][2], v[c][3]);
    lds_barrier();
    constexpr int PER = (NV + WPT - 1) / WPT;   // float4 values per summing wave (the last waves may have none)
#pragma unroll
    for (int q = 0; q < PER; ++q) {
        const int c = wr * PER + q;
        if (c >= NV) break;                      // wave-uniform
        float4 s = red[((size_t)(tg * WPT + 0) * NV + c) * 64 + lane];
#pragma unroll
        for (int w = 1; w < WPT; ++w) {
            const float4 t = red[((size_t)(tg * WPT + w) * NV + c) * 64 + lane];
            s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w;
        }
        sums[((size_t)tg * NV + c) * 64 + lane] = s;
    }
    lds_barrier();
#pragma unroll
    for (int c = 0; c < NV; ++c) {
        const float4 s = sums[((size_t)tg * NV + c) * 64 + lane];
        v[c][0] = s.x; v[c][1] = s.y; v[c][2] = s.z; v[c][3] = s.w;
    }
}

struct FwdArgs {
    const void *x, *xT;          // (B, D, L) in (h,w) / (w,h) order, dtype T
    const float *Wx, *Wdt, *dtb; // (4,3,D), (4,D), (4,D)
    const float *Alog, *Dsk;     // (4D), (4D)
    float *state;                // (B, 4D, ntiles, 2): per-tile aggregates -> (after the carry kernel) end-of-tile states
    float *out02, *out13;        // (B, D, L) fp32, memory order of x / xT
};

// MODE 0: per-tile aggregates of both directions -> state      MODE 1: apply (carry-in from state) -> out
// Workgroup = TPG tiles x WPT waves; wave `wr` of a tile owns rows d0 .. d0+RW-1; blockIdx.y = pair (0: dirs 0,2 on x;
// 1: dirs 1,3 on xT).  Direction p scans the tile's positions upwards, p+2 downwards; scan-order tile of the latter
// is ntiles-1-j.
// (launch bounds: 4 waves per SIMD = two 512-thread or four 256-thread workgroups per CU, so that one workgroup's
//  load + LDS-reduce latency is covered by another's scan arithmetic; the apply kernel sits at 132-140 VGPRs without it)
template <typename T, int RW, int WPT, int TPG, int MODE>
__global__ __launch_bounds__(64 * WPT * TPG, 4) void ss2d_fwd_kernel(const FwdArgs a, const Geo g) {
    extern __shared__ __attribute__((aligned(16))) float4 s_lds[];
    float4 *red = s_lds, *sums = s_lds + (WPT > 1 ? (size_t)TPG * WPT * 6 * 64 : 0);
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int tg = wave / WPT, wr = wave % WPT;
    const int pair = blockIdx.y;
    const int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int b = bid / g.tile_groups, j = (bid % g.tile_groups) * TPG + tg;
    const int d0 = wr * RW, D = g.D, L = g.L;
    const T *src = static_cast<const T *>(pair ? a.xT : a.x) + ((size_t)b * D + d0) * L + (size_t)j * kTile + lane * kItems;

    float u[RW][4];
#pragma unroll
    for (int r = 0; r < RW; ++r) load_tile4<T>(src + (size_t)r * L, u[r]);

    // x_proj for both directions of the pair: S[kk*3 + c][i] = sum_d Wx[k][c][d] u[d][i]
    float S[6][4];
#pragma unroll
    for (int kc = 0; kc < 6; ++kc) {
        const int k = pair + 2 * (kc / 3), c = kc % 3;
#pragma unroll
        for (int i = 0; i < 4; ++i) S[kc][i] = 0.f;
#pragma unroll
        for (int r = 0; r < RW; ++r) {
            const float w = a.Wx[((size_t)k * 3 + c) * D + d0 + r];
#pragma unroll
            for (int i = 0; i < 4; ++i) S[kc][i] = fmaf(w, u[r][i], S[kc][i]);
        }
    }
    reduce_over_waves<6, WPT>(S, red, sums, tg, wr, lane);

    float y[RW][4];
#pragma unroll
    for (int r = 0; r < RW; ++r)
#pragma unroll
        for (int i = 0; i < 4; ++i) y[r][i] = 0.f;

#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
        const int k = pair + 2 * kk;
        const int s = kk ? g.ntiles - 1 - j : j;   // scan-order index of this tile for direction k
#pragma unroll
        for (int r = 0; r < RW; ++r) {
            const int kd = k * D + d0 + r;
            const float wdt = a.Wdt[kd], bias = a.dtb[kd], Dk = a.Dsk[kd];
            const float An = -expf(a.Alog[kd]);
            float av[4], bv[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float dl = softplus_f(fmaf(wdt, S[kk * 3 + 0][i], bias));
                av[i] = decay_f(dl, An);
                bv[i] = dl * u[r][i] * S[kk * 3 + 1][i];
            }
            Pair agg, excl, tot;
            if (kk == 0) {
                agg = Pair{av[0], bv[0]};
#pragma unroll
                for (int i = 1; i < 4; ++i) agg = then(agg, Pair{av[i], bv[i]});
                wave_scan_fwd(agg, lane, excl, tot);
            } else {
                agg = Pair{av[3], bv[3]};
#pragma unroll
                for (int i = 2; i >= 0; --i) agg = then(agg, Pair{av[i], bv[i]});
                wave_scan_rev(agg, lane, excl, tot);
            }
            float *st = a.state + (((size_t)b * 4 * D + kd) * g.ntiles) * 2;
            if constexpr (MODE == 0) {
                if (lane == 0) *reinterpret_cast<float2 *>(st + (size_t)s * 2) = make_float2(tot.a, tot.b);
            } else {
                const float hin = s > 0 ? st[(size_t)(s - 1) * 2 + 1] : 0.f;
                float h = fmaf(excl.a, hin, excl.b);
                if (kk == 0) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        h = fmaf(av[i], h, bv[i]);
                        y[r][i] += fmaf(h, S[kk * 3 + 2][i], Dk * u[r][i]);
                    }
                } else {
#pragma unroll
                    for (int i = 3; i >= 0; --i) {
                        h = fmaf(av[i], h, bv[i]);
                        y[r][i] += fmaf(h, S[kk * 3 + 2][i], Dk * u[r][i]);
                    }
                }
            }
        }
    }
    if constexpr (MODE == 1) {
        float *dst = (pair ? a.out13 : a.out02) + ((size_t)b * D + d0) * L + (size_t)j * kTile + lane * kItems;
#pragma unroll
        for (int r = 0; r < RW; ++r)
            *reinterpret_cast<float4 *>(dst + (size_t)r * L) = make_float4(y[r][0], y[r][1], y[r][2], y[r][3]);
    }
}

// In-place scan of the per-tile aggregates of every (batch, direction, row) sequence: x[c] <- x[0] then ... then x[c]
// (REVERSE: exclusive from the right, .y <- carry entering tile c from tile c+1).  One wave per sequence.
template <bool REVERSE>
__global__ __launch_bounds__(256) void ss2d_carry_kernel(float *__restrict__ x, const int nseq, const int n_chunks) {
    const int lane = threadIdx.x & 63;
    const int seq = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (seq >= nseq) return;
    float *base = x + (size_t)seq * n_chunks * 2;
    const int per = (n_chunks + 63) / 64;
    Pair excl, tot;
    if constexpr (!REVERSE) {
        const int c0 = lane * per, c1 = min(n_chunks, c0 + per);
        Pair agg{1.f, 0.f};
        for (int c = c0; c < c1; ++c) agg = then(agg, Pair{base[c * 2], base[c * 2 + 1]});
        wave_scan_fwd(agg, lane, excl, tot);
        Pair run = excl;
        for (int c = c0; c < c1; ++c) {
            run = then(run, Pair{base[c * 2], base[c * 2 + 1]});
            base[c * 2] = run.a;
            base[c * 2 + 1] = run.b;
        }
    } else {
        const int c1 = n_chunks - lane * per, c0 = max(0, c1 - per);
        Pair agg{1.f, 0.f};
        for (int c = c1 - 1; c >= c0; --c) agg = then(agg, Pair{base[c * 2], base[c * 2 + 1]});
        wave_scan_fwd(agg, lane, excl, tot);  // lane order == right-to-left order
        Pair run = excl;
        for (int c = c1 - 1; c >= c0; --c) {
            const Pair mine{base[c * 2], base[c * 2 + 1]};
            base[c * 2 + 1] = run.b;  // carry entering tile c from the right (0 at the end)
            run = then(run, mine);
        }
    }
}

// ---- backward ---------------------------------------------------------------------------------------------------
// Adjoint of h_t = a_t h_{t-1} + b_t, y_t = C_t h_t:  g_t = dout_t C_t + a_{t+1} g_{t+1}.  With G_t := a_t g_t the
// recurrence G_t = a_t (beta_t + G_{t+1}) uses only the step's OWN decay, so a tile needs nothing from its neighbour
// but the scalar G entering it: scan elements (a_t, a_t beta_t), g_t = beta_t + G_{t+1}.
struct BwdArgs {
    FwdArgs f;                   // x, xT, weights, state (end-of-tile states of the forward)
    const float *dy, *dyT;       // (B, D, L) fp32: gradient of the merged output in (h,w) / (w,h) order
    float *adj;                  // (B, 4D, ntiles, 2): adjoint aggregates -> carries
    float *dx02, *dx13;          // (B, D, L) fp32: gradient wrt x from each pair, memory order of x / xT
    float *part;                 // (B * ntiles * 2 pairs, 2 dirs, D, 8): per-workgroup partial sums of the weight gradients
};
constexpr int kNPart = 8;  // dWx[0..2], dWdt, dbias, dA, dD, (pad)

// (the backward keeps ~60 live values per row: two rows per wave -> 137-150 VGPRs, 3 waves per SIMD; d_inner 32 keeps
//  four rows per wave (8 waves, 250 VGPRs, one workgroup per CU): as a 1024-thread workgroup of two-row waves it would
//  have to fit 128 VGPRs and spills — measured 3x the HBM traffic through scratch for no gain in time)
template <typename T, int RW, int WPT, int TPG, int MODE>
__global__ __launch_bounds__(64 * WPT * TPG, (64 * WPT * TPG) >= 512 ? 2 : 3) void ss2d_bwd_kernel(const BwdArgs q, const Geo g) {
    extern __shared__ __attribute__((aligned(16))) float4 s_lds[];
    float4 *red = s_lds, *sums = s_lds + (WPT > 1 ? (size_t)TPG * WPT * 6 * 64 : 0);
    const FwdArgs &a = q.f;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int tg = wave / WPT, wr = wave % WPT;
    const int pair = blockIdx.y;
    const int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int b = bid / g.tile_groups, j = (bid % g.tile_groups) * TPG + tg;
    const int d0 = wr * RW, D = g.D, L = g.L;
    const size_t off = ((size_t)b * D + d0) * L + (size_t)j * kTile + lane * kItems;
    const T *src = static_cast<const T *>(pair ? a.xT : a.x) + off;
    const float *gsrc = (pair ? q.dyT : q.dy) + off;

    float u[RW][4], dout[RW][4];
#pragma unroll
    for (int r = 0; r < RW; ++r) {
        load_tile4<T>(src + (size_t)r * L, u[r]);
        load_tile4<float>(gsrc + (size_t)r * L, dout[r]);
    }
    float S[6][4];
#pragma unroll
    for (int kc = 0; kc < 6; ++kc) {
        const int k = pair + 2 * (kc / 3), c = kc % 3;
#pragma unroll
        for (int i = 0; i < 4; ++i) S[kc][i] = 0.f;
#pragma unroll
        for (int r = 0; r < RW; ++r) {
            const float w = a.Wx[((size_t)k * 3 + c) * D + d0 + r];
#pragma unroll
            for (int i = 0; i < 4; ++i) S[kc][i] = fmaf(w, u[r][i], S[kc][i]);
        }
    }
    reduce_over_waves<6, WPT>(S, red, sums, tg, wr, lane);

    float du[RW][4], G6[6][4];   // G6: gradient wrt (dt, B, C) of both directions, summed over this wave's rows
#pragma unroll
    for (int kc = 0; kc < 6; ++kc)
#pragma unroll
        for (int i = 0; i < 4; ++i) G6[kc][i] = 0.f;
#pragma unroll
    for (int r = 0; r < RW; ++r)
#pragma unroll
        for (int i = 0; i < 4; ++i) du[r][i] = 0.f;
    float *part = q.part + ((((size_t)b * g.ntiles + j) * 2 + pair) * 2) * D * kNPart;

#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
        const int k = pair + 2 * kk;
        const int s = kk ? g.ntiles - 1 - j : j;
#pragma unroll
        for (int r = 0; r < RW; ++r) {
            const int kd = k * D + d0 + r;
            const float wdt = a.Wdt[kd], bias = a.dtb[kd], Dk = a.Dsk[kd];
            const float Araw = -expf(a.Alog[kd]), An = Araw;
            float dl[4], sig[4], av[4], bv[4], be[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                softplus_sigmoid_f(fmaf(wdt, S[kk * 3 + 0][i], bias), dl[i], sig[i]);
                av[i] = decay_f(dl[i], An);
                bv[i] = dl[i] * u[r][i] * S[kk * 3 + 1][i];
                be[i] = dout[r][i] * S[kk * 3 + 2][i];
            }
            // adjoint elements (a_t, a_t beta_t), composed AGAINST the direction's scan order
            Pair ragg, rexcl, rtot;
            if (kk == 0) {   // scan upwards -> adjoint downwards: items 3..0, lanes 63..0
                ragg = Pair{av[3], av[3] * be[3]};
#pragma unroll
                for (int i = 2; i >= 0; --i) ragg = then(ragg, Pair{av[i], av[i] * be[i]});
                wave_scan_rev(ragg, lane, rexcl, rtot);
            } else {
                ragg = Pair{av[0], av[0] * be[0]};
#pragma unroll
                for (int i = 1; i < 4; ++i) ragg = then(ragg, Pair{av[i], av[i] * be[i]});
                wave_scan_fwd(ragg, lane, rexcl, rtot);
            }
            float *adj = q.adj + (((size_t)b * 4 * D + kd) * g.ntiles) * 2;
            if constexpr (MODE == 0) {
                if (lane == 0) *reinterpret_cast<float2 *>(adj + (size_t)s * 2) = make_float2(rtot.a, rtot.b);
                continue;
            }
            // forward recurrence of the tile restarted from the saved state
            const float *st = a.state + (((size_t)b * 4 * D + kd) * g.ntiles) * 2;
            const float hin = s > 0 ? st[(size_t)(s - 1) * 2 + 1] : 0.f;
            const float Gin = adj[(size_t)s * 2 + 1];   // carry entering this tile from scan-tile s+1
            Pair agg, excl, tot;
            float hv[4];
            if (kk == 0) {
                agg = Pair{av[0], bv[0]};
#pragma unroll
                for (int i = 1; i < 4; ++i) agg = then(agg, Pair{av[i], bv[i]});
                wave_scan_fwd(agg, lane, excl, tot);
                float h = fmaf(excl.a, hin, excl.b);
#pragma unroll
                for (int i = 0; i < 4; ++i) { h = fmaf(av[i], h, bv[i]); hv[i] = h; }
            } else {
                agg = Pair{av[3], bv[3]};
#pragma unroll
                for (int i = 2; i >= 0; --i) agg = then(agg, Pair{av[i], bv[i]});
                wave_scan_rev(agg, lane, excl, tot);
                float h = fmaf(excl.a, hin, excl.b);
#pragma unroll
                for (int i = 3; i >= 0; --i) { h = fmaf(av[i], h, bv[i]); hv[i] = h; }
            }
            float Gnext = fmaf(rexcl.a, Gin, rexcl.b);   // G of the step scanned right after this lane's last one
            float accA = 0.f, accD = 0.f, accDt = 0.f, accBias = 0.f;
#pragma unroll
            for (int ii = 0; ii < 4; ++ii) {
                const int i = kk == 0 ? 3 - ii : ii;      // against the scan order
                const float gcur = be[i] + Gnext;         // adjoint of h at this step
                Gnext = av[i] * gcur;
                const float gB = gcur * S[kk * 3 + 1][i];
                const float ax = hv[i] - bv[i];           // a_t h_{t-1}
                du[r][i] += fmaf(gB, dl[i], Dk * dout[r][i]);
                const float dd = fmaf(gB, u[r][i], gcur * Araw * ax) * sig[i];   // wrt the pre-softplus delta
                accA = fmaf(gcur * dl[i], ax, accA);
                accD = fmaf(dout[r][i], u[r][i], accD);
                accDt = fmaf(dd, S[kk * 3 + 0][i], accDt);
                accBias += dd;
                G6[kk * 3 + 0][i] = fmaf(wdt, dd, G6[kk * 3 + 0][i]);                      // d dt
                G6[kk * 3 + 1][i] = fmaf(gcur * dl[i], u[r][i], G6[kk * 3 + 1][i]);        // d B
                G6[kk * 3 + 2][i] = fmaf(dout[r][i], hv[i], G6[kk * 3 + 2][i]);            // d C
            }
            // per-row parameter gradients of this tile (dWx comes after the cross-row reduction below)
            // (one butterfly for the four sums: lane l < 4 ends with the total of value l)
            const float four[4] = {accDt, accBias, accA * Araw, accD};   // dA_log = dA * A (A = -exp(A_log))
            const float tot4 = wave_sum4(four, lane);
            if (lane < 4) part[((size_t)kk * D + d0 + r) * kNPart + 3 + lane] = tot4;
        }
    }
    if constexpr (MODE == 1) {
        reduce_over_waves<6, WPT>(G6, red, sums, tg, wr, lane);
        float *dst = (pair ? q.dx13 : q.dx02) + off;
#pragma unroll
        for (int r = 0; r < RW; ++r) {
            float acc8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kc = 0; kc < 6; ++kc) {
                const int k = pair + 2 * (kc / 3), c = kc % 3;
                const float w = a.Wx[((size_t)k * 3 + c) * D + d0 + r];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    du[r][i] = fmaf(w, G6[kc][i], du[r][i]);
                    acc8[kc] = fmaf(G6[kc][i], u[r][i], acc8[kc]);
                }
            }
            const float sW = wave_sum8(acc8, lane);   // lane kc < 6: dWx partial of (direction kc / 3, component kc % 3)
            if (lane < 6) part[((size_t)(lane / 3) * D + d0 + r) * kNPart + lane % 3] = sW;
            *reinterpret_cast<float4 *>(dst + (size_t)r * L) = make_float4(du[r][0], du[r][1], du[r][2], du[r][3]);
        }
    }
}

// sum the per-workgroup partials: part (nwg, 2 pairs, 2 dirs, D, 8) -> dWx (4,3,D), dWdt (4,D), ddtb (4,D), dAlog (4D), dD (4D)
__global__ __launch_bounds__(256) void ss2d_bwd_reduce_kernel(const float *__restrict__ part, const int nwg, const int D,
                                                              float *__restrict__ dWx, float *__restrict__ dWdt,
                                                              float *__restrict__ ddtb, float *__restrict__ dAlog,
                                                              float *__restrict__ dDs) {
    // one wave per (pair, dir, d); lanes stride over the workgroups, 8 values each
    const int lane = threadIdx.x & 63;
    const int id = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (id >= 4 * D) return;
    const int d = id % D, kk = (id / D) % 2, pair = id / (2 * D);
    float s[7] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int w = lane; w < nwg; w += 64) {
        const float *pp = part + ((((size_t)w * 2 + pair) * 2 + kk) * D + d) * kNPart;
        const float4 p0 = *reinterpret_cast<const float4 *>(pp), p1 = *reinterpret_cast<const float4 *>(pp + 4);
        s[0] += p0.x; s[1] += p0.y; s[2] += p0.z; s[3] += p0.w; s[4] += p1.x; s[5] += p1.y; s[6] += p1.z;
    }
#pragma unroll
    for (int i = 0; i < 7; ++i) s[i] = wave_sum(s[i]);
    if (lane == 0) {
        const int k = pair + 2 * kk, kd = k * D + d;
        dWx[((size_t)k * 3 + 0) * D + d] = s[0];
        dWx[((size_t)k * 3 + 1) * D + d] = s[1];
        dWx[((size_t)k * 3 + 2) * D + d] = s[2];
        dWdt[kd] = s[3];
        ddtb[kd] = s[4];
        dAlog[kd] = s[5];
        dDs[kd] = s[6];
    }
}

// ---- host side ----------------------------------------------------------------------------------------------------
int check(const vmasr_ss2d_params &p, const char *what) {
    VMASR_REQUIRE(p.B > 0 && p.D > 0 && p.H > 0 && p.W > 0, VMASR_EINVAL, "%s: non-positive size", what);
    VMASR_REQUIRE(p.D == 2 || p.D == 4 || p.D == 8 || p.D == 16 || p.D == 32, VMASR_EINVAL,
                  "%s: d_inner must be 2, 4, 8, 16 or 32 (got %d)", what, p.D);
    VMASR_REQUIRE(((long)p.H * p.W) % kTile == 0, VMASR_EINVAL, "%s: H*W must be a multiple of %d", what, kTile);
    VMASR_REQUIRE((long)p.B * p.D <= 65535, VMASR_EINVAL, "%s: batch * d_inner too large", what);
    VMASR_REQUIRE(p.dtype == VMASR_F32 || p.dtype == VMASR_F16 || p.dtype == VMASR_BF16, VMASR_EINVAL, "%s: bad dtype", what);
    VMASR_REQUIRE(p.x && p.Wx && p.Wdt && p.dtb && p.Alog && p.Ds, VMASR_EINVAL, "%s: null tensor", what);
    return 0;
}

struct Cfg {
    int RW, WPT, TPG;
};
Cfg cfg_for(int D, bool bwd = false) {
    const int RW = (bwd && D < 32) ? 2 : (D >= 4 ? 4 : D);
    const int WPT = D / RW;
    return {RW, WPT, WPT >= 4 ? 1 : 4 / WPT};
}
size_t lds_bytes(const Cfg &c) {
    return c.WPT > 1 ? ((size_t)c.TPG * c.WPT * 6 * 64 + (size_t)c.TPG * 6 * 64) * sizeof(float4) : 0;
}

template <typename TI, typename TO>
void launch_transpose(const TI *x, TO *xt, int planes, int H, int W, hipStream_t st) {
    const dim3 grid(((W + kXT - 1) / kXT) * ((H + kXT - 1) / kXT), planes);
    VMASR_LAUNCH(VMASR_K_SS2D_TRANSPOSE, (double)planes * H * W * (sizeof(TI) + sizeof(TO)), (transpose_hw_kernel<TI, TO>), grid,
                 dim3(256), 0, st, x, xt, H, W);
}

// backward: RW = 2 -> WPT = D/2 in {1, 2, 4, 8}; d_inner 32: RW = 4, WPT = 8
#define SS2D_DISPATCH_BWD(KERNEL, MODE, KID, BYTES, ARGS)                                                               \
    do {                                                                                                                \
        const dim3 grid(p.B * geo.tile_groups, 2);                                                                      \
        const dim3 block(64 * c.WPT * c.TPG);                                                                           \
        if (c.WPT == 1)                                                                                                 \
            VMASR_LAUNCH(KID, BYTES, (KERNEL<T, 2, 1, 4, MODE>), grid, block, sm, st, ARGS, geo);                        \
        else if (c.WPT == 2)                                                                                            \
            VMASR_LAUNCH(KID, BYTES, (KERNEL<T, 2, 2, 2, MODE>), grid, block, sm, st, ARGS, geo);                        \
        else if (c.WPT == 4)                                                                                            \
            VMASR_LAUNCH(KID, BYTES, (KERNEL<T, 2, 4, 1, MODE>), grid, block, sm, st, ARGS, geo);                        \
        else if (c.WPT == 8 && c.RW == 2)                                                                               \
            VMASR_LAUNCH(KID, BYTES, (KERNEL<T, 2, 8, 1, MODE>), grid, block, sm, st, ARGS, geo);                        \
        else   /* d_inner 32: four rows per wave */                                                                     \
            VMASR_LAUNCH(KID, BYTES, (KERNEL<T, 4, 8, 1, MODE>), grid, block, sm, st, ARGS, geo);                        \
    } while (0)

#define SS2D_DISPATCH(KERNEL, MODE, KID, BYTES, ARGS)                                                                   \
    do {                                                                                                                \
        const dim3 grid(p.B * geo.tile_groups, 2);                                                                      \
        const dim3 block(64 * c.WPT * c.TPG);                                                                           \
        if (c.RW == 2)                                                                                                  \
            VMASR_LAUNCH(KID, BYTES, (KERNEL<T, 2, 1, 4, MODE>), grid, block, sm, st, ARGS, geo);                        \
        else if (c.WPT == 1)                                                                                            \
            VMASR_LAUNCH(KID, BYTES, (KERNEL<T, 4, 1, 4, MODE>), grid, block, sm, st, ARGS, geo);                        \
        else if (c.WPT == 2)                                                                                            \
            VMASR_LAUNCH(KID, BYTES, (KERNEL<T, 4, 2, 2, MODE>), grid, block, sm, st, ARGS, geo);                        \
        else if (c.WPT == 4)                                                                                            \
            VMASR_LAUNCH(KID, BYTES, (KERNEL<T, 4, 4, 1, MODE>), grid, block, sm, st, ARGS, geo);                        \
        else                                                                                                            \
            VMASR_LAUNCH(KID, BYTES, (KERNEL<T, 4, 8, 1, MODE>), grid, block, sm, st, ARGS, geo);                        \
    } while (0)

template <typename T>
int run_fwd(const vmasr_ss2d_params &p, hipStream_t st) {
    const int L = p.H * p.W, ntiles = L / kTile;
    const Cfg c = cfg_for(p.D);
    VMASR_REQUIRE(ntiles % c.TPG == 0, VMASR_EINVAL, "ss2d_fwd: tile count must be a multiple of %d", c.TPG);
    const Geo geo{p.B, p.D, L, ntiles, ntiles / c.TPG};
    const size_t sm = lds_bytes(c);
    launch_transpose<T, T>(static_cast<const T *>(p.x), static_cast<T *>(p.xT), p.B * p.D, p.H, p.W, st);
    const FwdArgs a{p.x, p.xT, p.Wx, p.Wdt, p.dtb, p.Alog, p.Ds, p.state, p.out02, p.out13};
    // Bytes reported to the profiler (bench.py's roofline): the apply kernels carry the ALGORITHMIC bytes of the
    // selective-scan call they perform as SURVEY.md §8(d) defines them for the reference's operator contract
    // (fwd (3 KD + 2 K N) L s, bwd (5 KD + 4 K N) L s with KD = 4 D, K = 4, N = 1, s = 4: Delta, B, C, the four
    // direction streams and the four outputs are part of that contract although this operator never materialises
    // them); the aggregate passes carry what they really re-read.  The kernels' OWN compulsory traffic is far
    // smaller (forward apply: 2 pairs x (x + out) = 2 (s_x + 4) B per (row, position)) — DESIGN.md §4 quotes both.
    const double el = (double)p.B * p.D * L, pos = (double)p.B * L;
    SS2D_DISPATCH(ss2d_fwd_kernel, 0, VMASR_K_SS2D_FWD_AGG, 2.0 * el * sizeof(T), a);
    const int nseq = p.B * 4 * p.D;
    VMASR_LAUNCH(VMASR_K_SS2D_CARRY, (double)nseq * ntiles * 16, (ss2d_carry_kernel<false>), dim3((nseq + 3) / 4), dim3(256), 0, st,
                 p.state, nseq, ntiles);
    SS2D_DISPATCH(ss2d_fwd_kernel, 1, VMASR_K_SS2D_FWD_APPLY, (3.0 * 4 * el + 2.0 * 4 * pos) * 4, a);
    if (p.flags & VMASR_SS2D_PAIRS) return check_launch("ss2d_fwd");      // the consumer adds out02 + out13^T itself
    const dim3 grid(((p.W + kXT - 1) / kXT) * ((p.H + kXT - 1) / kXT), p.B * p.D);
    VMASR_LAUNCH(VMASR_K_SS2D_MERGE, 12.0 * el, (merge_pairs_kernel<float>), grid, dim3(256), 0, st, p.out02, p.out13, p.y, p.H, p.W);
    return check_launch("ss2d_fwd");
}

template <typename T>
int run_bwd(const vmasr_ss2d_params &p, hipStream_t st) {
    const int L = p.H * p.W, ntiles = L / kTile;
    const Cfg c = cfg_for(p.D, true);
    VMASR_REQUIRE(ntiles % c.TPG == 0, VMASR_EINVAL, "ss2d_bwd: tile count must be a multiple of %d", c.TPG);
    const Geo geo{p.B, p.D, L, ntiles, ntiles / c.TPG};
    const size_t sm = lds_bytes(c);
    if (!(p.flags & VMASR_SS2D_PAIRS)) launch_transpose<float, float>(p.dy, p.dyT, p.B * p.D, p.H, p.W, st);
    BwdArgs q{{p.x, p.xT, p.Wx, p.Wdt, p.dtb, p.Alog, p.Ds, p.state, nullptr, nullptr}, p.dy, p.dyT, p.adj, p.out02, p.out13, p.part};
    const double el = (double)p.B * p.D * L, pos = (double)p.B * L;
    SS2D_DISPATCH_BWD(ss2d_bwd_kernel, 0, VMASR_K_SS2D_BWD_AGG, 2.0 * el * (sizeof(T) + 4), q);
    const int nseq = p.B * 4 * p.D;
    VMASR_LAUNCH(VMASR_K_SS2D_CARRY, (double)nseq * ntiles * 16, (ss2d_carry_kernel<true>), dim3((nseq + 3) / 4), dim3(256), 0, st,
                 p.adj, nseq, ntiles);
    SS2D_DISPATCH_BWD(ss2d_bwd_kernel, 1, VMASR_K_SS2D_BWD_APPLY, (5.0 * 4 * el + 4.0 * 4 * pos) * 4, q);
    const int nwg = p.B * ntiles;
    VMASR_LAUNCH(VMASR_K_SS2D_CARRY, (double)nwg * 4 * p.D * kNPart * 4, ss2d_bwd_reduce_kernel, dim3((4 * p.D + 3) / 4), dim3(256), 0, st,
                 p.part, nwg, p.D, p.dWx, p.dWdt, p.ddtb, p.dAlog, p.dDs);
    // dx = dx02 + transpose(dx13), in the dtype of x
    const dim3 grid(((p.W + kXT - 1) / kXT) * ((p.H + kXT - 1) / kXT), p.B * p.D);
    VMASR_LAUNCH(VMASR_K_SS2D_MERGE, el * (8.0 + sizeof(T)), (merge_pairs_kernel<T>), grid, dim3(256), 0, st, p.out02, p.out13,
                 static_cast<T *>(p.dx), p.H, p.W);
    return check_launch("ss2d_bwd");
}

}  // namespace
}  // namespace vmasr

using namespace vmasr;

VMASR_EXPORT int vmasr_ss2d_supported(int32_t d_state, int32_t dt_rank, int32_t d_inner, int32_t H, int32_t W) {
    if (d_state != 1 || dt_rank != 1) return 0;
    if (!(d_inner == 2 || d_inner == 4 || d_inner == 8 || d_inner == 16 || d_inner == 32)) return 0;
    const long L = (long)H * W;
    if (L % kTile) return 0;
    const Cfg c = cfg_for(d_inner), cb = cfg_for(d_inner, true);
    return ((L / kTile) % c.TPG == 0 && (L / kTile) % cb.TPG == 0) ? 1 : 0;
}

VMASR_EXPORT size_t vmasr_ss2d_part_floats(int32_t B, int32_t D, int32_t H, int32_t W) {
    return (size_t)B * ((size_t)H * W / kTile) * 4 * D * kNPart;
}

VMASR_EXPORT int vmasr_ss2d_fwd(const vmasr_ss2d_params *pp, vmasr_stream_t stream) {
    VMASR_REQUIRE(pp, VMASR_EINVAL, "ss2d_fwd: null params");
    const vmasr_ss2d_params &p = *pp;
    if (int e = check(p, "ss2d_fwd")) return e;
    VMASR_REQUIRE(p.xT && p.state && p.out02 && p.out13 && (p.y || (p.flags & VMASR_SS2D_PAIRS)), VMASR_EINVAL, "ss2d_fwd: null buffer");
    hipStream_t st = static_cast<hipStream_t>(stream);
    switch (p.dtype) {
        case VMASR_F32: return run_fwd<float>(p, st);
        case VMASR_F16: return run_fwd<f16_t>(p, st);
        default: return run_fwd<bf16_t>(p, st);
    }
}

VMASR_EXPORT int vmasr_ss2d_bwd(const vmasr_ss2d_params *pp, vmasr_stream_t stream) {
    VMASR_REQUIRE(pp, VMASR_EINVAL, "ss2d_bwd: null params");
    const vmasr_ss2d_params &p = *pp;
    if (int e = check(p, "ss2d_bwd")) return e;
    VMASR_REQUIRE(p.xT && p.state && p.out02 && p.out13 && p.dy && p.dyT && p.adj && p.part && p.dx && p.dWx && p.dWdt &&
                      p.ddtb && p.dAlog && p.dDs, VMASR_EINVAL, "ss2d_bwd: null buffer");
    hipStream_t st = static_cast<hipStream_t>(stream);
    switch (p.dtype) {
        case VMASR_F32: return run_bwd<float>(p, st);
        case VMASR_F16: return run_bwd<f16_t>(p, st);
        default: return run_bwd<bf16_t>(p, st);
    }
}
