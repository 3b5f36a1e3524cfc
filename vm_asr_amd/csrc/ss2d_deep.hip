// ss2d_deep.hip — the SS2D core of the DEEP stages as one operator for gfx950: d_state 1, dt_rank 2 / 4 / 8 (16 for H*W <= 512),
// d_inner 64 .. 512, H*W in {256, 512, 1024, 2048, 4096} (the 64x64, 32x32 and 16x16 stages of every shipped config: 18 of the
// 28 SS2D calls of a training step).
//
// Replaces, for those calls, the chain of SS2D.forward_corev2 (model/vmamba.py:1472-1497):
//     xs = CrossScan(x)                                     model/csm_triton.py:7-79
//     x_dbl = einsum(xs, x_proj_weight); dts = einsum(dts, dt_projs_weight)      (:1473-1477)
//     ys = selective_scan(xs, dts, A, Bs, Cs, Ds, dt_bias, softplus)             cus/selective_scan_fwd_kernel.cuh:61-172
//     y = CrossMerge(ys)                                    model/csm_triton.py:82-154
// and its backward (cus/selective_scan_bwd_kernel.cuh:66-273), which ran as 4 launches forward and 6 backward per call over
// tensors of 0.5 - 2 MB: latency, not bandwidth.
//
// The high-resolution fused core (ss2d.hip) gives a workgroup ALL rows of a 256-position tile (the x_proj contraction over the
// rows stays on chip) and pays for it with a three-phase scan (aggregates / carry / apply).  Here the roles are swapped,
// because the sequences are short and the rows are many:
//   * a workgroup owns WHOLE ROWS (one (b, d) image of H*W <= 4096 positions = <= 16 wave tiles): the scan of a row needs no
//     other workgroup — tile totals cross the waves of the row through LDS, one launch, no carry kernel;
//   * the row's image sits in LDS, so the directions that walk it column-wise (1 and 3) read it TRANSPOSED from LDS and add
//     their outputs back into an LDS image: CrossScan / CrossMerge are two LDS index computations, no transpose kernels;
//   * what does cross rows — x_proj (a contraction over d_inner) — is a small position-parallel kernel in front
//     (`deep_xproj_kernel`: x -> x_dbl (B, 4, R+2, L) fp32, 0.25 - 1 MB; delta = softplus(W_dt dt + bias) is then computed in
//     the scan kernel's registers: the (B, 4 D, L) delta tensor never exists), and its adjoint a small kernel behind
//     (`deep_xg_kernel`: the scan's per-row terms summed over d_inner; `deep_dx_kernel`: dx = du + W_x^T d(x_dbl)).
// forward = 2 launches, backward = 3 launches (+ one small GEMM for dW_x and one sum on the host side).
//
// x_dbl layout: (B, 4, R+2, L); directions 0 / 2 indexed by the row-major position p = h W + w, directions 1 / 3 by the
// column-major position q = w H + h; directions 2 / 3 scan the same data downwards.
// Numerics: as ss2d.hip (fp32 everywhere, activations converted on load, decay_f / softplus_f of scan_prims.h).
#include "common.h"
#include "scan_prims.h"

#include <mutex>
#include <unordered_set>

namespace vmasr {
namespace {

struct DeepGeo {
    int B, D, H, W, L;
};

__device__ __forceinline__ int rfl(int v) { return __builtin_amdgcn_readfirstlane(v); }

// p[0] + p[stride] + ... (N a power of two) as a balanced tree: the partial sums of the waves are of equal size
template <int N>
__device__ __forceinline__ float tree_sum(const float *p, const int stride) {
    if constexpr (N == 1) return p[0];
    else return tree_sum<N / 2>(p, stride) + tree_sum<N / 2>(p + (N / 2) * stride, stride);
}

// ---- x_proj: x (B, D, L) -> x_dbl (B, 4, C, L) ----------------------------------------------------------------------------
// grid (L / 64, B, 4 directions); NW waves split the rows, lane = position (row-major); the C sums of the direction are reduced
// over the waves in LDS.  Wx (4, C, D) is x_proj_weight AS STORED (scalar loads at stride D: a transposed copy per call cost a launch).
template <typename T, int R, int NW>
__global__ __launch_bounds__(64 * NW) void deep_xproj_kernel(const T *__restrict__ x, const float *__restrict__ WxT,
                                                            float *__restrict__ xdbl, const DeepGeo g) {
    constexpr int C = R + 2;
    extern __shared__ float s_red[];   // [NW][C][64]
    const int lane = threadIdx.x & 63, wave = rfl(threadIdx.x >> 6);
    const int b = blockIdx.y, k = blockIdx.z, p = blockIdx.x * 64 + lane;
    const int rows = g.D / NW, d0 = wave * rows;
    float acc[C];
#pragma unroll
    for (int c = 0; c < C; ++c) acc[c] = 0.f;
    const T *xp = x + ((size_t)b * g.D + d0) * g.L + p;
    const float *wp = WxT + (size_t)k * C * g.D + d0;
#pragma unroll 4
    for (int r = 0; r < rows; ++r) {
        const float v = to_f32(xp[(size_t)r * g.L]);
#pragma unroll
        for (int c = 0; c < C; ++c) acc[c] = fmaf(wp[(size_t)c * g.D + r], v, acc[c]);
    }
#pragma unroll
    for (int c = 0; c < C; ++c) s_red[(wave * C + c) * 64 + lane] = acc[c];
    lds_barrier();
    const int h = p / g.W, w = p - h * g.W, q = w * g.H + h;
    for (int c = wave; c < C; c += NW) {
        const float s = tree_sum<NW>(s_red + c * 64 + lane, C * 64);
        xdbl[(((size_t)b * 4 + k) * C + c) * g.L + ((k & 1) ? q : p)] = s;
    }
}

// ---- pieces shared by the two scan kernels ---------------------------------------------------------------------------------
template <int R>
struct XD {          // x_dbl of one direction at the lane's 4 scan positions
    float dt[R][4], Bv[4], Cv[4];
};
template <int R>
__device__ __forceinline__ void load_xd(const float *__restrict__ base /* (C, L) of (b, k) */, const int L, const int l0, XD<R> &s) {
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const float4 v = *reinterpret_cast<const float4 *>(base + (size_t)r * L + l0);
        s.dt[r][0] = v.x; s.dt[r][1] = v.y; s.dt[r][2] = v.z; s.dt[r][3] = v.w;
    }
    const float4 vb = *reinterpret_cast<const float4 *>(base + (size_t)R * L + l0);
    const float4 vc = *reinterpret_cast<const float4 *>(base + (size_t)(R + 1) * L + l0);
    s.Bv[0] = vb.x; s.Bv[1] = vb.y; s.Bv[2] = vb.z; s.Bv[3] = vb.w;
    s.Cv[0] = vc.x; s.Cv[1] = vc.y; s.Cv[2] = vc.z; s.Cv[3] = vc.w;
}

template <int R>
struct RowP {        // parameters of one (direction, row): wave-uniform
    float wdt[R], bias, A, Dk;
};
struct Weights {
    const float *Wdt, *dtb, *Alog, *Ds;   // (4, D, R), (4, D), (4 D), (4 D)
};
template <int R>
__device__ __forceinline__ RowP<R> load_row(const Weights &w, const int kd) {
    RowP<R> p;
#pragma unroll
    for (int r = 0; r < R; ++r) p.wdt[r] = w.Wdt[(size_t)kd * R + r];
    p.bias = w.dtb[kd];
    p.A = -expf(w.Alog[kd]);
    p.Dk = w.Ds[kd];
    return p;
}

// carry entering tile q from the tiles scanned before it: `up`: tiles 0 .. q-1 in that order, else WR-1 .. q+1
template <int WR>
__device__ __forceinline__ float carry_in(const float2 *tot, const int q, const bool up) {
    float h = 0.f;
    if (up) {
        for (int j = 0; j < q; ++j) { const float2 e = tot[j]; h = fmaf(e.x, h, e.y); }
    } else {
        for (int j = WR - 1; j > q; --j) { const float2 e = tot[j]; h = fmaf(e.x, h, e.y); }
    }
    return h;
}

// padded LDS image of one row: element (h, w) at h W + w + (h >> 2) S with S = max(1, 128 / H) — the column-wise readers (4
// consecutive h per lane, 32 lanes = 128 / H columns of H / 4 lanes) land on 32 different banks instead of one
__host__ __device__ __forceinline__ int img_pad(const int H) { return H >= 128 ? 1 : 128 / H; }
__host__ __device__ __forceinline__ int img_floats(const int H, const int W) { return (H * W + (H / 4) * img_pad(H) + 7) & ~3; }
struct Img {
    int base_r, idx_c[4];   // row-phase: base_r + i ; column-phase: idx_c[i]
};
__device__ __forceinline__ Img img_index(const DeepGeo &g, const int l0) {
    Img m;
    const int S = img_pad(g.H);
    const int hr = l0 / g.W;
    m.base_r = l0 + (hr >> 2) * S;
    const int wc = l0 / g.H, h0 = l0 - wc * g.H;
#pragma unroll
    for (int i = 0; i < 4; ++i) m.idx_c[i] = (h0 + i) * g.W + wc + (h0 >> 2) * S;
    return m;
}

// ---- forward ----------------------------------------------------------------------------------------------------------------
struct DeepFwdArgs {
    const void *x;      // (B, D, L) dtype T
    const float *xdbl;  // (B, 4, C, L)
    Weights w;
    float *y;           // (B, D, L) fp32: the merged output
};

// one direction of one row tile: y += C h + D u
template <int R, int WR>
__device__ __forceinline__ void fwd_dir(const int k, const int q, const int lane, const float (&u)[4], const XD<R> &s, const RowP<R> &w,
                                        float2 *tot, float (&y)[4]) {
    float av[4], bv[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        float pre = w.bias;
#pragma unroll
        for (int r = 0; r < R; ++r) pre = fmaf(w.wdt[r], s.dt[r][i], pre);
        const float dl = softplus_f(pre);
        av[i] = decay_f(dl, w.A);
        bv[i] = dl * u[i] * s.Bv[i];
    }
    const bool up = k < 2;
    Pair agg, excl, t;
    if (up) {
        agg = Pair{av[0], bv[0]};
#pragma unroll
        for (int i = 1; i < 4; ++i) agg = then(agg, Pair{av[i], bv[i]});
        wave_scan_fwd(agg, lane, excl, t);
    } else {
        agg = Pair{av[3], bv[3]};
#pragma unroll
        for (int i = 2; i >= 0; --i) agg = then(agg, Pair{av[i], bv[i]});
        wave_scan_rev(agg, lane, excl, t);
    }
    float hin = 0.f;
    if constexpr (WR > 1) {
        if (lane == 0) tot[q] = make_float2(t.a, t.b);
        lds_barrier();
        hin = carry_in<WR>(tot, q, up);
    }
    float h = fmaf(excl.a, hin, excl.b);
    if (up) {
#pragma unroll
        for (int i = 0; i < 4; ++i) { h = fmaf(av[i], h, bv[i]); y[i] += fmaf(h, s.Cv[i], w.Dk * u[i]); }
    } else {
#pragma unroll
        for (int i = 3; i >= 0; --i) { h = fmaf(av[i], h, bv[i]); y[i] += fmaf(h, s.Cv[i], w.Dk * u[i]); }
    }
}

// grid (D / RG, B); workgroup = RG rows x WR waves (WR = L / 256: one 256-position tile per wave)
template <typename T, int R, int WR, int RG>
__global__ __launch_bounds__(64 * WR * RG) void deep_fwd_kernel(const DeepFwdArgs a, const DeepGeo g) {
    constexpr int C = R + 2;
    extern __shared__ __attribute__((aligned(16))) float s_lds[];
    const int lane = threadIdx.x & 63, wave = rfl(threadIdx.x >> 6);
    const int rg = wave / WR, q = wave % WR;
    const int IMG = img_floats(g.H, g.W);
    float *uimg = s_lds + (size_t)rg * 2 * IMG, *yimg = uimg + IMG;
    float2 *tot = reinterpret_cast<float2 *>(s_lds + (size_t)RG * 2 * IMG) + (size_t)rg * 4 * WR;
    const int b = blockIdx.y, d = blockIdx.x * RG + rg, D = g.D, L = g.L;
    const int l0 = q * kTile + lane * kItems;
    const Img m = img_index(g, l0);
    const size_t row = ((size_t)b * D + d) * L;

    float u[4], y[4] = {0.f, 0.f, 0.f, 0.f};
    load4<T, true>(static_cast<const T *>(a.x) + row, l0, L, u);
#pragma unroll
    for (int i = 0; i < 4; ++i) uimg[m.base_r + i] = u[i];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
        const int k = 2 * kk;
        XD<R> s;
        load_xd<R>(a.xdbl + ((size_t)b * 4 + k) * C * L, L, l0, s);
        const RowP<R> w = load_row<R>(a.w, k * D + d);
        fwd_dir<R, WR>(k, q, lane, u, s, w, tot + k * WR, y);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) yimg[m.base_r + i] = y[i];
    lds_barrier();
    float uc[4], yc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 4; ++i) uc[i] = uimg[m.idx_c[i]];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
        const int k = 1 + 2 * kk;
        XD<R> s;
        load_xd<R>(a.xdbl + ((size_t)b * 4 + k) * C * L, L, l0, s);
        const RowP<R> w = load_row<R>(a.w, k * D + d);
        fwd_dir<R, WR>(k, q, lane, uc, s, w, tot + k * WR, yc);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) yimg[m.idx_c[i]] += yc[i];
    lds_barrier();
    *reinterpret_cast<float4 *>(a.y + row + l0) =
        make_float4(yimg[m.base_r], yimg[m.base_r + 1], yimg[m.base_r + 2], yimg[m.base_r + 3]);
}

// ---- backward ---------------------------------------------------------------------------------------------------------------
// Per (direction, row, position) the scan's backward yields du and three TERMS whose sums over the rows are the gradient of
// x_dbl:  tp = d(loss)/d(pre-softplus delta)  (d dt_r = sum_d W_dt[d][r] tp),  tb = g delta u (dB = sum_d tb),
// tc = dy h (dC = sum_d tc).  The terms are written in the direction's own scan order (row-major for 0 / 2, column-major for
// 1 / 3: contiguous stores either way); the adjoint of x_proj (deep_xg_kernel) reads them in that order, coalesced, and puts its
// small result g at the row-major position.
constexpr int kPG = 20;   // per-(b, direction, row, wave) parameter sums: dWdt[0..R-1], dbias, dAlog, dD (R + 3 <= 19)
struct DeepBwdArgs {
    const void *x;
    const float *xdbl, *dy;   // dy (B, D, L) fp32: gradient of the merged output
    Weights w;
    float *du;                // (B, D, L) fp32: the scan's own gradient wrt x, merged over the directions
    void *tp, *tb, *tc;       // (B, 4, D, L) dtype T each
    float *pg;                // (B, WR, 4, D, kPG): slab-major, so that csrc/wgrad.hip sums the B * WR slabs of a (4 D, kPG) matrix
};

template <int R, int WR>
__device__ __forceinline__ void bwd_dir(const int k, const int q, const int lane, const float (&u)[4], const float (&dout)[4],
                                        const XD<R> &s, const RowP<R> &w, float2 *tot, float2 *adj, float (&du)[4], float (&tp)[4],
                                        float (&tb)[4], float (&tc)[4], float *__restrict__ pg) {
    float dl[4], sig[4], av[4], bv[4], be[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        float pre = w.bias;
#pragma unroll
        for (int r = 0; r < R; ++r) pre = fmaf(w.wdt[r], s.dt[r][i], pre);
        softplus_sigmoid_f(pre, dl[i], sig[i]);
        av[i] = decay_f(dl[i], w.A);
        bv[i] = dl[i] * u[i] * s.Bv[i];
        be[i] = dout[i] * s.Cv[i];
    }
    const bool up = k < 2;
    // adjoint elements (a_t, a_t beta_t) composed AGAINST the scan order (ss2d.hip: G_t = a_t g_t)
    Pair ragg, rexcl, rtot, agg, excl, t;
    if (up) {
        ragg = Pair{av[3], av[3] * be[3]};
#pragma unroll
        for (int i = 2; i >= 0; --i) ragg = then(ragg, Pair{av[i], av[i] * be[i]});
        wave_scan_rev(ragg, lane, rexcl, rtot);
        agg = Pair{av[0], bv[0]};
#pragma unroll
        for (int i = 1; i < 4; ++i) agg = then(agg, Pair{av[i], bv[i]});
        wave_scan_fwd(agg, lane, excl, t);
    } else {
        ragg = Pair{av[0], av[0] * be[0]};
#pragma unroll
        for (int i = 1; i < 4; ++i) ragg = then(ragg, Pair{av[i], av[i] * be[i]});
        wave_scan_fwd(ragg, lane, rexcl, rtot);
        agg = Pair{av[3], bv[3]};
#pragma unroll
        for (int i = 2; i >= 0; --i) agg = then(agg, Pair{av[i], bv[i]});
        wave_scan_rev(agg, lane, excl, t);
    }
    float hin = 0.f, Gin = 0.f;
    if constexpr (WR > 1) {
        if (lane == 0) { tot[q] = make_float2(t.a, t.b); adj[q] = make_float2(rtot.a, rtot.b); }
        lds_barrier();
        hin = carry_in<WR>(tot, q, up);
        Gin = carry_in<WR>(adj, q, !up);
    }
    float hv[4];
    float h = fmaf(excl.a, hin, excl.b);
    if (up) {
#pragma unroll
        for (int i = 0; i < 4; ++i) { h = fmaf(av[i], h, bv[i]); hv[i] = h; }
    } else {
#pragma unroll
        for (int i = 3; i >= 0; --i) { h = fmaf(av[i], h, bv[i]); hv[i] = h; }
    }
    float Gnext = fmaf(rexcl.a, Gin, rexcl.b);
    float accDt[R], accBias = 0.f, accA = 0.f, accD = 0.f;
#pragma unroll
    for (int r = 0; r < R; ++r) accDt[r] = 0.f;
#pragma unroll
    for (int ii = 0; ii < 4; ++ii) {
        const int i = up ? 3 - ii : ii;            // against the scan order
        const float gcur = be[i] + Gnext;          // adjoint of h at this step
        Gnext = av[i] * gcur;
        const float gB = gcur * s.Bv[i];
        const float ax = hv[i] - bv[i];            // a_t h_{t-1}
        du[i] += fmaf(gB, dl[i], w.Dk * dout[i]);
        const float dd = fmaf(gB, u[i], gcur * w.A * ax) * sig[i];
        accA = fmaf(gcur * dl[i], ax, accA);
        accD = fmaf(dout[i], u[i], accD);
#pragma unroll
        for (int r = 0; r < R; ++r) accDt[r] = fmaf(dd, s.dt[r][i], accDt[r]);
        accBias += dd;
        tp[i] = dd;
        tb[i] = gcur * dl[i] * u[i];
        tc[i] = dout[i] * hv[i];
    }
    if constexpr (R <= 4) {
        float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int r = 0; r < R; ++r) v[r] = accDt[r];
        v[R] = accBias; v[R + 1] = accA * w.A; v[R + 2] = accD;    // dA_log = dA * A
        const float s8 = wave_sum8(v, lane);
        if (lane < R + 3) pg[lane] = s8;
    } else {                                   // R = 8, 16: butterflies of 8 for dWdt, one of 4 for the rest
#pragma unroll
        for (int r0 = 0; r0 < R; r0 += 8) {
            float v[8];
#pragma unroll
            for (int r = 0; r < 8; ++r) v[r] = accDt[r0 + r];
            const float s8 = wave_sum8(v, lane);
            if (lane < 8) pg[r0 + lane] = s8;
        }
        const float v4[4] = {accBias, accA * w.A, accD, 0.f};
        const float s4 = wave_sum4(v4, lane);
        if (lane < 3) pg[R + lane] = s4;       // lane l holds the total of v4[l & 3]
    }
}

template <typename T, int R, int WR, int RG>
__global__ __launch_bounds__(64 * WR * RG) void deep_bwd_kernel(const DeepBwdArgs a, const DeepGeo g) {
    constexpr int C = R + 2;
    extern __shared__ __attribute__((aligned(16))) float s_lds[];
    const int lane = threadIdx.x & 63, wave = rfl(threadIdx.x >> 6);
    const int rg = wave / WR, q = wave % WR;
    const int IMG = img_floats(g.H, g.W);
    float *uimg = s_lds + (size_t)rg * 3 * IMG, *dyimg = uimg + IMG, *duimg = uimg + 2 * IMG;
    float2 *tot = reinterpret_cast<float2 *>(s_lds + (size_t)RG * 3 * IMG) + (size_t)rg * 8 * WR, *adj = tot + 4 * WR;
    const int b = blockIdx.y, d = blockIdx.x * RG + rg, D = g.D, L = g.L;
    const int l0 = q * kTile + lane * kItems;
    const Img m = img_index(g, l0);
    const size_t row = ((size_t)b * D + d) * L;

    float u[4], dout[4], du[4] = {0.f, 0.f, 0.f, 0.f};
    load4<T, true>(static_cast<const T *>(a.x) + row, l0, L, u);
    load4<float, true>(a.dy + row, l0, L, dout);
#pragma unroll
    for (int i = 0; i < 4; ++i) { uimg[m.base_r + i] = u[i]; dyimg[m.base_r + i] = dout[i]; }
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
        const int k = 2 * kk;
        XD<R> s;
        load_xd<R>(a.xdbl + ((size_t)b * 4 + k) * C * L, L, l0, s);
        const RowP<R> w = load_row<R>(a.w, k * D + d);
        float tp[4], tb[4], tc[4];
        bwd_dir<R, WR>(k, q, lane, u, dout, s, w, tot + k * WR, adj + k * WR, du, tp, tb, tc,
                       a.pg + ((((size_t)b * WR + q) * 4 + k) * D + d) * kPG);
        const size_t trow = (((size_t)b * 4 + k) * D + d) * L;
        store4<T, true>(static_cast<T *>(a.tp) + trow, l0, L, tp);
        store4<T, true>(static_cast<T *>(a.tb) + trow, l0, L, tb);
        store4<T, true>(static_cast<T *>(a.tc) + trow, l0, L, tc);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) duimg[m.base_r + i] = du[i];
    lds_barrier();
    float uc[4], dc[4], duc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 4; ++i) { uc[i] = uimg[m.idx_c[i]]; dc[i] = dyimg[m.idx_c[i]]; }
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
        const int k = 1 + 2 * kk;
        XD<R> s;
        load_xd<R>(a.xdbl + ((size_t)b * 4 + k) * C * L, L, l0, s);
        const RowP<R> w = load_row<R>(a.w, k * D + d);
        float tp[4], tb[4], tc[4];
        bwd_dir<R, WR>(k, q, lane, uc, dc, s, w, tot + k * WR, adj + k * WR, duc, tp, tb, tc,
                       a.pg + ((((size_t)b * WR + q) * 4 + k) * D + d) * kPG);
        const size_t trow = (((size_t)b * 4 + k) * D + d) * L;      // (column-major order, like the scan: deep_xg_kernel maps back)
        store4<T, true>(static_cast<T *>(a.tp) + trow, l0, L, tp);
        store4<T, true>(static_cast<T *>(a.tb) + trow, l0, L, tb);
        store4<T, true>(static_cast<T *>(a.tc) + trow, l0, L, tc);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) duimg[m.idx_c[i]] += duc[i];
    lds_barrier();
    *reinterpret_cast<float4 *>(a.du + row + l0) =
        make_float4(duimg[m.base_r], duimg[m.base_r + 1], duimg[m.base_r + 2], duimg[m.base_r + 3]);
}

// ---- adjoint of x_proj: terms -> d(x_dbl) per position -> dx ----------------------------------------------------------------
// deep_xg_kernel, grid (L / 64, B, 4 directions): g[k][c] = sum over the rows of the direction's terms (d dt_r through W_dt),
// NW waves split the rows, LDS reduce; written (B, 4 C, L) in fp32 (for dx) and in dtype T (dW_x[k][c][d] =
// sum_{b,p} g[b][k][c][p] x[b][d][p] is one small GEMM on the host side).
// deep_dx_kernel, grid (L / 64, B, D / 16): dx = du + sum_kc W_x[k][c][d] g[k][c] — no reduction, 4 rows per wave.
struct DeepXgArgs {
    const void *tp, *tb, *tc;
    const float *Wdt;
    float *g32;
    void *gpos;
};
template <typename T, int R, int NW>
__global__ __launch_bounds__(64 * NW) void deep_xg_kernel(const DeepXgArgs a, const DeepGeo g) {
    constexpr int C = R + 2;
    extern __shared__ float s_red[];   // [NW][C][64]
    const int lane = threadIdx.x & 63, wave = rfl(threadIdx.x >> 6);
    const int b = blockIdx.y, k = blockIdx.z, p = blockIdx.x * 64 + lane;
    const int D = g.D, L = g.L, rows = D / NW, d0 = wave * rows;
    float acc[2][C];   // even / odd rows: two shorter chains
#pragma unroll
    for (int c = 0; c < C; ++c) acc[0][c] = acc[1][c] = 0.f;
    const size_t o0 = (((size_t)b * 4 + k) * D + d0) * L + p;
    const T *tp = static_cast<const T *>(a.tp) + o0, *tb = static_cast<const T *>(a.tb) + o0, *tc = static_cast<const T *>(a.tc) + o0;
    const float *wp = a.Wdt + ((size_t)k * D + d0) * R;
#pragma unroll 2
    for (int r = 0; r < rows; r += 2) {
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const size_t o = (size_t)(r + e) * L;
            const float vp = to_f32(tp[o]), vb = to_f32(tb[o]), vc = to_f32(tc[o]);
#pragma unroll
            for (int rr = 0; rr < R; ++rr) acc[e][rr] = fmaf(wp[(r + e) * R + rr], vp, acc[e][rr]);
            acc[e][R] += vb;
            acc[e][R + 1] += vc;
        }
    }
#pragma unroll
    for (int c = 0; c < C; ++c) s_red[(wave * C + c) * 64 + lane] = acc[0][c] + acc[1][c];
    lds_barrier();
    T *gp = static_cast<T *>(a.gpos);
    // directions 1 / 3: the terms (and this tile) are in column-major order q = w H + h; g goes to the row-major position h W + w
    const int pos = (k & 1) ? (p % g.H) * g.W + p / g.H : p;
    for (int c = wave; c < C; c += NW) {
        const float s = tree_sum<NW>(s_red + c * 64 + lane, C * 64);
        const size_t o = (((size_t)b * 4 + k) * C + c) * L + pos;
        a.g32[o] = s;
        gp[o] = from_f32<T>(s);
    }
}

template <typename T, int R>
__global__ __launch_bounds__(256) void deep_dx_kernel(const float *__restrict__ g32, const float *__restrict__ du,
                                                       const float *__restrict__ WxT, T *__restrict__ dx, const DeepGeo g) {
    constexpr int C = R + 2, KC = 4 * C;
    const int lane = threadIdx.x & 63, wave = rfl(threadIdx.x >> 6);
    const int b = blockIdx.y, p = blockIdx.x * 64 + lane, d0 = blockIdx.z * 16 + wave * 4;
    const int D = g.D, L = g.L;
    float gv[KC];
#pragma unroll
    for (int kc = 0; kc < KC; ++kc) gv[kc] = g32[((size_t)b * KC + kc) * L + p];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int d = d0 + r;
        const size_t o = ((size_t)b * D + d) * L + p;
        // per direction first, then the four directions pairwise as CrossMerge adds them, then the scan's own du: the
        // association of the reference chain (einsum backward per direction -> merge), small partial sums before large ones
        float sk[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float *wr = WxT + (size_t)k * C * D + d;          // x_proj_weight as stored: (4, C, D)
            float s0 = wr[0] * gv[k * C], s1 = wr[D] * gv[k * C + 1];
#pragma unroll
            for (int c = 2; c < C; c += 2) { s0 = fmaf(wr[(size_t)c * D], gv[k * C + c], s0); s1 = fmaf(wr[(size_t)(c + 1) * D], gv[k * C + c + 1], s1); }
            sk[k] = s0 + s1;
        }
        dx[o] = from_f32<T>(du[o] + ((sk[0] + sk[2]) + (sk[1] + sk[3])));
    }
}

// ---- host side -------------------------------------------------------------------------------------------------------------
struct DeepCfg {
    int WR, RG, NW;
};
bool deep_cfg(int R, int D, int H, int W, DeepCfg &c) {
    if (!(R == 2 || R == 4 || R == 8 || R == 16)) return false;
    if (D < 64 || D > 512 || D % 32) return false;
    if (H % 4 || W % 4) return false;
    const long L = (long)H * W;
    if (L % kTile || L > 4096) return false;
    const int WR = (int)(L / kTile);
    if (!(WR == 1 || WR == 2 || WR == 4 || WR == 8 || WR == 16)) return false;
    if (R == 16 && WR > 2) return false;     // (dt_rank 16 = d_inner 512: the 16 x 16 stage of the DIMS-32 configs)
    c.WR = WR;
    c.RG = WR >= 8 ? 1 : (WR >= 2 ? 2 : 4);
    c.NW = D >= 256 ? 16 : 8;
    return true;
}
size_t scan_lds(const DeepCfg &c, int H, int W, int images, int tots) {
    const size_t IMG = (size_t)img_floats(H, W);
    return ((size_t)c.RG * images * IMG) * 4 + (size_t)c.RG * tots * c.WR * 8;
}

// workgroups above 64 KB of dynamic LDS (the backward's six row images at H*W = 4096: 99 KB of the CU's 160 KB) need the
// function attribute; set once per kernel
template <typename K>
void allow_lds(K kernel, size_t bytes) {
    if (bytes <= 32 * 1024) return;
    static std::mutex mu;
    static std::unordered_set<const void *> done;
    const void *f = reinterpret_cast<const void *>(kernel);
    std::lock_guard<std::mutex> lk(mu);
    if (done.insert(f).second) (void)hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
}

#define DEEP_WRRG(KERNEL, T, R, ...)                                              \
    do {                                                                          \
        if (c.WR == 16) { DEEP_GO((KERNEL<T, R, 16, 1>), __VA_ARGS__); }          \
        else if (c.WR == 8) { DEEP_GO((KERNEL<T, R, 8, 1>), __VA_ARGS__); }       \
        else if (c.WR == 4) { DEEP_GO((KERNEL<T, R, 4, 2>), __VA_ARGS__); }       \
        else if (c.WR == 2) { DEEP_GO((KERNEL<T, R, 2, 2>), __VA_ARGS__); }       \
        else { DEEP_GO((KERNEL<T, R, 1, 4>), __VA_ARGS__); }                      \
    } while (0)
#define DEEP_R(KERNEL, T, ...)                                                    \
    do {                                                                          \
        if (p.R == 2) DEEP_WRRG(KERNEL, T, 2, __VA_ARGS__);                       \
        else if (p.R == 4) DEEP_WRRG(KERNEL, T, 4, __VA_ARGS__);                  \
        else if (p.R == 8) DEEP_WRRG(KERNEL, T, 8, __VA_ARGS__);                  \
        else if (c.WR == 1) { DEEP_GO((KERNEL<T, 16, 1, 4>), __VA_ARGS__); }      \
        else { DEEP_GO((KERNEL<T, 16, 2, 2>), __VA_ARGS__); }   /* dt_rank 16: H*W <= 512 only (deep_cfg) */ \
    } while (0)
#define DEEP_NW(KERNEL, T, R, ...)                                                \
    do {                                                                          \
        if (c.NW == 16) { DEEP_GO((KERNEL<T, R, 16>), __VA_ARGS__); }             \
        else { DEEP_GO((KERNEL<T, R, 8>), __VA_ARGS__); }                         \
    } while (0)
#define DEEP_RNW(KERNEL, T, ...)                                                  \
    do {                                                                          \
        if (p.R == 2) DEEP_NW(KERNEL, T, 2, __VA_ARGS__);                         \
        else if (p.R == 4) DEEP_NW(KERNEL, T, 4, __VA_ARGS__);                    \
        else if (p.R == 8) DEEP_NW(KERNEL, T, 8, __VA_ARGS__);                    \
        else DEEP_NW(KERNEL, T, 16, __VA_ARGS__);                                 \
    } while (0)

template <typename T>
int deep_fwd(const vmasr_ss2d_deep_params &p, const DeepCfg &c, hipStream_t st) {
    const DeepGeo g{p.B, p.D, p.H, p.W, p.H * p.W};
    const int C = p.R + 2;
    const double el = (double)p.B * p.D * g.L, pos = (double)p.B * g.L;
    {
        const dim3 grid(g.L / 64, p.B, 4), block(64 * c.NW);
        const size_t sm = (size_t)c.NW * C * 64 * 4;
#define DEEP_GO(K, ...) do { allow_lds(K, sm); VMASR_LAUNCH(VMASR_K_SS2D_DEEP_XPROJ, el * sizeof(T) + pos * 4 * C * 4, K, grid, block, sm, st, __VA_ARGS__); } while (0)
        DEEP_RNW(deep_xproj_kernel, T, static_cast<const T *>(p.x), p.WxT, p.xdbl, g);
#undef DEEP_GO
    }
    {
        const dim3 grid(p.D / c.RG, p.B), block(64 * c.WR * c.RG);
        const size_t sm = scan_lds(c, p.H, p.W, 2, 4);
        const DeepFwdArgs a{p.x, p.xdbl, {p.Wdt, p.dtb, p.Alog, p.Ds}, p.y};
        // bytes for the profiler: the ALGORITHMIC bytes of the selective-scan call this launch performs (as
        // ss2d.hip reports them for its apply kernels)
#define DEEP_GO(K, ...) do { allow_lds(K, sm); VMASR_LAUNCH(VMASR_K_SS2D_DEEP_FWD, (3.0 * 4 * el + 2.0 * 4 * pos) * 4, K, grid, block, sm, st, __VA_ARGS__); } while (0)
        DEEP_R(deep_fwd_kernel, T, a, g);
#undef DEEP_GO
    }
    return check_launch("ss2d_deep_fwd");
}

template <typename T>
int deep_bwd(const vmasr_ss2d_deep_params &p, const DeepCfg &c, hipStream_t st) {
    const DeepGeo g{p.B, p.D, p.H, p.W, p.H * p.W};
    const int C = p.R + 2;
    const double el = (double)p.B * p.D * g.L, pos = (double)p.B * g.L;
    {
        const dim3 grid(p.D / c.RG, p.B), block(64 * c.WR * c.RG);
        const size_t sm = scan_lds(c, p.H, p.W, 3, 8);
        const DeepBwdArgs a{p.x, p.xdbl, p.dy, {p.Wdt, p.dtb, p.Alog, p.Ds}, p.du, p.tp, p.tb, p.tc, p.pg};
#define DEEP_GO(K, ...) do { allow_lds(K, sm); VMASR_LAUNCH(VMASR_K_SS2D_DEEP_BWD, (5.0 * 4 * el + 4.0 * 4 * pos) * 4, K, grid, block, sm, st, __VA_ARGS__); } while (0)
        DEEP_R(deep_bwd_kernel, T, a, g);
#undef DEEP_GO
    }
    {
        const dim3 grid(g.L / 64, p.B, 4), block(64 * c.NW);
        const size_t sm = (size_t)c.NW * C * 64 * 4;
        const DeepXgArgs a{p.tp, p.tb, p.tc, p.Wdt, p.g32, p.gpos};
#define DEEP_GO(K, ...) do { allow_lds(K, sm); VMASR_LAUNCH(VMASR_K_SS2D_DEEP_XBWD, el * 12.0 * sizeof(T), K, grid, block, sm, st, __VA_ARGS__); } while (0)
        DEEP_RNW(deep_xg_kernel, T, a, g);
#undef DEEP_GO
    }
    {
        const dim3 grid(g.L / 64, p.B, p.D / 16), block(256);
        const double bytes = el * (4.0 + sizeof(T));
        T *dx = static_cast<T *>(p.dx);
        if (p.R == 2) VMASR_LAUNCH(VMASR_K_SS2D_DEEP_XBWD, bytes, (deep_dx_kernel<T, 2>), grid, block, 0, st, p.g32, p.du, p.WxT, dx, g);
        else if (p.R == 4) VMASR_LAUNCH(VMASR_K_SS2D_DEEP_XBWD, bytes, (deep_dx_kernel<T, 4>), grid, block, 0, st, p.g32, p.du, p.WxT, dx, g);
        else if (p.R == 16) VMASR_LAUNCH(VMASR_K_SS2D_DEEP_XBWD, bytes, (deep_dx_kernel<T, 16>), grid, block, 0, st, p.g32, p.du, p.WxT, dx, g);
        else VMASR_LAUNCH(VMASR_K_SS2D_DEEP_XBWD, bytes, (deep_dx_kernel<T, 8>), grid, block, 0, st, p.g32, p.du, p.WxT, dx, g);
    }
    return check_launch("ss2d_deep_bwd");
}

int deep_check(const vmasr_ss2d_deep_params &p, DeepCfg &c, const char *what) {
    VMASR_REQUIRE(p.B > 0 && p.B <= 65535, VMASR_EINVAL, "%s: bad batch %d", what, p.B);
    VMASR_REQUIRE(deep_cfg(p.R, p.D, p.H, p.W, c), VMASR_EINVAL, "%s: unsupported shape (dt_rank %d, d_inner %d, %d x %d)", what, p.R,
                  p.D, p.H, p.W);
    VMASR_REQUIRE(p.D % (c.RG) == 0 && p.D % (2 * c.NW) == 0 && p.D % 16 == 0, VMASR_EINVAL, "%s: d_inner %d not divisible", what, p.D);
    VMASR_REQUIRE(p.dtype == VMASR_F32 || p.dtype == VMASR_BF16, VMASR_EINVAL, "%s: dtype must be fp32 or bf16", what);
    VMASR_REQUIRE(p.x && p.WxT && p.Wdt && p.dtb && p.Alog && p.Ds && p.xdbl, VMASR_EINVAL, "%s: null tensor", what);
    return 0;
}

}  // namespace
}  // namespace vmasr

using namespace vmasr;

VMASR_EXPORT int vmasr_ss2d_deep_supported(int32_t d_state, int32_t dt_rank, int32_t d_inner, int32_t H, int32_t W) {
    DeepCfg c;
    return (d_state == 1 && deep_cfg(dt_rank, d_inner, H, W, c) && d_inner % c.RG == 0 && d_inner % (2 * c.NW) == 0) ? 1 : 0;
}

VMASR_EXPORT int32_t vmasr_ss2d_deep_waves_per_row(int32_t H, int32_t W) { return (int32_t)(((long)H * W) / kTile); }

VMASR_EXPORT int vmasr_ss2d_deep_fwd(const vmasr_ss2d_deep_params *pp, vmasr_stream_t stream) {
    VMASR_REQUIRE(pp, VMASR_EINVAL, "ss2d_deep_fwd: null params");
    const vmasr_ss2d_deep_params &p = *pp;
    DeepCfg c;
    if (int e = deep_check(p, c, "ss2d_deep_fwd")) return e;
    VMASR_REQUIRE(p.y, VMASR_EINVAL, "ss2d_deep_fwd: null output");
    hipStream_t st = static_cast<hipStream_t>(stream);
    return p.dtype == VMASR_F32 ? deep_fwd<float>(p, c, st) : deep_fwd<bf16_t>(p, c, st);
}

VMASR_EXPORT int vmasr_ss2d_deep_bwd(const vmasr_ss2d_deep_params *pp, vmasr_stream_t stream) {
    VMASR_REQUIRE(pp, VMASR_EINVAL, "ss2d_deep_bwd: null params");
    const vmasr_ss2d_deep_params &p = *pp;
    DeepCfg c;
    if (int e = deep_check(p, c, "ss2d_deep_bwd")) return e;
    VMASR_REQUIRE(p.dy && p.du && p.tp && p.tb && p.tc && p.pg && p.dx && p.gpos && p.g32, VMASR_EINVAL, "ss2d_deep_bwd: null buffer");
    hipStream_t st = static_cast<hipStream_t>(stream);
    return p.dtype == VMASR_F32 ? deep_bwd<float>(p, c, st) : deep_bwd<bf16_t>(p, c, st);
}
