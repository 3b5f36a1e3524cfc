// ln.hip — channel-last LayerNorm over the last dimension, forward and backward, for gfx950.
//
// VM-ASR normalises (B,H,W,C) activations with C between 1 and 512 and up to 10^6 rows per
// call: SS2D.out_norm (model/vmamba.py:767-769,1528-1529), VSSBlock.norm / norm2 (:1793,1817),
// PatchMerging2D.norm (model/model.py:70), PatchExpanding.norm (:105-108) and the patch-embed
// norms (:620,631).  ATen's LayerNorm kernels are built for wide rows: on these shapes they cost
// 58 us (forward), 58 + 110 us (backward) and 2.9 ms for a 1M x 4 call (profiles/r01_*), 16 % of
// the GPU time of a training step.  This file is the HBM-bound replacement:
//
//   forward : y = (x - mean) * rstd * gamma + beta         reads x, writes y (+ mean, rstd fp32)
//   backward: dx = rstd * (g*gamma - mean_c(g*gamma) - xhat * mean_c(g*gamma*xhat)),
//             dgamma = sum_rows g*xhat,  dbeta = sum_rows g   reads x, g; writes dx (+ partials)
//
// A row is owned by LPR = 1..64 adjacent lanes (a power of two), each holding a 4-element
// vector per 4*LPR columns; the two row reductions are xor-shuffles inside the lane group.
// Rows are walked grid-stride so that the per-column dgamma/dbeta partial sums stay in
// registers; one partial row per workgroup goes to scratch and a second small kernel sums them
// (deterministic, no atomics).
#include "common.h"

#include <algorithm>

namespace vmasr {
namespace {

constexpr int kMaxVecPerLane = 4;  // C <= 64 lanes * 4 elems * 4 vectors = 1024

// sum over the LPR adjacent lanes of a row group (LPR | 64), in every lane of the group: DPP for distances <= 8
// (common.h: quad_perm / row shifts under bank masks), ds_bpermute only for 16 and 32
template <int LPR>
__device__ __forceinline__ float group_sum(float v) {
    if constexpr (LPR >= 2) v += xor1(v);
    if constexpr (LPR >= 4) v += xor2(v);
    if constexpr (LPR >= 8) v += xor4(v);
    if constexpr (LPR >= 16) v += xor8(v);
    if constexpr (LPR >= 32) v += __shfl_xor(v, 16);
    if constexpr (LPR >= 64) v += __shfl_xor(v, 32);
    return v;
}

struct LnGeom {
    int rows, C, nvec;  // nvec = vectors per lane = ceil(C / (4*LPR))
    float eps;
};

// loads this lane's columns of one row into v[nv][4] (zeros beyond C or when !row_ok)
template <typename T, int LPR, bool VEC>
__device__ __forceinline__ void load_cols(const T *__restrict__ row, bool row_ok, int sub, const LnGeom g,
                                          float (&v)[kMaxVecPerLane][4]) {
#pragma unroll
    for (int k = 0; k < kMaxVecPerLane; ++k) {
        const int c0 = (k * LPR + sub) * 4;
        if (k < g.nvec && row_ok) load4<T, VEC>(row, c0, g.C, v[k]);
        else { v[k][0] = v[k][1] = v[k][2] = v[k][3] = 0.f; }
    }
}

template <typename T, typename TO, int LPR, bool VEC>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const T *__restrict__ x, const float *__restrict__ gamma,
                                                     const float *__restrict__ beta, TO *__restrict__ y,
                                                     float *__restrict__ mean, float *__restrict__ rstd, const LnGeom g) {
    constexpr int RPB = 256 / LPR;  // rows per block iteration
    const int sub = threadIdx.x % LPR, rloc = threadIdx.x / LPR;
    float gm[kMaxVecPerLane][4], bt[kMaxVecPerLane][4];
#pragma unroll
    for (int k = 0; k < kMaxVecPerLane; ++k)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int c = (k * LPR + sub) * 4 + i;
            const bool ok = k < g.nvec && c < g.C;
            gm[k][i] = ok ? (gamma ? gamma[c] : 1.f) : 0.f;
            bt[k][i] = ok ? (beta ? beta[c] : 0.f) : 0.f;
        }
    const float invC = 1.f / (float)g.C;
    for (long r = (long)blockIdx.x * RPB + rloc; r - rloc < g.rows; r += (long)gridDim.x * RPB) {
        const bool ok = r < g.rows;
        float v[kMaxVecPerLane][4];
        load_cols<T, LPR, VEC>(x + r * g.C, ok, sub, g, v);
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < kMaxVecPerLane; ++k) s += (v[k][0] + v[k][1]) + (v[k][2] + v[k][3]);
        const float mu = group_sum<LPR>(s) * invC;
        float q = 0.f;
#pragma unroll
        for (int k = 0; k < kMaxVecPerLane; ++k)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int c = (k * LPR + sub) * 4 + i;
                const float d = (k < g.nvec && c < g.C) ? v[k][i] - mu : 0.f;
                q = fmaf(d, d, q);
            }
        const float rs = rsqrtf(group_sum<LPR>(q) * invC + g.eps);
        if (ok) {
#pragma unroll
            for (int k = 0; k < kMaxVecPerLane; ++k)
                if (k < g.nvec) {
                    float o[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) o[i] = fmaf((v[k][i] - mu) * rs, gm[k][i], bt[k][i]);
                    store4<TO, VEC>(y + r * g.C, (k * LPR + sub) * 4, g.C, o);
                }
            if (sub == 0) { mean[r] = mu; rstd[r] = rs; }
        }
    }
}

// `residual`: optional (rows, C) tensor of x's dtype added to dx — the gradient that arrives over the residual connection
// around a pre-norm branch (vm_asr_amd/mlp.py).
// (dgamma / dbeta stay per-workgroup partials + ln_bwd_reduce_kernel.  Measured in round 3: folding the reduction into this
//  kernel — "last workgroup sums the partials", or device-scope float atomics into per-stream accumulators with a ticket —
//  is SLOWER than the 9 us reduce launch it removes: an agent-scope __threadfence() writes the L2 back (70 us per launch),
//  and without it 1 024 workgroups' atomics serialise per cache line (20-36 us per launch against 10 + 9).)
template <typename T, typename TG, int LPR, bool VEC>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const T *__restrict__ x, const TG *__restrict__ gy,
                                                     const float *__restrict__ gamma, const float *__restrict__ mean,
                                                     const float *__restrict__ rstd, T *__restrict__ dx,
                                                     float *__restrict__ part, const LnGeom g, const T *__restrict__ residual) {
    constexpr int RPB = 256 / LPR;
    __shared__ float s_acc[2][1024];  // reused per vector slot
    const int sub = threadIdx.x % LPR, rloc = threadIdx.x / LPR;
    float gm[kMaxVecPerLane][4], dg[kMaxVecPerLane][4], db[kMaxVecPerLane][4];
#pragma unroll
    for (int k = 0; k < kMaxVecPerLane; ++k)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int c = (k * LPR + sub) * 4 + i;
            gm[k][i] = (k < g.nvec && c < g.C) ? (gamma ? gamma[c] : 1.f) : 0.f;
            dg[k][i] = 0.f;
            db[k][i] = 0.f;
        }
    const float invC = 1.f / (float)g.C;
    for (long r = (long)blockIdx.x * RPB + rloc; r - rloc < g.rows; r += (long)gridDim.x * RPB) {
        const bool ok = r < g.rows;
        float v[kMaxVecPerLane][4], gv[kMaxVecPerLane][4];
        load_cols<T, LPR, VEC>(x + r * g.C, ok, sub, g, v);
        load_cols<TG, LPR, VEC>(gy + r * g.C, ok, sub, g, gv);
        const float mu = ok ? mean[r] : 0.f, rs = ok ? rstd[r] : 0.f;
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int k = 0; k < kMaxVecPerLane; ++k)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float xh = (v[k][i] - mu) * rs;  // 0 for padded columns / rows (rs = 0 or gm = 0 below)
                const float gw = gv[k][i] * gm[k][i];
                v[k][i] = xh;
                s1 += gw;
                s2 = fmaf(gw, xh, s2);
                dg[k][i] = fmaf(gv[k][i], xh, dg[k][i]);
                db[k][i] += gv[k][i];
            }
        s1 = group_sum<LPR>(s1) * invC;
        s2 = group_sum<LPR>(s2) * invC;
        if (ok) {
#pragma unroll
            for (int k = 0; k < kMaxVecPerLane; ++k)
                if (k < g.nvec) {
                    float o[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) o[i] = rs * (gv[k][i] * gm[k][i] - s1 - v[k][i] * s2);
                    if (residual) {
                        float rv[4];
                        load4<T, VEC>(residual + r * g.C, (k * LPR + sub) * 4, g.C, rv);
#pragma unroll
                        for (int i = 0; i < 4; ++i) o[i] += rv[i];
                    }
                    store4<T, VEC>(dx + r * g.C, (k * LPR + sub) * 4, g.C, o);
                }
        }
    }
    // column sums over the block's rows: lanes with equal `sub` own the same columns
    if (part) {
        float *pg = part + (size_t)blockIdx.x * 2 * g.C, *pb = pg + g.C;
#pragma unroll
        for (int k = 0; k < kMaxVecPerLane; ++k) {
            if (k >= g.nvec) break;
            __syncthreads();
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                s_acc[0][threadIdx.x * 4 + i] = dg[k][i];
                s_acc[1][threadIdx.x * 4 + i] = db[k][i];
            }
            __syncthreads();
            // thread t < LPR*4 sums column (t/4 -> sub, t%4 -> i) over the RPB row slots
            for (int t = threadIdx.x; t < LPR * 4; t += 256) {
                const int sb = t / 4, i = t % 4;
                float a = 0.f, bsum = 0.f;
                for (int rr = 0; rr < RPB; ++rr) {
                    a += s_acc[0][(rr * LPR + sb) * 4 + i];
                    bsum += s_acc[1][(rr * LPR + sb) * 4 + i];
                }
                const int c = (k * LPR + sb) * 4 + i;
                if (c < g.C) { pg[c] = a; pb[c] = bsum; }
            }
        }
    }
}

// dgamma[c] = sum over blocks of part[blk][0][c]; dbeta likewise
__global__ __launch_bounds__(256) void ln_bwd_reduce_kernel(const float *__restrict__ part, const int nblk, const int C,
                                                            float *__restrict__ dgamma, float *__restrict__ dbeta) {
    const int c = blockIdx.x * 4 + (threadIdx.x >> 6);  // one wave per column
    const int lane = threadIdx.x & 63;
    if (c >= C) return;
    float a = 0.f, b = 0.f;
    for (int k = lane; k < nblk; k += 64) {
        a += part[(size_t)k * 2 * C + c];
        b += part[(size_t)k * 2 * C + C + c];
    }
    a = wave_sum(a);
    b = wave_sum(b);
    if (lane == 0) {
        if (dgamma) dgamma[c] = a;
        if (dbeta) dbeta[c] = b;
    }
}

// dgamma / dbeta of MANY LayerNorm backward calls in one launch (blockIdx.y = call): the calls of one backward pass write
// their per-workgroup partials only, the host queues (partials, nblk, C, dgamma, dbeta) and flushes the queue once at the end of
// the pass — one launch instead of 65 (9 us each at <= 1 MB: pure launch floor).  The table travels by value (kernel argument).
constexpr int kMaxReduceItems = 96;
struct ReduceItem {
    const float *part;
    float *dgamma, *dbeta;
    int nblk, C;
};
struct ReduceTable {
    ReduceItem it[kMaxReduceItems];
};

__global__ __launch_bounds__(256) void ln_bwd_reduce_multi_kernel(const ReduceTable t) {
    const ReduceItem r = t.it[blockIdx.y];
    const int lane = threadIdx.x & 63;
    for (int c = blockIdx.x * 4 + (threadIdx.x >> 6); c < r.C; c += gridDim.x * 4) {   // one wave per column
        float a = 0.f, b = 0.f;
        for (int k = lane; k < r.nblk; k += 64) {
            a += r.part[(size_t)k * 2 * r.C + c];
            b += r.part[(size_t)k * 2 * r.C + r.C + c];
        }
        a = wave_sum(a);
        b = wave_sum(b);
        if (lane == 0) {
            if (r.dgamma) r.dgamma[c] = a;
            if (r.dbeta) r.dbeta[c] = b;
        }
    }
}

int lpr_for(int C) {
    int lpr = 1;
    while (lpr < 64 && lpr * 4 * 1 < C && lpr * 4 < C) lpr *= 2;  // smallest power of two with 4*LPR >= C (cap 64)
    return lpr;
}

int grid_for(int rows, int lpr) {
    const long rpb = 256 / lpr;
    long nblk = (rows + rpb - 1) / rpb;
    if (nblk > 2048) nblk = 2048;  // grid-stride above 8 blocks per CU
    return (int)nblk;
}

int check(const void *x, int rows, int C, int dtype, const char *what) {
    VMASR_REQUIRE(x, VMASR_EINVAL, "%s: null tensor", what);
    VMASR_REQUIRE(rows > 0 && C > 0 && C <= 1024, VMASR_EINVAL, "%s: need rows > 0 and 0 < C <= 1024", what);
    VMASR_REQUIRE(dtype == VMASR_F32 || dtype == VMASR_F16 || dtype == VMASR_BF16, VMASR_EINVAL,
                  "%s: dtype must be fp32/fp16/bf16", what);
    return 0;
}

template <typename T, typename TO, bool VEC, int KIND>  // KIND 0 fwd (TO = output type), 1 bwd (TO = grad type)
int launch_lpr(int lpr, dim3 grid, hipStream_t st, double bytes, const void *x, const void *gy, const float *gamma,
               const float *beta, void *y, float *mean, float *rstd, float *part, const LnGeom g, const void *residual) {
#define VMASR_LN_CASE(L)                                                                                              \
    case L:                                                                                                           \
        if (KIND == 0)                                                                                                \
            VMASR_LAUNCH(VMASR_K_LN_FWD, bytes, (ln_fwd_kernel<T, TO, L, VEC>), grid, dim3(256), 0, st, (const T *)x, gamma, \
                         beta, (TO *)y, mean, rstd, g);                                                               \
        else                                                                                                          \
            VMASR_LAUNCH(VMASR_K_LN_BWD, bytes, (ln_bwd_kernel<T, TO, L, VEC>), grid, dim3(256), 0, st, (const T *)x,   \
                         (const TO *)gy, gamma, mean, rstd, (T *)y, part, g, (const T *)residual);                    \
        break;
    switch (lpr) {
        VMASR_LN_CASE(1) VMASR_LN_CASE(2) VMASR_LN_CASE(4) VMASR_LN_CASE(8) VMASR_LN_CASE(16) VMASR_LN_CASE(32)
        VMASR_LN_CASE(64)
        default: set_error("layer_norm: bad lane group"); return VMASR_EINVAL;
    }
#undef VMASR_LN_CASE
    return 0;
}

template <int KIND>
int dispatch(int dtype, int odt, bool vec, int lpr, dim3 grid, hipStream_t st, double bytes, const void *x, const void *gy,
             const float *gamma, const float *beta, void *y, float *mean, float *rstd, float *part, const LnGeom g,
             const void *residual = nullptr) {
#define VMASR_LN_T(TT, TO)                                                                                             \
    (vec ? launch_lpr<TT, TO, true, KIND>(lpr, grid, st, bytes, x, gy, gamma, beta, y, mean, rstd, part, g, residual)  \
         : launch_lpr<TT, TO, false, KIND>(lpr, grid, st, bytes, x, gy, gamma, beta, y, mean, rstd, part, g, residual))
    // (x dtype, y/gy dtype) pairs: equal, 16-bit x with fp32 y (what autocast gives F.layer_norm), and
    // fp32 x with 16-bit y (the fp32 residual stream normalised straight into a GEMM operand)
    if (dtype == VMASR_F32 && odt == VMASR_F32) return VMASR_LN_T(float, float);
    if (dtype == VMASR_BF16 && odt == VMASR_BF16) return VMASR_LN_T(bf16_t, bf16_t);
    if (dtype == VMASR_BF16 && odt == VMASR_F32) return VMASR_LN_T(bf16_t, float);
    if (dtype == VMASR_F32 && odt == VMASR_BF16) return VMASR_LN_T(float, bf16_t);
    if (dtype == VMASR_F16 && odt == VMASR_F16) return VMASR_LN_T(f16_t, f16_t);
    if (dtype == VMASR_F16 && odt == VMASR_F32) return VMASR_LN_T(f16_t, float);
    if (dtype == VMASR_F32 && odt == VMASR_F16) return VMASR_LN_T(float, f16_t);
    set_error("layer_norm: unsupported dtype pair (%d, %d)", dtype, odt);
    return VMASR_EINVAL;
#undef VMASR_LN_T
}

}  // namespace
}  // namespace vmasr

using namespace vmasr;

VMASR_EXPORT int vmasr_layer_norm_fwd(const void *x, const float *gamma, const float *beta, void *y, float *mean,
                                      float *rstd, int32_t rows, int32_t C, float eps, int32_t dtype, int32_t y_dtype,
                                      vmasr_stream_t stream) {
    if (int e = check(x, rows, C, dtype, "layer_norm_fwd")) return e;
    VMASR_REQUIRE(y && mean && rstd, VMASR_EINVAL, "layer_norm_fwd: null output");
    const int lpr = lpr_for(C);
    const LnGeom g{rows, C, (C + 4 * lpr - 1) / (4 * lpr), eps};
    const size_t al = dtype == VMASR_F32 ? 16 : 8;
    const bool vec = C % 4 == 0 && aligned_to(x, al) && aligned_to(y, 16);
    const double es = dtype == VMASR_F32 ? 4 : 2, eo = y_dtype == VMASR_F32 ? 4 : 2;
    const double bytes = rows * (double)C * (es + eo) + 8.0 * rows;
    if (int e = dispatch<0>(dtype, y_dtype, vec, lpr, dim3(grid_for(rows, lpr)), static_cast<hipStream_t>(stream), bytes, x,
                            nullptr, gamma, beta, y, mean, rstd, nullptr, g))
        return e;
    return check_launch("layer_norm_fwd");
}

VMASR_EXPORT size_t vmasr_layer_norm_bwd_workspace(int32_t rows, int32_t C) {
    if (rows <= 0 || C <= 0 || C > 1024) return 0;
    return (size_t)grid_for(rows, lpr_for(C)) * 2 * C * sizeof(float);
}

VMASR_EXPORT int vmasr_layer_norm_bwd(const void *x, const void *gy, const float *gamma, const float *mean,
                                      const float *rstd, void *dx, float *dgamma, float *dbeta, float *ws, int32_t rows,
                                      int32_t C, int32_t dtype, int32_t gy_dtype, vmasr_stream_t stream) {
    return vmasr_layer_norm_bwd_res(x, gy, gamma, mean, rstd, nullptr, dx, dgamma, dbeta, ws, rows, C, dtype, gy_dtype, stream);
}

VMASR_EXPORT int vmasr_layer_norm_bwd_res(const void *x, const void *gy, const float *gamma, const float *mean,
                                          const float *rstd, const void *residual, void *dx, float *dgamma, float *dbeta,
                                          float *ws, int32_t rows, int32_t C, int32_t dtype, int32_t gy_dtype,
                                          vmasr_stream_t stream) {
    if (int e = check(x, rows, C, dtype, "layer_norm_bwd")) return e;
    VMASR_REQUIRE(gy && mean && rstd && dx, VMASR_EINVAL, "layer_norm_bwd: null tensor");
    // ws without dgamma / dbeta: write the per-workgroup partials only (the caller reduces them later, together with other
    // calls': vmasr_layer_norm_bwd_reduce_multi)
    const bool affine = dgamma || dbeta || ws;
    VMASR_REQUIRE(!affine || ws, VMASR_ENOSPACE, "layer_norm_bwd: workspace required for dgamma/dbeta");
    const int lpr = lpr_for(C);
    const LnGeom g{rows, C, (C + 4 * lpr - 1) / (4 * lpr), 0.f};
    const size_t al = dtype == VMASR_F32 ? 16 : 8;
    const bool vec = C % 4 == 0 && aligned_to(x, al) && aligned_to(gy, 16) && aligned_to(dx, al) && (!residual || aligned_to(residual, al));
    const double es = dtype == VMASR_F32 ? 4 : 2, eg = gy_dtype == VMASR_F32 ? 4 : 2;
    const double bytes = rows * (double)C * (2 * es + eg) + 8.0 * rows;
    const int nblk = grid_for(rows, lpr);
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (int e = dispatch<1>(dtype, gy_dtype, vec, lpr, dim3(nblk), st, bytes, x, gy, gamma, nullptr, dx, const_cast<float *>(mean),
                            const_cast<float *>(rstd), affine ? ws : nullptr, g, residual))
        return e;
    if (dgamma || dbeta)
        VMASR_LAUNCH(VMASR_K_LN_BWD_REDUCE, (double)nblk * 2 * C * 4, ln_bwd_reduce_kernel, dim3((C + 3) / 4), dim3(256), 0,
                     st, ws, nblk, C, dgamma, dbeta);
    return check_launch("layer_norm_bwd");
}

VMASR_EXPORT int32_t vmasr_layer_norm_bwd_blocks(int32_t rows, int32_t C) {
    return (rows <= 0 || C <= 0 || C > 1024) ? 0 : grid_for(rows, lpr_for(C));
}

VMASR_EXPORT int vmasr_layer_norm_bwd_reduce_multi(const float *const *parts, float *const *dgammas, float *const *dbetas,
                                                   const int32_t *nblks, const int32_t *Cs, int32_t n, vmasr_stream_t stream) {
    VMASR_REQUIRE(parts && dgammas && dbetas && nblks && Cs && n > 0, VMASR_EINVAL, "layer_norm_bwd_reduce_multi: null argument");
    hipStream_t st = static_cast<hipStream_t>(stream);
    for (int base = 0; base < n; base += kMaxReduceItems) {
        const int m = std::min(kMaxReduceItems, n - base);
        ReduceTable t{};
        int maxC = 0;
        double bytes = 0;
        for (int i = 0; i < m; ++i) {
            VMASR_REQUIRE(parts[base + i] && nblks[base + i] > 0 && Cs[base + i] > 0 && Cs[base + i] <= 1024, VMASR_EINVAL,
                          "layer_norm_bwd_reduce_multi: bad item %d", base + i);
            t.it[i] = ReduceItem{parts[base + i], dgammas[base + i], dbetas[base + i], nblks[base + i], Cs[base + i]};
            maxC = std::max(maxC, Cs[base + i]);
            bytes += (double)nblks[base + i] * 2 * Cs[base + i] * 4;
        }
        VMASR_LAUNCH(VMASR_K_LN_BWD_REDUCE, bytes, ln_bwd_reduce_multi_kernel, dim3(std::min((maxC + 3) / 4, 32), m), dim3(256), 0, st, t);
    }
    return check_launch("layer_norm_bwd_reduce_multi");
}
