// mlp.hip — the Mlp branch of a VSSBlock as ONE kernel on the gfx950 matrix cores:
//
//     y = x + s * fc2(GELU(fc1(LayerNorm(x))))          model/vmamba.py:1832-1837 (VSSBlock._forward, pre-norm),
//                                                         :483-509 (Mlp), timm DropPath (s = per-sample keep mask / keep)
//
// for the residual stream x (rows, d) fp32 or bf16 (under autocast the stream is bf16 behind every PatchMerging /
// skip convolution, fp32 elsewhere) with d in {8, 16, 32, 64, 128} and hidden = 4 d — every Mlp of every shipped
// config except the d_model = 1 block (csrc/linear.hip) and dims-32's deepest stage.  Under bf16 autocast the reference
// runs this as LayerNorm -> cast -> GEMM -> bias -> GELU -> GEMM -> bias -> add: 6-7 launches forward and ~13 backward
// per block over tensors of 0.1-8 MB, i.e. at the launch-latency floor (28 blocks per generator pass).
//
// MI355X-first design (not a GEMM library call): a WAVE owns 32 rows of x and never talks to another wave.  Both
// products run TRANSPOSED on v_mfma_f32_32x32x16_bf16 so that the data row is the accumulator's lane (column) index:
//     H^T (hidden x rows) = W1 (hidden x d) . xn^T (d x rows)         A = W1 rows as stored,   B = the lane's own row of xn
//     Y^T (d x rows)      = W2 (d x hidden) . act^T (hidden x rows)   A = W2 rows as stored,   B = GELU(H^T) straight from
// the accumulator registers — a 32x32 accumulator tile has its column on the lane and its rows in the 16 registers, which is
// exactly the B-operand layout of a product that sums over the tile's ROW index (here: hidden), so the hidden activations
// never leave the register file: no LDS, no barrier, no HBM.  (The k order inside such a step is permuted — register
// 8s + j of lane half h is row 16s + 8(j>>2) + 4h + (j&3) — so W2's fragment is gathered in that order: two 8-byte loads.)
// LayerNorm is two-pass in registers (a row's features sit in lanes l and l + 32), bias + exact-erf GELU on the fp32
// accumulators, residual + DropPath scale in the epilogue.  HBM traffic: read x, write y (8 d bytes per row).
//
// Backward (recompute, nothing saved but x): the same wave re-runs LayerNorm and H^T, forms
//     dact^T = W2^T . (s gy)^T,   gpre^T = dact^T * GELU'(H^T),   dxn^T = W1^T . gpre^T      (again register -> operand)
// and writes dxn (bf16) plus the operands of the two weight-gradient GEMMs — act, gpre, xn, s gy in bf16, `act` and `xn`
// with a trailing ones column so that the bias gradients are a column of the same GEMM (host: vm_asr_amd/mlp.py).
//
// Numerics: bf16 operands, fp32 accumulation — the reference's autocast dtype; fc1's output is NOT rounded to bf16 before
// GELU and fc2's not before the residual add (the reference rounds both), so the result is closer to the fp32 model.
#include "common.h"

namespace vmasr {
namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

struct MlpArgs {
    const void *x;                  // (rows, D) fp32 or bf16 (the residual stream's dtype: TX)
    const float *gamma, *beta;      // (D)
    const bf16_t *w1;               // (4D, D)   fc1.weight
    const float *b1;                // (4D)
    const bf16_t *w2;               // (D, 4D)   fc2.weight
    const float *b2;                // (D)
    const float *scale;             // per-sample residual scale (DropPath) or null
    void *y;                        // (rows, D) TX
    long rows;
    int rows_per_sample;
    float eps;
    // backward only
    const void *gy;                 // (rows, D) TX
    const bf16_t *w1t;              // (D, 4D)   fc1.weight^T
    const bf16_t *w2t;              // (4D, D)   fc2.weight^T
    bf16_t *dxn;                    // (rows, D)
    bf16_t *xn_aug;                 // (rows, D + 8):  xn | 1 0 0 0 0 0 0 0
    bf16_t *gys;                    // (rows, D):      s * gy
    bf16_t *act_aug;                // (rows, 4D + 8): GELU(h) | 1 0 ...
    bf16_t *gpre;                   // (rows, 4D)
    float *mean, *rstd;             // (rows)
    unsigned *det = nullptr;        // deterministic mode (common.h): the hidden-split waves add their partial tiles to LDS in wave order
};

__device__ __forceinline__ f32x16 mfma_bf16(const bf16x8 a, const bf16x8 b, const f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

__device__ __forceinline__ bf16x8 zero8() {
    bf16x8 z;
#pragma unroll
    for (int j = 0; j < 8; ++j) z[j] = (bf16_t)0.f;
    return z;
}

// The lane's row of x in operand order: element j of k-step s is feature 16 s + 8 h + j (0 beyond D / past the end).
template <typename TX>
__device__ __forceinline__ void load4x(const TX *__restrict__ p, float (&v)[4]) {
    if constexpr (sizeof(TX) == 4) {
        const float4 q = *reinterpret_cast<const float4 *>(p);
        v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
    } else {
        const bf16x4 q = *reinterpret_cast<const bf16x4 *>(p);
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = (float)q[j];
    }
}

template <typename TX>
__device__ __forceinline__ void store4x(TX *__restrict__ p, const float v0, const float v1, const float v2, const float v3) {
    if constexpr (sizeof(TX) == 4) {
        *reinterpret_cast<float4 *>(p) = make_float4(v0, v1, v2, v3);
    } else {
        bf16x4 q;
        q[0] = (bf16_t)v0; q[1] = (bf16_t)v1; q[2] = (bf16_t)v2; q[3] = (bf16_t)v3;
        *reinterpret_cast<bf16x4 *>(p) = q;
    }
}

template <int D, int KS, typename TX>
__device__ __forceinline__ void load_row(const TX *__restrict__ p, const bool ok, const int h, float (&v)[KS][8]) {
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        const int f0 = 16 * s + 8 * h;
        if (ok && f0 < D) {
            if constexpr (sizeof(TX) == 4) {
                const float4 q0 = *reinterpret_cast<const float4 *>(p + f0), q1 = *reinterpret_cast<const float4 *>(p + f0 + 4);
                v[s][0] = q0.x; v[s][1] = q0.y; v[s][2] = q0.z; v[s][3] = q0.w;
                v[s][4] = q1.x; v[s][5] = q1.y; v[s][6] = q1.z; v[s][7] = q1.w;
            } else {
                const bf16x8 q = *reinterpret_cast<const bf16x8 *>(p + f0);
#pragma unroll
                for (int j = 0; j < 8; ++j) v[s][j] = (float)q[j];
            }
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) v[s][j] = 0.f;
        }
    }
}

// two-pass LayerNorm statistics of the row shared by lanes l and l ^ 32
template <int D, int KS>
__device__ __forceinline__ void row_stats(const float (&v)[KS][8], const int h, const float eps, float &mean, float &rstd) {
    float s1 = 0.f;
#pragma unroll
    for (int s = 0; s < KS; ++s)
#pragma unroll
        for (int j = 0; j < 8; ++j) s1 += v[s][j];
    s1 += __shfl_xor(s1, 32);
    mean = s1 * (1.f / D);
    float s2 = 0.f;
#pragma unroll
    for (int s = 0; s < KS; ++s)
        if (16 * s + 8 * h < D) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float c = v[s][j] - mean;
                s2 = fmaf(c, c, s2);
            }
        }
    s2 += __shfl_xor(s2, 32);
    rstd = rsqrtf(s2 * (1.f / D) + eps);
}

template <int D, int KS>
__device__ __forceinline__ void normalise(const float (&v)[KS][8], const int h, const float mean, const float rstd,
                                          const float *__restrict__ gamma, const float *__restrict__ beta, bf16x8 (&xn)[KS]) {
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        const int f0 = 16 * s + 8 * h;
        if (f0 < D) {
#pragma unroll
            for (int j = 0; j < 8; ++j) xn[s][j] = (bf16_t)fmaf((v[s][j] - mean) * rstd, gamma[f0 + j], beta[f0 + j]);
        } else {
            xn[s] = zero8();
        }
    }
}

// H^T tile t (32 hidden units x 32 rows) = W1[32 t .. 32 t + 31, :] . xn^T
template <int D, int KS>
__device__ __forceinline__ f32x16 hidden_tile(const bf16_t *__restrict__ w1, const int t, const int r, const int h,
                                              const bf16x8 (&xn)[KS]) {
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    const bf16_t *wr = w1 + (size_t)(32 * t + r) * D;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        const int f0 = 16 * s + 8 * h;
        const bf16x8 a = f0 < D ? *reinterpret_cast<const bf16x8 *>(wr + f0) : zero8();
        acc = mfma_bf16(a, xn[s], acc);
    }
    return acc;
}

// fragment of a row-major (n_rows, ld) bf16 matrix W for the PERMUTED k order of an accumulator tile used as operand:
// element j <-> column c0 + 8 (j >> 2) + (j & 3), c0 = 32 t + 16 s2 + 4 h  (two 8-byte loads); zero row when !valid
__device__ __forceinline__ bf16x8 perm_frag(const bf16_t *__restrict__ row_ptr, const int c0, const bool valid) {
    bf16x8 w = zero8();
    if (valid) {
        const bf16x4 lo = *reinterpret_cast<const bf16x4 *>(row_ptr + c0), hi = *reinterpret_cast<const bf16x4 *>(row_ptr + c0 + 8);
#pragma unroll
        for (int j = 0; j < 4; ++j) { w[j] = lo[j]; w[4 + j] = hi[j]; }
    }
    return w;
}

constexpr float kRsqrt2 = 0.70710678118654752f, kInvSqrt2Pi = 0.3989422804014327f;

// Hidden-split workgroups (HS > 1: the deep stages have 1 024 .. 4 096 rows, i.e. 32 .. 128 row tiles — one wave per tile
// would leave the chip empty and walk 8 .. 16 hidden tiles serially): the HS waves of a workgroup share ONE row tile and
// each takes every HS-th hidden tile; their partial output tiles (feature x row, fp32) are summed in LDS with ds_add_f32
// and the epilogue is spread over the waves by 4-feature groups.   s_out[feature * 32 + row]
template <int OT>
__device__ __forceinline__ void lds_accumulate(float *s_out, const f32x16 (&acc)[OT], const int r, const int h) {
#pragma unroll
    for (int u = 0; u < OT; ++u)
#pragma unroll
        for (int i = 0; i < 16; ++i) atomicAdd(s_out + (32 * u + (i & 3) + 8 * (i >> 2) + 4 * h) * 32 + r, acc[u][i]);
}

template <int D, typename TX, int HS>
__global__ __launch_bounds__(HS > 4 ? 64 * HS : 256) void mlp_fwd_kernel(const MlpArgs a) {
    constexpr int KS = (D + 15) / 16, HD = 4 * D, HT = HD / 32, OT = (D + 31) / 32;
    constexpr int TPW = HS > 1 ? 1 : 4;                          // row tiles per workgroup
    __shared__ float s_out[HS > 1 ? OT * 32 * 32 : 1];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int split = HS > 1 ? wave : 0;
    for (long tile = (long)blockIdx.x * TPW + (HS > 1 ? 0 : wave); tile * 32 < a.rows; tile += (long)gridDim.x * TPW) {
        const long row = tile * 32 + r;
        const bool ok = row < a.rows;
        const TX *xr = static_cast<const TX *>(a.x) + row * D;
        float xv[KS][8];
        load_row<D, KS, TX>(xr, ok, h, xv);
        float mean, rstd;
        row_stats<D, KS>(xv, h, a.eps, mean, rstd);
        bf16x8 xn[KS];
        normalise<D, KS>(xv, h, mean, rstd, a.gamma, a.beta, xn);

        f32x16 out[OT];
#pragma unroll
        for (int u = 0; u < OT; ++u)
#pragma unroll
            for (int i = 0; i < 16; ++i) out[u][i] = 0.f;
        if constexpr (HS > 1) {
            for (int i = threadIdx.x; i < OT * 32 * 32; i += 64 * HS) s_out[i] = 0.f;
            __syncthreads();
        }
#pragma unroll 1
        for (int t = split; t < HT; t += HS) {
            const f32x16 hacc = hidden_tile<D, KS>(a.w1, t, r, h, xn);
            bf16x8 act[2];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 bq = *reinterpret_cast<const float4 *>(a.b1 + 32 * t + 8 * g + 4 * h);
                const float bb[4] = {bq.x, bq.y, bq.z, bq.w};
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const float v = hacc[4 * g + c] + bb[c];
                    act[g >> 1][4 * (g & 1) + c] = (bf16_t)(0.5f * v * (1.f + erff(v * kRsqrt2)));
                }
            }
#pragma unroll
            for (int u = 0; u < OT; ++u) {
                const int o = 32 * u + r;
                const bf16_t *wr = a.w2 + (size_t)o * HD;
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2)
                    out[u] = mfma_bf16(perm_frag(wr, 32 * t + 16 * s2 + 4 * h, o < D), act[s2], out[u]);
            }
        }
        const float sc = a.scale ? a.scale[ok ? row / a.rows_per_sample : 0] : 1.f;
        if constexpr (HS > 1) {
            VMASR_DET_WAVE_ORDER(a.det, HS, lds_accumulate<OT>(s_out, out, r, h));
            __syncthreads();
        }
#pragma unroll
        for (int u = 0; u < OT; ++u)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                if (HS > 1 && (4 * u + g) % HS != split) continue;            // the 4-feature groups go round the waves
                const int o0 = 32 * u + 8 * g + 4 * h;
                if (ok && o0 < D) {
                    float xq[4], acc[4];
                    load4x<TX>(xr + o0, xq);
#pragma unroll
                    for (int c = 0; c < 4; ++c) acc[c] = HS > 1 ? s_out[(o0 + c) * 32 + r] : out[u][4 * g + c];
                    const float4 bq = *reinterpret_cast<const float4 *>(a.b2 + o0);
                    store4x<TX>(static_cast<TX *>(a.y) + row * D + o0, fmaf(sc, acc[0] + bq.x, xq[0]), fmaf(sc, acc[1] + bq.y, xq[1]),
                                fmaf(sc, acc[2] + bq.z, xq[2]), fmaf(sc, acc[3] + bq.w, xq[3]));
                }
            }
        if constexpr (HS > 1) __syncthreads();                               // s_out is reused by the next row tile
    }
}

__device__ __forceinline__ void store_bf16x4(bf16_t *p, const float v0, const float v1, const float v2, const float v3) {
    bf16x4 q;
    q[0] = (bf16_t)v0; q[1] = (bf16_t)v1; q[2] = (bf16_t)v2; q[3] = (bf16_t)v3;
    *reinterpret_cast<bf16x4 *>(p) = q;
}

template <int D, typename TX, int HS>
__global__ __launch_bounds__(HS > 4 ? 64 * HS : 256) void mlp_bwd_kernel(const MlpArgs a) {
    constexpr int KS = (D + 15) / 16, HD = 4 * D, HT = HD / 32, OT = (D + 31) / 32;
    constexpr int TPW = HS > 1 ? 1 : 4;
    __shared__ float s_out[HS > 1 ? OT * 32 * 32 : 1];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int split = HS > 1 ? wave : 0;
    for (long tile = (long)blockIdx.x * TPW + (HS > 1 ? 0 : wave); tile * 32 < a.rows; tile += (long)gridDim.x * TPW) {
        const long row = tile * 32 + r;
        const bool ok = row < a.rows;
        float xv[KS][8];
        load_row<D, KS, TX>(static_cast<const TX *>(a.x) + row * D, ok, h, xv);
        float mean, rstd;
        row_stats<D, KS>(xv, h, a.eps, mean, rstd);
        bf16x8 xn[KS];
        normalise<D, KS>(xv, h, mean, rstd, a.gamma, a.beta, xn);
        load_row<D, KS, TX>(static_cast<const TX *>(a.gy) + row * D, ok, h, xv);      // xv <- gy
        const float sc = a.scale ? a.scale[ok ? row / a.rows_per_sample : 0] : 1.f;
        bf16x8 gyf[KS];
#pragma unroll
        for (int s = 0; s < KS; ++s) {
#pragma unroll
            for (int j = 0; j < 8; ++j) gyf[s][j] = (bf16_t)(sc * xv[s][j]);
            const int f0 = 16 * s + 8 * h;
            if (ok && f0 < D && split == 0) {
                *reinterpret_cast<bf16x8 *>(a.xn_aug + row * (D + 8) + f0) = xn[s];
                *reinterpret_cast<bf16x8 *>(a.gys + row * D + f0) = gyf[s];
            }
        }
        if (ok && h == 0 && split == 0) {
            bf16x8 one = zero8();
            one[0] = (bf16_t)1.f;
            *reinterpret_cast<bf16x8 *>(a.xn_aug + row * (D + 8) + D) = one;
            *reinterpret_cast<bf16x8 *>(a.act_aug + row * (HD + 8) + HD) = one;
            a.mean[row] = mean;
            a.rstd[row] = rstd;
        }

        f32x16 dx[OT];
#pragma unroll
        for (int u = 0; u < OT; ++u)
#pragma unroll
            for (int i = 0; i < 16; ++i) dx[u][i] = 0.f;
        if constexpr (HS > 1) {
            for (int i = threadIdx.x; i < OT * 32 * 32; i += 64 * HS) s_out[i] = 0.f;
            __syncthreads();
        }
#pragma unroll 1
        for (int t = split; t < HT; t += HS) {
            const f32x16 hacc = hidden_tile<D, KS>(a.w1, t, r, h, xn);
            const f32x16 dacc = hidden_tile<D, KS>(a.w2t, t, r, h, gyf);   // dact^T = W2^T[32 t .., :] . (s gy)^T
            bf16x8 gp[2];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int hid0 = 32 * t + 8 * g + 4 * h;
                const float4 bq = *reinterpret_cast<const float4 *>(a.b1 + hid0);
                const float bb[4] = {bq.x, bq.y, bq.z, bq.w};
                float av[4], gv[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const float v = hacc[4 * g + c] + bb[c];
                    const float cdf = 0.5f * (1.f + erff(v * kRsqrt2));
                    const float pdf = kInvSqrt2Pi * __expf(-0.5f * v * v);
                    av[c] = v * cdf;
                    gv[c] = dacc[4 * g + c] * fmaf(v, pdf, cdf);
                    gp[g >> 1][4 * (g & 1) + c] = (bf16_t)gv[c];
                }
                if (ok) {
                    store_bf16x4(a.act_aug + row * (HD + 8) + hid0, av[0], av[1], av[2], av[3]);
                    store_bf16x4(a.gpre + row * HD + hid0, gv[0], gv[1], gv[2], gv[3]);
                }
            }
#pragma unroll
            for (int u = 0; u < OT; ++u) {
                const int f = 32 * u + r;
                const bf16_t *wr = a.w1t + (size_t)f * HD;
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2)
                    dx[u] = mfma_bf16(perm_frag(wr, 32 * t + 16 * s2 + 4 * h, f < D), gp[s2], dx[u]);
            }
        }
        if constexpr (HS > 1) {
            VMASR_DET_WAVE_ORDER(a.det, HS, lds_accumulate<OT>(s_out, dx, r, h));
            __syncthreads();
        }
#pragma unroll
        for (int u = 0; u < OT; ++u)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                if (HS > 1 && (4 * u + g) % HS != split) continue;
                const int f0 = 32 * u + 8 * g + 4 * h;
                if (ok && f0 < D) {
                    float acc[4];
#pragma unroll
                    for (int c = 0; c < 4; ++c) acc[c] = HS > 1 ? s_out[(f0 + c) * 32 + r] : dx[u][4 * g + c];
                    store_bf16x4(a.dxn + row * D + f0, acc[0], acc[1], acc[2], acc[3]);
                }
            }
        if constexpr (HS > 1) __syncthreads();
    }
}

// ---- the input side of SS2D.forwardv2 as one kernel (model/vmamba.py:1826-1827 pre-norm, :1535-1542) ------------------------
//     xz = in_proj(LayerNorm(x));  x', z = xz.chunk(2);  xT = x'.permute(0,3,1,2) (channel-first, for the depthwise conv);
//     sz = SiLU(z)                                                       (= LayerNorm + Linear + vmasr_ss2d_pre_fwd)
// Same wave-owns-32-rows / transposed-product scheme as the Mlp kernel: H^T = W_in (4d x d) . xn^T puts the position on the
// lane and the output channel on the accumulator register, which is exactly how both outputs want to be written — the
// channel-first x' as 64-byte runs along the positions (one 2-byte store per register), SiLU(z) as 8-byte runs along the
// channels of the lane's own row.  `norm` = 0: the block has no LayerNorm (output layers 0 and 2 use nn.Identity).
// Backward (recompute): gpre^T = [dxT | dsz * SiLU'(z)] tile by tile, dxn^T = W_in^T . gpre^T from the registers; writes dxn,
// xn and gpre (bf16) for LayerNorm's backward and the weight-gradient GEMM dW = gpre^T . xn.
struct InProjArgs {
    const void *x;                 // (rows, D) TX
    const float *gamma, *beta;     // (D) or unused when !norm
    const bf16_t *w;               // (4D, D)  in_proj.weight
    const bf16_t *wt;              // (D, 4D)  its transpose (backward)
    bf16_t *xT;                    // (B, 2D, L) channel-first x half          (backward: dxT, read)
    bf16_t *sz;                    // (rows, 2D) SiLU(z)                        (backward: dsz, read)
    bf16_t *dxn, *xn, *gpre;       // backward outputs: (rows, D), (rows, D), (rows, 4D)
    float *mean, *rstd;            // (rows) backward outputs (LayerNorm statistics)
    long rows;
    int L, norm;
    float eps;
};

template <int D, typename TX, bool BWD>
__global__ __launch_bounds__(256) void inproj_kernel(const InProjArgs a) {
    constexpr int KS = (D + 15) / 16, HD = 4 * D, HT = HD / 32, OT = (D + 31) / 32, DI = 2 * D;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 31, h = lane >> 5;
    for (long tile = (long)blockIdx.x * 4 + wave; tile * 32 < a.rows; tile += (long)gridDim.x * 4) {
        const long row = tile * 32 + r;
        const bool ok = row < a.rows;
        const long b = row / a.L, l = row % a.L;
        float xv[KS][8];
        load_row<D, KS, TX>(static_cast<const TX *>(a.x) + row * D, ok, h, xv);
        float mean = 0.f, rstd = 1.f;
        bf16x8 xn[KS];
        if (a.norm) {
            row_stats<D, KS>(xv, h, a.eps, mean, rstd);
            normalise<D, KS>(xv, h, mean, rstd, a.gamma, a.beta, xn);
        } else {
#pragma unroll
            for (int s = 0; s < KS; ++s)
#pragma unroll
                for (int j = 0; j < 8; ++j) xn[s][j] = (bf16_t)xv[s][j];
        }
        f32x16 dx[OT];
        if constexpr (BWD) {
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                const int f0 = 16 * s + 8 * h;
                if (ok && f0 < D) *reinterpret_cast<bf16x8 *>(a.xn + row * D + f0) = xn[s];
            }
            if (ok && h == 0 && a.norm) { a.mean[row] = mean; a.rstd[row] = rstd; }
#pragma unroll
            for (int u = 0; u < OT; ++u)
#pragma unroll
                for (int i = 0; i < 16; ++i) dx[u][i] = 0.f;
        }
#pragma unroll 1
        for (int t = 0; t < HT; ++t) {
            // x-half tiles need no product in the backward (their gradient arrives as dxT); D = 8 has both halves in tile 0
            const bool need_h = !BWD || 32 * t + 31 >= DI;
            f32x16 hacc;
            if (need_h) hacc = hidden_tile<D, KS>(a.w, t, r, h, xn);
            bf16x8 gp[2];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int hid0 = 32 * t + 8 * g + 4 * h;           // 4 consecutive channels, all in one half (DI is a multiple of 16)
                if (hid0 < DI) {                                    // x half: channel-first
                    bf16_t *col = a.xT + ((size_t)b * DI + hid0) * a.L + l;
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        if constexpr (BWD) gp[g >> 1][4 * (g & 1) + c] = ok ? col[(size_t)c * a.L] : (bf16_t)0.f;
                        else if (ok) col[(size_t)c * a.L] = (bf16_t)hacc[4 * g + c];
                    }
                } else {                                            // z half: channel-last, SiLU
                    bf16_t *zp = a.sz + row * DI + (hid0 - DI);
                    float v[4];
#pragma unroll
                    for (int c = 0; c < 4; ++c) v[c] = hacc[4 * g + c];
                    if constexpr (BWD) {
                        bf16x4 gq;
#pragma unroll
                        for (int c = 0; c < 4; ++c) gq[c] = (bf16_t)0.f;
                        if (ok) gq = *reinterpret_cast<const bf16x4 *>(zp);
#pragma unroll
                        for (int c = 0; c < 4; ++c) {
                            const float sg = 1.f / (1.f + __expf(-v[c]));
                            gp[g >> 1][4 * (g & 1) + c] = (bf16_t)((float)gq[c] * sg * fmaf(v[c], 1.f - sg, 1.f));
                        }
                    } else if (ok) {
                        float o[4];
#pragma unroll
                        for (int c = 0; c < 4; ++c) o[c] = v[c] / (1.f + __expf(-v[c]));
                        store_bf16x4(zp, o[0], o[1], o[2], o[3]);
                    }
                }
                if constexpr (BWD) {
                    if (ok) {
                        bf16x4 q;
#pragma unroll
                        for (int c = 0; c < 4; ++c) q[c] = gp[g >> 1][4 * (g & 1) + c];
                        *reinterpret_cast<bf16x4 *>(a.gpre + row * HD + hid0) = q;
                    }
                }
            }
            if constexpr (BWD) {
#pragma unroll
                for (int u = 0; u < OT; ++u) {
                    const int f = 32 * u + r;
                    const bf16_t *wr = a.wt + (size_t)f * HD;
#pragma unroll
                    for (int s2 = 0; s2 < 2; ++s2)
                        dx[u] = mfma_bf16(perm_frag(wr, 32 * t + 16 * s2 + 4 * h, f < D), gp[s2], dx[u]);
                }
            }
        }
        if constexpr (BWD) {
#pragma unroll
            for (int u = 0; u < OT; ++u)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int f0 = 32 * u + 8 * g + 4 * h;
                    if (ok && f0 < D)
                        store_bf16x4(a.dxn + row * D + f0, dx[u][4 * g], dx[u][4 * g + 1], dx[u][4 * g + 2], dx[u][4 * g + 3]);
                }
        }
    }
}

int grid_for(long rows, int tiles_per_wg) {
    const long tiles = ((rows + 31) / 32 + tiles_per_wg - 1) / tiles_per_wg;
    return (int)(tiles < 2048 ? (tiles < 1 ? 1 : tiles) : 2048);
}

// d = 128 (1 024 rows at batch 4) is built but not offered: with 16 hidden tiles per row tile and 32 row tiles the kernel is a
// chain of dependent weight-fragment loads (62 us forward against 30 us for the hipBLASLt path, tools/bench_mlp.py); it needs
// the weights staged through LDS for the workgroup — until then the module path runs there.
bool supported_d(int d) { return d == 8 || d == 16 || d == 32 || d == 64; }

template <bool BWD, typename TX>
int launch(const MlpArgs &a, int d, hipStream_t st) {
    const double sx = sizeof(TX);
    // bytes: forward x in + y out; backward x, gy in + dxn, xn, gys (2 B) + act, gpre (2 B x 4 d) out
    const double bytes = (double)a.rows * d * (BWD ? 2 * sx + 2.0 * 3 + 2.0 * 8 : 2 * sx);
    // waves per row tile: 1 while the rows alone fill the chip (d <= 32: >= 16 384 rows at batch 4), hidden-split below that
#define VMASR_MLP_CASE(DD, HS)                                                                                          \
    case DD: {                                                                                                          \
        const dim3 grid(grid_for(a.rows, HS > 1 ? 1 : 4)), block(HS > 4 ? 64 * HS : 256);                              \
        if (BWD) VMASR_LAUNCH(VMASR_K_MLP_BWD, bytes, (mlp_bwd_kernel<DD, TX, HS>), grid, block, 0, st, a);            \
        else VMASR_LAUNCH(VMASR_K_MLP_FWD, bytes, (mlp_fwd_kernel<DD, TX, HS>), grid, block, 0, st, a);                \
    } break;
    switch (d) {
        VMASR_MLP_CASE(8, 1) VMASR_MLP_CASE(16, 1) VMASR_MLP_CASE(32, 1) VMASR_MLP_CASE(64, 4) VMASR_MLP_CASE(128, 8)
        default: set_error("mlp: unsupported width %d", d); return VMASR_EINVAL;
    }
#undef VMASR_MLP_CASE
    return check_launch(BWD ? "mlp_bwd" : "mlp_fwd");
}

}  // namespace
}  // namespace vmasr

using namespace vmasr;

VMASR_EXPORT int vmasr_mlp_supported(int32_t d, int32_t hidden) { return supported_d(d) && hidden == 4 * d; }

VMASR_EXPORT int vmasr_mlp_fwd(const void *x, const float *gamma, const float *beta, float eps, const void *w1, const float *b1,
                               const void *w2, const float *b2, const float *scale, int32_t rows_per_sample, void *y,
                               int64_t rows, int32_t d, int32_t x_dtype, vmasr_stream_t stream) {
    VMASR_REQUIRE(x_dtype == VMASR_F32 || x_dtype == VMASR_BF16, VMASR_EINVAL, "mlp_fwd: x must be fp32 or bf16");
    VMASR_REQUIRE(x && gamma && beta && w1 && b1 && w2 && b2 && y, VMASR_EINVAL, "mlp_fwd: null tensor");
    VMASR_REQUIRE(supported_d(d) && rows > 0, VMASR_EINVAL, "mlp_fwd: need d in {8,16,32,64} (got %d) and rows > 0", d);
    VMASR_REQUIRE(!scale || rows_per_sample > 0, VMASR_EINVAL, "mlp_fwd: rows_per_sample must be positive with a scale vector");
    VMASR_REQUIRE(aligned_to(x, 16) && aligned_to(y, 16) && aligned_to(w1, 16) && aligned_to(w2, 16) && aligned_to(b1, 16) &&
                      aligned_to(b2, 16), VMASR_EINVAL, "mlp_fwd: tensors must be 16-byte aligned");
    MlpArgs a{};
    a.det = det_ticket(VMASR_K_MLP_FWD);
    a.x = x; a.gamma = gamma; a.beta = beta; a.eps = eps;
    a.w1 = static_cast<const bf16_t *>(w1); a.b1 = b1; a.w2 = static_cast<const bf16_t *>(w2); a.b2 = b2;
    a.scale = scale; a.rows_per_sample = rows_per_sample > 0 ? rows_per_sample : 1; a.y = y; a.rows = rows;
    return x_dtype == VMASR_F32 ? launch<false, float>(a, d, static_cast<hipStream_t>(stream))
                                : launch<false, bf16_t>(a, d, static_cast<hipStream_t>(stream));
}

VMASR_EXPORT int vmasr_mlp_bwd(const void *x, const void *gy, const float *gamma, const float *beta, float eps, const void *w1,
                               const void *w1t, const float *b1, const void *w2t, const float *scale, int32_t rows_per_sample,
                               void *dxn, void *xn_aug, void *gys, void *act_aug, void *gpre, float *mean, float *rstd,
                               int64_t rows, int32_t d, int32_t x_dtype, vmasr_stream_t stream) {
    VMASR_REQUIRE(x_dtype == VMASR_F32 || x_dtype == VMASR_BF16, VMASR_EINVAL, "mlp_bwd: x / gy must be fp32 or bf16");
    VMASR_REQUIRE(x && gy && gamma && beta && w1 && w1t && b1 && w2t && dxn && xn_aug && gys && act_aug && gpre && mean && rstd,
                  VMASR_EINVAL, "mlp_bwd: null tensor");
    VMASR_REQUIRE(supported_d(d) && rows > 0, VMASR_EINVAL, "mlp_bwd: need d in {8,16,32,64} (got %d) and rows > 0", d);
    VMASR_REQUIRE(!scale || rows_per_sample > 0, VMASR_EINVAL, "mlp_bwd: rows_per_sample must be positive with a scale vector");
    VMASR_REQUIRE(aligned_to(x, 16) && aligned_to(gy, 16) && aligned_to(w1, 16) && aligned_to(w1t, 16) && aligned_to(w2t, 16) &&
                      aligned_to(b1, 16) && aligned_to(dxn, 16) && aligned_to(xn_aug, 16) && aligned_to(gys, 16) &&
                      aligned_to(act_aug, 16) && aligned_to(gpre, 16), VMASR_EINVAL, "mlp_bwd: tensors must be 16-byte aligned");
    MlpArgs a{};
    a.det = det_ticket(VMASR_K_MLP_BWD);
    a.x = x; a.gy = gy; a.gamma = gamma; a.beta = beta; a.eps = eps;
    a.w1 = static_cast<const bf16_t *>(w1); a.w1t = static_cast<const bf16_t *>(w1t); a.b1 = b1;
    a.w2t = static_cast<const bf16_t *>(w2t);
    a.scale = scale; a.rows_per_sample = rows_per_sample > 0 ? rows_per_sample : 1; a.rows = rows;
    a.dxn = static_cast<bf16_t *>(dxn); a.xn_aug = static_cast<bf16_t *>(xn_aug); a.gys = static_cast<bf16_t *>(gys);
    a.act_aug = static_cast<bf16_t *>(act_aug); a.gpre = static_cast<bf16_t *>(gpre); a.mean = mean; a.rstd = rstd;
    return x_dtype == VMASR_F32 ? launch<true, float>(a, d, static_cast<hipStream_t>(stream))
                                : launch<true, bf16_t>(a, d, static_cast<hipStream_t>(stream));
}

template <bool BWD, typename TX>
static int launch_inproj(const InProjArgs &a, int d, hipStream_t st) {
    const dim3 grid(grid_for(a.rows, 4)), block(256);
    const double sx = sizeof(TX);
    const double bytes = (double)a.rows * d * (BWD ? sx + 2.0 * 2 + 2.0 * 4 * 2 : sx + 2.0 * 4);
#define VMASR_INP_CASE(DD)                                                                                          \
    case DD:                                                                                                        \
        VMASR_LAUNCH(BWD ? VMASR_K_INPROJ_BWD : VMASR_K_INPROJ_FWD, bytes, (inproj_kernel<DD, TX, BWD>), grid, block, 0, st, a); \
        break;
    switch (d) {
        VMASR_INP_CASE(8) VMASR_INP_CASE(16) VMASR_INP_CASE(32) VMASR_INP_CASE(64)
        default: set_error("inproj: unsupported width %d", d); return VMASR_EINVAL;
    }
#undef VMASR_INP_CASE
    return check_launch(BWD ? "inproj_bwd" : "inproj_fwd");
}

VMASR_EXPORT int vmasr_inproj_supported(int32_t d, int32_t d_proj, int64_t L) {
    return (d == 8 || d == 16 || d == 32 || d == 64) && d_proj == 4 * d && L > 0 && L % 32 == 0;
}

VMASR_EXPORT int vmasr_inproj_fwd(const void *x, const float *gamma, const float *beta, float eps, const void *w, void *xT, void *sz,
                                  int64_t rows, int32_t L, int32_t d, int32_t x_dtype, vmasr_stream_t stream) {
    VMASR_REQUIRE(x && w && xT && sz, VMASR_EINVAL, "inproj_fwd: null tensor");
    VMASR_REQUIRE(vmasr_inproj_supported(d, 4 * d, L) && rows > 0 && rows % L == 0, VMASR_EINVAL,
                  "inproj_fwd: need d in {8,16,32,64}, L a multiple of 32 and rows a multiple of L");
    VMASR_REQUIRE(x_dtype == VMASR_F32 || x_dtype == VMASR_BF16, VMASR_EINVAL, "inproj_fwd: x must be fp32 or bf16");
    VMASR_REQUIRE((!gamma) == (!beta), VMASR_EINVAL, "inproj_fwd: gamma and beta go together");
    VMASR_REQUIRE(aligned_to(x, 16) && aligned_to(w, 16) && aligned_to(sz, 8) && aligned_to(xT, 2), VMASR_EINVAL, "inproj_fwd: unaligned");
    InProjArgs a{};
    a.x = x; a.gamma = gamma; a.beta = beta; a.norm = gamma ? 1 : 0; a.eps = eps; a.w = static_cast<const bf16_t *>(w);
    a.xT = static_cast<bf16_t *>(xT); a.sz = static_cast<bf16_t *>(sz); a.rows = rows; a.L = L;
    return x_dtype == VMASR_F32 ? launch_inproj<false, float>(a, d, static_cast<hipStream_t>(stream))
                                : launch_inproj<false, bf16_t>(a, d, static_cast<hipStream_t>(stream));
}

VMASR_EXPORT int vmasr_inproj_bwd(const void *x, const float *gamma, const float *beta, float eps, const void *w, const void *wt,
                                  const void *dxT, const void *dsz, void *dxn, void *xn, void *gpre, float *mean, float *rstd,
                                  int64_t rows, int32_t L, int32_t d, int32_t x_dtype, vmasr_stream_t stream) {
    VMASR_REQUIRE(x && w && wt && dxT && dsz && dxn && xn && gpre, VMASR_EINVAL, "inproj_bwd: null tensor");
    VMASR_REQUIRE(vmasr_inproj_supported(d, 4 * d, L) && rows > 0 && rows % L == 0, VMASR_EINVAL,
                  "inproj_bwd: need d in {8,16,32,64}, L a multiple of 32 and rows a multiple of L");
    VMASR_REQUIRE(x_dtype == VMASR_F32 || x_dtype == VMASR_BF16, VMASR_EINVAL, "inproj_bwd: x must be fp32 or bf16");
    VMASR_REQUIRE((!gamma) == (!beta) && (!gamma || (mean && rstd)), VMASR_EINVAL, "inproj_bwd: gamma, beta, mean, rstd go together");
    VMASR_REQUIRE(aligned_to(x, 16) && aligned_to(w, 16) && aligned_to(wt, 16) && aligned_to(dsz, 8) && aligned_to(dxn, 8) &&
                      aligned_to(xn, 16) && aligned_to(gpre, 8), VMASR_EINVAL, "inproj_bwd: unaligned");
    InProjArgs a{};
    a.x = x; a.gamma = gamma; a.beta = beta; a.norm = gamma ? 1 : 0; a.eps = eps;
    a.w = static_cast<const bf16_t *>(w); a.wt = static_cast<const bf16_t *>(wt);
    a.xT = static_cast<bf16_t *>(const_cast<void *>(dxT)); a.sz = static_cast<bf16_t *>(const_cast<void *>(dsz));
    a.dxn = static_cast<bf16_t *>(dxn); a.xn = static_cast<bf16_t *>(xn); a.gpre = static_cast<bf16_t *>(gpre);
    a.mean = mean; a.rstd = rstd; a.rows = rows; a.L = L;
    return x_dtype == VMASR_F32 ? launch_inproj<true, float>(a, d, static_cast<hipStream_t>(stream))
                                : launch_inproj<true, bf16_t>(a, d, static_cast<hipStream_t>(stream));
}

// ---- the output side of SS2D.forwardv2: out_proj + DropPath + residual on the same scheme ----------------------------------------
//     y = x + s * (g . W_out^T)        model/vmamba.py:1551 (out_proj, no bias; dropout p = 0) + :1826-1827 (VSSBlock residual)
// g (rows, 2D) bf16 = the gated LayerNorm output of ln_gate, W_out (D, 2D) bf16, x / y (rows, D) the residual stream (TX).
// Y^T (D x rows) = W_out . g^T: A = rows of W_out as stored, B = the lane's own row of g (contiguous 16-byte fragments); the
// accumulator has the data row on the lane, so the residual add and the store are the Mlp kernel's epilogue.  Replaces GEMM +
// add (+ DropPath multiply).  Backward: dg^T (2D x rows) = W_out^T . (s gy)^T -> dg (rows, 2D) bf16 (ln_gate's incoming
// gradient) and gys = s gy in bf16 (operand of dW = gys^T . g, one split-K GEMM on the host side); the stream's gradient is gy.
namespace vmasr {
namespace {

struct OutProjArgs {
    const bf16_t *g;      // (rows, 2D)
    const bf16_t *w;      // (D, 2D)   out_proj.weight           (backward: (2D, D) its transpose)
    const void *x;        // (rows, D) TX residual stream        (backward: gy)
    const float *scale;   // per-sample residual scale or null
    void *y;              // (rows, D) TX                        (backward: unused)
    bf16_t *dg, *gys;     // backward outputs (rows, 2D), (rows, D)
    long rows;
    int rows_per_sample;
};

template <int D, typename TX, bool BWD>
__global__ __launch_bounds__(256) void outproj_kernel(const OutProjArgs a) {
    constexpr int DI = 2 * D;
    constexpr int KIN = BWD ? D : DI, NOUT = BWD ? DI : D;          // contraction length, output features
    constexpr int KS = (KIN + 15) / 16, OT = (NOUT + 31) / 32;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 31, h = lane >> 5;
    for (long tile = (long)blockIdx.x * 4 + wave; tile * 32 < a.rows; tile += (long)gridDim.x * 4) {
        const long row = tile * 32 + r;
        const bool ok = row < a.rows;
        const float sc = a.scale ? a.scale[ok ? row / a.rows_per_sample : 0] : 1.f;
        bf16x8 b[KS];
        if constexpr (BWD) {
            float gv[KS][8];
            load_row<D, KS, TX>(static_cast<const TX *>(a.x) + row * D, ok, h, gv);
#pragma unroll
            for (int s = 0; s < KS; ++s) {
#pragma unroll
                for (int j = 0; j < 8; ++j) b[s][j] = (bf16_t)(sc * gv[s][j]);
                const int f0 = 16 * s + 8 * h;
                if (ok && f0 < D) *reinterpret_cast<bf16x8 *>(a.gys + row * D + f0) = b[s];
            }
        } else {
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                const int f0 = 16 * s + 8 * h;
                b[s] = (ok && f0 < DI) ? *reinterpret_cast<const bf16x8 *>(a.g + row * DI + f0) : zero8();
            }
        }
#pragma unroll
        for (int u = 0; u < OT; ++u) {
            f32x16 acc;
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] = 0.f;
            const int o = 32 * u + r;                                 // the A operand's row = output feature
            const bf16_t *wr = a.w + (size_t)o * KIN;
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                const int f0 = 16 * s + 8 * h;
                const bf16x8 av = (o < NOUT && f0 < KIN) ? *reinterpret_cast<const bf16x8 *>(wr + f0) : zero8();
                acc = mfma_bf16(av, b[s], acc);
            }
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const int o0 = 32 * u + 8 * g4 + 4 * h;
                if (!ok || o0 >= NOUT) continue;
                if constexpr (BWD) {
                    store_bf16x4(a.dg + row * DI + o0, acc[4 * g4], acc[4 * g4 + 1], acc[4 * g4 + 2], acc[4 * g4 + 3]);
                } else {
                    float xq[4];
                    load4x<TX>(static_cast<const TX *>(a.x) + row * D + o0, xq);
                    store4x<TX>(static_cast<TX *>(a.y) + row * D + o0, fmaf(sc, acc[4 * g4], xq[0]), fmaf(sc, acc[4 * g4 + 1], xq[1]),
                                fmaf(sc, acc[4 * g4 + 2], xq[2]), fmaf(sc, acc[4 * g4 + 3], xq[3]));
                }
            }
        }
    }
}

template <bool BWD, typename TX>
int launch_outproj(const OutProjArgs &a, int d, hipStream_t st) {
    const dim3 grid(grid_for(a.rows, 4)), block(256);
    const double sx = sizeof(TX);
    const double bytes = (double)a.rows * d * (BWD ? sx + 2.0 + 4.0 : 4.0 + 2 * sx);
#define VMASR_OUTP_CASE(DD)                                                                                           \
    case DD:                                                                                                          \
        VMASR_LAUNCH(BWD ? VMASR_K_OUTPROJ_BWD : VMASR_K_OUTPROJ_FWD, bytes, (outproj_kernel<DD, TX, BWD>), grid, block, 0, st, a); \
        break;
    switch (d) {
        VMASR_OUTP_CASE(8) VMASR_OUTP_CASE(16) VMASR_OUTP_CASE(32) VMASR_OUTP_CASE(64)
        default: set_error("outproj: unsupported width %d", d); return VMASR_EINVAL;
    }
#undef VMASR_OUTP_CASE
    return check_launch(BWD ? "outproj_bwd" : "outproj_fwd");
}

}  // namespace
}  // namespace vmasr

VMASR_EXPORT int vmasr_outproj_supported(int32_t d, int32_t d_inner) {
    return ((d == 8 || d == 16 || d == 32 || d == 64) && d_inner == 2 * d) ? 1 : 0;
}

VMASR_EXPORT int vmasr_outproj_fwd(const void *g, const void *w, const void *x, const float *scale, int32_t rows_per_sample, void *y,
                                   int64_t rows, int32_t d, int32_t x_dtype, vmasr_stream_t stream) {
    VMASR_REQUIRE(g && w && x && y, VMASR_EINVAL, "outproj_fwd: null tensor");
    VMASR_REQUIRE(vmasr_outproj_supported(d, 2 * d) && rows > 0 && (!scale || rows_per_sample > 0), VMASR_EINVAL,
                  "outproj_fwd: need d in {8,16,32,64}");
    VMASR_REQUIRE(x_dtype == VMASR_F32 || x_dtype == VMASR_BF16, VMASR_EINVAL, "outproj_fwd: x must be fp32 or bf16");
    VMASR_REQUIRE(aligned_to(g, 16) && aligned_to(w, 16) && aligned_to(x, 8) && aligned_to(y, 8), VMASR_EINVAL, "outproj_fwd: unaligned");
    OutProjArgs a{};
    a.g = static_cast<const bf16_t *>(g); a.w = static_cast<const bf16_t *>(w); a.x = x; a.scale = scale; a.y = y; a.rows = rows;
    a.rows_per_sample = rows_per_sample;
    return x_dtype == VMASR_F32 ? launch_outproj<false, float>(a, d, static_cast<hipStream_t>(stream))
                                : launch_outproj<false, bf16_t>(a, d, static_cast<hipStream_t>(stream));
}

VMASR_EXPORT int vmasr_outproj_bwd(const void *gy, const void *wt, const float *scale, int32_t rows_per_sample, void *dg, void *gys,
                                   int64_t rows, int32_t d, int32_t x_dtype, vmasr_stream_t stream) {
    VMASR_REQUIRE(gy && wt && dg && gys, VMASR_EINVAL, "outproj_bwd: null tensor");
    VMASR_REQUIRE(vmasr_outproj_supported(d, 2 * d) && rows > 0 && (!scale || rows_per_sample > 0), VMASR_EINVAL,
                  "outproj_bwd: need d in {8,16,32,64}");
    VMASR_REQUIRE(x_dtype == VMASR_F32 || x_dtype == VMASR_BF16, VMASR_EINVAL, "outproj_bwd: gy must be fp32 or bf16");
    VMASR_REQUIRE(aligned_to(gy, 16) && aligned_to(wt, 16) && aligned_to(dg, 8) && aligned_to(gys, 16), VMASR_EINVAL, "outproj_bwd: unaligned");
    OutProjArgs a{};
    a.w = static_cast<const bf16_t *>(wt); a.x = gy; a.scale = scale; a.dg = static_cast<bf16_t *>(dg); a.gys = static_cast<bf16_t *>(gys);
    a.rows = rows; a.rows_per_sample = rows_per_sample;
    return x_dtype == VMASR_F32 ? launch_outproj<true, float>(a, d, static_cast<hipStream_t>(stream))
                                : launch_outproj<true, bf16_t>(a, d, static_cast<hipStream_t>(stream));
}
