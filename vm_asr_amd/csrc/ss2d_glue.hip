// ss2d_glue.hip — the memory-bound glue of SS2D.forwardv2 around the scan core, fused into two operators, gfx950.
//
// model/vmamba.py:1533-1552 per block:   xz = in_proj(x); x, z = chunk(xz); z = SiLU(z);
//     x = x.permute(0,3,1,2).contiguous(); x = SiLU(conv2d(x)); y = core(x)            [(B,D,L) merged scan output]
//     y = out_norm(y.transpose(1,2).contiguous()); y = y.to(x.dtype); y = y * z; out_proj(y)
// The lines around conv / core are, as ATen, 2 + 4 kernels forward (SiLU on a strided view, a layout copy; a layout
// copy, LayerNorm, a cast, a multiply) and 3 + 6 backward, each a full pass over (B, L, D) — 12 launches per block
// and ~50 B of traffic per element for ~14 B of information.  Here:
//
//   ss2d_pre   : xz (B L, 2D) -> xT (B, D, L) = channel-first copy of the x half, sz (B L, D) = SiLU(z half)
//   ln_gate    : y (B, D, L) fp32, sz -> out (B L, D) = LayerNorm_D(y^T) * sz        (+ mean, rstd for the backward)
// and their backwards (one launch each).  A workgroup owns P consecutive positions x all D channels: the
// channel-first side is read / written as P-long runs per channel (coalesced along l), the channel-last side as
// 16-byte vectors per position, and the (D x P) tile crosses between the two layouts through LDS (padded rows:
// conflict-free both ways).  LayerNorm statistics are two-pass (mean, then centred squares) over the LDS tile.
#include "common.h"

#include <algorithm>

namespace vmasr {
namespace {

struct GlueGeom {
    int B, D, L, P, nch;   // P positions per workgroup, nch = D / CH channel chunks per position
    float eps;
    unsigned *det = nullptr;   // deterministic mode: the ticket word of this kernel id (common.h), else null
};

// accurate exp and a true division: these kernels are memory-bound, and the generator ends in a LayerNorm over two
// channels that amplifies every rounding error upstream (tests/test_fullsize.py) — no fast-math shortcuts here
__device__ __forceinline__ float sigmoid_f(float z) { return 1.f / (1.f + expf(-z)); }

template <typename T, int CH>
__device__ __forceinline__ void load_chunk(const T *__restrict__ p, float (&v)[CH]) {
    if constexpr (sizeof(T) == 4) {
        if constexpr (CH == 4) { const float4 q = *reinterpret_cast<const float4 *>(p); v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w; }
        else if constexpr (CH == 2) { const float2 q = *reinterpret_cast<const float2 *>(p); v[0] = q.x; v[1] = q.y; }
        else { v[0] = p[0]; }
    } else {
        union { uint4 r4; uint2 r2; uint32_t r1; T e[8]; } q;
        if constexpr (CH == 8) q.r4 = *reinterpret_cast<const uint4 *>(p);
        else if constexpr (CH == 4) q.r2 = *reinterpret_cast<const uint2 *>(p);
        else if constexpr (CH == 2) q.r1 = *reinterpret_cast<const uint32_t *>(p);
        else q.e[0] = p[0];
#pragma unroll
        for (int i = 0; i < CH; ++i) v[i] = to_f32(q.e[i]);
    }
}

template <typename T, int CH>
__device__ __forceinline__ void store_chunk(T *__restrict__ p, const float (&v)[CH]) {
    if constexpr (sizeof(T) == 4) {
        if constexpr (CH == 4) *reinterpret_cast<float4 *>(p) = make_float4(v[0], v[1], v[2], v[3]);
        else if constexpr (CH == 2) *reinterpret_cast<float2 *>(p) = make_float2(v[0], v[1]);
        else p[0] = v[0];
    } else {
        union { uint4 r4; uint2 r2; uint32_t r1; T e[8]; } q;
#pragma unroll
        for (int i = 0; i < CH; ++i) q.e[i] = from_f32<T>(v[i]);
        if constexpr (CH == 8) *reinterpret_cast<uint4 *>(p) = q.r4;
        else if constexpr (CH == 4) *reinterpret_cast<uint2 *>(p) = q.r2;
        else if constexpr (CH == 2) *reinterpret_cast<uint32_t *>(p) = q.r1;
        else p[0] = q.e[0];
    }
}

// (D x P) LDS tile <-> channel-first rows  rows[(b*D + d)*L + l0 + p]   (coalesced along p)
template <typename T>
__device__ __forceinline__ void tile_from_rows(float *tile, const T *__restrict__ rows, const GlueGeom &g, const int b, const int l0) {
    for (int i = threadIdx.x; i < g.D * g.P; i += blockDim.x) {
        const int d = i / g.P, p = i % g.P;
        tile[d * (g.P + 1) + p] = to_f32(rows[((size_t)b * g.D + d) * g.L + l0 + p]);
    }
}
template <typename T>
__device__ __forceinline__ void tile_to_rows(const float *tile, T *__restrict__ rows, const GlueGeom &g, const int b, const int l0) {
    for (int i = threadIdx.x; i < g.D * g.P; i += blockDim.x) {
        const int d = i / g.P, p = i % g.P;
        rows[((size_t)b * g.D + d) * g.L + l0 + p] = from_f32<T>(tile[d * (g.P + 1) + p]);
    }
}

// ---- ss2d_pre ----------------------------------------------------------------------------------------------------
template <typename T, int CH>
__global__ __launch_bounds__(256) void ss2d_pre_fwd_kernel(const T *__restrict__ xz, T *__restrict__ xT, T *__restrict__ sz,
                                                           const GlueGeom g) {
    extern __shared__ float tile[];
    const int b = blockIdx.y, l0 = blockIdx.x * g.P;
    for (int i = threadIdx.x; i < g.P * g.nch; i += blockDim.x) {
        const int p = i / g.nch, c = i % g.nch;
        const size_t row = (size_t)b * g.L + l0 + p;
        float xv[CH], zv[CH];
        load_chunk<T, CH>(xz + row * 2 * g.D + c * CH, xv);
        load_chunk<T, CH>(xz + row * 2 * g.D + g.D + c * CH, zv);
#pragma unroll
        for (int e = 0; e < CH; ++e) {
            tile[(c * CH + e) * (g.P + 1) + p] = xv[e];
            zv[e] = zv[e] * sigmoid_f(zv[e]);
        }
        store_chunk<T, CH>(sz + row * g.D + c * CH, zv);
    }
    __syncthreads();
    tile_to_rows<T>(tile, xT, g, b, l0);
}

template <typename T, int CH>
__global__ __launch_bounds__(256) void ss2d_pre_bwd_kernel(const T *__restrict__ xz, const T *__restrict__ dxT,
                                                           const T *__restrict__ dsz, T *__restrict__ dxz, const GlueGeom g) {
    extern __shared__ float tile[];
    const int b = blockIdx.y, l0 = blockIdx.x * g.P;
    tile_from_rows<T>(tile, dxT, g, b, l0);
    __syncthreads();
    for (int i = threadIdx.x; i < g.P * g.nch; i += blockDim.x) {
        const int p = i / g.nch, c = i % g.nch;
        const size_t row = (size_t)b * g.L + l0 + p;
        float dx[CH], zv[CH], gz[CH];
        load_chunk<T, CH>(xz + row * 2 * g.D + g.D + c * CH, zv);
        load_chunk<T, CH>(dsz + row * g.D + c * CH, gz);
#pragma unroll
        for (int e = 0; e < CH; ++e) {
            dx[e] = tile[(c * CH + e) * (g.P + 1) + p];
            const float s = sigmoid_f(zv[e]);
            gz[e] *= s * fmaf(zv[e], 1.f - s, 1.f);     // d SiLU = s (1 + z (1 - s))
        }
        store_chunk<T, CH>(dxz + row * 2 * g.D + c * CH, dx);
        store_chunk<T, CH>(dxz + row * 2 * g.D + g.D + c * CH, gz);
    }
}

// ---- ln_gate -------------------------------------------------------------------------------------------------------
// LDS: tile (D x (P+1)) | red (256 floats) | stat (2 P) | acc (2 D, backward)
template <typename T, int CH>
__global__ __launch_bounds__(256) void ln_gate_fwd_kernel(const float *__restrict__ y, const T *__restrict__ sz,
                                                          const float *__restrict__ gamma, const float *__restrict__ beta,
                                                          T *__restrict__ out, float *__restrict__ mean, float *__restrict__ rstd,
                                                          const GlueGeom g) {
    extern __shared__ float lds[];
    float *tile = lds, *red = lds + g.D * (g.P + 1), *stat = red + 256;
    const int b = blockIdx.y, l0 = blockIdx.x * g.P;
    tile_from_rows<float>(tile, y, g, b, l0);
    __syncthreads();
    // statistics: position p = tid % P, T_ = 256 / P threads per position take the channels t, t + T_, ...
    const int p = threadIdx.x % g.P, t = threadIdx.x / g.P, TP = blockDim.x / g.P;
    float s = 0.f;
    for (int d = t; d < g.D; d += TP) s += tile[d * (g.P + 1) + p];
    red[t * g.P + p] = s;
    __syncthreads();
    float mu = 0.f;
    for (int k = 0; k < TP; ++k) mu += red[k * g.P + p];
    mu /= (float)g.D;
    __syncthreads();
    float q = 0.f;
    for (int d = t; d < g.D; d += TP) { const float c = tile[d * (g.P + 1) + p] - mu; q = fmaf(c, c, q); }
    red[t * g.P + p] = q;
    __syncthreads();
    if (t == 0) {
        float v = 0.f;
        for (int k = 0; k < TP; ++k) v += red[k * g.P + p];
        const float rs = rsqrtf(v / (float)g.D + g.eps);
        stat[p] = mu; stat[g.P + p] = rs;
        mean[(size_t)b * g.L + l0 + p] = mu;
        rstd[(size_t)b * g.L + l0 + p] = rs;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < g.P * g.nch; i += blockDim.x) {
        const int pp = i / g.nch, c = i % g.nch;
        const size_t row = (size_t)b * g.L + l0 + pp;
        float zv[CH], o[CH];
        load_chunk<T, CH>(sz + row * g.D + c * CH, zv);
        const float m = stat[pp], r = stat[g.P + pp];
#pragma unroll
        for (int e = 0; e < CH; ++e) {
            const int d = c * CH + e;
            o[e] = fmaf((tile[d * (g.P + 1) + pp] - m) * r, gamma[d], beta[d]) * zv[e];
        }
        store_chunk<T, CH>(out + row * g.D + c * CH, o);
    }
}

template <typename T, int CH>
__global__ __launch_bounds__(256) void ln_gate_bwd_kernel(const float *__restrict__ y, const T *__restrict__ sz,
                                                          const T *__restrict__ dout, const float *__restrict__ gamma,
                                                          const float *__restrict__ beta, const float *__restrict__ mean,
                                                          const float *__restrict__ rstd, float *__restrict__ dy,
                                                          T *__restrict__ dsz, float *__restrict__ dgamma,
                                                          float *__restrict__ dbeta, float *__restrict__ part, const GlueGeom g) {
    extern __shared__ float lds[];
    float *tile = lds, *acc = lds + g.D * (g.P + 1);   // acc: dgamma[D], dbeta[D]
    const int b = blockIdx.y, l0 = blockIdx.x * g.P;
    for (int i = threadIdx.x; i < 2 * g.D; i += blockDim.x) acc[i] = 0.f;
    tile_from_rows<float>(tile, y, g, b, l0);
    __syncthreads();
    const float invD = 1.f / (float)g.D;
    float dg[CH], db[CH];
#pragma unroll
    for (int e = 0; e < CH; ++e) { dg[e] = 0.f; db[e] = 0.f; }
    const int c = threadIdx.x % g.nch;                   // 256 % nch == 0: the chunk of a thread is loop-invariant
    for (int i = threadIdx.x; i < g.P * g.nch; i += blockDim.x) {
        const int pp = i / g.nch;
        const size_t row = (size_t)b * g.L + l0 + pp;
        const float m = mean[row], r = rstd[row];
        float zv[CH], go[CH], xh[CH], gg[CH];
        load_chunk<T, CH>(sz + row * g.D + c * CH, zv);
        load_chunk<T, CH>(dout + row * g.D + c * CH, go);
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int e = 0; e < CH; ++e) {
            const int d = c * CH + e;
            xh[e] = (tile[d * (g.P + 1) + pp] - m) * r;
            const float a = fmaf(xh[e], gamma[d], beta[d]);      // LayerNorm output
            const float gl = go[e] * zv[e];                      // gradient wrt the LayerNorm output
            zv[e] = go[e] * a;                                   // gradient wrt sz
            dg[e] = fmaf(gl, xh[e], dg[e]);
            db[e] += gl;
            gg[e] = gl * gamma[d];
            s1 += gg[e];
            s2 = fmaf(gg[e], xh[e], s2);
        }
        store_chunk<T, CH>(dsz + row * g.D + c * CH, zv);
        for (int off = g.nch >> 1; off > 0; off >>= 1) {         // the nch lanes of a position are adjacent (nch | 64)
            s1 += __shfl_xor(s1, off);
            s2 += __shfl_xor(s2, off);
        }
#pragma unroll
        for (int e = 0; e < CH; ++e) tile[(c * CH + e) * (g.P + 1) + pp] = r * (gg[e] - s1 * invD - xh[e] * s2 * invD);
    }
    // lanes l, l + nch, l + 2 nch, ... of a wave hold the same channels: fold them first (one LDS atomic per wave and channel)
#pragma unroll
    for (int e = 0; e < CH; ++e) {
        for (int off = g.nch; off < 64; off <<= 1) {
            dg[e] += __shfl_xor(dg[e], off);
            db[e] += __shfl_xor(db[e], off);
        }
    }
    // (deterministic mode: the waves' LDS adds in wave order, the workgroups' global atomics in workgroup order — common.h)
    VMASR_DET_WAVE_ORDER(g.det, 4, {
        if ((threadIdx.x & 63) < g.nch) {
            _Pragma("unroll") for (int e = 0; e < CH; ++e) {
                atomicAdd(&acc[c * CH + e], dg[e]);
                atomicAdd(&acc[g.D + c * CH + e], db[e]);
            }
        }
    });
    __syncthreads();
    tile_to_rows<float>(tile, dy, g, b, l0);
    if (part) {
        // per-workgroup partials [dgamma (D) | dbeta (D)], summed by ln.hip's reduce kernel: with up to 1 024 workgroups adding to
        // the same D addresses the atomics serialise (~13 ns each per address: 13 us of a 27 us launch at 64 x 64 x 64)
        float *dst = part + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 2 * g.D;
        for (int d = threadIdx.x; d < 2 * g.D; d += blockDim.x) dst[d] = acc[d];
    } else {
        det_enter(g.det);
        for (int d = threadIdx.x; d < g.D; d += blockDim.x) {
            atomicAdd(dgamma + d, acc[d]);
            atomicAdd(dbeta + d, acc[g.D + d]);
        }
        det_leave(g.det);
    }
}

// ---- direct variants for d_inner <= 32 (the high-resolution stages: most of the bytes) ---------------------------------
// One THREAD owns one position and keeps its D channels in registers: the channel-first side is read / written with
// one coalesced 4-byte access per channel (64 consecutive positions per wave), the channel-last side as the thread's
// own contiguous row; LayerNorm needs no cross-thread reduction and there is no LDS tile and no barrier on the data
// path.  The backward walks kIter x 256 positions per workgroup so that the per-channel dgamma / dbeta partials are
// wave-reduced once per workgroup.
template <typename T, int D>
struct RowIO {
    static constexpr int V = 16 / (int)sizeof(T);
    static constexpr int CH = D < V ? D : V;
    __device__ static __forceinline__ void load(const T *__restrict__ p, float (&v)[D]) {
#pragma unroll
        for (int c = 0; c < D / CH; ++c) {
            float t[CH];
            load_chunk<T, CH>(p + c * CH, t);
#pragma unroll
            for (int e = 0; e < CH; ++e) v[c * CH + e] = t[e];
        }
    }
    __device__ static __forceinline__ void store(T *__restrict__ p, const float (&v)[D]) {
#pragma unroll
        for (int c = 0; c < D / CH; ++c) {
            float t[CH];
#pragma unroll
            for (int e = 0; e < CH; ++e) t[e] = v[c * CH + e];
            store_chunk<T, CH>(p + c * CH, t);
        }
    }
};

template <typename T, int D>
__global__ __launch_bounds__(256) void ss2d_pre_fwd_direct(const T *__restrict__ xz, T *__restrict__ xT, T *__restrict__ sz,
                                                           const int L) {
    const int b = blockIdx.y, l = blockIdx.x * 256 + threadIdx.x;
    if (l >= L) return;
    const size_t row = (size_t)b * L + l;
    float xv[D], zv[D];
    RowIO<T, D>::load(xz + row * 2 * D, xv);
    RowIO<T, D>::load(xz + row * 2 * D + D, zv);
#pragma unroll
    for (int d = 0; d < D; ++d) {
        xT[((size_t)b * D + d) * L + l] = from_f32<T>(xv[d]);
        zv[d] = zv[d] * sigmoid_f(zv[d]);
    }
    RowIO<T, D>::store(sz + row * D, zv);
}

template <typename T, int D>
__global__ __launch_bounds__(256) void ss2d_pre_bwd_direct(const T *__restrict__ xz, const T *__restrict__ dxT,
                                                           const T *__restrict__ dsz, T *__restrict__ dxz, const int L) {
    const int b = blockIdx.y, l = blockIdx.x * 256 + threadIdx.x;
    if (l >= L) return;
    const size_t row = (size_t)b * L + l;
    float dx[D], zv[D], gz[D];
    RowIO<T, D>::load(xz + row * 2 * D + D, zv);
    RowIO<T, D>::load(dsz + row * D, gz);
#pragma unroll
    for (int d = 0; d < D; ++d) {
        dx[d] = to_f32(dxT[((size_t)b * D + d) * L + l]);
        const float s = sigmoid_f(zv[d]);
        gz[d] *= s * fmaf(zv[d], 1.f - s, 1.f);
    }
    RowIO<T, D>::store(dxz + row * 2 * D, dx);
    RowIO<T, D>::store(dxz + row * 2 * D + D, gz);
}

template <typename T, int D>
__global__ __launch_bounds__(256) void ln_gate_fwd_direct(const float *__restrict__ y, const T *__restrict__ sz,
                                                          const float *__restrict__ gamma, const float *__restrict__ beta,
                                                          T *__restrict__ out, float *__restrict__ mean, float *__restrict__ rstd,
                                                          const int L, const float eps) {
    const int b = blockIdx.y, l = blockIdx.x * 256 + threadIdx.x;
    if (l >= L) return;
    const size_t row = (size_t)b * L + l;
    float v[D], zv[D];
    float s = 0.f;
#pragma unroll
    for (int d = 0; d < D; ++d) { v[d] = y[((size_t)b * D + d) * L + l]; s += v[d]; }
    const float mu = s * (1.f / D);
    float q = 0.f;
#pragma unroll
    for (int d = 0; d < D; ++d) { v[d] -= mu; q = fmaf(v[d], v[d], q); }
    const float rs = rsqrtf(q * (1.f / D) + eps);
    RowIO<T, D>::load(sz + row * D, zv);
#pragma unroll
    for (int d = 0; d < D; ++d) zv[d] *= fmaf(v[d] * rs, gamma[d], beta[d]);
    RowIO<T, D>::store(out + row * D, zv);
    mean[row] = mu;
    rstd[row] = rs;
}

constexpr int kGlueMaxBlocks = 64;   // workgroups per batch element of the direct backward (bounds the same-address
                                     // atomics of dgamma / dbeta: 2 D per workgroup)

template <typename T, int D>
__global__ __launch_bounds__(256) void ln_gate_bwd_direct(const float *__restrict__ y, const T *__restrict__ sz,
                                                          const T *__restrict__ dout, const float *__restrict__ gamma,
                                                          const float *__restrict__ beta, const float *__restrict__ mean,
                                                          const float *__restrict__ rstd, float *__restrict__ dy,
                                                          T *__restrict__ dsz, float *__restrict__ dgamma,
                                                          float *__restrict__ dbeta, const int L, const int iters, unsigned *det) {
    __shared__ float acc[2 * D];
    const int b = blockIdx.y;
    if (threadIdx.x < 2 * D) acc[threadIdx.x] = 0.f;
    __syncthreads();
    float dg[D], db[D];
#pragma unroll
    for (int d = 0; d < D; ++d) { dg[d] = 0.f; db[d] = 0.f; }
#pragma unroll 1
    for (int k = 0; k < iters; ++k) {
        const int l = (k * gridDim.x + blockIdx.x) * 256 + threadIdx.x;   // consecutive workgroups take consecutive chunks
        if (l >= L) break;
        const size_t row = (size_t)b * L + l;
        const float m = mean[row], r = rstd[row];
        float xh[D], zv[D], go[D];
        RowIO<T, D>::load(sz + row * D, zv);
        RowIO<T, D>::load(dout + row * D, go);
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int d = 0; d < D; ++d) {
            xh[d] = (y[((size_t)b * D + d) * L + l] - m) * r;
            const float a = fmaf(xh[d], gamma[d], beta[d]);
            const float gl = go[d] * zv[d];
            zv[d] = go[d] * a;                   // d sz
            dg[d] = fmaf(gl, xh[d], dg[d]);
            db[d] += gl;
            go[d] = gl * gamma[d];               // g
            s1 += go[d];
            s2 = fmaf(go[d], xh[d], s2);
        }
        RowIO<T, D>::store(dsz + row * D, zv);
        s1 *= (1.f / D);
        s2 *= (1.f / D);
#pragma unroll
        for (int d = 0; d < D; ++d) dy[((size_t)b * D + d) * L + l] = r * (go[d] - s1 - xh[d] * s2);
    }
    const int lane = threadIdx.x & 63;
    // (deterministic mode: the four waves' LDS adds in wave order, the workgroups' global atomics in workgroup order — common.h)
    VMASR_DET_WAVE_ORDER(det, 4, {
        _Pragma("unroll") for (int d = 0; d < D; ++d) {
            const float a = wave_sum(dg[d]), c = wave_sum(db[d]);
            if (lane == 0) { atomicAdd(&acc[d], a); atomicAdd(&acc[D + d], c); }
        }
    });
    __syncthreads();
    det_enter(det);
    if (threadIdx.x < D) atomicAdd(dgamma + threadIdx.x, acc[threadIdx.x]);
    else if (threadIdx.x < 2 * D) atomicAdd(dbeta + threadIdx.x - D, acc[threadIdx.x]);
    det_leave(det);
}

// ---- pair variants: y = y02 + transpose(y13) formed on the fly (the last step of CrossMerge never becomes a tensor) ------
// A workgroup owns a 16 x 16 (h, w) tile, thread (th, tw): the (h,w)-ordered operand is read as 64-byte runs along w, the
// (w,h)-ordered one as 16-byte runs along h that the workgroup's four waves complete to 64 bytes (L1 hits) — instead of a
// merge kernel that reads both and writes y (12 B per (row, position)) and of this kernel reading y back.
struct PairGeom {
    int H, W, tiles_w, ntiles;
};

template <typename T, int D>
__global__ __launch_bounds__(256) void ln_gate_pair_fwd_kernel(const float *__restrict__ y02, const float *__restrict__ y13,
                                                               const T *__restrict__ sz, const float *__restrict__ gamma,
                                                               const float *__restrict__ beta, T *__restrict__ out,
                                                               float *__restrict__ mean, float *__restrict__ rstd,
                                                               const PairGeom g, const float eps) {
    const int b = blockIdx.y, L = g.H * g.W;
    const int h = (blockIdx.x / g.tiles_w) * 16 + (threadIdx.x >> 4), w = (blockIdx.x % g.tiles_w) * 16 + (threadIdx.x & 15);
    const int l = h * g.W + w, lt = w * g.H + h;
    const size_t row = (size_t)b * L + l;
    float v[D], zv[D];
    float s = 0.f;
#pragma unroll
    for (int d = 0; d < D; ++d) {
        const size_t base = ((size_t)b * D + d) * L;
        v[d] = y02[base + l] + y13[base + lt];
        s += v[d];
    }
    const float mu = s * (1.f / D);
    float q = 0.f;
#pragma unroll
    for (int d = 0; d < D; ++d) { v[d] -= mu; q = fmaf(v[d], v[d], q); }
    const float rs = rsqrtf(q * (1.f / D) + eps);
    RowIO<T, D>::load(sz + row * D, zv);
#pragma unroll
    for (int d = 0; d < D; ++d) zv[d] *= fmaf(v[d] * rs, gamma[d], beta[d]);
    RowIO<T, D>::store(out + row * D, zv);
    mean[row] = mu;
    rstd[row] = rs;
}

template <typename T, int D>
__global__ __launch_bounds__(256) void ln_gate_pair_bwd_kernel(const float *__restrict__ y02, const float *__restrict__ y13,
                                                               const T *__restrict__ sz, const T *__restrict__ dout,
                                                               const float *__restrict__ gamma, const float *__restrict__ beta,
                                                               const float *__restrict__ mean, const float *__restrict__ rstd,
                                                               float *__restrict__ dy02, float *__restrict__ dy13,
                                                               T *__restrict__ dsz, float *__restrict__ dgamma,
                                                               float *__restrict__ dbeta, const PairGeom g, const int iters, unsigned *det) {
    __shared__ float acc[2 * D];
    const int b = blockIdx.y, L = g.H * g.W;
    if (threadIdx.x < 2 * D) acc[threadIdx.x] = 0.f;
    __syncthreads();
    float dg[D], db[D];
#pragma unroll
    for (int d = 0; d < D; ++d) { dg[d] = 0.f; db[d] = 0.f; }
#pragma unroll 1
    for (int k = 0; k < iters; ++k) {
        const int tile = k * gridDim.x + blockIdx.x;
        if (tile >= g.ntiles) break;
        const int h = (tile / g.tiles_w) * 16 + (threadIdx.x >> 4), w = (tile % g.tiles_w) * 16 + (threadIdx.x & 15);
        const int l = h * g.W + w, lt = w * g.H + h;
        const size_t row = (size_t)b * L + l;
        const float m = mean[row], r = rstd[row];
        float xh[D], zv[D], go[D];
        RowIO<T, D>::load(sz + row * D, zv);
        RowIO<T, D>::load(dout + row * D, go);
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int d = 0; d < D; ++d) {
            const size_t base = ((size_t)b * D + d) * L;
            xh[d] = (y02[base + l] + y13[base + lt] - m) * r;
            const float a = fmaf(xh[d], gamma[d], beta[d]);
            const float gl = go[d] * zv[d];
            zv[d] = go[d] * a;                   // d sz
            dg[d] = fmaf(gl, xh[d], dg[d]);
            db[d] += gl;
            go[d] = gl * gamma[d];               // g
            s1 += go[d];
            s2 = fmaf(go[d], xh[d], s2);
        }
        RowIO<T, D>::store(dsz + row * D, zv);
        s1 *= (1.f / D);
        s2 *= (1.f / D);
#pragma unroll
        for (int d = 0; d < D; ++d) {
            const size_t base = ((size_t)b * D + d) * L;
            const float gv = r * (go[d] - s1 - xh[d] * s2);
            dy02[base + l] = gv;
            dy13[base + lt] = gv;
        }
    }
    const int lane = threadIdx.x & 63;
    // (deterministic mode: the four waves' LDS adds in wave order, the workgroups' global atomics in workgroup order — common.h)
    VMASR_DET_WAVE_ORDER(det, 4, {
        _Pragma("unroll") for (int d = 0; d < D; ++d) {
            const float a = wave_sum(dg[d]), c = wave_sum(db[d]);
            if (lane == 0) { atomicAdd(&acc[d], a); atomicAdd(&acc[D + d], c); }
        }
    });
    __syncthreads();
    det_enter(det);
    if (threadIdx.x < D) atomicAdd(dgamma + threadIdx.x, acc[threadIdx.x]);
    else if (threadIdx.x < 2 * D) atomicAdd(dbeta + threadIdx.x - D, acc[threadIdx.x]);
    det_leave(det);
}

#define GLUE_DIRECT_D(KERNEL, T, KID, BYTES, GRIDX, ...)                                                              \
    do {                                                                                                             \
        const dim3 grid(GRIDX, B);                                                                                   \
        switch (D) {                                                                                                 \
            case 2: VMASR_LAUNCH(KID, BYTES, (KERNEL<T, 2>), grid, dim3(256), 0, st, __VA_ARGS__); break;            \
            case 4: VMASR_LAUNCH(KID, BYTES, (KERNEL<T, 4>), grid, dim3(256), 0, st, __VA_ARGS__); break;            \
            case 8: VMASR_LAUNCH(KID, BYTES, (KERNEL<T, 8>), grid, dim3(256), 0, st, __VA_ARGS__); break;            \
            case 16: VMASR_LAUNCH(KID, BYTES, (KERNEL<T, 16>), grid, dim3(256), 0, st, __VA_ARGS__); break;          \
            default: VMASR_LAUNCH(KID, BYTES, (KERNEL<T, 32>), grid, dim3(256), 0, st, __VA_ARGS__);                 \
        }                                                                                                            \
    } while (0)

#define GLUE_DIRECT(KERNEL, KID, BYTES, GRIDX)                                                                       \
    do {                                                                                                             \
        if (dtype == VMASR_F32) GLUE_DIRECT_D(KERNEL, float, KID, BYTES, GRIDX, GLUE_ARGS(float));                   \
        else if (dtype == VMASR_F16) GLUE_DIRECT_D(KERNEL, f16_t, KID, BYTES, GRIDX, GLUE_ARGS(f16_t));              \
        else GLUE_DIRECT_D(KERNEL, bf16_t, KID, BYTES, GRIDX, GLUE_ARGS(bf16_t));                                    \
    } while (0)

inline bool direct_ok(int D) { return D == 2 || D == 4 || D == 8 || D == 16 || D == 32; }

int plan(int B, int D, int L, int esz, GlueGeom &g, int &CH, const char *what) {
    VMASR_REQUIRE(B > 0 && D > 0 && L > 0, VMASR_EINVAL, "%s: non-positive size", what);
    VMASR_REQUIRE(B <= 65535, VMASR_EINVAL, "%s: batch too large", what);
    const int V = 16 / esz;
    CH = D % V == 0 ? V : (D % 4 == 0 ? 4 : (D % 2 == 0 ? 2 : 1));
    if (esz == 4 && CH > 4) CH = 4;
    const int nch = D / CH;
    VMASR_REQUIRE(nch <= 64 && (nch & (nch - 1)) == 0, VMASR_EINVAL, "%s: d_inner / %d must be a power of two <= 64 (d_inner %d)", what, CH, D);
    // (the LDS-tile kernels serve d_inner >= 64, the deep stages with few positions: 16 positions per workgroup
    //  keep >= 128 workgroups in flight there; d_inner <= 32 runs the direct kernels)
    const int P = D <= 32 ? 64 : 16;
    VMASR_REQUIRE(D <= 512 && L % P == 0, VMASR_EINVAL, "%s: needs d_inner <= 512 and H*W %% %d == 0", what, P);
    g = GlueGeom{B, D, L, P, nch, 0.f};
    return 0;
}

#define GLUE_DISPATCH_CH(KERNEL, T, KID, BYTES, SM, ...)                                                             \
    do {                                                                                                             \
        const dim3 grid(g.L / g.P, g.B);                                                                             \
        if (CH == 8) VMASR_LAUNCH(KID, BYTES, (KERNEL<T, 8>), grid, dim3(256), SM, st, __VA_ARGS__, g);              \
        else if (CH == 4) VMASR_LAUNCH(KID, BYTES, (KERNEL<T, 4>), grid, dim3(256), SM, st, __VA_ARGS__, g);         \
        else if (CH == 2) VMASR_LAUNCH(KID, BYTES, (KERNEL<T, 2>), grid, dim3(256), SM, st, __VA_ARGS__, g);         \
        else VMASR_LAUNCH(KID, BYTES, (KERNEL<T, 1>), grid, dim3(256), SM, st, __VA_ARGS__, g);                      \
    } while (0)

#define GLUE_DISPATCH(KERNEL, KID, BYTES, SM, ...)                                                                   \
    do {                                                                                                             \
        if (dtype == VMASR_F32) { using T = float; GLUE_DISPATCH_CH(KERNEL, T, KID, BYTES, SM, GLUE_ARGS(float)); }   \
        else if (dtype == VMASR_F16) { using T = f16_t; GLUE_DISPATCH_CH(KERNEL, T, KID, BYTES, SM, GLUE_ARGS(f16_t)); } \
        else { using T = bf16_t; GLUE_DISPATCH_CH(KERNEL, T, KID, BYTES, SM, GLUE_ARGS(bf16_t)); }                    \
    } while (0)

}  // namespace
}  // namespace vmasr

using namespace vmasr;

VMASR_EXPORT int vmasr_ss2d_glue_supported(int32_t D, int32_t L, int32_t dtype) {
    GlueGeom g;
    int CH;
    const int esz = dtype == VMASR_F32 ? 4 : 2;
    if (D <= 0 || L <= 0 || D > 512) return 0;
    const int V = 16 / esz;
    int ch = D % V == 0 ? V : (D % 4 == 0 ? 4 : (D % 2 == 0 ? 2 : 1));
    if (esz == 4 && ch > 4) ch = 4;
    const int nch = D / ch;
    if (nch > 64 || (nch & (nch - 1))) return 0;
    const int P = D <= 32 ? 64 : 16;
    (void)g; (void)CH;
    return L % P == 0 ? 1 : 0;
}

VMASR_EXPORT int vmasr_ss2d_pre_fwd(const void *xz, void *xT, void *sz, int32_t B, int32_t D, int32_t L, int32_t dtype,
                                    vmasr_stream_t stream) {
    GlueGeom g;
    int CH;
    const int esz = dtype == VMASR_F32 ? 4 : 2;
    if (int e = plan(B, D, L, esz, g, CH, "ss2d_pre_fwd")) return e;
    VMASR_REQUIRE(xz && xT && sz, VMASR_EINVAL, "ss2d_pre_fwd: null tensor");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const size_t sm = (size_t)D * (g.P + 1) * sizeof(float);
    const double bytes = (double)B * L * D * esz * 4;
    if (direct_ok(D)) {
#define GLUE_ARGS(TT) static_cast<const TT *>(xz), static_cast<TT *>(xT), static_cast<TT *>(sz), L
        GLUE_DIRECT(ss2d_pre_fwd_direct, VMASR_K_SS2D_PRE, bytes, (L + 255) / 256);
#undef GLUE_ARGS
        return check_launch("ss2d_pre_fwd");
    }
#define GLUE_ARGS(TT) static_cast<const TT *>(xz), static_cast<TT *>(xT), static_cast<TT *>(sz)
    GLUE_DISPATCH(ss2d_pre_fwd_kernel, VMASR_K_SS2D_PRE, bytes, sm);
#undef GLUE_ARGS
    return check_launch("ss2d_pre_fwd");
}

VMASR_EXPORT int vmasr_ss2d_pre_bwd(const void *xz, const void *dxT, const void *dsz, void *dxz, int32_t B, int32_t D, int32_t L,
                                    int32_t dtype, vmasr_stream_t stream) {
    GlueGeom g;
    int CH;
    const int esz = dtype == VMASR_F32 ? 4 : 2;
    if (int e = plan(B, D, L, esz, g, CH, "ss2d_pre_bwd")) return e;
    VMASR_REQUIRE(xz && dxT && dsz && dxz, VMASR_EINVAL, "ss2d_pre_bwd: null tensor");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const size_t sm = (size_t)D * (g.P + 1) * sizeof(float);
    const double bytes = (double)B * L * D * esz * 5;
    if (direct_ok(D)) {
#define GLUE_ARGS(TT) static_cast<const TT *>(xz), static_cast<const TT *>(dxT), static_cast<const TT *>(dsz), static_cast<TT *>(dxz), L
        GLUE_DIRECT(ss2d_pre_bwd_direct, VMASR_K_SS2D_PRE, bytes, (L + 255) / 256);
#undef GLUE_ARGS
        return check_launch("ss2d_pre_bwd");
    }
#define GLUE_ARGS(TT) static_cast<const TT *>(xz), static_cast<const TT *>(dxT), static_cast<const TT *>(dsz), static_cast<TT *>(dxz)
    GLUE_DISPATCH(ss2d_pre_bwd_kernel, VMASR_K_SS2D_PRE, bytes, sm);
#undef GLUE_ARGS
    return check_launch("ss2d_pre_bwd");
}

VMASR_EXPORT int vmasr_ln_gate_fwd(const float *y, const void *sz, const float *gamma, const float *beta, void *out, float *mean,
                                   float *rstd, int32_t B, int32_t D, int32_t L, float eps, int32_t dtype, vmasr_stream_t stream) {
    GlueGeom g;
    int CH;
    const int esz = dtype == VMASR_F32 ? 4 : 2;
    if (int e = plan(B, D, L, esz, g, CH, "ln_gate_fwd")) return e;
    VMASR_REQUIRE(y && sz && gamma && beta && out && mean && rstd, VMASR_EINVAL, "ln_gate_fwd: null tensor");
    g.eps = eps;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const size_t sm = ((size_t)D * (g.P + 1) + 256 + 2 * g.P) * sizeof(float);
    const double bytes = (double)B * L * D * (4 + 2 * esz);
    if (direct_ok(D)) {
#define GLUE_ARGS(TT) y, static_cast<const TT *>(sz), gamma, beta, static_cast<TT *>(out), mean, rstd, L, eps
        GLUE_DIRECT(ln_gate_fwd_direct, VMASR_K_LN_GATE, bytes, (L + 255) / 256);
#undef GLUE_ARGS
        return check_launch("ln_gate_fwd");
    }
#define GLUE_ARGS(TT) y, static_cast<const TT *>(sz), gamma, beta, static_cast<TT *>(out), mean, rstd
    GLUE_DISPATCH(ln_gate_fwd_kernel, VMASR_K_LN_GATE, bytes, sm);
#undef GLUE_ARGS
    return check_launch("ln_gate_fwd");
}

VMASR_EXPORT int64_t vmasr_ln_gate_bwd_workspace(int32_t B, int32_t D, int32_t L, int32_t dtype) {
    GlueGeom g;
    int CH;
    if (direct_ok(D) || B <= 0 || D <= 0 || L <= 0) return 0;      // (the direct variants fold per workgroup and use few atomics)
    if (plan(B, D, L, dtype == VMASR_F32 ? 4 : 2, g, CH, "ln_gate_bwd_workspace")) return 0;
    return (int64_t)B * (L / g.P) * 2 * D;
}

static int ln_gate_bwd_impl(const float *y, const void *sz, const void *dout, const float *gamma, const float *beta, const float *mean,
                            const float *rstd, float *dy, void *dsz, float *dgamma, float *dbeta, float *part, int32_t B, int32_t D,
                            int32_t L, int32_t dtype, vmasr_stream_t stream);

VMASR_EXPORT int vmasr_ln_gate_bwd(const float *y, const void *sz, const void *dout, const float *gamma, const float *beta,
                                   const float *mean, const float *rstd, float *dy, void *dsz, float *dgamma, float *dbeta,
                                   int32_t B, int32_t D, int32_t L, int32_t dtype, vmasr_stream_t stream) {
    VMASR_REQUIRE(dgamma && dbeta, VMASR_EINVAL, "ln_gate_bwd: null tensor");
    return ln_gate_bwd_impl(y, sz, dout, gamma, beta, mean, rstd, dy, dsz, dgamma, dbeta, nullptr, B, D, L, dtype, stream);
}

VMASR_EXPORT int vmasr_ln_gate_bwd_ws(const float *y, const void *sz, const void *dout, const float *gamma, const float *beta,
                                      const float *mean, const float *rstd, float *dy, void *dsz, float *dgamma, float *dbeta, float *ws,
                                      int32_t B, int32_t D, int32_t L, int32_t dtype, vmasr_stream_t stream) {
    VMASR_REQUIRE(ws && vmasr_ln_gate_bwd_workspace(B, D, L, dtype) > 0, VMASR_EINVAL, "ln_gate_bwd_ws: no workspace variant for this shape");
    if (int e = ln_gate_bwd_impl(y, sz, dout, gamma, beta, mean, rstd, dy, dsz, nullptr, nullptr, ws, B, D, L, dtype, stream)) return e;
    if (!dgamma && !dbeta) return VMASR_OK;                        // partials only: the caller reduces them later
    const float *parts[1] = {ws};
    float *dgs[1] = {dgamma}, *dbs[1] = {dbeta};
    const int32_t nblk[1] = {(int32_t)(vmasr_ln_gate_bwd_workspace(B, D, L, dtype) / (2 * D))}, Cs[1] = {D};
    return vmasr_layer_norm_bwd_reduce_multi(parts, dgs, dbs, nblk, Cs, 1, stream);
}

static int ln_gate_bwd_impl(const float *y, const void *sz, const void *dout, const float *gamma, const float *beta, const float *mean,
                            const float *rstd, float *dy, void *dsz, float *dgamma, float *dbeta, float *part, int32_t B, int32_t D,
                            int32_t L, int32_t dtype, vmasr_stream_t stream) {
    GlueGeom g;
    int CH;
    const int esz = dtype == VMASR_F32 ? 4 : 2;
    if (int e = plan(B, D, L, esz, g, CH, "ln_gate_bwd")) return e;
    VMASR_REQUIRE(y && sz && dout && gamma && beta && mean && rstd && dy && dsz && (part || (dgamma && dbeta)), VMASR_EINVAL,
                  "ln_gate_bwd: null tensor");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const size_t sm = ((size_t)D * (g.P + 1) + 2 * D) * sizeof(float);
    const double bytes = (double)B * L * D * (8 + 3 * esz);
    g.det = det_ticket(VMASR_K_LN_GATE);
    if (direct_ok(D)) {
        const int chunks = (L + 255) / 256, nblk = std::min(chunks, kGlueMaxBlocks), iters = (chunks + nblk - 1) / nblk;
#define GLUE_ARGS(TT) y, static_cast<const TT *>(sz), static_cast<const TT *>(dout), gamma, beta, mean, rstd, dy, static_cast<TT *>(dsz), dgamma, dbeta, L, iters, det_ticket(VMASR_K_LN_GATE)
        GLUE_DIRECT(ln_gate_bwd_direct, VMASR_K_LN_GATE, bytes, nblk);
#undef GLUE_ARGS
        return check_launch("ln_gate_bwd");
    }
#define GLUE_ARGS(TT) y, static_cast<const TT *>(sz), static_cast<const TT *>(dout), gamma, beta, mean, rstd, dy, static_cast<TT *>(dsz), dgamma, dbeta, part
    GLUE_DISPATCH(ln_gate_bwd_kernel, VMASR_K_LN_GATE, bytes, sm);
#undef GLUE_ARGS
    return check_launch("ln_gate_bwd");
}

VMASR_EXPORT int vmasr_ln_gate_pair_supported(int32_t D, int32_t H, int32_t W) {
    return (direct_ok(D) && H > 0 && W > 0 && H % 16 == 0 && W % 16 == 0) ? 1 : 0;
}

VMASR_EXPORT int vmasr_ln_gate_pair_fwd(const float *y02, const float *y13, const void *sz, const float *gamma, const float *beta,
                                        void *out, float *mean, float *rstd, int32_t B, int32_t D, int32_t H, int32_t W, float eps,
                                        int32_t dtype, vmasr_stream_t stream) {
    VMASR_REQUIRE(vmasr_ln_gate_pair_supported(D, H, W) && B > 0 && B <= 65535, VMASR_EINVAL,
                  "ln_gate_pair_fwd: need d_inner in {2,4,8,16,32} and H, W multiples of 16");
    VMASR_REQUIRE(dtype == VMASR_F32 || dtype == VMASR_F16 || dtype == VMASR_BF16, VMASR_EINVAL, "ln_gate_pair_fwd: bad dtype");
    VMASR_REQUIRE(y02 && y13 && sz && gamma && beta && out && mean && rstd, VMASR_EINVAL, "ln_gate_pair_fwd: null tensor");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int L = H * W, esz = dtype == VMASR_F32 ? 4 : 2;
    const PairGeom g{H, W, W / 16, (H / 16) * (W / 16)};
    const double bytes = (double)B * L * D * (8 + 2 * esz);
#define GLUE_ARGS(TT) y02, y13, static_cast<const TT *>(sz), gamma, beta, static_cast<TT *>(out), mean, rstd, g, eps
    GLUE_DIRECT(ln_gate_pair_fwd_kernel, VMASR_K_LN_GATE, bytes, g.ntiles);
#undef GLUE_ARGS
    return check_launch("ln_gate_pair_fwd");
}

VMASR_EXPORT int vmasr_ln_gate_pair_bwd(const float *y02, const float *y13, const void *sz, const void *dout, const float *gamma,
                                        const float *beta, const float *mean, const float *rstd, float *dy02, float *dy13, void *dsz,
                                        float *dgamma, float *dbeta, int32_t B, int32_t D, int32_t H, int32_t W, int32_t dtype,
                                        vmasr_stream_t stream) {
    VMASR_REQUIRE(vmasr_ln_gate_pair_supported(D, H, W) && B > 0 && B <= 65535, VMASR_EINVAL,
                  "ln_gate_pair_bwd: need d_inner in {2,4,8,16,32} and H, W multiples of 16");
    VMASR_REQUIRE(dtype == VMASR_F32 || dtype == VMASR_F16 || dtype == VMASR_BF16, VMASR_EINVAL, "ln_gate_pair_bwd: bad dtype");
    VMASR_REQUIRE(y02 && y13 && sz && dout && gamma && beta && mean && rstd && dy02 && dy13 && dsz && dgamma && dbeta, VMASR_EINVAL,
                  "ln_gate_pair_bwd: null tensor");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int L = H * W, esz = dtype == VMASR_F32 ? 4 : 2;
    const PairGeom g{H, W, W / 16, (H / 16) * (W / 16)};
    const int nblk = std::min(g.ntiles, kGlueMaxBlocks), iters = (g.ntiles + nblk - 1) / nblk;
    const double bytes = (double)B * L * D * (16 + 3 * esz);
#define GLUE_ARGS(TT) y02, y13, static_cast<const TT *>(sz), static_cast<const TT *>(dout), gamma, beta, mean, rstd, dy02, dy13, static_cast<TT *>(dsz), dgamma, dbeta, g, iters, det_ticket(VMASR_K_LN_GATE)
    GLUE_DIRECT(ln_gate_pair_bwd_kernel, VMASR_K_LN_GATE, bytes, nblk);
#undef GLUE_ARGS
    return check_launch("ln_gate_pair_bwd");
}
