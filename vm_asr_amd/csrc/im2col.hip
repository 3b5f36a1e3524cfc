// im2col.hip — column / scatter kernels of the period discriminator's (k,1) convolutions, gfx950.
//
// The reference's MultiPeriodDiscriminator (model/discriminator.py:21-147) is five stacks of
// Conv2d((5,1), stride (3,1)) / ((5,1), 1) / ((3,1), 1) over the signal folded to (T/p, p).  MIOpen has no
// tuned solver for them on gfx950 (naive_conv_* fallbacks, 40+ ms per call), so the host side runs them as
// GEMMs on a channel-last (N = B*p sequences, H, C) layout; these kernels build the GEMM operand and
// scatter its gradient back — pure HBM-bound data movement, previously ATen strided copies
// (`x.unfold(...).reshape` at ~1 TB/s) and `_unfold_backward` (2.7 ms per step):
//
//   im2col : cols[n, h1, j, c] = x[n, h1*s + j - pad, c]   (0 outside [0,H))     -> (N*H1, k*C) row-major
//   col2im : dx[n, h, c] = sum_{j : (h + pad - j) % s == 0, h1 = (h + pad - j)/s in [0,H1)} dcols[n, h1, j, c]
//
// col2im is a gather (<= ceil(k/s) terms per element): no atomics, deterministic.
// Algorithmic bytes: im2col reads N*H*C, writes N*H1*k*C; col2im reads N*H1*k*C, writes N*H*C.
#include <algorithm>
#include <type_traits>

#include "common.h"

namespace vmasr {
namespace {

struct ColGeom {
    int N, H, C, k, stride, pad, H1;
    long rows_out;  // im2col: rows written (>= N*H1; the surplus is zero-filled: padding of a stacked operand)
};

// V = elements per 16-byte vector (4 fp32 / 8 bf16); C % V == 0 on the vector path
template <typename T, int V>
__global__ __launch_bounds__(256) void im2col_kernel(const T *__restrict__ x, T *__restrict__ cols, const ColGeom g) {
    const int cv = g.C / V;                                   // vectors per tap
    const long total = g.rows_out * g.k * cv;
    using Vec = typename std::conditional<V == 1, T, uint4>::type;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % cv);
        long r = i / cv;
        const int j = (int)(r % g.k);
        r /= g.k;
        const int h1 = (int)(r % g.H1);
        const int n = (int)(r / g.H1);
        const int h = h1 * g.stride + j - g.pad;
        Vec v{};
        if (n < g.N && h >= 0 && h < g.H) v = reinterpret_cast<const Vec *>(x + ((size_t)n * g.H + h) * g.C)[c];
        reinterpret_cast<Vec *>(cols)[i] = v;
    }
}

// im2col of fp32 input with the error-compensated bf16 split of split.hip fused into the store: hi = bf16(v),
// lo = bf16(v - hi) go to two (rows_out, k*C) bf16 operands (4 B written per element instead of 4 B written + 4 B
// re-read + 4 B written by im2col followed by vmasr_split_bf16).
__global__ __launch_bounds__(256) void im2col_split_kernel(const float *__restrict__ x, bf16_t *__restrict__ hi,
                                                           bf16_t *__restrict__ lo, const ColGeom g) {
    const int cv = g.C / 4;
    const long total = g.rows_out * g.k * cv;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % cv);
        long r = i / cv;
        const int j = (int)(r % g.k);
        r /= g.k;
        const int h1 = (int)(r % g.H1);
        const int n = (int)(r / g.H1);
        const int h = h1 * g.stride + j - g.pad;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (n < g.N && h >= 0 && h < g.H) v = reinterpret_cast<const float4 *>(x + ((size_t)n * g.H + h) * g.C)[c];
        const float e[4] = {v.x, v.y, v.z, v.w};
        union { uint2 raw; bf16_t b[4]; } H, L;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            H.b[q] = (bf16_t)e[q];
            L.b[q] = (bf16_t)(e[q] - (float)H.b[q]);
        }
        reinterpret_cast<uint2 *>(hi)[i] = H.raw;
        reinterpret_cast<uint2 *>(lo)[i] = L.raw;
    }
}

template <typename T, int V>
__global__ __launch_bounds__(256) void col2im_kernel(const T *__restrict__ dcols, T *__restrict__ dx, const ColGeom g) {
    const int cv = g.C / V;
    const long total = (long)g.N * g.H * cv;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % cv);
        const long r = i / cv;
        const int h = (int)(r % g.H);
        const int n = (int)(r / g.H);
        float acc[V];
#pragma unroll
        for (int e = 0; e < V; ++e) acc[e] = 0.f;
        for (int j = (h + g.pad) % g.stride; j < g.k; j += g.stride) {
            const int q = h + g.pad - j;
            if (q < 0) break;
            const int h1 = q / g.stride;
            if (h1 >= g.H1) continue;
            const T *src = dcols + ((((size_t)n * g.H1 + h1) * g.k + j) * g.C) + (size_t)c * V;
            if constexpr (V == 1) {
                acc[0] += to_f32(src[0]);
            } else {
                union { uint4 raw; T e[V]; } q4;
                q4.raw = *reinterpret_cast<const uint4 *>(src);
#pragma unroll
                for (int e = 0; e < V; ++e) acc[e] += to_f32(q4.e[e]);
            }
        }
        T *dst = dx + ((size_t)n * g.H + h) * g.C + (size_t)c * V;
        if constexpr (V == 1) {
            dst[0] = from_f32<T>(acc[0]);
        } else {
            union { uint4 raw; T e[V]; } o;
#pragma unroll
            for (int e = 0; e < V; ++e) o.e[e] = from_f32<T>(acc[e]);
            *reinterpret_cast<uint4 *>(dst) = o.raw;
        }
    }
}


// ---- multi-slot variants: the stacked discriminator pass runs n <= kMaxSlots (k,1) convolutions of identical
// (C, k, stride, pad) but different (N_i, H_i) per layer; one launch with blockIdx.y = slot replaces n launches
// (the per-slot kernels last 3-20 us each, the same order as the launch gap between them).
constexpr int kMaxSlots = 8;

struct SlotTable {
    const void *src[kMaxSlots];
    void *dst[kMaxSlots];
    int N[kMaxSlots], H[kMaxSlots], H1[kMaxSlots];
    long rows[kMaxSlots];        // stack_rows: valid rows of the slot
};

// cat3: ONE (n, rows, 3 K) operand [hi | lo | hi] (hi = the base pointer, lo unused) — the A side of the K-concatenated
// product [hi | lo | hi] . [w_hi; w_hi; w_lo], which accumulates the three bf16 products of an fp32 GEMM inside one GEMM
__global__ __launch_bounds__(256) void im2col_split_multi_kernel(const SlotTable t, bf16_t *__restrict__ hi,
                                                                 bf16_t *__restrict__ lo, const ColGeom g0, const int cat3) {
    const int s = blockIdx.y;
    const float *__restrict__ x = static_cast<const float *>(t.src[s]);
    const int N = t.N[s], H = t.H[s], H1 = t.H1[s];
    const int cv = g0.C / 4;
    const long total = g0.rows_out * g0.k * cv;
    uint2 *__restrict__ oh = reinterpret_cast<uint2 *>(hi) + (size_t)s * total;
    uint2 *__restrict__ ol = reinterpret_cast<uint2 *>(lo) + (size_t)s * total;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % cv);
        long r = i / cv;
        const int j = (int)(r % g0.k);
        r /= g0.k;
        const int h1 = (int)(r % H1);
        const long n = r / H1;
        const int h = h1 * g0.stride + j - g0.pad;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (n < N && h >= 0 && h < H) v = reinterpret_cast<const float4 *>(x + ((size_t)n * H + h) * g0.C)[c];
        const float e[4] = {v.x, v.y, v.z, v.w};
        union { uint2 raw; bf16_t b[4]; } Hh, L;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            Hh.b[q] = (bf16_t)e[q];
            L.b[q] = (bf16_t)(e[q] - (float)Hh.b[q]);
        }
        if (cat3) {
            const long kq = (long)g0.k * cv;                 // 8-byte units per row of one block
            uint2 *__restrict__ o3 = reinterpret_cast<uint2 *>(hi) + ((size_t)s * g0.rows_out + i / kq) * 3 * kq + i % kq;
            o3[0] = Hh.raw;
            o3[kq] = L.raw;
            o3[2 * kq] = Hh.raw;
        } else {
            oh[i] = Hh.raw;
            ol[i] = L.raw;
        }
    }
}

// dcols (n, rows, k*C) -> dx_s (N_s, H_s, C) for every slot with a destination
template <typename T, int V>
__global__ __launch_bounds__(256) void col2im_multi_kernel(const T *__restrict__ dcols_all, const SlotTable t, const ColGeom g0) {
    const int s = blockIdx.y;
    T *__restrict__ dx = static_cast<T *>(t.dst[s]);
    if (!dx) return;
    const int N = t.N[s], H = t.H[s], H1 = t.H1[s];
    const T *__restrict__ dcols = dcols_all + (size_t)s * g0.rows_out * g0.k * g0.C;
    const int cv = g0.C / V;
    const long valid = (long)N * H * cv;
    const long total = t.rows[s] ? t.rows[s] * cv : valid;      // rows[s]: rows of the destination slot (zero-filled past N*H), 0 = exact
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % cv);
        const long r = i / cv;
        const int h = (int)(r % H);
        const int n = (int)(r / H);
        float acc[V];
#pragma unroll
        for (int e = 0; e < V; ++e) acc[e] = 0.f;
        for (int j = (h + g0.pad) % g0.stride; j < g0.k && i < valid; j += g0.stride) {
            const int q = h + g0.pad - j;
            if (q < 0) break;
            const int h1 = q / g0.stride;
            if (h1 >= H1) continue;
            const T *src = dcols + ((((size_t)n * H1 + h1) * g0.k + j) * g0.C) + (size_t)c * V;
            if constexpr (V == 1) {
                acc[0] += to_f32(src[0]);
            } else {
                union { uint4 raw; T e[V]; } q4;
                q4.raw = *reinterpret_cast<const uint4 *>(src);
#pragma unroll
                for (int e = 0; e < V; ++e) acc[e] += to_f32(q4.e[e]);
            }
        }
        T *dst = dx + ((size_t)n * H + h) * g0.C + (size_t)c * V;
        if constexpr (V == 1) {
            dst[0] = from_f32<T>(acc[0]);
        } else {
            union { uint4 raw; T e[V]; } o;
#pragma unroll
            for (int e = 0; e < V; ++e) o.e[e] = from_f32<T>(acc[e]);
            *reinterpret_cast<uint4 *>(dst) = o.raw;
        }
    }
}

// full (n, rows, W bytes): slot s = its source's first rows[s]*W bytes, zeros below (and all zeros without a source).
// U = uint4 (16-byte path) or unsigned char.
template <typename U>
__global__ __launch_bounds__(256) void stack_rows_kernel(const SlotTable t, U *__restrict__ full, const long slot_units) {
    const int s = blockIdx.y;
    const U *__restrict__ src = static_cast<const U *>(t.src[s]);
    const long valid = src ? t.rows[s] : 0;                // in units of U
    U *__restrict__ out = full + (size_t)s * slot_units;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < slot_units; i += (long)gridDim.x * blockDim.x) {
        U v{};
        if (i < valid) v = src[i];
        out[i] = v;
    }
}

template <typename T>
int launch_col2im_multi(const void *dcols, const SlotTable &t, int n, long max_total_elems, const ColGeom &g, bool vec_ok, double bytes,
                        hipStream_t st) {
    constexpr int V = 16 / sizeof(T);
    const bool vec = vec_ok && (g.C % V == 0);
    const long total = max_total_elems / (vec ? V : 1);
    const int blocks = (int)std::min<long>((total + 255) / 256, 256L * 16);
    if (vec) VMASR_LAUNCH(VMASR_K_COL2IM, bytes, (col2im_multi_kernel<T, V>), dim3(blocks, n), dim3(256), 0, st, static_cast<const T *>(dcols), t, g);
    else VMASR_LAUNCH(VMASR_K_COL2IM, bytes, (col2im_multi_kernel<T, 1>), dim3(blocks, n), dim3(256), 0, st, static_cast<const T *>(dcols), t, g);
    return check_launch("col2im_kx1_multi");
}

template <typename T>
int launch(bool fwd, const void *src, void *dst, const ColGeom &g, hipStream_t st) {
    constexpr int V = 16 / sizeof(T);
    const bool vec = (g.C % V == 0) && aligned_to(src, 16) && aligned_to(dst, 16);
    const long cvn = vec ? g.C / V : g.C;
    const long total = fwd ? g.rows_out * g.k * cvn : (long)g.N * g.H * cvn;
    const int blocks = (int)std::min<long>((total + 255) / 256, 256L * 64);   // grid-stride: ~64 workgroups per CU at most
    const double bytes = ((double)g.N * g.H * g.C + (double)g.N * g.H1 * g.k * g.C) * sizeof(T);
    if (fwd) {
        if (vec) VMASR_LAUNCH(VMASR_K_IM2COL, bytes, (im2col_kernel<T, V>), dim3(blocks), dim3(256), 0, st, static_cast<const T *>(src), static_cast<T *>(dst), g);
        else VMASR_LAUNCH(VMASR_K_IM2COL, bytes, (im2col_kernel<T, 1>), dim3(blocks), dim3(256), 0, st, static_cast<const T *>(src), static_cast<T *>(dst), g);
    } else {
        if (vec) VMASR_LAUNCH(VMASR_K_COL2IM, bytes, (col2im_kernel<T, V>), dim3(blocks), dim3(256), 0, st, static_cast<const T *>(src), static_cast<T *>(dst), g);
        else VMASR_LAUNCH(VMASR_K_COL2IM, bytes, (col2im_kernel<T, 1>), dim3(blocks), dim3(256), 0, st, static_cast<const T *>(src), static_cast<T *>(dst), g);
    }
    return check_launch(fwd ? "im2col_kx1" : "col2im_kx1");
}

int run(bool fwd, const void *src, void *dst, int64_t N, int32_t H, int32_t C, int32_t k, int32_t stride, int32_t pad,
        int64_t rows_out, int32_t dtype, hipStream_t st) {
    VMASR_REQUIRE(src && dst, VMASR_EINVAL, "im2col_kx1: null tensor");
    VMASR_REQUIRE(N > 0 && H > 0 && C > 0 && k > 0 && stride > 0 && pad >= 0 && H + 2 * pad >= k, VMASR_EINVAL,
                  "im2col_kx1: bad geometry (N=%ld H=%d C=%d k=%d stride=%d pad=%d)", (long)N, H, C, k, stride, pad);
    VMASR_REQUIRE(N <= 0x7fffffff, VMASR_EINVAL, "im2col_kx1: too many sequences");
    const int H1 = (H + 2 * pad - k) / stride + 1;
    VMASR_REQUIRE(rows_out == 0 || rows_out >= N * H1, VMASR_EINVAL, "im2col_kx1: rows_out smaller than N*H1");
    const ColGeom g{(int)N, H, C, k, stride, pad, H1, rows_out ? (long)rows_out : (long)N * H1};
    switch (dtype) {
        case VMASR_F32: return launch<float>(fwd, src, dst, g, st);
        case VMASR_F16: return launch<f16_t>(fwd, src, dst, g, st);
        case VMASR_BF16: return launch<bf16_t>(fwd, src, dst, g, st);
    }
    set_error("im2col_kx1: unsupported dtype %d", dtype);
    return VMASR_EINVAL;
}

}  // namespace
}  // namespace vmasr

using namespace vmasr;

VMASR_EXPORT int vmasr_im2col_kx1(const void *x, void *cols, int64_t N, int32_t H, int32_t C, int32_t k, int32_t stride,
                                  int32_t pad, int64_t rows_out, int32_t dtype, vmasr_stream_t stream) {
    return run(true, x, cols, N, H, C, k, stride, pad, rows_out, dtype, static_cast<hipStream_t>(stream));
}

VMASR_EXPORT int vmasr_im2col_kx1_split(const float *x, void *hi, void *lo, int64_t N, int32_t H, int32_t C, int32_t k,
                                        int32_t stride, int32_t pad, int64_t rows_out, vmasr_stream_t stream) {
    VMASR_REQUIRE(x && hi && lo, VMASR_EINVAL, "im2col_kx1_split: null tensor");
    VMASR_REQUIRE(N > 0 && H > 0 && C > 0 && k > 0 && stride > 0 && pad >= 0 && H + 2 * pad >= k, VMASR_EINVAL,
                  "im2col_kx1_split: bad geometry (N=%ld H=%d C=%d k=%d stride=%d pad=%d)", (long)N, H, C, k, stride, pad);
    VMASR_REQUIRE(N <= 0x7fffffff, VMASR_EINVAL, "im2col_kx1_split: too many sequences");
    VMASR_REQUIRE(C % 4 == 0 && aligned_to(x, 16) && aligned_to(hi, 8) && aligned_to(lo, 8), VMASR_EINVAL,
                  "im2col_kx1_split: needs C %% 4 == 0 and aligned operands");
    const int H1 = (H + 2 * pad - k) / stride + 1;
    VMASR_REQUIRE(rows_out == 0 || rows_out >= N * H1, VMASR_EINVAL, "im2col_kx1_split: rows_out smaller than N*H1");
    const ColGeom g{(int)N, H, C, k, stride, pad, H1, rows_out ? (long)rows_out : (long)N * H1};
    const long total = g.rows_out * k * (C / 4);
    const int blocks = (int)std::min<long>((total + 255) / 256, 256L * 64);
    hipStream_t st = static_cast<hipStream_t>(stream);
    const double bytes = (double)N * H * C * 4 + (double)g.rows_out * k * C * 4;
    VMASR_LAUNCH(VMASR_K_IM2COL, bytes, im2col_split_kernel, dim3(blocks), dim3(256), 0, st, x, static_cast<bf16_t *>(hi),
                 static_cast<bf16_t *>(lo), g);
    return check_launch("im2col_kx1_split");
}

VMASR_EXPORT int vmasr_col2im_kx1(const void *dcols, void *dx, int64_t N, int32_t H, int32_t C, int32_t k, int32_t stride,
                                  int32_t pad, int32_t dtype, vmasr_stream_t stream) {
    return run(false, dcols, dx, N, H, C, k, stride, pad, 0, dtype, static_cast<hipStream_t>(stream));
}

static int im2col_split_multi_impl(const float *const *xs, const int64_t *Ns, const int32_t *Hs, int32_t n, void *hi,
                                   void *lo, int32_t C, int32_t k, int32_t stride, int32_t pad, int64_t rows_out, int cat3,
                                   vmasr_stream_t stream);

VMASR_EXPORT int vmasr_im2col_kx1_split_multi(const float *const *xs, const int64_t *Ns, const int32_t *Hs, int32_t n, void *hi,
                                              void *lo, int32_t C, int32_t k, int32_t stride, int32_t pad, int64_t rows_out,
                                              vmasr_stream_t stream) {
    return im2col_split_multi_impl(xs, Ns, Hs, n, hi, lo, C, k, stride, pad, rows_out, 0, stream);
}

VMASR_EXPORT int vmasr_im2col_kx1_split3_multi(const float *const *xs, const int64_t *Ns, const int32_t *Hs, int32_t n, void *cat3,
                                               int32_t C, int32_t k, int32_t stride, int32_t pad, int64_t rows_out,
                                               vmasr_stream_t stream) {
    return im2col_split_multi_impl(xs, Ns, Hs, n, cat3, cat3, C, k, stride, pad, rows_out, 1, stream);
}

static int im2col_split_multi_impl(const float *const *xs, const int64_t *Ns, const int32_t *Hs, int32_t n, void *hi,
                                   void *lo, int32_t C, int32_t k, int32_t stride, int32_t pad, int64_t rows_out, int cat3,
                                   vmasr_stream_t stream) {
    VMASR_REQUIRE(xs && Ns && Hs && hi && lo, VMASR_EINVAL, "im2col_kx1_split_multi: null argument");
    VMASR_REQUIRE(n > 0 && n <= kMaxSlots, VMASR_EINVAL, "im2col_kx1_split_multi: 1..%d slots (got %d)", kMaxSlots, n);
    VMASR_REQUIRE(C > 0 && C % 4 == 0 && k > 0 && stride > 0 && pad >= 0 && rows_out > 0, VMASR_EINVAL,
                  "im2col_kx1_split_multi: bad geometry (C=%d k=%d stride=%d pad=%d rows_out=%ld)", C, k, stride, pad, (long)rows_out);
    VMASR_REQUIRE(aligned_to(hi, 8) && aligned_to(lo, 8) && ((size_t)rows_out * k * C) % 4 == 0, VMASR_EINVAL,
                  "im2col_kx1_split_multi: unaligned operands");
    SlotTable t{};
    double bytes = 0;
    for (int s = 0; s < n; ++s) {
        VMASR_REQUIRE(xs[s] && aligned_to(xs[s], 16), VMASR_EINVAL, "im2col_kx1_split_multi: slot %d null or unaligned", s);
        VMASR_REQUIRE(Ns[s] > 0 && Ns[s] <= 0x7fffffff && Hs[s] > 0 && Hs[s] + 2 * pad >= k, VMASR_EINVAL,
                      "im2col_kx1_split_multi: slot %d bad geometry (N=%ld H=%d)", s, (long)Ns[s], Hs[s]);
        const int H1 = (Hs[s] + 2 * pad - k) / stride + 1;
        VMASR_REQUIRE(rows_out >= Ns[s] * H1, VMASR_EINVAL, "im2col_kx1_split_multi: rows_out smaller than N*H1 of slot %d", s);
        t.src[s] = xs[s];
        t.N[s] = (int)Ns[s];
        t.H[s] = Hs[s];
        t.H1[s] = H1;
        bytes += (double)Ns[s] * Hs[s] * C * 4 + (double)rows_out * k * C * (cat3 ? 6 : 4);
    }
    const ColGeom g{0, 0, C, k, stride, pad, 0, (long)rows_out};
    const long total = (long)rows_out * k * (C / 4);
    const int blocks = (int)std::min<long>((total + 255) / 256, 256L * 16);
    hipStream_t st = static_cast<hipStream_t>(stream);
    VMASR_LAUNCH(VMASR_K_IM2COL, bytes, im2col_split_multi_kernel, dim3(blocks, n), dim3(256), 0, st, t, static_cast<bf16_t *>(hi),
                 static_cast<bf16_t *>(lo), g, cat3);
    return check_launch("im2col_kx1_split_multi");
}

static int col2im_multi_impl(const void *dcols, void *const *dxs, const int64_t *Ns, const int32_t *Hs, int32_t n, int32_t C, int32_t k,
                             int32_t stride, int32_t pad, int64_t rows, int64_t dx_rows, int32_t dtype, vmasr_stream_t stream) {
    VMASR_REQUIRE(dcols && dxs && Ns && Hs, VMASR_EINVAL, "col2im_kx1_multi: null argument");
    VMASR_REQUIRE(n > 0 && n <= kMaxSlots, VMASR_EINVAL, "col2im_kx1_multi: 1..%d slots (got %d)", kMaxSlots, n);
    VMASR_REQUIRE(C > 0 && k > 0 && stride > 0 && pad >= 0 && rows > 0, VMASR_EINVAL,
                  "col2im_kx1_multi: bad geometry (C=%d k=%d stride=%d pad=%d rows=%ld)", C, k, stride, pad, (long)rows);
    const size_t esz = dtype == VMASR_F32 ? 4 : 2;
    SlotTable t{};
    bool vec_ok = aligned_to(dcols, 16) && ((size_t)rows * k * C * esz) % 16 == 0;
    long max_total = 0;
    double bytes = 0;
    bool any = false;
    for (int s = 0; s < n; ++s) {
        VMASR_REQUIRE(Ns[s] > 0 && Ns[s] <= 0x7fffffff && Hs[s] > 0 && Hs[s] + 2 * pad >= k, VMASR_EINVAL,
                      "col2im_kx1_multi: slot %d bad geometry (N=%ld H=%d)", s, (long)Ns[s], Hs[s]);
        const int H1 = (Hs[s] + 2 * pad - k) / stride + 1;
        VMASR_REQUIRE(rows >= Ns[s] * H1, VMASR_EINVAL, "col2im_kx1_multi: rows smaller than N*H1 of slot %d", s);
        t.dst[s] = dxs[s];
        t.rows[s] = dx_rows;
        VMASR_REQUIRE(dx_rows == 0 || dx_rows >= Ns[s] * Hs[s], VMASR_EINVAL, "col2im_kx1_multi: destination slot %d has fewer rows than N*H", s);
        t.N[s] = (int)Ns[s];
        t.H[s] = Hs[s];
        t.H1[s] = H1;
        if (!dxs[s]) continue;
        any = true;
        vec_ok = vec_ok && aligned_to(dxs[s], 16);
        max_total = std::max(max_total, (dx_rows ? (long)dx_rows : (long)Ns[s] * Hs[s]) * C);
        bytes += ((double)Ns[s] * Hs[s] * C + (double)Ns[s] * H1 * k * C) * esz;
    }
    if (!any) return VMASR_OK;
    const ColGeom g{0, 0, C, k, stride, pad, 0, (long)rows};
    hipStream_t st = static_cast<hipStream_t>(stream);
    switch (dtype) {
        case VMASR_F32: return launch_col2im_multi<float>(dcols, t, n, max_total, g, vec_ok, bytes, st);
        case VMASR_F16: return launch_col2im_multi<f16_t>(dcols, t, n, max_total, g, vec_ok, bytes, st);
        case VMASR_BF16: return launch_col2im_multi<bf16_t>(dcols, t, n, max_total, g, vec_ok, bytes, st);
    }
    set_error("col2im_kx1_multi: unsupported dtype %d", dtype);
    return VMASR_EINVAL;
}


VMASR_EXPORT int vmasr_col2im_kx1_multi(const void *dcols, void *const *dxs, const int64_t *Ns, const int32_t *Hs, int32_t n,
                                        int32_t C, int32_t k, int32_t stride, int32_t pad, int64_t rows, int32_t dtype,
                                        vmasr_stream_t stream) {
    return col2im_multi_impl(dcols, dxs, Ns, Hs, n, C, k, stride, pad, rows, 0, dtype, stream);
}

// the same into ONE stacked destination dx (n, dx_rows, C): slot s = the gradient of its N_s * H_s rows, zeros below
VMASR_EXPORT int vmasr_col2im_kx1_stacked(const void *dcols, void *dx, const int64_t *Ns, const int32_t *Hs, int32_t n, int32_t C, int32_t k,
                                          int32_t stride, int32_t pad, int64_t rows, int64_t dx_rows, int32_t dtype, vmasr_stream_t stream) {
    VMASR_REQUIRE(dx && n > 0 && n <= kMaxSlots && dx_rows > 0 && C > 0, VMASR_EINVAL, "col2im_kx1_stacked: bad arguments");
    void *ptrs[kMaxSlots];
    const size_t esz = dtype == VMASR_F32 ? 4 : 2;
    for (int s = 0; s < n; ++s) ptrs[s] = static_cast<char *>(dx) + (size_t)s * dx_rows * C * esz;
    return col2im_multi_impl(dcols, ptrs, Ns, Hs, n, C, k, stride, pad, rows, dx_rows, dtype, stream);
}

VMASR_EXPORT int vmasr_stack_rows(const void *const *srcs, const int64_t *Ms, int32_t n, void *full, int64_t rows, int64_t row_bytes,
                                  vmasr_stream_t stream) {
    VMASR_REQUIRE(srcs && Ms && full, VMASR_EINVAL, "stack_rows: null argument");
    VMASR_REQUIRE(n > 0 && n <= kMaxSlots, VMASR_EINVAL, "stack_rows: 1..%d slots (got %d)", kMaxSlots, n);
    VMASR_REQUIRE(rows > 0 && row_bytes > 0, VMASR_EINVAL, "stack_rows: bad shape (rows=%ld row_bytes=%ld)", (long)rows, (long)row_bytes);
    SlotTable t{};
    const long slot_bytes = (long)rows * row_bytes;
    bool vec = aligned_to(full, 16) && slot_bytes % 16 == 0;
    double bytes = (double)n * slot_bytes;
    for (int s = 0; s < n; ++s) {
        VMASR_REQUIRE(Ms[s] >= 0 && Ms[s] <= rows, VMASR_EINVAL, "stack_rows: slot %d has %ld rows of %ld", s, (long)Ms[s], (long)rows);
        t.src[s] = srcs[s];
        t.rows[s] = (long)Ms[s] * row_bytes;
        if (srcs[s]) {
            vec = vec && aligned_to(srcs[s], 16) && t.rows[s] % 16 == 0;
            bytes += (double)t.rows[s];
        }
    }
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (vec) {
        for (int s = 0; s < n; ++s) t.rows[s] /= 16;
        const long units = slot_bytes / 16;
        const int blocks = (int)std::min<long>((units + 255) / 256, 256L * 16);
        VMASR_LAUNCH(VMASR_K_STACK_ROWS, bytes, stack_rows_kernel<uint4>, dim3(blocks, n), dim3(256), 0, st, t, static_cast<uint4 *>(full), units);
    } else {
        const int blocks = (int)std::min<long>((slot_bytes + 255) / 256, 256L * 16);
        VMASR_LAUNCH(VMASR_K_STACK_ROWS, bytes, stack_rows_kernel<unsigned char>, dim3(blocks, n), dim3(256), 0, st, t,
                     static_cast<unsigned char *>(full), slot_bytes);
    }
    return check_launch("stack_rows");
}

// ---- 2-D im2col straight into GEMM ROWS (the generator's patch embedding: model/model.py:603-633, two 3x3 stride-2 convolutions) -------------
// F.unfold writes (B, C kh kw, Ho Wo); the GEMM wants pixels as rows, (B Ho Wo, C kh kw), so the host path paid unfold + a transposing copy
// (+ a cast under autocast) per convolution and the mirror image in the backward.  Here: one gather pass x (B, C, H, W; any strides) ->
// rows in the GEMM's dtype, column order (c, i, j) = weight.flatten(1)'s; and its adjoint as a gather over the input pixels (each sums the
// <= ceil(kh/sh) ceil(kw/sw) row entries that read it: no atomics).
namespace vmasr {
namespace {

struct I2dGeom {
    int B, C, H, W, kh, kw, sh, sw, ph, pw, Ho, Wo;
    long xs_b, xs_c, xs_h, xs_w;       // element strides of x / dx
};

template <typename TI, typename TO>
__global__ __launch_bounds__(256) void im2col2d_rows_kernel(const TI *__restrict__ x, TO *__restrict__ cols, const I2dGeom g, const long total) {
    const int K = g.C * g.kh * g.kw;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const long r = idx / K;
        const int kk = (int)(idx - r * K);
        const int c = kk / (g.kh * g.kw), ij = kk - c * (g.kh * g.kw), i = ij / g.kw, j = ij - i * g.kw;
        const int wo = (int)(r % g.Wo);
        const long t = r / g.Wo;
        const int ho = (int)(t % g.Ho), b = (int)(t / g.Ho);
        const int h = ho * g.sh + i - g.ph, w = wo * g.sw + j - g.pw;
        float v = 0.f;
        if (h >= 0 && h < g.H && w >= 0 && w < g.W) v = (float)x[b * g.xs_b + c * g.xs_c + h * g.xs_h + w * g.xs_w];
        cols[idx] = (TO)v;
    }
}

template <typename TI, typename TO>
__global__ __launch_bounds__(256) void col2im2d_rows_kernel(const TI *__restrict__ gcols, TO *__restrict__ dx, const I2dGeom g, const long total) {
    const int K = g.C * g.kh * g.kw;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        // idx runs over (b, h, w, c) with c fastest: neighbouring threads read neighbouring row entries' column blocks
        const int c = (int)(idx % g.C);
        long t = idx / g.C;
        const int w = (int)(t % g.W);
        t /= g.W;
        const int h = (int)(t % g.H), b = (int)(t / g.H);
        float s = 0.f;
        for (int i = 0; i < g.kh; ++i) {
            const int hn = h + g.ph - i;
            if (hn < 0 || hn % g.sh) continue;
            const int ho = hn / g.sh;
            if (ho >= g.Ho) continue;
            for (int j = 0; j < g.kw; ++j) {
                const int wn = w + g.pw - j;
                if (wn < 0 || wn % g.sw) continue;
                const int wo = wn / g.sw;
                if (wo >= g.Wo) continue;
                s += (float)gcols[(((long)b * g.Ho + ho) * g.Wo + wo) * K + (c * g.kh + i) * g.kw + j];
            }
        }
        dx[b * g.xs_b + c * g.xs_c + h * g.xs_h + w * g.xs_w] = (TO)s;
    }
}

template <typename F>
int i2d_dispatch(int a, int b, F &&f) {        // (VMASR_F32 | VMASR_BF16) x (VMASR_F32 | VMASR_BF16)
    if (a == VMASR_F32 && b == VMASR_F32) return f((const float *)nullptr, (float *)nullptr);
    if (a == VMASR_F32 && b == VMASR_BF16) return f((const float *)nullptr, (bf16_t *)nullptr);
    if (a == VMASR_BF16 && b == VMASR_F32) return f((const bf16_t *)nullptr, (float *)nullptr);
    if (a == VMASR_BF16 && b == VMASR_BF16) return f((const bf16_t *)nullptr, (bf16_t *)nullptr);
    return VMASR_EINVAL;
}

}  // namespace
}  // namespace vmasr

static int i2d_geom(vmasr::I2dGeom &g, int32_t B, int32_t C, int32_t H, int32_t W, int32_t kh, int32_t kw, int32_t sh, int32_t sw, int32_t ph, int32_t pw,
                    const int64_t *strides) {
    if (B <= 0 || C <= 0 || H <= 0 || W <= 0 || kh <= 0 || kw <= 0 || sh <= 0 || sw <= 0 || ph < 0 || pw < 0 || !strides) return 0;
    g = {B, C, H, W, kh, kw, sh, sw, ph, pw, (H + 2 * ph - kh) / sh + 1, (W + 2 * pw - kw) / sw + 1, strides[0], strides[1], strides[2], strides[3]};
    return g.Ho > 0 && g.Wo > 0;
}

VMASR_EXPORT int vmasr_im2col2d_rows(const void *x, void *cols, int32_t B, int32_t C, int32_t H, int32_t W, int32_t kh, int32_t kw, int32_t sh, int32_t sw,
                                     int32_t ph, int32_t pw, const int64_t *x_strides, int32_t x_dtype, int32_t cols_dtype, vmasr_stream_t stream) {
    vmasr::I2dGeom g;
    VMASR_REQUIRE(x && cols && i2d_geom(g, B, C, H, W, kh, kw, sh, sw, ph, pw, x_strides), VMASR_EINVAL, "im2col2d_rows: bad argument");
    const long total = (long)B * g.Ho * g.Wo * C * kh * kw;
    const int blocks = (int)std::min<long>((total + 255) / 256, 256L * 32);
    hipStream_t st = static_cast<hipStream_t>(stream);
    const double bytes = (double)B * C * H * W * (x_dtype == VMASR_F32 ? 4 : 2) + (double)total * (cols_dtype == VMASR_F32 ? 4 : 2);
    return vmasr::i2d_dispatch(x_dtype, cols_dtype, [&](auto *pi, auto *po) {
        using TI = std::remove_const_t<std::remove_pointer_t<decltype(pi)>>;
        using TO = std::remove_pointer_t<decltype(po)>;
        VMASR_LAUNCH(VMASR_K_IM2COL, bytes, (vmasr::im2col2d_rows_kernel<TI, TO>), dim3(blocks), dim3(256), 0, st, static_cast<const TI *>(x),
                     static_cast<TO *>(cols), g, total);
        return vmasr::check_launch("im2col2d_rows");
    });
}

VMASR_EXPORT int vmasr_col2im2d_rows(const void *gcols, void *dx, int32_t B, int32_t C, int32_t H, int32_t W, int32_t kh, int32_t kw, int32_t sh, int32_t sw,
                                     int32_t ph, int32_t pw, const int64_t *dx_strides, int32_t cols_dtype, int32_t dx_dtype, vmasr_stream_t stream) {
    vmasr::I2dGeom g;
    VMASR_REQUIRE(gcols && dx && i2d_geom(g, B, C, H, W, kh, kw, sh, sw, ph, pw, dx_strides), VMASR_EINVAL, "col2im2d_rows: bad argument");
    const long total = (long)B * C * H * W;
    const int blocks = (int)std::min<long>((total + 255) / 256, 256L * 32);
    hipStream_t st = static_cast<hipStream_t>(stream);
    const double bytes = (double)total * (dx_dtype == VMASR_F32 ? 4 : 2) + (double)B * g.Ho * g.Wo * C * kh * kw * (cols_dtype == VMASR_F32 ? 4 : 2);
    return vmasr::i2d_dispatch(cols_dtype, dx_dtype, [&](auto *pi, auto *po) {
        using TI = std::remove_const_t<std::remove_pointer_t<decltype(pi)>>;
        using TO = std::remove_pointer_t<decltype(po)>;
        VMASR_LAUNCH(VMASR_K_COL2IM, bytes, (vmasr::col2im2d_rows_kernel<TI, TO>), dim3(blocks), dim3(256), 0, st, static_cast<const TI *>(gcols),
                     static_cast<TO *>(dx), g, total);
        return vmasr::check_launch("col2im2d_rows");
    });
}
