// spectral.hip — the matrix-vector products of spectral-norm power iteration, for gfx950.
//
// torch.nn.utils.parametrizations.spectral_norm (what the reference's MPD gets through the
// inverted ternary at model/discriminator.py:37) runs u <- normalize(W v), v <- normalize(W^T u)
// on every training forward, for 30 conv weights of up to 1024 x 5120 fp32.  As hipBLASLt GEMMs
// with one column these take 34 us each (0.6 TB/s) and 8 launches per iteration; `torch.mv`
// costs 4 ms of HOST time per call on ROCm 7.2.  Here: HBM-bound kernels.
//
//   gemv_rows : t[r] = sum_c W[r,c] v[c]      one wave per row, 16-B loads
//   gemv_cols : s[c] += sum_{r in chunk} W[r,c] u[r]   thread per 4 columns, 32-row chunks, atomics
//   l2_normalize : y = x / max(||x||, eps)    one workgroup
#include "common.h"

namespace vmasr {
namespace {

__global__ __launch_bounds__(256) void gemv_rows_kernel(const float *__restrict__ W, const float *__restrict__ v,
                                                        float *__restrict__ t, const int R, const int C) {
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= R) return;
    const float *row = W + (size_t)r * C;
    float acc = 0.f;
    if ((C & 3) == 0) {
        for (int c = lane * 4; c < C; c += 256) {
            const float4 a = *reinterpret_cast<const float4 *>(row + c);
            const float4 b = *reinterpret_cast<const float4 *>(v + c);
            acc += (a.x * b.x + a.y * b.y) + (a.z * b.z + a.w * b.w);
        }
    } else {
        for (int c = lane; c < C; c += 64) acc = fmaf(row[c], v[c], acc);
    }
    acc = wave_sum(acc);
    if (lane == 0) t[r] = acc;
}

constexpr int kRowChunk = 32;

__global__ __launch_bounds__(256) void gemv_cols_kernel(const float *__restrict__ W, const float *__restrict__ u,
                                                        float *__restrict__ s, const int R, const int C, unsigned *det) {
    const int c0 = (blockIdx.x * 256 + threadIdx.x) * 4;
    const int r0 = blockIdx.y * kRowChunk;
    int r1 = min(R, r0 + kRowChunk);
    float a[4] = {0.f, 0.f, 0.f, 0.f};
    if (c0 >= C) {
        r1 = r0;                                          // (no early return: every thread reaches the ordered tail below)
    } else if ((C & 3) == 0) {
        for (int r = r0; r < r1; ++r) {
            const float4 w = *reinterpret_cast<const float4 *>(W + (size_t)r * C + c0);
            const float ur = u[r];
            a[0] = fmaf(w.x, ur, a[0]); a[1] = fmaf(w.y, ur, a[1]); a[2] = fmaf(w.z, ur, a[2]); a[3] = fmaf(w.w, ur, a[3]);
        }
    } else {
        for (int r = r0; r < r1; ++r) {
            const float ur = u[r];
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (c0 + i < C) a[i] = fmaf(W[(size_t)r * C + c0 + i], ur, a[i]);
        }
    }
    det_enter(det);                                       // deterministic mode: workgroup order (common.h)
#pragma unroll
    for (int i = 0; i < 4; ++i)
        if (c0 + i < C) atomicAdd(s + c0 + i, a[i]);
    det_leave(det);
}

// y = x / max(||x||_2, eps); also zeroes `clear` (n_clear floats) for the next gemv_cols accumulation
__global__ __launch_bounds__(1024) void l2_normalize_kernel(const float *__restrict__ x, float *__restrict__ y, const int n,
                                                            const float eps, float *__restrict__ clear, const int n_clear) {
    __shared__ float s_part[16];
    float acc = 0.f;
    for (int i = threadIdx.x; i < n; i += blockDim.x) acc = fmaf(x[i], x[i], acc);
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = acc;
    __syncthreads();
    float tot = 0.f;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) tot += s_part[w];
    const float inv = 1.f / fmaxf(sqrtf(tot), eps);
    for (int i = threadIdx.x; i < n; i += blockDim.x) y[i] = x[i] * inv;
    for (int i = threadIdx.x; i < n_clear; i += blockDim.x) clear[i] = 0.f;
}

// ---- all matrices of a model in one launch per phase -------------------------------------------
// 30 spectrally normalised weights x 3 iterations x 4 phases = 360 launches of ~6 us per training
// step when run one matrix at a time; here the four phases each cover every matrix (12 launches).
// A workgroup finds its matrix by scanning the (<= 64 entry) prefix table in the descriptor array.

__device__ __forceinline__ int find_item(const vmasr_spectral_item *__restrict__ it, const int n, const int bid,
                                         const bool cols) {
    int m = 0;
    for (int i = 1; i < n; ++i) m = (bid >= (cols ? it[i].col_tile_start : it[i].row_block_start)) ? i : m;
    return m;
}

__global__ __launch_bounds__(256) void gemv_rows_multi_kernel(const vmasr_spectral_item *__restrict__ items, const int n) {
    const int m = find_item(items, n, blockIdx.x, false);
    const vmasr_spectral_item it = items[m];
    const int lane = threadIdx.x & 63;
    const int r = (blockIdx.x - it.row_block_start) * 4 + (threadIdx.x >> 6);
    if (r >= it.R) return;
    const float *row = it.W + (size_t)r * it.C;
    const float *v = it.v;
    float acc = 0.f;
    if ((it.C & 3) == 0) {
        for (int c = lane * 4; c < it.C; c += 256) {
            const float4 a = *reinterpret_cast<const float4 *>(row + c);
            const float4 b = *reinterpret_cast<const float4 *>(v + c);
            acc += (a.x * b.x + a.y * b.y) + (a.z * b.z + a.w * b.w);
        }
    } else {
        for (int c = lane; c < it.C; c += 64) acc = fmaf(row[c], v[c], acc);
    }
    acc = wave_sum(acc);
    if (lane == 0) it.t[r] = acc;
}

__global__ __launch_bounds__(256) void gemv_cols_multi_kernel(const vmasr_spectral_item *__restrict__ items, const int n, unsigned *det) {
    const int m = find_item(items, n, blockIdx.x, true);
    const vmasr_spectral_item it = items[m];
    const int local = blockIdx.x - it.col_tile_start;
    const int ctiles = (it.C + 1023) / 1024;
    const int c0 = ((local % ctiles) * 256 + threadIdx.x) * 4;
    const int r0 = (local / ctiles) * kRowChunk;
    int r1 = min(it.R, r0 + kRowChunk);
    const float *W = it.W, *u = it.u;
    const int C = it.C;
    float a[4] = {0.f, 0.f, 0.f, 0.f};
    if (c0 >= C) {
        r1 = r0;                                          // (no early return: every thread reaches the ordered tail below)
    } else if ((C & 3) == 0) {
        for (int r = r0; r < r1; ++r) {
            const float4 w = *reinterpret_cast<const float4 *>(W + (size_t)r * C + c0);
            const float ur = u[r];
            a[0] = fmaf(w.x, ur, a[0]); a[1] = fmaf(w.y, ur, a[1]); a[2] = fmaf(w.z, ur, a[2]); a[3] = fmaf(w.w, ur, a[3]);
        }
    } else {
        for (int r = r0; r < r1; ++r) {
            const float ur = u[r];
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (c0 + i < C) a[i] = fmaf(W[(size_t)r * C + c0 + i], ur, a[i]);
        }
    }
    det_enter(det);
#pragma unroll
    for (int i = 0; i < 4; ++i)
        if (c0 + i < C) atomicAdd(it.s + c0 + i, a[i]);
    det_leave(det);
}

// one workgroup per matrix: COLS ? (v = normalize(s), s = 0) : (u = normalize(t))
template <bool COLS>
__global__ __launch_bounds__(1024) void l2_normalize_multi_kernel(const vmasr_spectral_item *__restrict__ items, const float eps) {
    const vmasr_spectral_item it = items[blockIdx.x];
    float *x = COLS ? it.s : it.t;
    float *y = COLS ? it.v : it.u;
    const int len = COLS ? it.C : it.R;
    __shared__ float s_part[16];
    float acc = 0.f;
    for (int i = threadIdx.x; i < len; i += blockDim.x) acc = fmaf(x[i], x[i], acc);
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = acc;
    __syncthreads();
    float tot = 0.f;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) tot += s_part[w];
    const float inv = 1.f / fmaxf(sqrtf(tot), eps);
    for (int i = threadIdx.x; i < len; i += blockDim.x) {
        y[i] = x[i] * inv;
        if (COLS) x[i] = 0.f;
    }
}

// sigma_m = u_m . t_m  (t = W v just computed): one workgroup per matrix
__global__ __launch_bounds__(256) void sigma_multi_kernel(const vmasr_spectral_item *__restrict__ items, float *__restrict__ sigma) {
    const vmasr_spectral_item it = items[blockIdx.x];
    __shared__ float s_part[4];
    float acc = 0.f;
    for (int i = threadIdx.x; i < it.R; i += blockDim.x) acc = fmaf(it.u[i], it.t[i], acc);
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) sigma[blockIdx.x] = (s_part[0] + s_part[1]) + (s_part[2] + s_part[3]);
}

}  // namespace
}  // namespace vmasr

using namespace vmasr;

// `items`: n descriptors in DEVICE memory (W, u, v, scratch t (R floats) and s (C floats, zero on entry and
// on exit), prefix sums of 4-row blocks and of (1024-column x 32-row) tiles); total_* = the two grid sizes;
// weight_bytes = sum of R*C*4 (the algorithmic bytes of one matrix-vector phase, for the profiler);
// sigma: null, or n floats that receive u^T W v of the final vectors.
VMASR_EXPORT int vmasr_spectral_power_iter_batched(const vmasr_spectral_item *items, int32_t n, int32_t total_row_blocks,
                                                   int32_t total_col_tiles, int64_t weight_bytes, int32_t n_iter,
                                                   float eps, float *sigma, vmasr_stream_t stream) {
    VMASR_REQUIRE(items && n > 0 && n <= 64, VMASR_EINVAL, "spectral_power_iter_batched: 1..64 matrices");
    VMASR_REQUIRE(total_row_blocks > 0 && total_col_tiles > 0 && n_iter >= 0, VMASR_EINVAL, "spectral_power_iter_batched: bad size");
    hipStream_t st = static_cast<hipStream_t>(stream);
    for (int it = 0; it < n_iter; ++it) {
        VMASR_LAUNCH(VMASR_K_SPECTRAL, (double)weight_bytes, gemv_rows_multi_kernel, dim3(total_row_blocks), dim3(256), 0, st, items, n);
        VMASR_LAUNCH(VMASR_K_SPECTRAL, 0.0, (l2_normalize_multi_kernel<false>), dim3(n), dim3(1024), 0, st, items, eps);
        VMASR_LAUNCH(VMASR_K_SPECTRAL, (double)weight_bytes, gemv_cols_multi_kernel, dim3(total_col_tiles), dim3(256), 0, st, items, n, det_ticket(VMASR_K_SPECTRAL));
        VMASR_LAUNCH(VMASR_K_SPECTRAL, 0.0, (l2_normalize_multi_kernel<true>), dim3(n), dim3(1024), 0, st, items, eps);
    }
    if (sigma) {  // sigma_m = u^T W v with the final u, v (what spectral_norm divides the weight by)
        VMASR_LAUNCH(VMASR_K_SPECTRAL, (double)weight_bytes, gemv_rows_multi_kernel, dim3(total_row_blocks), dim3(256), 0, st, items, n);
        VMASR_LAUNCH(VMASR_K_SPECTRAL, 0.0, sigma_multi_kernel, dim3(n), dim3(256), 0, st, items, sigma);
    }
    return check_launch("spectral_power_iter_batched");
}

// n_iter rounds of  u <- normalize(W v);  v <- normalize(W^T u)  in place.  W (R,C) fp32 row-major,
// u (R), v (C); ws: (R + C) floats of scratch.
VMASR_EXPORT int vmasr_spectral_power_iter(const float *W, float *u, float *v, float *ws, int32_t R, int32_t C,
                                           int32_t n_iter, float eps, vmasr_stream_t stream) {
    VMASR_REQUIRE(W && u && v && ws, VMASR_EINVAL, "spectral_power_iter: null tensor");
    VMASR_REQUIRE(R > 0 && C > 0 && n_iter >= 0, VMASR_EINVAL, "spectral_power_iter: bad size");
    VMASR_REQUIRE(aligned_to(W, 16) && aligned_to(v, 16), VMASR_EALIGN, "spectral_power_iter: 16-byte alignment required");
    hipStream_t st = static_cast<hipStream_t>(stream);
    float *t = ws, *s = ws + R;
    const double wb = (double)R * C * 4;
    if (n_iter > 0) (void)hipMemsetAsync(s, 0, (size_t)C * sizeof(float), st);
    for (int it = 0; it < n_iter; ++it) {
        VMASR_LAUNCH(VMASR_K_SPECTRAL, wb, gemv_rows_kernel, dim3((R + 3) / 4), dim3(256), 0, st, W, v, t, R, C);
        VMASR_LAUNCH(VMASR_K_SPECTRAL, R * 8.0, l2_normalize_kernel, dim3(1), dim3(1024), 0, st, t, u, R, eps, (float *)nullptr, 0);
        VMASR_LAUNCH(VMASR_K_SPECTRAL, wb, gemv_cols_kernel, dim3((C + 1023) / 1024, (R + kRowChunk - 1) / kRowChunk), dim3(256), 0,
                     st, W, u, s, R, C, det_ticket(VMASR_K_SPECTRAL));
        // normalise into v and clear the accumulator for the next round
        VMASR_LAUNCH(VMASR_K_SPECTRAL, C * 8.0, l2_normalize_kernel, dim3(1), dim3(1024), 0, st, s, v, C, eps, s, C);
    }
    return check_launch("spectral_power_iter");
}

// ---- spectrally normalised, stacked GEMM weights of one discriminator layer (all n period discriminators) -----------
// model/discriminator.py:26-45 wraps every convolution in spectral_norm: W / sigma, sigma = u^T W v with u, v from the
// power iteration (constants for autograd).  The batched discriminator pass wants the n weights of a layer as ONE
// (n, N, K) operand in (tap, channel) column order.  As ATen ops that is n divisions, a stack and — in the backward —
// per weight a dot product, an outer-product update and a division (25+ launches per layer).  Here:
//   fwd  : out[s, o, j*Cin + c] = W_s[o, c, j] / sigma_s                                   one launch per layer
//   dot  : partial sums of <dW_s, out_s> (fp64 per workgroup)                              one launch
//   bwd  : gW_s[o, c, j] = (dW[s, o, j*Cin + c] - <dW_s, out_s> u_s[o] v_s[c*k + j]) / sigma_s   one launch
// One workgroup per (s, o) row; the (Cin x k) <-> (k x Cin) permutation of a row goes through LDS so that both the
// read and the write are contiguous.  Pure streaming: 8 B per element forward, 8 + 8 B backward.
namespace vmasr {
namespace {

constexpr int kSnMaxSlots = 8;
constexpr int kSnDotBlocks = 64;

struct SnSlots {
    const float *w[kSnMaxSlots];       // fwd: original weights (N, Cin, k); bwd: unused
    float *gw[kSnMaxSlots];            // bwd: gradient wrt the original weights
    const float *sigma[kSnMaxSlots];   // device scalars
    const float *u[kSnMaxSlots];       // (N)
    const float *v[kSnMaxSlots];       // (Cin * k)
};

__global__ __launch_bounds__(256) void sn_stack_fwd_kernel(const SnSlots t, float *__restrict__ out, const int N, const int Cin, const int k) {
    extern __shared__ float row[];
    const int s = blockIdx.y, o = blockIdx.x, K = Cin * k;
    const float inv = 1.f / t.sigma[s][0];
    const float *src = t.w[s] + (size_t)o * K;
    for (int i = threadIdx.x; i < K; i += 256) row[i] = src[i] * inv;          // (c, j) order
    __syncthreads();
    float *dst = out + ((size_t)s * N + o) * K;
    for (int e = threadIdx.x; e < K; e += 256) dst[e] = row[(e % Cin) * k + e / Cin];   // (j, c) order
}

__global__ __launch_bounds__(256) void sn_dot_kernel(const float *__restrict__ dW, const float *__restrict__ Wn, double *__restrict__ partials,
                                                     const size_t per_slot) {
    const int s = blockIdx.y;
    const float4 *a = reinterpret_cast<const float4 *>(dW + (size_t)s * per_slot), *b = reinterpret_cast<const float4 *>(Wn + (size_t)s * per_slot);
    const size_t n4 = per_slot / 4;
    double acc = 0.0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        const float4 x = a[i], y = b[i];
        acc += (double)((x.x * y.x + x.y * y.y) + (x.z * y.z + x.w * y.w));
    }
    if (blockIdx.x == 0)
        for (size_t i = n4 * 4 + threadIdx.x; i < per_slot; i += 256) acc += (double)dW[(size_t)s * per_slot + i] * Wn[(size_t)s * per_slot + i];
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
    __shared__ double wsum[4];
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) partials[(size_t)s * gridDim.x + blockIdx.x] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}

__global__ __launch_bounds__(256) void sn_stack_bwd_kernel(const SnSlots t, const float *__restrict__ dW, const double *__restrict__ partials,
                                                           const int N, const int Cin, const int k) {
    extern __shared__ float row[];
    __shared__ float s_dot;
    const int s = blockIdx.y, o = blockIdx.x, K = Cin * k;
    if (threadIdx.x < 64) {
        double d = threadIdx.x < kSnDotBlocks ? partials[(size_t)s * kSnDotBlocks + threadIdx.x] : 0.0;
        for (int off = 32; off > 0; off >>= 1) d += __shfl_down(d, off, 64);
        if (threadIdx.x == 0) s_dot = (float)d;
    }
    const float *src = dW + ((size_t)s * N + o) * K;
    for (int e = threadIdx.x; e < K; e += 256) row[(e % Cin) * k + e / Cin] = src[e];   // (j, c) -> (c, j)
    __syncthreads();
    const float inv = 1.f / t.sigma[s][0], su = s_dot * t.u[s][o];
    const float *v = t.v[s];
    float *dst = t.gw[s] + (size_t)o * K;
    for (int i = threadIdx.x; i < K; i += 256) dst[i] = (row[i] - su * v[i]) * inv;
}

int sn_fill(SnSlots &t, const void *const *w, void *const *gw, const void *const *sigma, const void *const *u, const void *const *v, int n,
            const char *what) {
    VMASR_REQUIRE(n > 0 && n <= kSnMaxSlots, VMASR_EINVAL, "%s: 1..%d slots (got %d)", what, kSnMaxSlots, n);
    for (int s = 0; s < n; ++s) {
        VMASR_REQUIRE(sigma[s] && (!w || w[s]) && (!gw || (gw[s] && u[s] && v[s])), VMASR_EINVAL, "%s: null pointer in slot %d", what, s);
        t.w[s] = w ? static_cast<const float *>(w[s]) : nullptr;
        t.gw[s] = gw ? static_cast<float *>(gw[s]) : nullptr;
        t.sigma[s] = static_cast<const float *>(sigma[s]);
        t.u[s] = u ? static_cast<const float *>(u[s]) : nullptr;
        t.v[s] = v ? static_cast<const float *>(v[s]) : nullptr;
    }
    return 0;
}

}  // namespace
}  // namespace vmasr

using namespace vmasr;

VMASR_EXPORT int32_t vmasr_sn_dot_blocks(void) { return kSnDotBlocks; }

VMASR_EXPORT int vmasr_sn_stack_fwd(const void *const *weights, const void *const *sigmas, int32_t n, float *out, int32_t N, int32_t Cin,
                                    int32_t k, vmasr_stream_t stream) {
    VMASR_REQUIRE(weights && sigmas && out, VMASR_EINVAL, "sn_stack_fwd: null argument");
    VMASR_REQUIRE(N > 0 && N <= 65535 && Cin > 0 && k > 0 && (size_t)Cin * k * 4 <= 60 * 1024, VMASR_EINVAL, "sn_stack_fwd: bad shape");
    SnSlots t{};
    if (int e = sn_fill(t, weights, nullptr, sigmas, nullptr, nullptr, n, "sn_stack_fwd")) return e;
    hipStream_t st = static_cast<hipStream_t>(stream);
    VMASR_LAUNCH(VMASR_K_SPECTRAL, 8.0 * n * (double)N * Cin * k, sn_stack_fwd_kernel, dim3(N, n), dim3(256), (size_t)Cin * k * 4, st, t, out, N,
                 Cin, k);
    return check_launch("sn_stack_fwd");
}

VMASR_EXPORT int vmasr_sn_stack_bwd(const float *dW, const float *Wn, void *const *gws, const void *const *sigmas, const void *const *us,
                                    const void *const *vs, int32_t n, double *partials, int32_t N, int32_t Cin, int32_t k,
                                    vmasr_stream_t stream) {
    VMASR_REQUIRE(dW && Wn && gws && sigmas && us && vs && partials, VMASR_EINVAL, "sn_stack_bwd: null argument");
    VMASR_REQUIRE(N > 0 && N <= 65535 && Cin > 0 && k > 0 && (size_t)Cin * k * 4 <= 60 * 1024, VMASR_EINVAL, "sn_stack_bwd: bad shape");
    VMASR_REQUIRE(aligned_to(dW, 16) && aligned_to(Wn, 16), VMASR_EINVAL, "sn_stack_bwd: dW and Wn must be 16-byte aligned");
    SnSlots t{};
    if (int e = sn_fill(t, nullptr, gws, sigmas, us, vs, n, "sn_stack_bwd")) return e;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const size_t per_slot = (size_t)N * Cin * k;
    VMASR_REQUIRE(per_slot % 4 == 0 || n == 1, VMASR_EINVAL, "sn_stack_bwd: N * Cin * k must be a multiple of 4 for stacked slots");
    VMASR_LAUNCH(VMASR_K_SPECTRAL, 8.0 * n * (double)per_slot, sn_dot_kernel, dim3(kSnDotBlocks, n), dim3(256), 0, st, dW, Wn, partials, per_slot);
    VMASR_LAUNCH(VMASR_K_SPECTRAL, 8.0 * n * (double)per_slot, sn_stack_bwd_kernel, dim3(N, n), dim3(256), (size_t)Cin * k * 4, st, t, dW, partials,
                 N, Cin, k);
    return check_launch("sn_stack_bwd");
}
