// featloss.hip — HiFi-GAN feature-matching loss over the stacked discriminator feature maps, gfx950.
//
// Reference: model/loss.py:227-235 — mean over all feature maps of mean |real - generated|.  The batched
// discriminator pass keeps every layer's maps of all n period discriminators in one (n, rows, N) fp32 tensor whose
// slot s holds valid[s] rows of signal followed by zero padding, so the whole layer's contribution is
//     sum_s scale[s] * sum_{r < valid[s], c} |gen[s, r, c] - real[s, r, c]|,   scale[s] = 1 / (valid[s] * N * n_maps).
// As ATen ops that is sub, abs, mul (mask), sum forward and sgn, mul, mul backward: 7 passes, ~52 B per element, over
// 170 M elements per training step.  Here: one forward pass (r 8 B, w 1 B: the sign, kept for the backward) and one
// backward pass (r 1 B, w 4 B — or, with the gradient arriving from the next layer as addend, r 5 B, w 4 B for the map's WHOLE
// gradient: vmasr_masked_l1_bwd_add).  Pure HBM-bound streaming; the valid part of a slot is one contiguous range in both
// tensors, so the kernels are flat grid-stride loops with 16-byte accesses.
//
// Determinism: per-workgroup partial sums in fp64 land in a workspace that the host side adds up (no atomics).
#include <algorithm>

#include "common.h"

namespace vmasr {
namespace {

constexpr int kMaxSlots = 8;

struct L1Slots {
    long valid_elems[kMaxSlots];   // valid[s] * N
    float scale[kMaxSlots];
};

__device__ __forceinline__ signed char sgn1(float d) { return (signed char)((d > 0.f) - (d < 0.f)); }

// grid (blocks, n); partials[(slot * gridDim.x + block)]
__global__ __launch_bounds__(256) void masked_l1_fwd_kernel(const float *__restrict__ real, const float *__restrict__ gen,
                                                            signed char *__restrict__ sgn, double *__restrict__ partials,
                                                            const L1Slots t, const size_t stride_r, const size_t stride_g) {
    const int s = blockIdx.y;
    const float *__restrict__ r = real + (size_t)s * stride_r;
    const float *__restrict__ g = gen + (size_t)s * stride_g;
    signed char *__restrict__ o = sgn ? sgn + (size_t)s * stride_g : nullptr;
    const long n = t.valid_elems[s], n4 = n / 4;
    double acc = 0.0;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        const float4 a = reinterpret_cast<const float4 *>(g)[i], b = reinterpret_cast<const float4 *>(r)[i];
        const float d0 = a.x - b.x, d1 = a.y - b.y, d2 = a.z - b.z, d3 = a.w - b.w;
        acc += (double)((fabsf(d0) + fabsf(d1)) + (fabsf(d2) + fabsf(d3)));
        if (o) {
            char4 q;
            q.x = sgn1(d0); q.y = sgn1(d1); q.z = sgn1(d2); q.w = sgn1(d3);
            reinterpret_cast<char4 *>(o)[i] = q;
        }
    }
    if (blockIdx.x == 0 && (int)threadIdx.x < (int)(n - n4 * 4)) {   // tail (valid * N not a multiple of 4)
        const long i = n4 * 4 + threadIdx.x;
        const float d = g[i] - r[i];
        acc += (double)fabsf(d);
        if (o) o[i] = sgn1(d);
    }
    // workgroup sum: wave shuffle, then LDS across the 4 waves
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
    __shared__ double wsum[4];
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) partials[(size_t)s * gridDim.x + blockIdx.x] = (wsum[0] + wsum[1] + wsum[2] + wsum[3]) * (double)t.scale[s];
}

// dgen[s, e] = add[s, e] + gout * scale[s] * sgn[s, e] for e < valid_elems[s], add[s, e] on the padding rows (add == NULL: 0);
// grid (blocks, n).  With `add` = the gradient arriving from the next layer this is the whole gradient of the feature map in
// one pass (r 5 B, w 4 B) where a separate loss backward (r 1, w 4) + autograd's sum of the two (r 8, w 4) moved 17 B.
__global__ __launch_bounds__(256) void masked_l1_bwd_kernel(const signed char *__restrict__ sgn, const float *__restrict__ gout,
                                                            const float *__restrict__ add, float *__restrict__ dgen, const L1Slots t,
                                                            const size_t stride_g) {
    const int s = blockIdx.y;
    const signed char *__restrict__ q = sgn + (size_t)s * stride_g;
    const float *__restrict__ a = add ? add + (size_t)s * stride_g : nullptr;
    float *__restrict__ o = dgen + (size_t)s * stride_g;
    const float k = gout[0] * t.scale[s];
    const long n = t.valid_elems[s], total4 = (long)(stride_g / 4);
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += (long)gridDim.x * blockDim.x) {
        float4 v = a ? reinterpret_cast<const float4 *>(a)[i] : make_float4(0.f, 0.f, 0.f, 0.f);
        const long e = i * 4;
        if (e + 3 < n) {
            const char4 c = reinterpret_cast<const char4 *>(q)[i];
            v.x = fmaf(k, (float)c.x, v.x); v.y = fmaf(k, (float)c.y, v.y); v.z = fmaf(k, (float)c.z, v.z); v.w = fmaf(k, (float)c.w, v.w);
        } else if (e < n) {
            float w[4] = {v.x, v.y, v.z, v.w};
            for (int j = 0; j < 4 && e + j < n; ++j) w[j] = fmaf(k, (float)q[e + j], w[j]);
            v = make_float4(w[0], w[1], w[2], w[3]);
        }
        reinterpret_cast<float4 *>(o)[i] = v;
    }
}

int fill_slots(L1Slots &t, const int64_t *valid, const float *scale, int n, int N, int64_t rows_g, int64_t rows_r, const char *what) {
    for (int s = 0; s < n; ++s) {
        if (valid[s] < 0 || valid[s] > rows_g || valid[s] > rows_r) {
            set_error("%s: slot %d has %ld valid rows of %ld / %ld", what, s, (long)valid[s], (long)rows_g, (long)rows_r);
            return VMASR_EINVAL;
        }
        t.valid_elems[s] = (long)valid[s] * N;
        t.scale[s] = scale[s];
    }
    return 0;
}

}  // namespace
}  // namespace vmasr

using namespace vmasr;

VMASR_EXPORT int32_t vmasr_masked_l1_blocks(void) { return 256 * 4; }

VMASR_EXPORT int vmasr_masked_l1_fwd(const float *real, const float *gen, void *sgn, double *partials, const int64_t *valid,
                                     const float *scale, int32_t n, int64_t rows_r, int64_t rows_g, int32_t N, vmasr_stream_t stream) {
    VMASR_REQUIRE(real && gen && partials && valid && scale, VMASR_EINVAL, "masked_l1_fwd: null argument");
    VMASR_REQUIRE(n > 0 && n <= kMaxSlots && rows_r > 0 && rows_g > 0 && N > 0, VMASR_EINVAL, "masked_l1_fwd: bad shape");
    VMASR_REQUIRE(((size_t)rows_r * N) % 4 == 0 && ((size_t)rows_g * N) % 4 == 0 && aligned_to(real, 16) && aligned_to(gen, 16) &&
                      (!sgn || aligned_to(sgn, 4)),
                  VMASR_EINVAL, "masked_l1_fwd: slots must be 16-byte aligned (rows * N %% 4 == 0)");
    L1Slots t{};
    if (int e = fill_slots(t, valid, scale, n, N, rows_g, rows_r, "masked_l1_fwd")) return e;
    double bytes = 0;
    for (int s = 0; s < n; ++s) bytes += (double)t.valid_elems[s] * (sgn ? 9.0 : 8.0);
    hipStream_t st = static_cast<hipStream_t>(stream);
    VMASR_LAUNCH(VMASR_K_FEAT_L1, bytes, masked_l1_fwd_kernel, dim3(vmasr_masked_l1_blocks(), n), dim3(256), 0, st, real, gen,
                 static_cast<signed char *>(sgn), partials, t, (size_t)rows_r * N, (size_t)rows_g * N);
    return check_launch("masked_l1_fwd");
}

VMASR_EXPORT int vmasr_masked_l1_bwd_add(const void *sgn, const float *gout, const float *add, float *dgen, const int64_t *valid,
                                         const float *scale, int32_t n, int64_t rows_g, int32_t N, vmasr_stream_t stream) {
    VMASR_REQUIRE(sgn && gout && dgen && valid && scale, VMASR_EINVAL, "masked_l1_bwd: null argument");
    VMASR_REQUIRE(n > 0 && n <= kMaxSlots && rows_g > 0 && N > 0, VMASR_EINVAL, "masked_l1_bwd: bad shape");
    VMASR_REQUIRE(((size_t)rows_g * N) % 4 == 0 && aligned_to(dgen, 16) && aligned_to(sgn, 4) && (!add || aligned_to(add, 16)), VMASR_EINVAL,
                  "masked_l1_bwd: slots must be 16-byte aligned (rows * N %% 4 == 0)");
    L1Slots t{};
    if (int e = fill_slots(t, valid, scale, n, N, rows_g, rows_g, "masked_l1_bwd")) return e;
    const long total4 = (long)rows_g * N / 4;
    const int blocks = (int)std::min<long>((total4 + 255) / 256, 256L * 8);
    hipStream_t st = static_cast<hipStream_t>(stream);
    VMASR_LAUNCH(VMASR_K_FEAT_L1, (add ? 9.0 : 5.0) * n * (double)rows_g * N, masked_l1_bwd_kernel, dim3(blocks, n), dim3(256), 0, st,
                 static_cast<const signed char *>(sgn), gout, add, dgen, t, (size_t)rows_g * N);
    return check_launch("masked_l1_bwd");
}

VMASR_EXPORT int vmasr_masked_l1_bwd(const void *sgn, const float *gout, float *dgen, const int64_t *valid, const float *scale, int32_t n,
                                     int64_t rows_g, int32_t N, vmasr_stream_t stream) {
    return vmasr_masked_l1_bwd_add(sgn, gout, nullptr, dgen, valid, scale, n, rows_g, N, stream);
}

// ---- LSGAN terms over a list of score tensors ---------------------------------------------------------------------------------
// model/loss.py:190-213: sum_i mean((t_i - c_i)^2) over the discriminators' score tensors (5 per signal; c = 1 for "real", 0 for
// "generated" targets).  As ATen ops: sub, pow, mean and an add per tensor forward, ~6 more backward — ~135 launches of 3-5 us per
// training step over ~6 000-element tensors.  One workgroup walks all tensors (fixed order: deterministic), one launch forms all
// gradients 2 (t_i - c_i) g / n_i.
namespace vmasr {
namespace {

constexpr int kGanMax = 16;
struct GanItems {
    const float *x[kGanMax];
    float *d[kGanMax];
    long n[kGanMax];
    float c[kGanMax];
    int count;
};

__global__ __launch_bounds__(1024) void lsgan_fwd_kernel(const GanItems t, float *__restrict__ out) {
    __shared__ double ws[16];
    double total = 0.0;                                  // (thread 0 only)
    for (int it = 0; it < t.count; ++it) {
        const float *__restrict__ x = t.x[it];
        const float c = t.c[it];
        double acc = 0.0;
        for (long i = threadIdx.x; i < t.n[it]; i += blockDim.x) {
            const float d = x[i] - c;
            acc += (double)(d * d);
        }
        for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
        if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = acc;
        __syncthreads();
        if (threadIdx.x == 0) {
            double s = 0.0;
            for (int w = 0; w < 16; ++w) s += ws[w];
            total += s / (double)t.n[it];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = (float)total;
}

// grid (blocks, count)
__global__ __launch_bounds__(256) void lsgan_bwd_kernel(const GanItems t, const float *__restrict__ gout) {
    const int it = blockIdx.y;
    const float *__restrict__ x = t.x[it];
    float *__restrict__ d = t.d[it];
    const float c = t.c[it], k = 2.f * gout[0] / (float)t.n[it];
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < t.n[it]; i += (long)gridDim.x * blockDim.x) d[i] = k * (x[i] - c);
}

int gan_fill(GanItems &t, const void *const *xs, void *const *ds, const int64_t *ns, const float *cs, int count, const char *what) {
    VMASR_REQUIRE(xs && ns && cs && count > 0 && count <= kGanMax, VMASR_EINVAL, "%s: 1..%d tensors", what, kGanMax);
    t.count = count;
    for (int i = 0; i < count; ++i) {
        VMASR_REQUIRE(xs[i] && ns[i] > 0 && (!ds || ds[i]), VMASR_EINVAL, "%s: tensor %d null / empty", what, i);
        t.x[i] = static_cast<const float *>(xs[i]);
        t.d[i] = ds ? static_cast<float *>(ds[i]) : nullptr;
        t.n[i] = ns[i];
        t.c[i] = cs[i];
    }
    return 0;
}

}  // namespace
}  // namespace vmasr

VMASR_EXPORT int vmasr_lsgan_fwd(const void *const *xs, const int64_t *ns, const float *targets, int32_t count, float *out,
                                 vmasr_stream_t stream) {
    VMASR_REQUIRE(out, VMASR_EINVAL, "lsgan_fwd: null output");
    GanItems t{};
    if (int e = gan_fill(t, xs, nullptr, ns, targets, count, "lsgan_fwd")) return e;
    double bytes = 0;
    for (int i = 0; i < count; ++i) bytes += 4.0 * ns[i];
    VMASR_LAUNCH(VMASR_K_FEAT_L1, bytes, lsgan_fwd_kernel, dim3(1), dim3(1024), 0, static_cast<hipStream_t>(stream), t, out);
    return check_launch("lsgan_fwd");
}

VMASR_EXPORT int vmasr_lsgan_bwd(const void *const *xs, void *const *ds, const int64_t *ns, const float *targets, int32_t count,
                                 const float *gout, vmasr_stream_t stream) {
    VMASR_REQUIRE(gout && ds, VMASR_EINVAL, "lsgan_bwd: null argument");
    GanItems t{};
    if (int e = gan_fill(t, xs, ds, ns, targets, count, "lsgan_bwd")) return e;
    long nmax = 0;
    double bytes = 0;
    for (int i = 0; i < count; ++i) { nmax = std::max<long>(nmax, ns[i]); bytes += 8.0 * ns[i]; }
    const int blocks = (int)std::min<long>((nmax + 255) / 256, 64);
    VMASR_LAUNCH(VMASR_K_FEAT_L1, bytes, lsgan_bwd_kernel, dim3(blocks, count), dim3(256), 0, static_cast<hipStream_t>(stream), t, gout);
    return check_launch("lsgan_bwd");
}
