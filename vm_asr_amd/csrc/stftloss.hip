// stftloss.hip — one resolution of the multi-resolution STFT loss on the (re, im) spectra, gfx950.
//
// Reference: model/loss.py:17-45, 137-184 (STFTLoss): for the generated signal x and the target y
//     mag = sqrt(clamp(re^2 + im^2, 1e-7));  sc = ||mag_y - mag_x||_F / ||mag_y||_F;  ml = mean |log mag_y - log mag_x|.
// As ATen ops that is pow, pow, add, clamp, sqrt (x 2 signals), sub, norm, norm, div, log, log, l1_loss: ~19 launches forward and
// ~35 backward over 2.1-2.5 M elements, three resolutions per step — ~170 launches of 5-6 us.  Here: one pass that reduces the
// three sums S1 = sum (mag_y - mag_x)^2, S2 = sum mag_y^2, S3 = sum |log mag_y - log mag_x| (fp64 partials per workgroup, no
// atomics), a one-workgroup finish (sc, ml and the sums kept for the backward) and one backward pass
//     d mag_x = g_sc (mag_x - mag_y) / sqrt(S1 S2) - g_ml sign(log mag_y - log mag_x) / (n mag_x),
//     d re_x = d mag_x re_x / mag_x, d im_x = d mag_x im_x / mag_x            (0 where re^2 + im^2 < 1e-7: clamp's gradient)
// with the upstream gradients read from device memory (graph replay).  HBM-bound streaming: forward r 16 B, backward r 16 B, w 8 B.
#include <algorithm>

#include "common.h"

namespace vmasr {
namespace {

constexpr int kSlBlocks = 1024;
constexpr float kSlMin = 1e-7f;

__device__ __forceinline__ float sl_mag(const float re, const float im) { return sqrtf(fmaxf(fmaf(re, re, im * im), kSlMin)); }

// partials[(block, 3)] fp64
__global__ __launch_bounds__(256) void stft_loss_fwd_kernel(const float *__restrict__ rx, const float *__restrict__ ix,
                                                            const float *__restrict__ ry, const float *__restrict__ iy, const long n4,
                                                            const long n, double *__restrict__ partials) {
    double s1 = 0.0, s2 = 0.0, s3 = 0.0;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        const float4 a = reinterpret_cast<const float4 *>(rx)[i], b = reinterpret_cast<const float4 *>(ix)[i];
        const float4 c = reinterpret_cast<const float4 *>(ry)[i], d = reinterpret_cast<const float4 *>(iy)[i];
        const float mx[4] = {sl_mag(a.x, b.x), sl_mag(a.y, b.y), sl_mag(a.z, b.z), sl_mag(a.w, b.w)};
        const float my[4] = {sl_mag(c.x, d.x), sl_mag(c.y, d.y), sl_mag(c.z, d.z), sl_mag(c.w, d.w)};
        float t1 = 0.f, t2 = 0.f, t3 = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float df = my[j] - mx[j];
            t1 = fmaf(df, df, t1);
            t2 = fmaf(my[j], my[j], t2);
            t3 += fabsf(logf(my[j]) - logf(mx[j]));
        }
        s1 += (double)t1; s2 += (double)t2; s3 += (double)t3;
    }
    if (blockIdx.x == 0 && (long)threadIdx.x < n - n4 * 4) {       // tail
        const long i = n4 * 4 + threadIdx.x;
        const float mx = sl_mag(rx[i], ix[i]), my = sl_mag(ry[i], iy[i]);
        s1 += (double)((my - mx) * (my - mx)); s2 += (double)(my * my); s3 += (double)fabsf(logf(my) - logf(mx));
    }
    for (int off = 32; off > 0; off >>= 1) {
        s1 += __shfl_down(s1, off, 64); s2 += __shfl_down(s2, off, 64); s3 += __shfl_down(s3, off, 64);
    }
    __shared__ double ws[4][3];
    if ((threadIdx.x & 63) == 0) { ws[threadIdx.x >> 6][0] = s1; ws[threadIdx.x >> 6][1] = s2; ws[threadIdx.x >> 6][2] = s3; }
    __syncthreads();
    if (threadIdx.x < 3) partials[(size_t)blockIdx.x * 3 + threadIdx.x] = (ws[0][threadIdx.x] + ws[1][threadIdx.x]) + (ws[2][threadIdx.x] + ws[3][threadIdx.x]);
}

// out[0] = sc, out[1] = ml, out[2] = 1 / sqrt(S1 S2) (0 when S1 == 0), out[3] = 1 / n       (one workgroup, fixed order)
__global__ __launch_bounds__(256) void stft_loss_finish_kernel(const double *__restrict__ partials, const int nblk, const long n,
                                                               float *__restrict__ out) {
    double s[3] = {0.0, 0.0, 0.0};
    for (int b = threadIdx.x; b < nblk; b += blockDim.x)
        for (int j = 0; j < 3; ++j) s[j] += partials[(size_t)b * 3 + j];
    __shared__ double ws[4][3];
    for (int j = 0; j < 3; ++j) {
        double v = s[j];
        for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
        if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6][j] = v;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const double S1 = (ws[0][0] + ws[1][0]) + (ws[2][0] + ws[3][0]), S2 = (ws[0][1] + ws[1][1]) + (ws[2][1] + ws[3][1]);
        const double S3 = (ws[0][2] + ws[1][2]) + (ws[2][2] + ws[3][2]);
        out[0] = (float)(sqrt(S1) / sqrt(S2));
        out[1] = (float)(S3 / (double)n);
        out[2] = S1 > 0.0 ? (float)(1.0 / (sqrt(S1) * sqrt(S2))) : 0.f;
        out[3] = (float)(1.0 / (double)n);
    }
}

__global__ __launch_bounds__(256) void stft_loss_bwd_kernel(const float *__restrict__ rx, const float *__restrict__ ix,
                                                            const float *__restrict__ ry, const float *__restrict__ iy, const long n,
                                                            const float *__restrict__ fin, const float *__restrict__ g_sc,
                                                            const float *__restrict__ g_ml, float *__restrict__ drx, float *__restrict__ dix) {
    const float ksc = (g_sc ? g_sc[0] : 0.f) * fin[2], kml = (g_ml ? g_ml[0] : 0.f) * fin[3];
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const float re = rx[i], im = ix[i];
        const float p = fmaf(re, re, im * im);
        float gr = 0.f, gi = 0.f;
        if (p >= kSlMin) {
            const float mx = sqrtf(p), my = sl_mag(ry[i], iy[i]);
            const float dl = logf(my) - logf(mx);
            const float sg = (float)((dl > 0.f) - (dl < 0.f));
            const float dm = ksc * (mx - my) - kml * sg / mx;
            const float q = dm / mx;
            gr = q * re;
            gi = q * im;
        }
        drx[i] = gr;
        dix[i] = gi;
    }
}

}  // namespace
}  // namespace vmasr

using namespace vmasr;

VMASR_EXPORT int32_t vmasr_stft_loss_blocks(void) { return kSlBlocks; }

VMASR_EXPORT int vmasr_stft_loss_fwd(const float *re_x, const float *im_x, const float *re_y, const float *im_y, int64_t n, double *partials,
                                     float *out, vmasr_stream_t stream) {
    VMASR_REQUIRE(re_x && im_x && re_y && im_y && partials && out && n > 0, VMASR_EINVAL, "stft_loss_fwd: null / empty argument");
    VMASR_REQUIRE(aligned_to(re_x, 16) && aligned_to(im_x, 16) && aligned_to(re_y, 16) && aligned_to(im_y, 16), VMASR_EINVAL,
                  "stft_loss_fwd: unaligned");
    hipStream_t st = static_cast<hipStream_t>(stream);
    VMASR_LAUNCH(VMASR_K_STFT_LOSS, 16.0 * n, stft_loss_fwd_kernel, dim3(kSlBlocks), dim3(256), 0, st, re_x, im_x, re_y, im_y, (long)(n / 4),
                 (long)n, partials);
    VMASR_LAUNCH(VMASR_K_STFT_LOSS, 24.0 * kSlBlocks, stft_loss_finish_kernel, dim3(1), dim3(256), 0, st, partials, kSlBlocks, (long)n, out);
    return check_launch("stft_loss_fwd");
}

VMASR_EXPORT int vmasr_stft_loss_bwd(const float *re_x, const float *im_x, const float *re_y, const float *im_y, int64_t n, const float *fin,
                                     const float *g_sc, const float *g_ml, float *d_re, float *d_im, vmasr_stream_t stream) {
    VMASR_REQUIRE(re_x && im_x && re_y && im_y && fin && d_re && d_im && n > 0, VMASR_EINVAL, "stft_loss_bwd: null / empty argument");
    const int blocks = (int)std::min<long>((n + 255) / 256, 256L * 16);
    VMASR_LAUNCH(VMASR_K_STFT_LOSS, 24.0 * n, stft_loss_bwd_kernel, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream), re_x, im_x,
                 re_y, im_y, (long)n, fin, g_sc, g_ml, d_re, d_im);
    return check_launch("stft_loss_bwd");
}
