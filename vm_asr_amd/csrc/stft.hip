// stft.hip — windowed STFT / iSTFT front-end for gfx950.
//
// Replaces wav2spectro / spectro2wav of the reference (utils/stft.py:22-68, :71-115),
// i.e. torch.stft(center=True(reflect), onesided, normalized) + log2|S| / angle(S) and
// exp2(mag) e^{i phase} -> torch.istft(center=True, normalized=True), plus the gradient
// of the inverse wrt (mag, phase) that training needs (the forward STFT input is the
// wave itself: no gradient flows there).
//
// One workgroup owns FR = 2 consecutive frames of one clip.  Per PAIR of frames one complex
// radix-2 Stockham FFT runs in LDS (two real frames packed as re/im and separated by
// Hermitian symmetry), twiddles and the hann window are built once per workgroup with
// sincospi, and the epilogue (log2/atan2, or the exp2/cos/sin chain) is fused.  Results
// are staged in LDS so that global writes along the frame axis are contiguous runs.
// HBM-bound: a clip reads T floats and writes 2*(n/2+1)*(1+T/hop) floats.
#include "common.h"

namespace vmasr {
namespace {

// frames per workgroup (one packed pair).  Measured at B = 4 (tools: make VARIANT=fr8 DEFS=-DVMASR_STFT_FR=8 + VMASR_LIB): with 8
// frames a launch is 240-1000 workgroups each walking four pair-FFTs (40 barriers) in sequence — stft 40.5 us, istft_bwd 67 us
// per launch; with 2 the same work is four times as many independent workgroups: 35 / 36 us (istft_frames 37.6 -> 31.5).  The
// shorter runs along the frame axis (8 B instead of 32 B per frequency row and workgroup) are merged in L2.
#ifndef VMASR_STFT_FR
#define VMASR_STFT_FR 2
#endif
constexpr int kFR = VMASR_STFT_FR;
constexpr float kLn2 = 0.6931471805599453f;

struct FftSmem {
    float2 *a, *b;  // ping-pong, n each
    float2 *tw;     // n/2: exp(-2 pi i m / n)
    float *win;     // n
    float *o0, *o1; // F * kFR each (STFT / iSTFT-bwd outputs)
};

__device__ __forceinline__ FftSmem carve(char *smem, int n, bool outs) {
    FftSmem s;
    s.a = reinterpret_cast<float2 *>(smem);
    s.b = s.a + n;
    s.tw = s.b + n;
    s.win = reinterpret_cast<float *>(s.tw + n / 2);
    s.o0 = s.win + n;
    s.o1 = outs ? s.o0 + (n / 2 + 1) * kFR : s.o0;
    return s;
}

size_t smem_bytes(int n, bool outs) {
    return (size_t)n * 8 * 2 + (size_t)n / 2 * 8 + (size_t)n * 4 + (outs ? (size_t)(n / 2 + 1) * kFR * 4 * 2 : 0);
}

// periodic hann(win) centred in n (torch.stft pads the window when win_length < n_fft)
__device__ __forceinline__ float hann_at(int i, int n, int win) {
    const int left = (n - win) / 2;
    const int k = i - left;
    return (k >= 0 && k < win) ? 0.5f - 0.5f * cospif(2.f * (float)k / (float)win) : 0.f;
}

__device__ __forceinline__ void setup_tables(const FftSmem &s, int n, int win) {
    for (int m = threadIdx.x; m < n / 2; m += blockDim.x) {
        float sn, cs;
        sincospif(2.f * (float)m / (float)n, &sn, &cs);
        s.tw[m] = make_float2(cs, -sn);
    }
    for (int i = threadIdx.x; i < n; i += blockDim.x) s.win[i] = hann_at(i, n, win);
}

// Stockham radix-2, natural order in/out.  Caller syncs after filling `src`.
__device__ __forceinline__ float2 *block_fft(float2 *src, float2 *dst, const float2 *tw, int n, bool inverse) {
    for (int ns = 1; ns < n; ns <<= 1) {
        const int tstride = n / (2 * ns);
        for (int j = threadIdx.x; j < n / 2; j += blockDim.x) {
            const int k = j & (ns - 1);
            float2 w = tw[k * tstride];
            if (inverse) w.y = -w.y;
            const float2 p = src[j], q = src[j + n / 2];
            const float2 wq = make_float2(w.x * q.x - w.y * q.y, w.x * q.y + w.y * q.x);
            const int j0 = ((j - k) << 1) + k;
            dst[j0] = make_float2(p.x + wq.x, p.y + wq.y);
            dst[j0 + ns] = make_float2(p.x - wq.x, p.y - wq.y);
        }
        __syncthreads();
        float2 *t = src; src = dst; dst = t;
    }
    return src;
}

__device__ __forceinline__ int reflect_idx(int i, int T) {
    if (T == 1) return 0;
    while (i < 0 || i >= T) {
        if (i < 0) i = -i;
        if (i >= T) i = 2 * (T - 1) - i;
    }
    return i;
}

// sum over frames of win^2 at padded position p (the istft normaliser)
__device__ __forceinline__ float envelope_at(int p, int n, int hop, int win, int M) {
    const int m_hi = min(M - 1, p / hop);
    const int m_lo = max(0, (p - n + hop) / hop);  // ceil((p-n+1)/hop)
    float e = 0.f;
    for (int m = m_lo; m <= m_hi; ++m) {
        const float w = hann_at(p - m * hop, n, win);
        e = fmaf(w, w, e);
    }
    return e;
}

// KIND 0: wav2spectro.  KIND 2: gradient of spectro2wav wrt (mag, phase).
// Both: real frames -> one-sided spectrum -> fused epilogue, outputs (B,F,M).
template <int KIND>
__global__ __launch_bounds__(256) void stft_like_kernel(const float *__restrict__ in, const float *__restrict__ mag,
                                                        const float *__restrict__ phase, float *__restrict__ out0,
                                                        float *__restrict__ out1, const int T, const int n,
                                                        const int hop, const int win, const int M,
                                                        const int normalized, const int logmag) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const FftSmem s = carve(smem, n, true);
    const int F = n / 2 + 1, pad = n / 2;
    // neighbouring frame groups read overlapping samples and write the same 128-byte lines of the (B, F, M) outputs:
    // keep them on one XCD (see istft_frames_kernel)
    const int lin = xcd_remap(blockIdx.y * gridDim.x + blockIdx.x, gridDim.x * gridDim.y);
    const int b = lin / gridDim.x, m0 = (lin % gridDim.x) * kFR;
    const float *x = in + (size_t)b * T;
    setup_tables(s, n, win);
    __syncthreads();
    const float nrm = rsqrtf((float)n);
    for (int pr = 0; pr < kFR; pr += 2) {
        const int ma = m0 + pr, mb = ma + 1;
        if (ma >= M) break;
        for (int i = threadIdx.x; i < n; i += blockDim.x) {
            float va, vb;
            if constexpr (KIND == 0) {
                va = x[reflect_idx(ma * hop + i - pad, T)];
                vb = mb < M ? x[reflect_idx(mb * hop + i - pad, T)] : 0.f;
            } else {
                const int pa = ma * hop + i, pb = mb * hop + i;
                const int ta = pa - pad, tb = pb - pad;
                va = (ta >= 0 && ta < T) ? x[ta] / envelope_at(pa, n, hop, win, M) : 0.f;
                vb = (mb < M && tb >= 0 && tb < T) ? x[tb] / envelope_at(pb, n, hop, win, M) : 0.f;
            }
            const float w = s.win[i];
            s.a[i] = make_float2(w * va, w * vb);
        }
        __syncthreads();
        const float2 *z = block_fft(s.a, s.b, s.tw, n, false);
        for (int f = threadIdx.x; f < F; f += blockDim.x) {
            const float2 zf = z[f], zc = z[(n - f) & (n - 1)];
            // X_a = (Z[f] + conj Z[n-f]) / 2 ; X_b = (Z[f] - conj Z[n-f]) / (2i)
            float2 xa = make_float2(0.5f * (zf.x + zc.x), 0.5f * (zf.y - zc.y));
            float2 xb = make_float2(0.5f * (zf.y + zc.y), 0.5f * (zc.x - zf.x));
#pragma unroll
            for (int which = 0; which < 2; ++which) {
                const float2 X = which ? xb : xa;
                const int m = which ? mb : ma;
                float r0, r1;
                if constexpr (KIND == 0) {
                    const float re = normalized ? X.x * nrm : X.x;
                    // frame 0 is an even sequence (reflect padding + symmetric window): its spectrum is
                    // exactly real; take +0 instead of the FFT's rounding noise so that angle() does not
                    // flip between +pi and -pi (see oracle/vmasr_oracle.c)
                    const float im = (m == 0) ? 0.f : (normalized ? X.y * nrm : X.y);
                    if (logmag) {
                        r0 = log2f(sqrtf(re * re + im * im) + 1e-8f);
                        r1 = atan2f(im, re);
                    } else { r0 = re; r1 = im; }
                } else {
                    // adjoint of c2r irfft (1/n, doubled interior bins) and of normalized=True
                    const float ck = (f == 0 || f == F - 1) ? 1.f : 2.f;
                    const float sc = ck * nrm;  // ck / n * sqrt(n)
                    const float gr = X.x * sc, gi = X.y * sc;
                    float a = 0.f, c = 1.f, sn = 0.f;
                    if (m < M) {
                        const size_t o = ((size_t)b * F + f) * M + m;
                        a = exp2f(mag[o]);
                        sincosf(phase[o], &sn, &c);
                    }
                    r0 = (gr * c + gi * sn) * a * kLn2;
                    r1 = a * (gi * c - gr * sn);
                }
                s.o0[f * kFR + pr + which] = r0;
                s.o1[f * kFR + pr + which] = r1;
            }
        }
        __syncthreads();
    }
    // contiguous runs of kFR frames per frequency row
    const int nfr = min(kFR, M - m0);
    for (int e = threadIdx.x; e < F * kFR; e += blockDim.x) {
        const int f = e / kFR, k = e % kFR;
        if (k < nfr) {
            const size_t o = ((size_t)b * F + f) * M + m0 + k;
            out0[o] = s.o0[e];
            out1[o] = s.o1[e];
        }
    }
}

// Stage 1 of an inverse transform: one-sided spectra -> windowed time frames (B, M, n).
//   SRC 0 (spectro2wav):   X = exp2(mag) e^{i phase};  c2r irfft, normalized=True undone
//   SRC 1 (STFT adjoint):  X = (gRe + i gIm) weighted 1/2 on interior bins (the adjoint of a real->
//                          one-sided transform), no 1/n; `normalized` multiplies by 1/sqrt(n)
template <int SRC>
__global__ __launch_bounds__(256) void istft_frames_kernel(const float *__restrict__ mag,
                                                           const float *__restrict__ phase,
                                                           float *__restrict__ frames, const int n, const int win,
                                                           const int M, const int normalized) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const FftSmem s = carve(smem, n, false);
    const int F = n / 2 + 1;
    // Frame groups that are neighbours in m read the same 128-byte lines of the (B, F, M) spectra (kFR frames = 8 bytes
    // per bin): under round-robin dispatch they would sit on different XCDs and each L2 would fetch the line again
    // (PMC: 243 MB fetched per launch for 17 MB of spectra).  Give each XCD a contiguous run of frame groups.
    const int lin = xcd_remap(blockIdx.y * gridDim.x + blockIdx.x, gridDim.x * gridDim.y);
    const int b = lin / gridDim.x, m0 = (lin % gridDim.x) * kFR;
    setup_tables(s, n, win);
    __syncthreads();
    // SRC 0: undo normalized=True, then irfft 1/n.  SRC 1: plain adjoint (optionally normalized)
    const float scale = SRC == 0 ? sqrtf((float)n) / (float)n : (normalized ? rsqrtf((float)n) : 1.f);
    for (int pr = 0; pr < kFR; pr += 2) {
        const int ma = m0 + pr, mb = ma + 1;
        if (ma >= M) break;
        for (int f = threadIdx.x; f < F; f += blockDim.x) {
            float2 xa = make_float2(0.f, 0.f), xb = make_float2(0.f, 0.f);
            if constexpr (SRC == 0) {
                {
                    const size_t o = ((size_t)b * F + f) * M + ma;
                    const float a = exp2f(mag[o]);
                    float sn, c;
                    sincosf(phase[o], &sn, &c);
                    xa = make_float2(a * c, a * sn);
                }
                if (mb < M) {
                    const size_t o = ((size_t)b * F + f) * M + mb;
                    const float a = exp2f(mag[o]);
                    float sn, c;
                    sincosf(phase[o], &sn, &c);
                    xb = make_float2(a * c, a * sn);
                }
            } else {
                const float wgt = (f == 0 || f == F - 1) ? 1.f : 0.5f;
                const size_t oa = ((size_t)b * F + f) * M + ma;
                xa = make_float2(wgt * mag[oa], wgt * phase[oa]);
                if (mb < M) xb = make_float2(wgt * mag[oa + 1], wgt * phase[oa + 1]);
            }
            if (f == 0 || f == F - 1) { xa.y = 0.f; xb.y = 0.f; }  // c2r ignores imag of DC / Nyquist
            // Z = X_a + i X_b ;  Z[n-f] = conj(X_a[f]) + i conj(X_b[f])
            s.a[f] = make_float2(xa.x - xb.y, xa.y + xb.x);
            if (f > 0 && f < F - 1) s.a[n - f] = make_float2(xa.x + xb.y, xb.x - xa.y);
        }
        __syncthreads();
        const float2 *z = block_fft(s.a, s.b, s.tw, n, true);
        float *fa = frames + ((size_t)b * M + ma) * n;
        float *fb = frames + ((size_t)b * M + mb) * n;
        for (int i = threadIdx.x; i < n; i += blockDim.x) {
            const float w = s.win[i] * scale;
            fa[i] = z[i].x * w;
            if (mb < M) fb[i] = z[i].y * w;
        }
        __syncthreads();
    }
}

// stage 2: overlap-add the <= ceil(n/hop) frames that cover each sample, divide by the
// window-square envelope, trim n/2 (center=True).  Gather form: deterministic, no atomics.
__device__ __forceinline__ float ola_at(const float *__restrict__ frames, int b, int p, int n, int hop, int M) {
    const int m_hi = min(M - 1, p / hop);
    const int m_lo = max(0, (p - n + hop) / hop);
    float acc = 0.f;
    for (int m = m_lo; m <= m_hi; ++m) acc += frames[((size_t)b * M + m) * n + (p - m * hop)];
    return acc;
}

// FOLD 0 (istft): overlap-add / window-square envelope, trim n/2.
// FOLD 1 (STFT adjoint): overlap-add, then fold the reflect padding back onto the signal
//        (x_pad[i] = x[reflect(i - n/2)]: sample t also collects padded positions n/2 - t and
//        n/2 + 2(T-1) - t when they exist).
template <int FOLD>
__global__ __launch_bounds__(256) void istft_ola_kernel(const float *__restrict__ frames, float *__restrict__ wav,
                                                        const int n, const int hop, const int win, const int M,
                                                        const int T) {
    const int b = blockIdx.y;
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= T) return;
    const int pad = n / 2;
    const int p = t + pad;
    if constexpr (FOLD == 0) {
        const int m_hi = min(M - 1, p / hop);
        const int m_lo = max(0, (p - n + hop) / hop);
        float env = 0.f;
        for (int m = m_lo; m <= m_hi; ++m) {
            const float w = hann_at(p - m * hop, n, win);
            env = fmaf(w, w, env);
        }
        wav[(size_t)b * T + t] = ola_at(frames, b, p, n, hop, M) / env;
    } else {
        float acc = ola_at(frames, b, p, n, hop, M);
        if (t >= 1 && t <= pad) acc += ola_at(frames, b, pad - t, n, hop, M);
        if (t <= T - 2 && t >= T - 1 - pad) acc += ola_at(frames, b, pad + 2 * (T - 1) - t, n, hop, M);
        wav[(size_t)b * T + t] = acc;
    }
}

int check_fft(int n, int hop, int win, const char *what) {
    VMASR_REQUIRE(n >= 64 && n <= 2048 && (n & (n - 1)) == 0, VMASR_EINVAL,
                  "%s: n_fft must be a power of two in [64, 2048] (got %d)", what, n);
    VMASR_REQUIRE(hop > 0 && win > 0 && win <= n, VMASR_EINVAL, "%s: need 0 < win <= n_fft and hop > 0", what);
    return 0;
}

// Raise the dynamic-LDS limit of `kernel` once, to the largest size this library ever asks for
// (n_fft 2048 with staged outputs).  Done on first use only: hipFuncSetAttribute is not a stream
// operation and is refused while a stream is capturing, so it must not run inside a HIP graph.
template <typename K>
int allow_smem(K kernel, size_t bytes, const char *what, bool *done) {
    if (*done || bytes <= 64 * 1024) return 0;
    const size_t cap = smem_bytes(2048, true);
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)cap);
    if (e != hipSuccess) {
        set_error("%s: cannot reserve %zu B of LDS: %s", what, cap, hipGetErrorString(e));
        return (int)e;
    }
    *done = true;
    return 0;
}

}  // namespace
}  // namespace vmasr

using namespace vmasr;

VMASR_EXPORT int vmasr_stft(const float *wav, float *out0, float *out1, int32_t B, int32_t T, int32_t n_fft,
                            int32_t hop, int32_t win, int32_t normalized, int32_t logmag, vmasr_stream_t stream) {
    if (int e = check_fft(n_fft, hop, win, "stft")) return e;
    VMASR_REQUIRE(wav && out0 && out1, VMASR_EINVAL, "stft: null tensor");
    VMASR_REQUIRE(B > 0 && B <= 65535 && T > n_fft / 2, VMASR_EINVAL,
                  "stft: need 0 < B <= 65535 and T > n_fft/2 (reflect padding)");
    const int M = 1 + T / hop;
    const size_t sm = smem_bytes(n_fft, true);
    static bool smem_ok = false;
    if (int e = allow_smem(stft_like_kernel<0>, sm, "stft", &smem_ok)) return e;
    const double bytes = (double)B * (T * 4.0 + 2.0 * (n_fft / 2 + 1) * M * 4.0);  // read wave, write 2 planes
    VMASR_LAUNCH(VMASR_K_STFT, bytes, stft_like_kernel<0>, dim3((M + kFR - 1) / kFR, B), dim3(256), sm,
                       static_cast<hipStream_t>(stream), wav, nullptr, nullptr, out0, out1, T, n_fft, hop, win, M,
                       normalized, logmag);
    return check_launch("stft");
}

VMASR_EXPORT size_t vmasr_istft_workspace(int32_t B, int32_t F, int32_t M, int32_t hop) {
    (void)hop;
    if (B <= 0 || F < 2 || M <= 0) return 0;
    return (size_t)B * M * (2 * (F - 1)) * sizeof(float);
}

VMASR_EXPORT int vmasr_istft(const float *mag, const float *phase, float *wav, int32_t B, int32_t F, int32_t M,
                             int32_t hop, int32_t win, void *ws, size_t ws_bytes, vmasr_stream_t stream) {
    const int n = 2 * (F - 1);
    if (int e = check_fft(n, hop, win, "istft")) return e;
    VMASR_REQUIRE(mag && phase && wav, VMASR_EINVAL, "istft: null tensor");
    VMASR_REQUIRE(B > 0 && B <= 65535 && M > 1, VMASR_EINVAL, "istft: need 0 < B <= 65535 and at least 2 frames");
    VMASR_REQUIRE(ws && ws_bytes >= vmasr_istft_workspace(B, F, M, hop), VMASR_ENOSPACE, "istft: workspace too small");
    const int T = hop * (M - 1);
    const size_t sm = smem_bytes(n, false);
    hipStream_t st = static_cast<hipStream_t>(stream);
    const double fbytes = (double)B * M * n * 4.0;
    VMASR_LAUNCH(VMASR_K_ISTFT_FRAMES, 2.0 * B * F * M * 4.0 + fbytes, istft_frames_kernel<0>, dim3((M + kFR - 1) / kFR, B), dim3(256), sm, st, mag, phase,
                       static_cast<float *>(ws), n, win, M, 1);
    VMASR_LAUNCH(VMASR_K_ISTFT_OLA, fbytes + (double)B * T * 4.0, istft_ola_kernel<0>, dim3((T + 255) / 256, B), dim3(256), 0, st, static_cast<const float *>(ws),
                       wav, n, hop, win, M, T);
    return check_launch("istft");
}

VMASR_EXPORT int vmasr_istft_bwd(const float *mag, const float *phase, const float *g, float *dmag, float *dphase,
                                 int32_t B, int32_t F, int32_t M, int32_t hop, int32_t win, vmasr_stream_t stream) {
    const int n = 2 * (F - 1);
    if (int e = check_fft(n, hop, win, "istft_bwd")) return e;
    VMASR_REQUIRE(mag && phase && g && dmag && dphase, VMASR_EINVAL, "istft_bwd: null tensor");
    VMASR_REQUIRE(B > 0 && B <= 65535 && M > 1, VMASR_EINVAL, "istft_bwd: need 0 < B <= 65535 and at least 2 frames");
    const int T = hop * (M - 1);
    const size_t sm = smem_bytes(n, true);
    static bool smem_ok = false;
    if (int e = allow_smem(stft_like_kernel<2>, sm, "istft_bwd", &smem_ok)) return e;
    const double bytes = (double)B * (T * 4.0 + 4.0 * F * M * 4.0);  // read g, mag, phase; write dmag, dphase
    VMASR_LAUNCH(VMASR_K_ISTFT_BWD, bytes, stft_like_kernel<2>, dim3((M + kFR - 1) / kFR, B), dim3(256), sm,
                       static_cast<hipStream_t>(stream), g, mag, phase, dmag, dphase, T, n, hop, win, M, 1, 0);
    return check_launch("istft_bwd");
}

VMASR_EXPORT size_t vmasr_stft_bwd_workspace(int32_t B, int32_t T, int32_t n_fft, int32_t hop) {
    if (B <= 0 || T <= 0 || n_fft <= 0 || hop <= 0) return 0;
    return (size_t)B * (1 + T / hop) * n_fft * sizeof(float);
}

VMASR_EXPORT int vmasr_stft_bwd(const float *gre, const float *gim, float *gwav, int32_t B, int32_t T, int32_t n_fft,
                                int32_t hop, int32_t win, int32_t normalized, void *ws, size_t ws_bytes,
                                vmasr_stream_t stream) {
    if (int e = check_fft(n_fft, hop, win, "stft_bwd")) return e;
    VMASR_REQUIRE(gre && gim && gwav, VMASR_EINVAL, "stft_bwd: null tensor");
    VMASR_REQUIRE(B > 0 && B <= 65535 && T > n_fft / 2, VMASR_EINVAL,
                  "stft_bwd: need 0 < B <= 65535 and T > n_fft/2 (reflect padding)");
    VMASR_REQUIRE(ws && ws_bytes >= vmasr_stft_bwd_workspace(B, T, n_fft, hop), VMASR_ENOSPACE,
                  "stft_bwd: workspace too small");
    const int M = 1 + T / hop, F = n_fft / 2 + 1;
    const size_t sm = smem_bytes(n_fft, false);
    hipStream_t st = static_cast<hipStream_t>(stream);
    const double fbytes = (double)B * M * n_fft * 4.0;
    VMASR_LAUNCH(VMASR_K_ISTFT_FRAMES, 2.0 * B * F * M * 4.0 + fbytes, istft_frames_kernel<1>, dim3((M + kFR - 1) / kFR, B),
                 dim3(256), sm, st, gre, gim, static_cast<float *>(ws), n_fft, win, M, normalized);
    VMASR_LAUNCH(VMASR_K_ISTFT_OLA, fbytes + (double)B * T * 4.0, istft_ola_kernel<1>, dim3((T + 255) / 256, B), dim3(256),
                 0, st, static_cast<const float *>(ws), gwav, n_fft, hop, win, M, T);
    return check_launch("stft_bwd");
}
