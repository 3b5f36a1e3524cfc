/*
 * vmasr_oracle.c — CPU restatement of the VM-ASR hot path.   TEST INFRASTRUCTURE ONLY.
 *
 * This file is the checker for the HIP kernels: only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load it.  The product path never calls it.
 *
 * Parity pinning: every function here is checked against golden vectors produced by
 * running the reference itself on CPU (tests/golden/make_golden.py ->
 * tests/golden/ (npz files); see tests/test_oracle.py).  Citations are file:line under the
 * reference checkout.
 *
 * Plain C99 + OpenMP, fp32 arithmetic in the order of the reference where it matters,
 * double accumulators for the reductions of the backward pass.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

#define VMASR_API __attribute__((visibility("default")))

/* float64 build of the SAME source (oracle/Makefile: libvmasr_oracle64.so, -DVMASR_ORACLE_F64): every `float`
 * below becomes a double and every fp32 math call its double form.  It is the ADJUDICATOR of the parity tests:
 * where two fp32 evaluations differ by more than the tolerance budget (ill-conditioned spots such as a LayerNorm over
 * two channels), the one closer to this float64 evaluation is the more accurate one.  Same entry points, arrays
 * of double.  (The keyword macro is defined after the last #include.) */
#ifdef VMASR_ORACLE_F64
#define float double
#define expf exp
#define exp2f exp2
#define log1pf log1p
#define log2f log2
#define sqrtf sqrt
#define sinf sin
#define cosf cos
#define atan2f atan2
#endif

/* F.softplus (threshold 20) as used by selective_scan_ref
 * (kernels/selective_scan/test_selective_scan.py:315-316) and by the CUDA kernel
 * (cus/selective_scan_fwd_kernel.cuh:115-118). */
static inline float softplus_f(float x) { return x <= 20.f ? log1pf(expf(x)) : x; }

VMASR_API int vmasr_oracle_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* ------------------------------------------------------------------------------------
 * Selective scan forward.  Follows selective_scan_ref
 * (kernels/selective_scan/test_selective_scan.py:287-367):
 *   delta = softplus(delta + bias);  x_t = exp(delta_t A) x_{t-1} + delta_t B_t u_t;
 *   y_t = sum_n x_t[n] C_t[n] + D u_t.
 * Layouts (contiguous): u,delta,out (batch,dim,L); A (dim,N); B,C (batch,G,N,L);
 * D,bias (dim) or NULL; last_state (batch,dim,N) or NULL.
 * ---------------------------------------------------------------------------------- */
VMASR_API void vmasr_oracle_sscan_fwd(const float *u, const float *delta, const float *A,
                                      const float *B, const float *C, const float *D,
                                      const float *bias, int softplus, int batch, int dim,
                                      int G, int N, int L, float *out, float *last_state) {
    const int rows_per_group = dim / G;
#pragma omp parallel for collapse(2) schedule(static)
    for (int b = 0; b < batch; ++b) {
        for (int d = 0; d < dim; ++d) {
            const int g = d / rows_per_group;
            const float *ur = u + ((size_t)b * dim + d) * L;
            const float *dr = delta + ((size_t)b * dim + d) * L;
            float *orow = out + ((size_t)b * dim + d) * L;
            const float *Bg = B + ((size_t)b * G + g) * N * L;
            const float *Cg = C + ((size_t)b * G + g) * N * L;
            const float Dv = D ? D[d] : 0.f;
            const float bv = bias ? bias[d] : 0.f;
            float *x = (float *)calloc((size_t)N, sizeof(float));
            for (int t = 0; t < L; ++t) {
                float dl = dr[t] + bv;
                if (softplus) dl = softplus_f(dl);
                const float uv = ur[t];
                float y = 0.f;
                for (int n = 0; n < N; ++n) {
                    const float a = expf(dl * A[(size_t)d * N + n]);
                    x[n] = a * x[n] + dl * Bg[(size_t)n * L + t] * uv;
                    y += x[n] * Cg[(size_t)n * L + t];
                }
                orow[t] = y + Dv * uv;
            }
            if (last_state)
                for (int n = 0; n < N; ++n) last_state[((size_t)b * dim + d) * N + n] = x[n];
            free(x);
        }
    }
}

/* ------------------------------------------------------------------------------------
 * Selective scan backward: the analytic adjoint of the recurrence above (the reference
 * obtains it by autograd through selective_scan_ref; the CUDA kernel computes the same
 * quantities at cus/selective_scan_bwd_kernel.cuh:139-241):
 *   g_t  = dout_t C_t + a_{t+1} g_{t+1}                 (reverse scan, per state n)
 *   du_t = D dout_t + sum_n g_t B_t delta_t
 *   ddelta_t = sum_n [ g_t B_t u_t + g_t A (x_t - delta_t B_t u_t) ]      (pre-softplus)
 *   dA[n] += g_t delta_t (x_t - delta_t B_t u_t);  dB_t[n] += g_t delta_t u_t (over rows
 *   of the group);  dC_t[n] += dout_t x_t;  dD += dout_t u_t;
 *   dDelta_raw = ddelta * sigmoid(raw) when softplus (raw<=20);  dbias += dDelta_raw.
 * Outputs: du,ddelta (batch,dim,L); dA (dim,N); dB,dC (batch,G,N,L); dD,dbias (dim)
 * (may be NULL).  Everything is written (no pre-zeroing needed).
 * ---------------------------------------------------------------------------------- */
VMASR_API void vmasr_oracle_sscan_bwd(const float *u, const float *delta, const float *A,
                                      const float *B, const float *C, const float *D,
                                      const float *bias, const float *dout, int softplus,
                                      int batch, int dim, int G, int N, int L, float *du,
                                      float *ddelta, float *dA, float *dB, float *dC,
                                      float *dD, float *dbias) {
    const int rpg = dim / G;
    double *dA_acc = (double *)calloc((size_t)dim * N, sizeof(double));
    double *dD_acc = (double *)calloc((size_t)dim, sizeof(double));
    double *db_acc = (double *)calloc((size_t)dim, sizeof(double));
    double *dB_acc = (double *)calloc((size_t)batch * G * N * L, sizeof(double));
    double *dC_acc = (double *)calloc((size_t)batch * G * N * L, sizeof(double));
    /* parallel over (batch, group): rows of one group are walked serially so the dB/dC
     * sums need no atomics and are deterministic. */
#pragma omp parallel for collapse(2) schedule(dynamic)
    for (int b = 0; b < batch; ++b) {
        for (int g = 0; g < G; ++g) {
            float *dl = (float *)malloc(sizeof(float) * (size_t)L);
            double *xs = (double *)malloc(sizeof(double) * (size_t)L); /* x_t for one n */
            double *dd = (double *)malloc(sizeof(double) * (size_t)L);
            double *duu = (double *)malloc(sizeof(double) * (size_t)L);
            const float *Bg = B + ((size_t)b * G + g) * N * L;
            const float *Cg = C + ((size_t)b * G + g) * N * L;
            double *dBg = dB_acc + ((size_t)b * G + g) * N * L;
            double *dCg = dC_acc + ((size_t)b * G + g) * N * L;
            for (int r = 0; r < rpg; ++r) {
                const int d = g * rpg + r;
                const size_t off = ((size_t)b * dim + d) * L;
                const float *ur = u + off, *dr = delta + off, *gor = dout + off;
                const float bv = bias ? bias[d] : 0.f, Dv = D ? D[d] : 0.f;
                double dDv = 0.0;
                for (int t = 0; t < L; ++t) {
                    float v = dr[t] + bv;
                    dl[t] = softplus ? softplus_f(v) : v;
                    dd[t] = 0.0;
                    duu[t] = (double)Dv * gor[t];
                    dDv += (double)gor[t] * ur[t];
                }
                for (int n = 0; n < N; ++n) {
                    const double An = A[(size_t)d * N + n];
                    const float *Bn = Bg + (size_t)n * L, *Cn = Cg + (size_t)n * L;
                    double x = 0.0;
                    for (int t = 0; t < L; ++t) {
                        x = exp((double)dl[t] * An) * x + (double)dl[t] * Bn[t] * ur[t];
                        xs[t] = x;
                    }
                    double gcar = 0.0, dAv = 0.0;
                    for (int t = L - 1; t >= 0; --t) {
                        const double a_next = (t + 1 < L) ? exp((double)dl[t + 1] * An) : 0.0;
                        gcar = (double)gor[t] * Cn[t] + a_next * gcar;
                        const double bt = (double)dl[t] * Bn[t] * ur[t];
                        const double ax = xs[t] - bt; /* a_t x_{t-1} */
                        duu[t] += gcar * Bn[t] * dl[t];
                        dd[t] += gcar * Bn[t] * ur[t] + gcar * An * ax;
                        dAv += gcar * dl[t] * ax;
                        dBg[(size_t)n * L + t] += gcar * dl[t] * ur[t];
                        dCg[(size_t)n * L + t] += (double)gor[t] * xs[t];
                    }
#pragma omp atomic
                    dA_acc[(size_t)d * N + n] += dAv;
                }
                double dbv = 0.0;
                for (int t = 0; t < L; ++t) {
                    double v = dd[t];
                    if (softplus) {
                        const float raw = dr[t] + bv;
                        if (raw <= 20.f) v = v / (1.0 + exp(-(double)raw));
                    }
                    ddelta[off + t] = (float)v;
                    du[off + t] = (float)duu[t];
                    dbv += v;
                }
#pragma omp atomic
                dD_acc[d] += dDv;
#pragma omp atomic
                db_acc[d] += dbv;
            }
            free(dl); free(xs); free(dd); free(duu);
        }
    }
    for (size_t i = 0; i < (size_t)dim * N; ++i) dA[i] = (float)dA_acc[i];
    for (size_t i = 0; i < (size_t)batch * G * N * L; ++i) { dB[i] = (float)dB_acc[i]; dC[i] = (float)dC_acc[i]; }
    if (dD) for (int i = 0; i < dim; ++i) dD[i] = (float)dD_acc[i];
    if (dbias) for (int i = 0; i < dim; ++i) dbias[i] = (float)db_acc[i];
    free(dA_acc); free(dD_acc); free(db_acc); free(dB_acc); free(dC_acc);
}

/* ------------------------------------------------------------------------------------
 * CrossScan (model/vmamba.py:27-47): x (B,C,H,W) -> xs (B,4,C,H*W)
 *   k=0 row-major, k=1 column-major (transpose), k=2/3 = k=0/1 reversed.
 * ---------------------------------------------------------------------------------- */
VMASR_API void vmasr_oracle_cross_scan(const float *x, int Bn, int C, int H, int W, float *xs) {
    const size_t L = (size_t)H * W;
#pragma omp parallel for collapse(2) schedule(static)
    for (int b = 0; b < Bn; ++b)
        for (int c = 0; c < C; ++c) {
            const float *xp = x + ((size_t)b * C + c) * L;
            float *y0 = xs + (((size_t)b * 4 + 0) * C + c) * L;
            float *y1 = xs + (((size_t)b * 4 + 1) * C + c) * L;
            float *y2 = xs + (((size_t)b * 4 + 2) * C + c) * L;
            float *y3 = xs + (((size_t)b * 4 + 3) * C + c) * L;
            for (int h = 0; h < H; ++h)
                for (int w = 0; w < W; ++w) {
                    const float v = xp[(size_t)h * W + w];
                    const size_t l0 = (size_t)h * W + w, l1 = (size_t)w * H + h;
                    y0[l0] = v; y1[l1] = v; y2[L - 1 - l0] = v; y3[L - 1 - l1] = v;
                }
        }
}

/* CrossMerge (model/vmamba.py:50-61): ys (B,4,C,H,W) -> y (B,C,H*W);
 *   y[h,w] = ys0[l0] + ys2[L-1-l0] + ys1[l1] + ys3[L-1-l1]
 * (same association as the reference: (ys0+flip ys2) + ((ys1+flip ys3))^T). */
VMASR_API void vmasr_oracle_cross_merge(const float *ys, int Bn, int C, int H, int W, float *y) {
    const size_t L = (size_t)H * W;
#pragma omp parallel for collapse(2) schedule(static)
    for (int b = 0; b < Bn; ++b)
        for (int c = 0; c < C; ++c) {
            const float *y0 = ys + (((size_t)b * 4 + 0) * C + c) * L;
            const float *y1 = ys + (((size_t)b * 4 + 1) * C + c) * L;
            const float *y2 = ys + (((size_t)b * 4 + 2) * C + c) * L;
            const float *y3 = ys + (((size_t)b * 4 + 3) * C + c) * L;
            float *yp = y + ((size_t)b * C + c) * L;
            for (int h = 0; h < H; ++h)
                for (int w = 0; w < W; ++w) {
                    const size_t l0 = (size_t)h * W + w, l1 = (size_t)w * H + h;
                    yp[l0] = (y0[l0] + y2[L - 1 - l0]) + (y1[l1] + y3[L - 1 - l1]);
                }
        }
}

/* ------------------------------------------------------------------------------------
 * Depthwise 3x3 conv (pad 1) + bias + SiLU  (model/vmamba.py:859-868,1543-1545).
 * x,y (B,C,H,W); w (C,3,3); bias (C) or NULL.  pre (optional) receives the pre-activation.
 * ---------------------------------------------------------------------------------- */
static inline float sigmoid_f(float v) { return 1.f / (1.f + expf(-v)); }

VMASR_API void vmasr_oracle_dwconv_silu_fwd(const float *x, const float *w, const float *bias,
                                            int Bn, int C, int H, int W, float *y, float *pre) {
#pragma omp parallel for collapse(2) schedule(static)
    for (int b = 0; b < Bn; ++b)
        for (int c = 0; c < C; ++c) {
            const float *xp = x + ((size_t)b * C + c) * H * W;
            const float *wp = w + (size_t)c * 9;
            float *yp = y + ((size_t)b * C + c) * H * W;
            for (int h = 0; h < H; ++h)
                for (int ww = 0; ww < W; ++ww) {
                    float acc = bias ? bias[c] : 0.f;
                    for (int i = -1; i <= 1; ++i)
                        for (int j = -1; j <= 1; ++j) {
                            const int hh = h + i, wj = ww + j;
                            if (hh < 0 || hh >= H || wj < 0 || wj >= W) continue;
                            acc += wp[(i + 1) * 3 + (j + 1)] * xp[(size_t)hh * W + wj];
                        }
                    if (pre) pre[((size_t)b * C + c) * H * W + (size_t)h * W + ww] = acc;
                    yp[(size_t)h * W + ww] = acc * sigmoid_f(acc);
                }
        }
}

/* backward: given g = dL/dy, produce dx (B,C,H,W), dw (C,3,3), db (C). */
VMASR_API void vmasr_oracle_dwconv_silu_bwd(const float *x, const float *w, const float *bias,
                                            const float *g, int Bn, int C, int H, int W,
                                            float *dx, float *dw, float *db) {
    const size_t HW = (size_t)H * W;
    float *pre = (float *)malloc(sizeof(float) * (size_t)Bn * C * HW);
    float *tmp = (float *)malloc(sizeof(float) * (size_t)Bn * C * HW);
    vmasr_oracle_dwconv_silu_fwd(x, w, bias, Bn, C, H, W, tmp, pre);
    /* gp = g * silu'(pre) */
    for (size_t i = 0; i < (size_t)Bn * C * HW; ++i) {
        const float s = sigmoid_f(pre[i]);
        tmp[i] = g[i] * (s * (1.f + pre[i] * (1.f - s)));
    }
#pragma omp parallel for schedule(static)
    for (int c = 0; c < C; ++c) {
        double dwacc[9] = {0}, dbacc = 0.0;
        const float *wp = w + (size_t)c * 9;
        for (int b = 0; b < Bn; ++b) {
            const float *xp = x + ((size_t)b * C + c) * HW;
            const float *gp = tmp + ((size_t)b * C + c) * HW;
            float *dxp = dx + ((size_t)b * C + c) * HW;
            for (int h = 0; h < H; ++h)
                for (int ww = 0; ww < W; ++ww) {
                    const float gv = gp[(size_t)h * W + ww];
                    dbacc += gv;
                    for (int i = -1; i <= 1; ++i)
                        for (int j = -1; j <= 1; ++j) {
                            const int hh = h + i, wj = ww + j;
                            if (hh < 0 || hh >= H || wj < 0 || wj >= W) continue;
                            dwacc[(i + 1) * 3 + (j + 1)] += (double)gv * xp[(size_t)hh * W + wj];
                        }
                }
            /* dx[h,w] = sum_{i,j} w[i,j] * gp[h-i, w-j] */
            for (int h = 0; h < H; ++h)
                for (int ww = 0; ww < W; ++ww) {
                    float acc = 0.f;
                    for (int i = -1; i <= 1; ++i)
                        for (int j = -1; j <= 1; ++j) {
                            const int h2 = h - i, w2 = ww - j;
                            if (h2 < 0 || h2 >= H || w2 < 0 || w2 >= W) continue;
                            acc += wp[(i + 1) * 3 + (j + 1)] * gp[(size_t)h2 * W + w2];
                        }
                    dxp[(size_t)h * W + ww] = acc;
                }
        }
        for (int k = 0; k < 9; ++k) dw[(size_t)c * 9 + k] = (float)dwacc[k];
        if (db) db[c] = (float)dbacc;
    }
    free(pre); free(tmp);
}

/* ------------------------------------------------------------------------------------
 * FFT helper (double precision, iterative radix-2, n a power of two).
 * ---------------------------------------------------------------------------------- */
static void fft_inplace(double *re, double *im, int n, int inverse) {
    for (int i = 1, j = 0; i < n; ++i) {
        int bit = n >> 1;
        for (; j & bit; bit >>= 1) j ^= bit;
        j ^= bit;
        if (i < j) { double t = re[i]; re[i] = re[j]; re[j] = t; t = im[i]; im[i] = im[j]; im[j] = t; }
    }
    for (int len = 2; len <= n; len <<= 1) {
        const double ang = (inverse ? 2.0 : -2.0) * M_PI / len;
        for (int i = 0; i < n; i += len)
            for (int k = 0; k < len / 2; ++k) {
                const double wr = cos(ang * k), wi = sin(ang * k);
                const int p = i + k, q = i + k + len / 2;
                const double xr = re[q] * wr - im[q] * wi, xi = re[q] * wi + im[q] * wr;
                re[q] = re[p] - xr; im[q] = im[p] - xi; re[p] += xr; im[p] += xi;
            }
    }
}

/* periodic hann of length win, centred inside n_fft (torch.stft pads the window when
 * win_length < n_fft).  utils/stft.py:36 uses torch.hann_window(win_length) (periodic). */
static void make_window(double *wnd, int n_fft, int win) {
    const int left = (n_fft - win) / 2;
    for (int i = 0; i < n_fft; ++i) wnd[i] = 0.0;
    for (int i = 0; i < win; ++i) wnd[left + i] = 0.5 - 0.5 * cos(2.0 * M_PI * i / win);
}

static inline int reflect_idx(int i, int T) { /* 'reflect' padding (no edge repeat) */
    if (T == 1) return 0;
    while (i < 0 || i >= T) { if (i < 0) i = -i; if (i >= T) i = 2 * (T - 1) - i; }
    return i;
}

VMASR_API int vmasr_oracle_stft_frames(int T, int hop) { return 1 + T / hop; }

/* ------------------------------------------------------------------------------------
 * wav2spectro (utils/stft.py:22-68): torch.stft(center=True, reflect, onesided,
 * normalized=`normalized`), then, when `logmag` != 0: mag = log2(|S| + 1e-8),
 * phase = angle(S) written to out0/out1; else (re, im) are written (used by the loss /
 * metric restatements).  wav (B,T) -> out0,out1 (B, n_fft/2+1, frames).
 * ---------------------------------------------------------------------------------- */
VMASR_API void vmasr_oracle_stft(const float *wav, int Bn, int T, int n_fft, int hop, int win,
                                 int normalized, int logmag, float *out0, float *out1) {
    const int F = n_fft / 2 + 1, M = 1 + T / hop, pad = n_fft / 2;
    double *wnd = (double *)malloc(sizeof(double) * n_fft);
    make_window(wnd, n_fft, win);
    const double scale = normalized ? 1.0 / sqrt((double)n_fft) : 1.0;
#pragma omp parallel
    {
        double *re = (double *)malloc(sizeof(double) * n_fft);
        double *im = (double *)malloc(sizeof(double) * n_fft);
#pragma omp for collapse(2) schedule(static)
        for (int b = 0; b < Bn; ++b)
            for (int m = 0; m < M; ++m) {
                const float *x = wav + (size_t)b * T;
                for (int n = 0; n < n_fft; ++n) {
                    re[n] = wnd[n] * (double)x[reflect_idx(m * hop + n - pad, T)];
                    im[n] = 0.0;
                }
                fft_inplace(re, im, n_fft, 0);
                /* a real-input FFT (what torch.stft runs) returns EXACTLY zero imaginary parts for
                 * the DC and Nyquist bins; angle() is then 0 or +pi, never -pi.  The network
                 * consumes the phase as is (it is not 2*pi-periodic), so the branch matters. */
                im[0] = 0.0; im[n_fft / 2] = 0.0;
                /* Frame 0 is centred on sample 0: reflect padding and the symmetric window make it
                 * an even sequence about n_fft/2, so its spectrum is exactly real.  A floating
                 * point FFT leaves +-1e-9 there and angle() flips between +pi and -pi with the
                 * FFT library's rounding (the reference's own CPU and GPU runs disagree).  The
                 * exact value is taken: imaginary part +0. */
                if (m == 0) for (int f = 0; f < F; ++f) im[f] = 0.0;
                for (int f = 0; f < F; ++f) {
                    const float r = (float)(re[f] * scale), i = (float)(im[f] * scale);
                    const size_t o = ((size_t)b * F + f) * M + m;
                    if (logmag) {
                        out0[o] = log2f(sqrtf(r * r + i * i) + (float)1e-8);
                        out1[o] = atan2f(i, r);
                    } else { out0[o] = r; out1[o] = i; }
                }
            }
        free(re); free(im);
    }
    free(wnd);
}

/* ------------------------------------------------------------------------------------
 * spectro2wav (utils/stft.py:71-115): S = exp2(mag) * exp(i phase);
 * torch.istft(n_fft=2F-2, hop, win, normalized=True, center=True): per-frame irfft,
 * window, overlap-add, divide by the window-square envelope, trim n_fft/2 each side.
 * mag,phase (B,F,M) -> wav (B, hop*(M-1)).
 * ---------------------------------------------------------------------------------- */
VMASR_API void vmasr_oracle_istft(const float *mag, const float *phase, int Bn, int F, int M,
                                  int hop, int win, float *wav) {
    const int n_fft = 2 * F - 2, pad = n_fft / 2;
    const int full = n_fft + hop * (M - 1), T = hop * (M - 1);
    double *wnd = (double *)malloc(sizeof(double) * n_fft);
    make_window(wnd, n_fft, win);
    double *env = (double *)calloc((size_t)full, sizeof(double));
    for (int m = 0; m < M; ++m)
        for (int n = 0; n < n_fft; ++n) env[(size_t)m * hop + n] += wnd[n] * wnd[n];
    const double scale = sqrt((double)n_fft); /* undo normalized=True */
#pragma omp parallel for schedule(static)
    for (int b = 0; b < Bn; ++b) {
        double *re = (double *)malloc(sizeof(double) * n_fft);
        double *im = (double *)malloc(sizeof(double) * n_fft);
        double *ola = (double *)calloc((size_t)full, sizeof(double));
        for (int m = 0; m < M; ++m) {
            for (int f = 0; f < F; ++f) {
                const size_t o = ((size_t)b * F + f) * M + m;
                const float a = exp2f(mag[o]);
                re[f] = (double)(a * cosf(phase[o])) * scale;
                im[f] = (double)(a * sinf(phase[o])) * scale;
            }
            im[0] = 0.0; im[F - 1] = 0.0; /* c2r ignores imag of DC / Nyquist */
            for (int f = 1; f < F - 1; ++f) { re[n_fft - f] = re[f]; im[n_fft - f] = -im[f]; }
            fft_inplace(re, im, n_fft, 1);
            for (int n = 0; n < n_fft; ++n) ola[(size_t)m * hop + n] += wnd[n] * re[n] / n_fft;
        }
        for (int t = 0; t < T; ++t) wav[(size_t)b * T + t] = (float)(ola[pad + t] / env[pad + t]);
        free(re); free(im); free(ola);
    }
    free(wnd); free(env);
}

/* Backward of spectro2wav wrt (mag, phase): adjoint of the linear iSTFT followed by the
 * chain rule through S = 2^mag (cos phase + i sin phase).  g (B,T) -> dmag,dphase (B,F,M). */
VMASR_API void vmasr_oracle_istft_bwd(const float *mag, const float *phase, const float *g,
                                      int Bn, int F, int M, int hop, int win, float *dmag,
                                      float *dphase) {
    const int n_fft = 2 * F - 2, pad = n_fft / 2;
    const int full = n_fft + hop * (M - 1), T = hop * (M - 1);
    double *wnd = (double *)malloc(sizeof(double) * n_fft);
    make_window(wnd, n_fft, win);
    double *env = (double *)calloc((size_t)full, sizeof(double));
    for (int m = 0; m < M; ++m)
        for (int n = 0; n < n_fft; ++n) env[(size_t)m * hop + n] += wnd[n] * wnd[n];
    const double scale = sqrt((double)n_fft);
#pragma omp parallel
    {
        double *re = (double *)malloc(sizeof(double) * n_fft);
        double *im = (double *)malloc(sizeof(double) * n_fft);
#pragma omp for collapse(2) schedule(static)
        for (int b = 0; b < Bn; ++b)
            for (int m = 0; m < M; ++m) {
                for (int n = 0; n < n_fft; ++n) {
                    const int p = m * hop + n, t = p - pad;
                    re[n] = (t >= 0 && t < T) ? wnd[n] * (double)g[(size_t)b * T + t] / env[p] : 0.0;
                    im[n] = 0.0;
                }
                fft_inplace(re, im, n_fft, 0);
                for (int f = 0; f < F; ++f) {
                    const double ck = (f == 0 || f == F - 1) ? 1.0 : 2.0;
                    const double gr = ck / n_fft * re[f] * scale, gi = ck / n_fft * im[f] * scale;
                    const size_t o = ((size_t)b * F + f) * M + m;
                    const double a = exp2((double)mag[o]), c = cos((double)phase[o]), s = sin((double)phase[o]);
                    dmag[o] = (float)((gr * c + gi * s) * a * M_LN2);
                    dphase[o] = (float)(a * (-gr * s + gi * c));
                }
            }
        free(re); free(im);
    }
    free(wnd); free(env);
}

/* ------------------------------------------------------------------------------------
 * Gradient of vmasr_oracle_stft(..., logmag=0) wrt the wave: the adjoint of
 * reflect-pad -> frame -> window -> one-sided DFT (what autograd does for torch.stft in the
 * MR-STFT loss, model/loss.py:17-45).  gre,gim (B,F,M) -> gwav (B,T).
 * ---------------------------------------------------------------------------------- */
VMASR_API void vmasr_oracle_stft_bwd(const float *gre, const float *gim, int Bn, int T, int n_fft,
                                     int hop, int win, int normalized, float *gwav) {
    const int F = n_fft / 2 + 1, M = 1 + T / hop, pad = n_fft / 2;
    double *wnd = (double *)malloc(sizeof(double) * n_fft);
    make_window(wnd, n_fft, win);
    const double scale = normalized ? 1.0 / sqrt((double)n_fft) : 1.0;
#pragma omp parallel for schedule(static)
    for (int b = 0; b < Bn; ++b) {
        double *re = (double *)malloc(sizeof(double) * n_fft);
        double *im = (double *)malloc(sizeof(double) * n_fft);
        double *acc = (double *)calloc((size_t)T, sizeof(double));
        for (int m = 0; m < M; ++m) {
            /* x_n = sum_f [gRe_f cos(2 pi f n / N) - gIm_f sin(...)] = Re(sum_f G_f e^{+i theta}) */
            for (int f = 0; f < n_fft; ++f) { re[f] = 0.0; im[f] = 0.0; }
            for (int f = 0; f < F; ++f) {
                const size_t o = ((size_t)b * F + f) * M + m;
                re[f] = gre[o]; im[f] = gim[o];
            }
            fft_inplace(re, im, n_fft, 1);
            for (int n = 0; n < n_fft; ++n)
                acc[reflect_idx(m * hop + n - pad, T)] += wnd[n] * re[n] * scale;
        }
        for (int t = 0; t < T; ++t) gwav[(size_t)b * T + t] = (float)acc[t];
        free(re); free(im); free(acc);
    }
    free(wnd);
}
