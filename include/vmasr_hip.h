/*
 * vmasr_hip.h — C ABI of libvmasr_hip.so, the MI355X (gfx950) implementation of the
 * VM-ASR data-parallel hot path.  Plain pointers and sizes only: no torch types.
 *
 * Every entry point replaces one reference interface (file:line under the reference):
 *
 *   vmasr_sscan_fwd      selective_scan_cuda_core.fwd  kernels/selective_scan/csrc/selective_scan/cus/selective_scan.cpp:157-239
 *                        (kernel cus/selective_scan_fwd_kernel.cuh:61-172; params selective_scan.h:26-62)
 *   vmasr_sscan_bwd      selective_scan_cuda_core.bwd  cus/selective_scan.cpp:241-349
 *                        (kernel cus/selective_scan_bwd_kernel.cuh:66-273; params selective_scan.h:64-90)
 *   vmasr_cross_scan     CrossScanTriton.forward / CrossMergeTriton.backward   model/csm_triton.py:311-337 (kernel :7-79), model/vmamba.py:27-47
 *   vmasr_cross_merge    CrossMergeTriton.forward / CrossScanTriton.backward   model/csm_triton.py:340-366 (kernel :82-154), model/vmamba.py:50-73
 *   vmasr_dwconv_silu_*  SS2D.conv2d + act (depthwise 3x3, pad 1)               model/vmamba.py:859-868,1543-1545
 *   vmasr_stft           wav2spectro                                            utils/stft.py:22-68
 *   vmasr_stft_bwd       autograd of torch.stft in the MR-STFT loss             model/loss.py:17-45
 *   vmasr_istft(_bwd)    spectro2wav (+ its autograd)                           utils/stft.py:71-115
 *
 * All functions are asynchronous on `stream` (a hipStream_t passed as void*; NULL = the
 * default stream), never synchronise the host, allocate nothing, and return 0 on
 * success, a negative VMASR_E* code for a contract violation (nothing launched), or a
 * positive hipError_t.  vmasr_last_error() gives a thread-local message.
 *
 * Device pointers must belong to the current HIP device of the calling thread.
 */
#ifndef VMASR_HIP_H
#define VMASR_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VMASR_ABI_VERSION 1

typedef void *vmasr_stream_t; /* hipStream_t */

enum { VMASR_F32 = 0, VMASR_F16 = 1, VMASR_BF16 = 2 };

enum {
    VMASR_OK = 0,
    VMASR_EINVAL = -1,   /* shape / dtype / stride contract violated            */
    VMASR_EALIGN = -2,   /* pointer or stride alignment not supported           */
    VMASR_ENOSPACE = -3  /* workspace too small                                 */
};

int vmasr_abi_version(void);
const char *vmasr_last_error(void);

/* Sequence positions per saved scan state.  The reference saves one (prod a, h) pair
 * per 2048-step chunk (cus/selective_scan.cpp:217-220); this library saves one per
 * VMASR_SSCAN_CHUNK steps so that the backward can restart every wave-tile
 * independently.  x has shape (batch, dim, n_chunks, 2*dstate) fp32 with
 * n_chunks = ceil(seqlen / vmasr_sscan_chunk()). */
#define VMASR_SSCAN_CHUNK 256
int vmasr_sscan_chunk(void);

/* POD mirror of SSMParamsBase (selective_scan.h:26-62).  Strides are in ELEMENTS.
 * u, delta, B, C, out share `dtype`; A, D, delta_bias, x are fp32.  Last-dim (seqlen)
 * stride is 1 for u, delta, B, C, out.  D_ptr / delta_bias_ptr may be NULL. */
typedef struct vmasr_sscan_params {
    int32_t batch, dim, seqlen, dstate, n_groups, n_chunks;
    int32_t dtype;          /* VMASR_F32 / F16 / BF16 */
    int32_t delta_softplus; /* bool */
    int64_t A_d_stride, A_dstate_stride;
    int64_t B_batch_stride, B_group_stride, B_dstate_stride;
    int64_t C_batch_stride, C_group_stride, C_dstate_stride;
    int64_t u_batch_stride, u_d_stride;
    int64_t delta_batch_stride, delta_d_stride;
    int64_t out_batch_stride, out_d_stride;
    const void *A_ptr, *B_ptr, *C_ptr, *D_ptr, *u_ptr, *delta_ptr, *delta_bias_ptr;
    void *out_ptr; /* (batch, dim, seqlen) dtype                       */
    void *x_ptr;   /* (batch, dim, n_chunks, 2*dstate) fp32 contiguous */
} vmasr_sscan_params;

/* POD mirror of SSMParamsBwd (selective_scan.h:64-90).  du, ddelta have `dtype`;
 * dA, dB, dC, dD, ddelta_bias are fp32 and MUST BE ZERO-INITIALISED by the caller
 * (they are accumulated, as in the reference: cus/selective_scan.cpp:319-327).
 * dB, dC are (batch, n_groups, dstate, seqlen) fp32 contiguous.  `ws_ptr` is a scratch
 * buffer of at least vmasr_sscan_bwd_workspace() bytes (may be NULL if that is 0). */
typedef struct vmasr_sscan_bwd_params {
    vmasr_sscan_params f; /* forward tensors; f.out_ptr unused; f.x_ptr = saved states */
    int64_t dout_batch_stride, dout_d_stride;
    int64_t du_batch_stride, du_d_stride;
    int64_t ddelta_batch_stride, ddelta_d_stride;
    int64_t dA_d_stride, dA_dstate_stride;
    const void *dout_ptr;
    void *du_ptr, *ddelta_ptr;
    void *dA_ptr, *dB_ptr, *dC_ptr, *dD_ptr, *ddelta_bias_ptr;
    void *ws_ptr;
    size_t ws_bytes;
} vmasr_sscan_bwd_params;

int vmasr_sscan_fwd(const vmasr_sscan_params *p, vmasr_stream_t stream);
size_t vmasr_sscan_bwd_workspace(const vmasr_sscan_bwd_params *p);
int vmasr_sscan_bwd(const vmasr_sscan_bwd_params *p, vmasr_stream_t stream);

/* Tuning knobs (process-wide; -1 = automatic).  rows: rows of one group handled per
 * wave (1/2/4); split: 0 = one wave walks a whole row, 1 = tile-parallel 3-phase scan. */
void vmasr_sscan_tune(int rows, int split);

/* x (B,C,H,W) contiguous -> xs (B,4,C,H*W) contiguous; same dtype. */
int vmasr_cross_scan(const void *x, void *xs, int32_t B, int32_t C, int32_t H, int32_t W,
                     int32_t dtype, vmasr_stream_t stream);
/* ys (B,4,C,H*W) contiguous -> y (B,C,H*W) contiguous; same dtype, fp32 accumulation. */
int vmasr_cross_merge(const void *ys, void *y, int32_t B, int32_t C, int32_t H, int32_t W,
                      int32_t dtype, vmasr_stream_t stream);

/* dtype-converting variants: 16-bit x -> fp32 xs (scan), fp32 ys -> 16-bit y (merge); equal dtypes
 * forward to the plain entry points. */
int vmasr_cross_scan_cvt(const void *x, void *xs, int32_t B, int32_t C, int32_t H, int32_t W,
                         int32_t in_dtype, int32_t out_dtype, vmasr_stream_t stream);
int vmasr_cross_merge_cvt(const void *ys, void *y, int32_t B, int32_t C, int32_t H, int32_t W,
                          int32_t in_dtype, int32_t out_dtype, vmasr_stream_t stream);

/* SS2D's x_proj / dt_proj einsums (model/vmamba.py:1473-1491) as one memory-bound map:
 *   xs (B,K,D,L) `dtype`; Wx (K, R+2N, D), Wdt (K, D, R) fp32  ->
 *   dts (B,K*D,L), Bs (B,K,N,L), Cs (B,K,N,L) fp32 contiguous (scan-ready), dtr (B,K,R,L) fp32 (the
 *   low-rank dt rows, kept by the caller for the backward).  R <= 8, R + 2N <= 16. */
/* The same projections for a general state dimension (d_state > 1: csrc/xproj_n.hip, fp32 MFMA products).  Same tensors and
 * meaning as vmasr_xproj_fwd / vmasr_xproj_bwd, except that `ws` of the backward holds only the gradient of the low-rank dt rows,
 * (B, K, R, L) fp32 = vmasr_xproj_n_ws_floats() floats.  d_inner even, dt_rank <= 16. */
int vmasr_xproj_n_supported(int32_t d_state, int32_t dt_rank, int32_t d_inner);
size_t vmasr_xproj_n_ws_floats(int32_t B, int32_t K, int32_t R, int32_t L);
int vmasr_xproj_n_fwd(const void *xs, const float *Wx, const float *Wdt, float *dts, float *Bs, float *Cs, float *dtr,
                      int32_t B, int32_t K, int32_t D, int32_t N, int32_t R, int32_t L, int32_t dtype, vmasr_stream_t stream);
int vmasr_xproj_n_bwd(const void *xs, const float *Wx, const float *Wdt, const float *dtr, const float *ddts, const float *dBs,
                      const float *dCs, const float *du, void *dxs, float *dWx, float *dWdt, float *ws,
                      int32_t B, int32_t K, int32_t D, int32_t N, int32_t R, int32_t L, int32_t dtype, vmasr_stream_t stream);
int vmasr_xproj_supported(int32_t d_state, int32_t dt_rank, int32_t d_inner);
int vmasr_xproj_fwd(const void *xs, const float *Wx, const float *Wdt, float *dts, float *Bs, float *Cs,
                    float *dtr, int32_t B, int32_t K, int32_t D, int32_t N, int32_t R, int32_t L,
                    int32_t dtype, vmasr_stream_t stream);
/* (ddts, dBs, dCs) fp32 [+ du (B,K*D,L) fp32 or NULL: added into dxs] -> dxs (B,K,D,L) `dtype`,
 * dWx, dWdt fp32, ZERO-INITIALISED by the caller (accumulated).  ws: B*K*(R+2N)*L floats of scratch. */
int vmasr_xproj_bwd(const void *xs, const float *Wx, const float *Wdt, const float *dtr, const float *ddts,
                    const float *dBs, const float *dCs, const float *du, void *dxs, float *dWx, float *dWdt,
                    float *ws, int32_t B, int32_t K, int32_t D, int32_t N, int32_t R, int32_t L,
                    int32_t dtype, vmasr_stream_t stream);

/* y = silu(dwconv3x3(x, w) + bias); x,y (B,C,H,W) contiguous `dtype`; w (C,3,3), bias (C)
 * fp32 (bias may be NULL). */
int vmasr_dwconv_silu_fwd(const void *x, const float *w, const float *bias, void *y, int32_t B,
                          int32_t C, int32_t H, int32_t W, int32_t dtype, vmasr_stream_t stream);
/* dx (B,C,H,W) `dtype`; dw (C,3,3), db (C) fp32, ZERO-INITIALISED by the caller.
 * ws: scratch of B*C*H*W floats (holds gy * silu'(pre)). */
int vmasr_dwconv_silu_bwd(const void *x, const float *w, const float *bias, const void *gy,
                          void *dx, float *dw, float *db, float *ws, int32_t B, int32_t C,
                          int32_t H, int32_t W, int32_t dtype, vmasr_stream_t stream);

/* wav (B,T) fp32 -> out0,out1 (B, n_fft/2+1, 1+T/hop) fp32.  center=True (reflect),
 * periodic hann(win) centred in n_fft, onesided.  logmag!=0: out0 = log2(|S|+1e-8),
 * out1 = angle(S); else out0 = Re S, out1 = Im S.  n_fft a power of two in [64, 4096]. */
int vmasr_stft(const float *wav, float *out0, float *out1, int32_t B, int32_t T, int32_t n_fft,
               int32_t hop, int32_t win, int32_t normalized, int32_t logmag,
               vmasr_stream_t stream);
/* gradient of vmasr_stft(..., logmag=0) wrt the wave: (gRe, gIm) (B,F,M) -> gwav (B,T).  Needed by
 * the multi-resolution STFT loss (model/loss.py:17-45 differentiates torch.stft).  ws: scratch of
 * vmasr_stft_bwd_workspace() bytes. */
size_t vmasr_stft_bwd_workspace(int32_t B, int32_t T, int32_t n_fft, int32_t hop);
int vmasr_stft_bwd(const float *gre, const float *gim, float *gwav, int32_t B, int32_t T, int32_t n_fft,
                   int32_t hop, int32_t win, int32_t normalized, void *ws, size_t ws_bytes,
                   vmasr_stream_t stream);
/* mag,phase (B,F,M) fp32 -> wav (B, hop*(M-1)) fp32; n_fft = 2F-2, normalized=True,
 * center=True.  ws: scratch of vmasr_istft_workspace() bytes. */
size_t vmasr_istft_workspace(int32_t B, int32_t F, int32_t M, int32_t hop);
int vmasr_istft(const float *mag, const float *phase, float *wav, int32_t B, int32_t F, int32_t M,
                int32_t hop, int32_t win, void *ws, size_t ws_bytes, vmasr_stream_t stream);
/* gradient of vmasr_istft wrt (mag, phase) given g = dL/dwav (B, hop*(M-1)). */
int vmasr_istft_bwd(const float *mag, const float *phase, const float *g, float *dmag,
                    float *dphase, int32_t B, int32_t F, int32_t M, int32_t hop, int32_t win,
                    vmasr_stream_t stream);

/* Channel-last LayerNorm over the last dimension (F.layer_norm on (rows, C) with C <= 1024):
 * SS2D.out_norm, VSSBlock.norm/norm2, PatchMerging2D.norm, PatchExpanding.norm
 * (model/vmamba.py:767-769,1793,1817; model/model.py:70,105-108,620,631).
 * x (rows,C) `dtype`; gamma/beta (C) fp32 or NULL; y (rows,C) `y_dtype` (equal to `dtype`, or fp32
 * for 16-bit x — what autocast gives —, or 16-bit for fp32 x); mean, rstd (rows) fp32 are saved for
 * the backward. */
int vmasr_layer_norm_fwd(const void *x, const float *gamma, const float *beta, void *y, float *mean,
                         float *rstd, int32_t rows, int32_t C, float eps, int32_t dtype, int32_t y_dtype,
                         vmasr_stream_t stream);
/* dx (rows,C) `dtype`; gy (rows,C) `gy_dtype` (same pairs); dgamma/dbeta (C) fp32 or NULL (plainly
 * written, no zero-init needed); ws: vmasr_layer_norm_bwd_workspace() bytes when dgamma/dbeta asked. */
size_t vmasr_layer_norm_bwd_workspace(int32_t rows, int32_t C);
int vmasr_layer_norm_bwd(const void *x, const void *gy, const float *gamma, const float *mean,
                         const float *rstd, void *dx, float *dgamma, float *dbeta, float *ws, int32_t rows,
                         int32_t C, int32_t dtype, int32_t gy_dtype, vmasr_stream_t stream);
/* Same, with `residual` (rows, C) fp32 or NULL added to dx: the gradient that arrives over the residual connection
 * around a pre-norm branch (x + branch(LayerNorm(x)), model/vmamba.py:1826-1837); residual has x's dtype. */
int vmasr_layer_norm_bwd_res(const void *x, const void *gy, const float *gamma, const float *mean,
                             const float *rstd, const void *residual, void *dx, float *dgamma, float *dbeta, float *ws,
                             int32_t rows, int32_t C, int32_t dtype, int32_t gy_dtype, vmasr_stream_t stream);
/* Deferred dgamma / dbeta: vmasr_layer_norm_bwd(_res) with ws but dgamma = dbeta = NULL writes the per-workgroup partials only
 * (vmasr_layer_norm_bwd_blocks(rows, C) rows of 2 C floats); vmasr_layer_norm_bwd_reduce_multi sums the partials of n such
 * calls in ONE launch (host arrays of device pointers / sizes; the table travels as a kernel argument). */
int32_t vmasr_layer_norm_bwd_blocks(int32_t rows, int32_t C);
int vmasr_layer_norm_bwd_reduce_multi(const float *const *parts, float *const *dgammas, float *const *dbetas, const int32_t *nblks,
                                      const int32_t *Cs, int32_t n, vmasr_stream_t stream);

/* nn.Linear with in/out features in {1,2,4,8} (in*out <= 32) over `rows` rows: the d_model = 1
 * VSS block and the 4->1 pointwise conv of the output layer (model/model.py:862-885,
 * model/vmamba.py:855,881,498-500).  x (rows,in) x_dtype; w (out,in), bias (out) fp32; y (rows,out)
 * y_dtype; supported dtype pairs: equal, or fp32 -> fp16/bf16.  16-byte aligned pointers. */
int vmasr_small_linear_supported(int32_t in_features, int32_t out_features);
int vmasr_small_linear_fwd(const void *x, const float *w, const float *bias, void *y, int64_t rows,
                           int32_t in_features, int32_t out_features, int32_t x_dtype, int32_t y_dtype,
                           vmasr_stream_t stream);
/* dx (rows,in) x_dtype or NULL; dw (out,in), db (out) fp32 or NULL (plainly written); gy (rows,out)
 * gy_dtype; ws: vmasr_small_linear_bwd_workspace() bytes when dw/db are requested. */
size_t vmasr_small_linear_bwd_workspace(int64_t rows, int32_t in_features, int32_t out_features);
int vmasr_small_linear_bwd(const void *x, const float *w, const void *gy, void *dx, float *dw, float *db,
                           float *ws, int64_t rows, int32_t in_features, int32_t out_features,
                           int32_t x_dtype, int32_t gy_dtype, vmasr_stream_t stream);

/* Power iteration of spectral normalisation (torch.nn.utils.parametrizations.spectral_norm, which
 * the reference's MPD uses: model/discriminator.py:37): n_iter rounds of u <- normalize(W v),
 * v <- normalize(W^T u), in place.  W (R,C) fp32 row-major, u (R), v (C) fp32; ws: R + C floats. */
int vmasr_spectral_power_iter(const float *W, float *u, float *v, float *ws, int32_t R, int32_t C,
                              int32_t n_iter, float eps, vmasr_stream_t stream);

/* The same for n matrices at once (one launch per phase instead of per matrix and phase).  `items`: n
 * descriptors in DEVICE memory; t (R floats) and s (C floats; zero on entry, left zero) are per-matrix scratch;
 * row_block_start / col_tile_start: exclusive prefix sums of ceil(R/4) and ceil(C/1024)*ceil(R/32);
 * total_row_blocks / total_col_tiles: their totals; weight_bytes = sum of R*C*4; sigma: null or n floats
 * receiving u^T W v of the final vectors.  n <= 64. */
typedef struct vmasr_spectral_item {
    const float *W;
    float *u, *v, *t, *s;
    int32_t R, C, row_block_start, col_tile_start;
} vmasr_spectral_item;
int vmasr_spectral_power_iter_batched(const vmasr_spectral_item *items, int32_t n, int32_t total_row_blocks,
                                      int32_t total_col_tiles, int64_t weight_bytes, int32_t n_iter, float eps,
                                      float *sigma, vmasr_stream_t stream);

/* Column / scatter operands of the period discriminator's (k,1) convolutions on channel-last sequences
 * (replaces the MIOpen convolutions behind model/discriminator.py:40-88 together with a GEMM):
 *   im2col: x (N, H, C) -> cols (N, H1, k, C), cols[n,h1,j,c] = x[n, h1*stride + j - pad, c] (0 outside),
 *           H1 = (H + 2*pad - k)/stride + 1;   col2im: the adjoint gather, dcols (N, H1, k, C) -> dx (N, H, C).
 * rows_out: 0, or the number of column rows to write (>= N*H1): the surplus rows are zero-filled (padding of a
 * stacked GEMM operand).  Contiguous tensors of `dtype` (VMASR_F32/F16/BF16). */
int vmasr_im2col_kx1(const void *x, void *cols, int64_t N, int32_t H, int32_t C, int32_t k, int32_t stride,
                     int32_t pad, int64_t rows_out, int32_t dtype, vmasr_stream_t stream);
/* im2col of fp32 x with the bf16 hi/lo split (vmasr_split_bf16) fused into the store: hi, lo (rows_out, k*C) bf16 */
int vmasr_im2col_kx1_split(const float *x, void *hi, void *lo, int64_t N, int32_t H, int32_t C, int32_t k, int32_t stride,
                           int32_t pad, int64_t rows_out, vmasr_stream_t stream);
int vmasr_col2im_kx1(const void *dcols, void *dx, int64_t N, int32_t H, int32_t C, int32_t k, int32_t stride,
                     int32_t pad, int32_t dtype, vmasr_stream_t stream);
/* Multi-slot variants for the stacked discriminator pass (n <= 8 convolutions of equal C/k/stride/pad, per-slot
 * N_i sequences of H_i): ONE launch per layer instead of n.  xs / dxs / srcs are HOST arrays of device pointers.
 *   im2col_split_multi: hi, lo (n, rows_out, k*C) bf16, slot i = split im2col of xs[i], zero rows below N_i*H1_i
 *   col2im_multi      : dcols (n, rows, k*C) -> dxs[i] (N_i, H_i, C); a NULL dxs[i] is skipped
 *   stack_rows        : full (n, rows, row_bytes) = srcs[i]'s first Ms[i] rows, zeros below (all zeros if NULL);
 *                       the gradient of `y[i, :M_i]` views for all slots (reference: autograd's slice backward). */
int vmasr_im2col_kx1_split_multi(const float *const *xs, const int64_t *Ns, const int32_t *Hs, int32_t n, void *hi, void *lo,
                                 int32_t C, int32_t k, int32_t stride, int32_t pad, int64_t rows_out, vmasr_stream_t stream);
/* Same gather, written as ONE (n, rows_out, 3 k C) bf16 operand [hi | lo | hi]: the A side of the K-concatenated triple
 * [hi | lo | hi] . [w_hi; w_hi; w_lo] (the three bf16 products of the fp32 GEMM accumulate inside one GEMM call). */
int vmasr_im2col_kx1_split3_multi(const float *const *xs, const int64_t *Ns, const int32_t *Hs, int32_t n, void *cat3, int32_t C,
                                  int32_t k, int32_t stride, int32_t pad, int64_t rows_out, vmasr_stream_t stream);
int vmasr_col2im_kx1_multi(const void *dcols, void *const *dxs, const int64_t *Ns, const int32_t *Hs, int32_t n, int32_t C,
                           int32_t k, int32_t stride, int32_t pad, int64_t rows, int32_t dtype, vmasr_stream_t stream);
/* col2im_multi into ONE stacked destination dx (n, dx_rows, C): slot s = gradient of its N_s * H_s rows, zeros below (the
 * gradient of the previous layer's stacked output, without the per-slot tensors + stack_rows in between) */
int vmasr_col2im_kx1_stacked(const void *dcols, void *dx, const int64_t *Ns, const int32_t *Hs, int32_t n, int32_t C, int32_t k, int32_t stride,
                             int32_t pad, int64_t rows, int64_t dx_rows, int32_t dtype, vmasr_stream_t stream);
int vmasr_stack_rows(const void *const *srcs, const int64_t *Ms, int32_t n, void *full, int64_t rows, int64_t row_bytes,
                     vmasr_stream_t stream);


/* Epilogues of the period discriminator's GEMMs (model/discriminator.py:100-104: conv -> GELU), slots x (M, N) fp32:
 *   vmasr_bias_gelu_fwd : acc (parts, slots, M, N): acc[0] = sum_p acc[p] + bias[slot, col] (the pre-activation, in place),
 *                         act = GELU(acc[0]) (exact erf form); parts = 1..8 (3: the products of a bf16 GEMM triple)
 *   vmasr_gelu_bwd_split: gx = g * GELU'(pre) (pre == NULL: gx = g), written only as its bf16 split (hi, lo);
 *                         cat3 (may be NULL): the same split as rows [hi | lo | hi] of width 3N (slots, M, 3N);
 *                         hi and lo may both be NULL when cat3 is given (its first two column blocks ARE hi and lo);
 *                         db[slot, col] += sum over rows of gx (db zero-initialised by the caller; may be NULL) */
int vmasr_bias_gelu_fwd(float *acc, const float *bias, float *act, int32_t slots, int64_t M, int32_t N, int32_t parts,
                        vmasr_stream_t stream);
/* w (n, N, K) fp32 -> out (n, K, 3N) bf16 = [hi^T | hi^T | lo^T] of the bf16 split of w: the B operands of the
 * forward GEMM triple (column blocks 0 and 2) and of the concatenated-contraction column-gradient GEMM (all of it) */
int vmasr_weight_prep_split(const float *w, void *out, int32_t n, int32_t N, int32_t K, vmasr_stream_t stream);
/* out (n, NK) = sum over p < P, s < S of parts (P, n, S, NK): the partial products of a split weight-gradient GEMM triple */
int vmasr_sum_parts(const float *parts, float *out, int32_t P, int32_t n, int32_t S, int64_t NK, vmasr_stream_t stream);
int vmasr_gelu_bwd_split(const float *pre, const float *g, void *hi, void *lo, void *cat3, float *db, int32_t slots, int64_t M,
                         int32_t N, vmasr_stream_t stream);
/* the same pass with the fp32 gradient as output (layers whose GEMMs stay fp32): gx = g * GELU'(pre), db += column sums */
int vmasr_gelu_bwd(const float *pre, const float *g, float *gx, float *db, int32_t slots, int64_t M, int32_t N, vmasr_stream_t stream);

/* The period discriminator's (k,1) convolutions as implicit GEMMs on the bf16 matrix cores, fp32 operands carried as
 * error-compensated bf16 pairs (hi, lo) — csrc/convgemm.hip; replaces im2col + three hipBLASLt GEMMs + epilogue passes
 * behind model/discriminator.py:21-147 (Conv2d (k,1), stride (s,1), zero padding (pad,0), GELU) for the stacked
 * discriminators ("slots": per slot nseq channel-last sequences of H positions).  Host array of n slot descriptors.
 *   fwd  : ah/al (>= nseq*H, Cin) bf16 pair of the input rows; bh/bl (Cout, k*Cin) pair of the weight in (tap, channel)
 *          column order; c0 (rows_out, Cout) pre-activation = conv + bias; act != 0: c1 = GELU(c0) fp32 (may be NULL),
 *          ch/cl its bf16 pair (may be NULL).  Rows below nseq*H1 (H1 = (H + 2 pad - k)/s + 1) up to rows_out are zeroed.
 *   dgrad: ah/al (>= nseq*H1, Cout) pair of the output gradient; bh/bl (Cin, k*Cout) pair of the weight transposed to
 *          (tap, output channel) column order; c0 = dx (rows_in, Cin) fp32, rows below nseq*H zeroed.  H = INPUT positions.
 *   wgrad: ah/al = pair of the output gradient g (>= nseq*H1, Cout); bh/bl = pair of the INPUT rows x (>= nseq*H, Cin);
 *          c0 = dW (splits, Cout, k*Cin) fp32 partial sums over `splits` row ranges ((tap, channel) column order).
 * Cin, Cout multiples of 128; k <= 8; s <= 3 (vmasr_conv_mfma_supported). */
typedef struct vmasr_cg_slot {
    const void *ah, *al;
    const void *bh, *bl;
    float *c0;
    float *c1;
    void *ch, *cl;
    const float *bias;
    int64_t nseq;
    int32_t H;
    int32_t reserved;
} vmasr_cg_slot;
int vmasr_conv_mfma_supported(int32_t Cin, int32_t Cout, int32_t k, int32_t stride);
/* The same for one stacked launch of `n` slots over `rows` operand rows (the largest of the layer's input and output row counts):
 * also checks the launchers' slot and 32-bit row-offset bounds (forward, input gradient and weight gradient).  Dispatch on this. */
int vmasr_conv_mfma_supported_launch(int32_t Cin, int32_t Cout, int32_t k, int32_t stride, int32_t n, int64_t rows);
/* CUs the conv_mfma kernels may occupy from now on (process-wide; 0 = all, the default): launched with fewer workgroups than
 * tiles they loop over the tiles.  Used by the two-stream train step while other kernels run beside them; results do
 * not depend on it. */
void vmasr_conv_set_cu_limit(int32_t cus);
int32_t vmasr_conv_get_cu_limit(void);
int vmasr_conv_mfma_fwd(const vmasr_cg_slot *slots, int32_t n, int32_t Cin, int32_t Cout, int32_t k, int32_t stride, int32_t pad,
                        int64_t rows_out, int32_t act, vmasr_stream_t stream);
int vmasr_conv_mfma_dgrad(const vmasr_cg_slot *slots, int32_t n, int32_t Cin, int32_t Cout, int32_t k, int32_t stride, int32_t pad,
                          int64_t rows_in, vmasr_stream_t stream);
/* The same stacked convolution with FP32 OPERANDS and exact-f32 products (v_mfma_f32_32x32x2_f32) for Cin = 32 — the 32 -> 128 layer of
 * model/discriminator.py:40-60, whose forward at the bf16 pair's 16-17 bits moved d(loss)/d(wave) out of its gate.  slots: ah = x fp32
 * (rows of Cin), bh = W fp32 (Cout, k Cin) in (tap, channel) order [dgrad: ah = g fp32 (rows of Cout), bh = W^T fp32 (Cin, k Cout) in
 * (tap, output channel) order, c0 = dx fp32]; al / bl are ignored; outputs, geometry and epilogue as the pair form.  Replace the fp32
 * GEMM over a materialised im2col operand + bias / GELU pass (forward) and the GEMM + col2im (input gradient). */
int vmasr_conv_f32_fwd(const vmasr_cg_slot *slots, int32_t n, int32_t Cin, int32_t Cout, int32_t k, int32_t stride, int32_t pad,
                       int64_t rows_out, int32_t act, vmasr_stream_t stream);
int vmasr_conv_f32_dgrad(const vmasr_cg_slot *slots, int32_t n, int32_t Cin, int32_t Cout, int32_t k, int32_t stride, int32_t pad,
                         int64_t rows_in, vmasr_stream_t stream);
int vmasr_conv_mfma_wgrad(const vmasr_cg_slot *slots, int32_t n, int32_t Cin, int32_t Cout, int32_t k, int32_t stride, int32_t pad,
                          int32_t splits, vmasr_stream_t stream);
/* The input gradient of layer l + 1 with the activation backward of layer l in its epilogue (round 5).  The reference's autograd runs,
 * per layer of model/discriminator.py:21-147, conv backward -> (+ the feature-matching term's gradient of model/loss.py) -> GELU backward;
 * vmasr_conv_mfma_dgrad + vmasr_masked_l1_bwd_add + vmasr_gelu_bwd_split are three passes over the feature map, this is one:
 *   t = dx tile (+ gtok[0] * epi[i].scale * sgn[row, col] for rows < epi[i].valid),   g = t * GELU'(pre[row, col])
 *   -> slots[i].c0 (fp32, may be NULL) and the bf16 pair slots[i].ch / cl (may be NULL; one of the two outputs is required);
 *   epi[i].db += column sums of g (fp32 atomics, one per column and tile — as vmasr_gelu_bwd_split does per workgroup).
 * slots as for vmasr_conv_mfma_dgrad (c1, bias unused); epi[i].pre (rows_in, Cin) fp32: layer l's pre-activation in slot i's stacked layout;
 * epi[i].sgn (rows_in, Cin) int8 or NULL: sign(generated - real) left by vmasr_masked_l1_fwd; gtok: DEVICE scalar, the loss term's upstream
 * gradient (required with a sign map).  Cin % 128 == 0. */
typedef struct vmasr_cg_gelu_bwd {
    const float *pre;
    const signed char *sgn;
    int64_t valid;
    float scale;
    int32_t reserved;
    float *db;                  /* (Cin) or NULL: the column sums of g are ADDED here (layer l's bias gradient; zero it first) */
} vmasr_cg_gelu_bwd;
int vmasr_conv_mfma_dgrad_gelu(const vmasr_cg_slot *slots, const vmasr_cg_gelu_bwd *epi, const float *gtok, int32_t n, int32_t Cin, int32_t Cout,
                               int32_t k, int32_t stride, int32_t pad, int64_t rows_in, vmasr_stream_t stream);

/* Finish of n split-K weight-gradient GEMMs in one launch (csrc/wgrad.hip): item i has partial products parts[i] (S_i, N_i, ld_i)
 * fp32; dws[i] (N_i, K_i) = sum over the S_i slabs of columns [0, K_i); dbs[i] (may be NULL) (N_i) = the same sum of column K_i
 * (the bias gradient when the GEMM's operand carried a ones column).  Host arrays of device pointers / sizes.
 * e1s / e2s (arrays may be NULL, entries may be NULL): (N_i) = columns K_i + 1 / K_i + 2 (the deep SS2D core's dA_log, dD). */
int vmasr_wgrad_finish_multi(const float *const *parts, float *const *dws, float *const *dbs, float *const *e1s, float *const *e2s,
                             const int32_t *Ss, const int32_t *Ns, const int32_t *Ks, const int32_t *lds, int32_t n, vmasr_stream_t stream);

/* y (rows, out) = x (rows, in) w^T (out, in) + bias, fp32 operands / result, the dot products accumulated in float64 and rounded once
 * (csrc/linear.hip): the Linear layers of the fp32 parity path (nn.Linear of model/vmamba.py:855,881,498-500, model/model.py:57-116). */
int vmasr_linear_f64acc(const float *x, const float *w, const float *bias, float *y, int64_t rows, int32_t out_features,
                        int32_t in_features, vmasr_stream_t stream);

/* AdamW step of many parameter tensors in one launch (torch.optim.AdamW semantics, utils/optimizer.py:16-50 of the
 * reference; non-amsgrad, decoupled weight decay, bias correction).  `items` is a DEVICE array, one entry per tensor;
 * `chunks` a DEVICE array of (item index, chunk index) int32 pairs, one per workgroup, chunk = vmasr_adamw_chunk() elements;
 * lr and step (the step count AFTER this update, as float) are DEVICE scalars read at run time.  lp (may be NULL): bf16
 * copy of the updated parameter, written in the same pass (lpt: the same transposed, for 2-D weights whose backward kernels read
 * W^T rows — vm_asr_amd/mlp.py, inproj.py, outproj.py).  vec: set when p, g, m, v are 16-byte and lp 8-byte aligned. */
typedef struct vmasr_adamw_item {
    float *p;
    const float *g;
    float *m;
    float *v;
    void *lp;
    int64_t n;
    float weight_decay;
    int32_t vec;
    void *lpt;            /* (may be NULL) bf16 copy of the updated 2-D parameter TRANSPOSED: lpt[c * rows + r] = p[r * cols + c] */
    int32_t rows, cols;
} vmasr_adamw_item;
int32_t vmasr_adamw_chunk(void);
int vmasr_adamw_step(const vmasr_adamw_item *items, const int32_t *chunks, int32_t nchunks, int64_t total_elems, const float *lr,
                     const float *step, float beta1, float beta2, float eps, vmasr_stream_t stream);

/* Spectrally normalised weights of one discriminator layer, stacked for the batched pass (model/discriminator.py:26-45:
 * spectral_norm around every convolution; sigma, u, v from the power iteration are constants for autograd):
 *   fwd: out (n, N, k*Cin) with out[s, o, j*Cin + c] = W_s[o, c, j] / sigma_s   (W_s: (N, Cin, k) fp32, HOST array of n pointers)
 *   bwd: gW_s[o, c, j] = (dW[s, o, j*Cin + c] - <dW_s, Wn_s> u_s[o] v_s[c*k + j]) / sigma_s, Wn = the forward's output;
 *        partials: n * vmasr_sn_dot_blocks() doubles of scratch.  sigmas / us / vs / gws: HOST arrays of device pointers. */
int32_t vmasr_sn_dot_blocks(void);
int vmasr_sn_stack_fwd(const void *const *weights, const void *const *sigmas, int32_t n, float *out, int32_t N, int32_t Cin, int32_t k,
                       vmasr_stream_t stream);
int vmasr_sn_stack_bwd(const float *dW, const float *Wn, void *const *gws, const void *const *sigmas, const void *const *us,
                       const void *const *vs, int32_t n, double *partials, int32_t N, int32_t Cin, int32_t k, vmasr_stream_t stream);

/* conv_post of the period discriminators (model/discriminator.py:45,106-109: Conv2d(C, 1, (3,1), 1, padding (1,0))) directly on
 * the previous layer's stacked output x (n, rows, C) fp32: slot s holds Ms[s] valid rows = whole sequences of Hs[s] positions
 * (zero padding at the sequence ends), w (n, 3, C) in (tap, channel) order, b (n); y (n, rows) (0 on the padding rows).
 * bwd: dx (n, rows, C) (zeros on the padding rows), dw (n, 3, C) and db (n) ACCUMULATED (zero-initialised by the caller);
 * each may be NULL.  Ms / Hs: HOST arrays.  C in {256, 512, 768, 1024}, kernel 3 (vmasr_conv_post_supported). */
int vmasr_conv_post_supported(int32_t C, int32_t k);
int vmasr_conv_post_fwd(const float *x, const float *w, const float *b, float *y, const int64_t *Ms, const int32_t *Hs, int32_t n, int64_t rows,
                        int32_t C, int32_t k, vmasr_stream_t stream);
int vmasr_conv_post_bwd(const float *x, const float *w, const float *gy, float *dx, float *dw, float *db, const int64_t *Ms, const int32_t *Hs,
                        int32_t n, int64_t rows, int32_t C, int32_t k, vmasr_stream_t stream);

/* First convolution of the period discriminators (model/discriminator.py:26-40,100-104: Conv2d(1, 32, (5,1), (3,1), padding
 * (2,0)) + GELU) for all n discriminators in one launch: xs[s] (Ns[s] sequences x Hs[s] samples, fp32; HOST arrays),
 * w (n, 32, 5), b (n, 32) -> pre, act (n, rows, 32) (zeros on the padding rows).  bwd: g = d loss / d act ->
 * dcols (n, rows, 5) = gradient of the (rows, 5) column operand (scatter it with vmasr_col2im_kx1_multi), dw (n, 32, 5) and
 * db (n, 32) ACCUMULATED (zero-initialised by the caller); each may be NULL. */
int vmasr_conv_first_fwd(const void *const *xs, const int64_t *Ns, const int32_t *Hs, int32_t n, const float *w, const float *b, float *pre,
                         float *act, int64_t rows, vmasr_stream_t stream);
int vmasr_conv_first_bwd(const void *const *xs, const int64_t *Ns, const int32_t *Hs, int32_t n, const float *w, const float *pre,
                         const float *g, float *dcols, float *dw, float *db, int64_t rows, vmasr_stream_t stream);

/* Feature-matching loss of the stacked discriminator pass (model/loss.py:227-235: mean over maps of mean |r - g|):
 *   real (n, rows_r, N), gen (n, rows_g, N) fp32; slot s compares its first valid[s] rows (valid, scale: HOST arrays);
 *   fwd: partials[s * vmasr_masked_l1_blocks() + b] = scale[s] * partial sum of |gen - real| (fp64; the caller adds them),
 *        sgn (n, rows_g, N) int8 (may be NULL) = sign(gen - real) on the valid part (the rest is not written);
 *   bwd: dgen (n, rows_g, N) = gout[0] * scale[s] * sgn on the valid part, 0 on the padding rows (gout: DEVICE scalar). */
int32_t vmasr_masked_l1_blocks(void);
int vmasr_masked_l1_fwd(const float *real, const float *gen, void *sgn, double *partials, const int64_t *valid, const float *scale,
                        int32_t n, int64_t rows_r, int64_t rows_g, int32_t N, vmasr_stream_t stream);
int vmasr_masked_l1_bwd(const void *sgn, const float *gout, float *dgen, const int64_t *valid, const float *scale, int32_t n,
                        int64_t rows_g, int32_t N, vmasr_stream_t stream);
/* the same with an addend: dgen = add + gout * scale[s] * sgn (add (n, rows_g, N) fp32 or NULL) — the feature map's whole
 * gradient (next layer's + the loss's) in one pass */
int vmasr_masked_l1_bwd_add(const void *sgn, const float *gout, const float *add, float *dgen, const int64_t *valid, const float *scale,
                            int32_t n, int64_t rows_g, int32_t N, vmasr_stream_t stream);
/* LSGAN terms (model/loss.py:190-213): out[0] = sum_i mean((x_i - targets[i])^2) over `count` <= 16 fp32 score tensors of ns[i]
 * elements (xs, ns, targets: HOST arrays), one launch; bwd: ds[i] (ns[i]) = 2 (x_i - targets[i]) gout[0] / ns[i], one launch. */
int vmasr_lsgan_fwd(const void *const *xs, const int64_t *ns, const float *targets, int32_t count, float *out, vmasr_stream_t stream);
int vmasr_lsgan_bwd(const void *const *xs, void *const *ds, const int64_t *ns, const float *targets, int32_t count, const float *gout,
                    vmasr_stream_t stream);

/* ---- fused SS2D core (vm_asr_amd/csrc/ss2d.hip) ---------------------------------------------------------
 * One operator for  y = CrossMerge(selective_scan(CrossScan(x), dt_proj(x_proj(.)), A, B, C, D, dt_bias, softplus))
 * of SS2D.forward_corev2 (model/vmamba.py:1472-1497; kernels: model/csm_triton.py:7-154,
 * cus/selective_scan_{fwd,bwd}_kernel.cuh) and its backward, for d_state 1, dt_rank 1, d_inner in {2,4,8,16,32},
 * H*W a multiple of 256 (vmasr_ss2d_supported).  All buffers are caller-owned device memory:
 *   x (B,D,H,W) `dtype`;  Wx (4,3,D) = x_proj_weight rows [dt, B, C];  Wdt (4,D) = dt_projs_weight;  dtb (4,D);
 *   Alog (4D) = A_logs (the operator applies A = -exp(Alog));  Ds (4D);            all weights fp32
 *   xT (B,D,W,H) `dtype` scratch (kept for the backward);  state (B,4D,H*W/256,2) fp32 (kept for the backward);
 *   out02, out13 (B,D,H*W) fp32 scratch;  y (B,D,H*W) fp32 = the merged output.
 * backward: dy (B,D,H*W) fp32 in, dyT / adj (like state) / part (vmasr_ss2d_part_floats) scratch,
 *   dx (B,D,H,W) `dtype`, dWx (4,3,D), dWdt (4,D), ddtb (4,D), dAlog (4D), dDs (4D) fp32 out (plain stores). */
#define VMASR_SS2D_PAIRS 1   /* flags: forward leaves the two pair outputs out02 (h,w order) / out13 (w,h order) to the caller
                              * (no merge launch, y unused); backward takes dy AND dyT as inputs (no transpose launch) —
                              * for a consumer / producer that handles both orders itself (vmasr_ln_gate_pair_*) */
typedef struct vmasr_ss2d_params {
    int32_t B, D, H, W, dtype, flags;
    const void *x;
    void *xT;
    const float *Wx, *Wdt, *dtb, *Alog, *Ds;
    float *state, *out02, *out13, *y;
    const float *dy;
    float *dyT, *adj, *part;
    void *dx;
    float *dWx, *dWdt, *ddtb, *dAlog, *dDs;
} vmasr_ss2d_params;
int vmasr_ss2d_supported(int32_t d_state, int32_t dt_rank, int32_t d_inner, int32_t H, int32_t W);
size_t vmasr_ss2d_part_floats(int32_t B, int32_t D, int32_t H, int32_t W);
int vmasr_ss2d_fwd(const vmasr_ss2d_params *p, vmasr_stream_t stream);
int vmasr_ss2d_bwd(const vmasr_ss2d_params *p, vmasr_stream_t stream);

/* ---- the SS2D core of the deep stages (vm_asr_amd/csrc/ss2d_deep.hip) -------------------------------------------------
 * The same operator as vmasr_ss2d_fwd/bwd (model/vmamba.py:1472-1497, model/csm_triton.py:7-154,
 * cus/selective_scan_{fwd,bwd}_kernel.cuh) for d_state 1, dt_rank R in {2,4,8} (16 for H*W <= 512), d_inner 64..512 (a multiple of 32),
 * H and W multiples of 4, H*W in {256,512,1024,2048,4096} (vmasr_ss2d_deep_supported): a workgroup owns whole (b, d) rows,
 * cross-scan / cross-merge are LDS index computations, x_proj runs as a small kernel in front (and its adjoint behind).
 * All buffers are caller-owned device memory; `dtype` (VMASR_F32 or VMASR_BF16) is the type of x, dx, tp, tb, tc, gpos:
 *   x (B,D,H,W);  WxT (4,R+2,D) = x_proj_weight AS STORED (the field keeps its round-3 name; since round 4 no transposed copy is made);  Wdt (4,D,R) = dt_projs_weight;  dtb (4,D);
 *   Alog (4D);  Ds (4D)                                                                                  weights fp32
 *   xdbl (B,4,R+2,H*W) fp32: written by the forward, read by the backward (directions 1/3 in (w,h) order);
 *   y (B,D,H*W) fp32 = the merged output.
 * backward: dy (B,D,H*W) fp32 in;  du (B,D,H*W) fp32, tp / tb / tc (B,4,D,H*W) scratch (per-row terms of d(x_dbl), each direction
 *   in its own scan order);
 *   pg (B,4,D,WR,20) fp32 out with WR = vmasr_ss2d_deep_waves_per_row: per-wave sums [dWdt[0..R-1], ddtb, dAlog, dDs] — the
 *   caller sums over (B, WR);  dx (B,D,H,W);  g32 (fp32 scratch) / gpos (`dtype`) (B,4(R+2),H*W): gradient of x_dbl per
 *   row-major position — dWx[k][c][d] = sum_{b,p} gpos[b][k(R+2)+c][p] x[b][d][p] (one GEMM on the caller's side). */
typedef struct vmasr_ss2d_deep_params {
    int32_t B, D, H, W, R, dtype;
    const void *x;
    const float *WxT, *Wdt, *dtb, *Alog, *Ds;
    float *xdbl, *y;
    const float *dy;
    float *du;
    void *tp, *tb, *tc;
    float *pg, *g32;
    void *dx, *gpos;
} vmasr_ss2d_deep_params;
int vmasr_ss2d_deep_supported(int32_t d_state, int32_t dt_rank, int32_t d_inner, int32_t H, int32_t W);
int32_t vmasr_ss2d_deep_waves_per_row(int32_t H, int32_t W);
int vmasr_ss2d_deep_fwd(const vmasr_ss2d_deep_params *p, vmasr_stream_t stream);
int vmasr_ss2d_deep_bwd(const vmasr_ss2d_deep_params *p, vmasr_stream_t stream);

/* ---- glue of SS2D.forwardv2 around the scan core (vm_asr_amd/csrc/ss2d_glue.hip; model/vmamba.py:1535-1550) ----------
 * ss2d_pre : xz (B*L, 2D) -> xT (B, D, L) = the x half channel-first, sz (B*L, D) = SiLU(z half)
 *            (replaces chunk + SiLU on a strided view + permute(0,3,1,2).contiguous()); _bwd: (dxT, dsz) -> dxz
 * ln_gate  : y (B, D, L) fp32, sz -> out (B*L, D) = LayerNorm_D(y^T; gamma, beta, eps) * sz, mean / rstd (B*L) fp32
 *            (replaces transpose.contiguous + LayerNorm + cast + multiply); _bwd: dout -> dy (B, D, L) fp32, dsz,
 *            dgamma / dbeta (D) fp32 ACCUMULATED with atomics (caller zero-initialises)
 * `dtype` is the type of xz / xT / sz / out / dout / dsz; D <= 512, L a multiple of 64 (D <= 32) or 16. */
int vmasr_ss2d_glue_supported(int32_t D, int32_t L, int32_t dtype);
int vmasr_ss2d_pre_fwd(const void *xz, void *xT, void *sz, int32_t B, int32_t D, int32_t L, int32_t dtype, vmasr_stream_t stream);
int vmasr_ss2d_pre_bwd(const void *xz, const void *dxT, const void *dsz, void *dxz, int32_t B, int32_t D, int32_t L, int32_t dtype,
                       vmasr_stream_t stream);
int vmasr_ln_gate_fwd(const float *y, const void *sz, const float *gamma, const float *beta, void *out, float *mean, float *rstd,
                      int32_t B, int32_t D, int32_t L, float eps, int32_t dtype, vmasr_stream_t stream);
int vmasr_ln_gate_bwd(const float *y, const void *sz, const void *dout, const float *gamma, const float *beta, const float *mean,
                      const float *rstd, float *dy, void *dsz, float *dgamma, float *dbeta, int32_t B, int32_t D, int32_t L,
                      int32_t dtype, vmasr_stream_t stream);
/* ln_gate_bwd for d_inner >= 64 with a workspace instead of atomics on dgamma / dbeta (up to 1 024 workgroups adding to the same D
 * addresses serialise): ws holds vmasr_ln_gate_bwd_workspace(...) floats = per-workgroup partials [dgamma (D) | dbeta (D)]
 * (0: this shape uses the direct variant, call vmasr_ln_gate_bwd).  dgamma / dbeta given: reduced at once into them (plain
 * stores, no zero-initialisation needed); both NULL: partials only, to be summed later (vmasr_layer_norm_bwd_reduce_multi with
 * nblk = workspace / (2 D), C = D). */
int64_t vmasr_ln_gate_bwd_workspace(int32_t B, int32_t D, int32_t L, int32_t dtype);
int vmasr_ln_gate_bwd_ws(const float *y, const void *sz, const void *dout, const float *gamma, const float *beta, const float *mean,
                         const float *rstd, float *dy, void *dsz, float *dgamma, float *dbeta, float *ws, int32_t B, int32_t D,
                         int32_t L, int32_t dtype, vmasr_stream_t stream);
/* ln_gate on the two PAIR outputs of the fused core: y = y02 (B, D, H*W in (h,w) order) + transpose(y13 (B, D, W*H in (w,h)
 * order)) is formed on the fly — what is left of CrossMerge (model/vmamba.py:50-73) never becomes a tensor; the backward
 * writes the gradient in both orders (dy02, dy13: the inputs of vmasr_ss2d_bwd with VMASR_SS2D_PAIRS).  D in {2,4,8,16,32},
 * H and W multiples of 16 (vmasr_ln_gate_pair_supported). */
int vmasr_ln_gate_pair_supported(int32_t D, int32_t H, int32_t W);
int vmasr_ln_gate_pair_fwd(const float *y02, const float *y13, const void *sz, const float *gamma, const float *beta, void *out,
                           float *mean, float *rstd, int32_t B, int32_t D, int32_t H, int32_t W, float eps, int32_t dtype,
                           vmasr_stream_t stream);
int vmasr_ln_gate_pair_bwd(const float *y02, const float *y13, const void *sz, const void *dout, const float *gamma,
                           const float *beta, const float *mean, const float *rstd, float *dy02, float *dy13, void *dsz,
                           float *dgamma, float *dbeta, int32_t B, int32_t D, int32_t H, int32_t W, int32_t dtype,
                           vmasr_stream_t stream);

/* x (n fp32) -> hi = bf16(x), lo = bf16(x - hi): the operands of an error-compensated 3-GEMM bf16 product that
 * reproduces the fp32 GEMM of the period discriminator's convolutions (model/discriminator.py:21-147) to ~1e-6
 * relative (vm_asr_amd/csrc/split.hip).  hi, lo: n bf16 each. */
int vmasr_split_bf16(const float *x, void *hi, void *lo, int64_t n, vmasr_stream_t stream);

/* ---- the Mlp branch of a VSSBlock as one MFMA kernel (vm_asr_amd/csrc/mlp.hip) ------------------------------------
 * Replaces, under bf16 autocast, `x + DropPath(Mlp(LayerNorm(x)))` of VSSBlock._forward (model/vmamba.py:1832-1837;
 * Mlp :483-509: fc1 -> GELU (exact erf) -> fc2) for the fp32 residual stream x (rows, d), d in {8,16,32,64},
 * hidden = 4 d (vmasr_mlp_supported).  All buffers are caller-owned device memory, 16-byte aligned:
 *   gamma, beta (d) fp32 = norm2;  w1 (4d, d) bf16 = fc1.weight, b1 (4d) fp32;  w2 (d, 4d) bf16 = fc2.weight, b2 (d) fp32;
 *   scale: per-sample residual scale (the DropPath keep mask / keep, one float per sample of rows_per_sample rows) or NULL;
 *   y (rows, d) = x + scale * (fc2(GELU(fc1(LN(x)) + b1)) + b2).
 * vmasr_mlp_bwd recomputes the forward from x and writes what the host needs for the remaining (library) steps:
 *   dxn (rows, d) bf16 = gradient wrt LayerNorm's output  (-> vmasr_layer_norm_bwd_res with residual = gy gives dx, dgamma, dbeta)
 *   xn_aug (rows, d + 8) bf16 = [LN(x) | 1 0 0 0 0 0 0 0],  gpre (rows, 4d) bf16 = gradient wrt fc1's output:
 *       gpre^T . xn_aug = [dW1 | db1 | 0]           (one GEMM with fp32 accumulation)
 *   gys (rows, d) bf16 = scale * gy,  act_aug (rows, 4d + 8) bf16 = [GELU(h) | 1 0 ...]:   gys^T . act_aug = [dW2 | db2 | 0]
 *   mean, rstd (rows) fp32 of the LayerNorm;   w1t (d, 4d) = fc1.weight^T, w2t (4d, d) = fc2.weight^T (bf16, contiguous). */
int vmasr_mlp_supported(int32_t d, int32_t hidden);
int vmasr_mlp_fwd(const void *x, const float *gamma, const float *beta, float eps, const void *w1, const float *b1,
                  const void *w2, const float *b2, const float *scale, int32_t rows_per_sample, void *y, int64_t rows,
                  int32_t d, int32_t x_dtype, vmasr_stream_t stream);
int vmasr_mlp_bwd(const void *x, const void *gy, const float *gamma, const float *beta, float eps, const void *w1,
                  const void *w1t, const float *b1, const void *w2t, const float *scale, int32_t rows_per_sample, void *dxn,
                  void *xn_aug, void *gys, void *act_aug, void *gpre, float *mean, float *rstd, int64_t rows, int32_t d,
                  int32_t x_dtype, vmasr_stream_t stream);

/* ---- the input side of SS2D.forwardv2 as one MFMA kernel (vm_asr_amd/csrc/mlp.hip: inproj_kernel) ----------------------
 * xz = in_proj(LayerNorm(x)) (model/vmamba.py:1826-1827, 1535); x', z = xz.chunk(2, -1); xT = x'.permute(0,3,1,2);
 * sz = SiLU(z) (:1537-1542) — LayerNorm + Linear + vmasr_ss2d_pre_fwd in one launch, under bf16 autocast, for x (rows, d) fp32 or
 * bf16 with rows = B * L, d in {8,16,32,64}, in_proj.weight w (4d, d) bf16 without bias, L a multiple of 32.  gamma = beta = NULL:
 * no LayerNorm (the output layers' blocks use nn.Identity).  xT (B, 2d, L) and sz (rows, 2d) bf16.
 * vmasr_inproj_bwd recomputes the forward and turns dxT / dsz into dxn (rows, d) bf16 (gradient wrt LayerNorm's output, or wrt x
 * without a norm), xn (rows, d) and gpre (rows, 4d) bf16 (dW = gpre^T . xn), mean / rstd (rows) for vmasr_layer_norm_bwd;
 * wt (d, 4d) = w^T contiguous. */
int vmasr_inproj_supported(int32_t d, int32_t d_proj, int64_t L);
int vmasr_inproj_fwd(const void *x, const float *gamma, const float *beta, float eps, const void *w, void *xT, void *sz, int64_t rows,
                     int32_t L, int32_t d, int32_t x_dtype, vmasr_stream_t stream);
int vmasr_inproj_bwd(const void *x, const float *gamma, const float *beta, float eps, const void *w, const void *wt, const void *dxT,
                     const void *dsz, void *dxn, void *xn, void *gpre, float *mean, float *rstd, int64_t rows, int32_t L, int32_t d,
                     int32_t x_dtype, vmasr_stream_t stream);

/* ---- the output side of SS2D.forwardv2 as one MFMA kernel (vm_asr_amd/csrc/mlp.hip: outproj_kernel) ----------------------
 * y = x + scale * (g . W^T): out_proj (model/vmamba.py:1551, no bias, dropout p = 0) + the VSSBlock's DropPath + residual add
 * (:1826-1827) for g (rows, 2d) bf16 = ln_gate's output, W (d, 2d) bf16 = out_proj.weight, x / y (rows, d) the residual stream
 * (fp32 or bf16), d in {8,16,32,64}; scale: per-sample residual scale (one float per rows_per_sample rows) or NULL.
 * vmasr_outproj_bwd: gy (rows, d) -> dg (rows, 2d) bf16 = scale * gy . W (ln_gate's incoming gradient) and gys (rows, d) bf16 =
 * scale * gy (dW = gys^T . g is one GEMM on the caller's side); wt (2d, d) = W^T contiguous.  The stream's own gradient is gy. */
int vmasr_outproj_supported(int32_t d, int32_t d_inner);
int vmasr_outproj_fwd(const void *g, const void *w, const void *x, const float *scale, int32_t rows_per_sample, void *y, int64_t rows,
                      int32_t d, int32_t x_dtype, vmasr_stream_t stream);
int vmasr_outproj_bwd(const void *gy, const void *wt, const float *scale, int32_t rows_per_sample, void *dg, void *gys, int64_t rows,
                      int32_t d, int32_t x_dtype, vmasr_stream_t stream);

/* ---- one resolution of the multi-resolution STFT loss on (re, im) spectra (vm_asr_amd/csrc/stftloss.hip) ----------------------
 * model/loss.py:17-45,137-184: mag = sqrt(clamp(re^2 + im^2, 1e-7)); sc = ||mag_y - mag_x||_F / ||mag_y||_F; ml = mean |log mag_y -
 * log mag_x| for the generated signal x and the target y, n elements each (fp32, 16-byte aligned).
 * fwd: partials (vmasr_stft_loss_blocks() x 3 fp64, scratch), out[0] = sc, out[1] = ml, out[2..3] = factors kept for the backward.
 * bwd: fin = the forward's `out`; g_sc / g_ml: device scalars (upstream gradients; NULL = 0); d_re / d_im (n) = gradient wrt x's
 * spectrum (exact zeros where re^2 + im^2 < 1e-7: clamp's gradient). */
int32_t vmasr_stft_loss_blocks(void);
int vmasr_stft_loss_fwd(const float *re_x, const float *im_x, const float *re_y, const float *im_y, int64_t n, double *partials, float *out,
                        vmasr_stream_t stream);
int vmasr_stft_loss_bwd(const float *re_x, const float *im_x, const float *re_y, const float *im_y, int64_t n, const float *fin,
                        const float *g_sc, const float *g_ml, float *d_re, float *d_im, vmasr_stream_t stream);

/* ---- in-library kernel timing (HIP events on the launch stream) ---------------------
 * When enabled, every kernel launch of this library is bracketed by two hipEvents
 * recorded on the stream the kernel is launched on; vmasr_prof_collect() waits for the
 * recorded events and returns, per kernel id, the launch count, the summed event time
 * (ms) and the summed ALGORITHMIC bytes of those launches (SURVEY.md §8d definition).
 * Used by bench.py for the `roofline` object; off by default (no events, no overhead). */
enum {
    VMASR_K_SSCAN_FWD = 0,      /* single-pass forward scan (mode 0)                    */
    VMASR_K_SSCAN_FWD_AGG,      /* split: per-tile aggregates                           */
    VMASR_K_SSCAN_FWD_CARRY,    /* split: scan of aggregates                            */
    VMASR_K_SSCAN_FWD_APPLY,    /* split: per-tile apply                                */
    VMASR_K_SSCAN_BWD,
    VMASR_K_SSCAN_BWD_AGG,
    VMASR_K_SSCAN_BWD_CARRY,
    VMASR_K_SSCAN_BWD_APPLY,
    VMASR_K_CROSS_SCAN,
    VMASR_K_CROSS_MERGE,
    VMASR_K_DWCONV_FWD,
    VMASR_K_DWCONV_BWD_A,
    VMASR_K_DWCONV_BWD_B,
    VMASR_K_STFT,
    VMASR_K_ISTFT_FRAMES,
    VMASR_K_ISTFT_OLA,
    VMASR_K_ISTFT_BWD,
    VMASR_K_LN_FWD,
    VMASR_K_LN_BWD,
    VMASR_K_LN_BWD_REDUCE,
    VMASR_K_SMALL_LINEAR_FWD,
    VMASR_K_SMALL_LINEAR_BWD,
    VMASR_K_SMALL_LINEAR_REDUCE,
    VMASR_K_XPROJ_FWD,
    VMASR_K_XPROJ_BWD_A,
    VMASR_K_XPROJ_BWD_B,
    VMASR_K_SPECTRAL,
    VMASR_K_IM2COL,
    VMASR_K_COL2IM,
    VMASR_K_SPLIT_BF16,
    VMASR_K_BIAS_GELU,          /* discriminator GEMM epilogues (bias + GELU; GELU' + split + bias gradient) */
    VMASR_K_SS2D_TRANSPOSE,     /* fused SS2D core: x -> x^T, dy -> dy^T                 */
    VMASR_K_SS2D_FWD_AGG,       /* x_proj + dt_proj + per-tile aggregates, 2 directions  */
    VMASR_K_SS2D_CARRY,         /* scan of the aggregates / adjoint carries / reduce     */
    VMASR_K_SS2D_FWD_APPLY,     /* x_proj + dt_proj + scan of both directions + add      */
    VMASR_K_SS2D_MERGE,         /* pair outputs: a + transpose(b)                        */
    VMASR_K_SS2D_BWD_AGG,
    VMASR_K_SS2D_BWD_APPLY,
    VMASR_K_SS2D_PRE,           /* xz -> (x channel-first, SiLU(z)) and its backward          */
    VMASR_K_LN_GATE,            /* LayerNorm_D(y^T) * SiLU(z) and its backward                */
    VMASR_K_STACK_ROWS,         /* gradient of the stacked discriminator views: copy + zero pad  */
    VMASR_K_FEAT_L1,            /* feature-matching loss over the stacked feature maps, fwd + bwd   */
    VMASR_K_ADAMW,              /* AdamW step of all parameters (+ bf16 shadow refresh), one launch */
    VMASR_K_CONV_POST,          /* the discriminators' 1024 -> 1 output convolution on the stacked maps */
    VMASR_K_MLP_FWD,            /* LayerNorm + fc1 + GELU + fc2 + residual of a VSS block as one MFMA kernel */
    VMASR_K_MLP_BWD,
    VMASR_K_INPROJ_FWD,         /* LayerNorm + in_proj + chunk + SiLU(z) + channel-first copy of a VSS block's SS2D as one MFMA kernel */
    VMASR_K_INPROJ_BWD,
    VMASR_K_SS2D_DEEP_XPROJ,    /* deep-stage SS2D core: x_proj (x -> x_dbl)                                    */
    VMASR_K_SS2D_DEEP_FWD,      /* dt_proj + the four directional scans of whole rows + cross-merge, one launch */
    VMASR_K_SS2D_DEEP_BWD,      /* its backward: du, per-row terms of d(x_dbl), parameter sums                  */
    VMASR_K_SS2D_DEEP_XBWD,     /* adjoint of x_proj: terms -> d(x_dbl) -> dx                                   */
    VMASR_K_OUTPROJ_FWD,        /* out_proj + DropPath + residual of a VSS block's SS2D branch as one MFMA kernel */
    VMASR_K_OUTPROJ_BWD,
    VMASR_K_STFT_LOSS,          /* one resolution of the MR-STFT loss: three sums in one pass, finish, one backward pass */
    VMASR_K_CONV_MFMA_FWD,      /* discriminator (k,1) convolution as an implicit bf16x3 MFMA GEMM + bias + GELU + split (csrc/convgemm.hip) */
    VMASR_K_CONV_MFMA_DGRAD,    /* its input gradient (residue classes of the stride, no col2im)                                               */
    VMASR_K_CONV_MFMA_WGRAD,    /* its weight gradient (transposed LDS reads)                                                                   */
    VMASR_K_WGRAD_FINISH,       /* sum over split-K slabs + bias column split-off of many weight gradients, one launch (csrc/wgrad.hip) */
    VMASR_K_SKINNY_LINEAR,      /* y = x W^T + b for >= 4096 rows and <= 96 features each side (csrc/skinny.hip) */
    VMASR_K_COUNT
};
/* Deterministic-reduction switch (debug aid, off by default; the Python side turns it on for VMASR_DETERMINISTIC=1): the kernels whose
 * parameter-gradient sums end in fp32 atomics take those atomics in workgroup order (csrc/common.h) -> bit-reproducible results, slower. */
void vmasr_set_deterministic(int on);   /* on: allocates the per-device ticket words NOW (call outside a stream capture); one stream only */
int vmasr_get_deterministic(void);
int64_t vmasr_det_timeouts(void);       /* ordered waits that ran out since the mode was switched on (must be 0); -1 on a HIP error */
void vmasr_prof_enable(int on);
void vmasr_prof_reset(void);
const char *vmasr_prof_name(int kernel_id);
/* returns 0, or a hipError_t; blocks the host until the recorded events have completed */
int vmasr_prof_collect(int kernel_id, int64_t *launches, double *total_ms, double *alg_bytes);
/* the same per CALL SHAPE: launches of `kernel_id` grouped by their algorithmic byte count (<= max_groups groups, in order of
 * first appearance); returns the number of groups filled, negative on a HIP error */
/* debug aid: *dst (device memory) = the device's constant-rate clock (100 MHz ticks) when `stream` reaches this point; a kernel launch,
 * hence capturable into a HIP graph (tools/phase_probe.py) */
int vmasr_mark_time(uint64_t *dst, vmasr_stream_t stream);

/* Linear layers with many rows and few features (csrc/skinny.hip; the reference's nn.Linear / 1x1 and patch-embedding nn.Conv2d of the U-Net glue at the
 * two highest resolutions, model/model.py:57-116,603-633, and their input gradients): y (rows, out) = x (rows, in) W^T + bias, rows >= 4096, in / out <= 96,
 * as one streaming pass (fp32 accumulation in k order).  W is read as w[o * w_stride_out + i * w_stride_in] (elements, fp32): (out, in) row-major =
 * (in, 1); the input gradient dx = g W of the same layer = the same call with in/out swapped and strides (1, in).  x / y: fp32 or bf16 (vmasr_dtype). */
/* 2-D im2col straight into GEMM rows and its adjoint (csrc/im2col.hip; the patch embedding's 3x3 stride-2 convolutions, model/model.py:603-633):
 * cols (B Ho Wo, C kh kw) with column order (c, i, j) = conv.weight.flatten(1)'s, from x (B, C, H, W) with element strides x_strides[4] (any layout);
 * dx (same logical shape / given strides) = sum of the row entries that read each pixel (a gather: no atomics).  dtypes: VMASR_F32 / VMASR_BF16 each side. */
int vmasr_im2col2d_rows(const void *x, void *cols, int32_t B, int32_t C, int32_t H, int32_t W, int32_t kh, int32_t kw, int32_t sh, int32_t sw,
                        int32_t ph, int32_t pw, const int64_t *x_strides, int32_t x_dtype, int32_t cols_dtype, vmasr_stream_t stream);
int vmasr_col2im2d_rows(const void *gcols, void *dx, int32_t B, int32_t C, int32_t H, int32_t W, int32_t kh, int32_t kw, int32_t sh, int32_t sw,
                        int32_t ph, int32_t pw, const int64_t *dx_strides, int32_t cols_dtype, int32_t dx_dtype, vmasr_stream_t stream);
int vmasr_skinny_linear_supported(int64_t rows, int32_t in_f, int32_t out_f);
int vmasr_skinny_linear(const void *x, const float *w, const float *bias, void *y, int64_t rows, int32_t in_f, int32_t out_f,
                        int64_t w_stride_out, int64_t w_stride_in, int32_t x_dtype, int32_t y_dtype, vmasr_stream_t stream);
int vmasr_prof_collect_shapes(int kernel_id, int max_groups, double *group_bytes, int64_t *group_launches, double *group_ms);

#ifdef __cplusplus
}
#endif
#endif /* VMASR_HIP_H */
