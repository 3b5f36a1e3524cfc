// convgemm.hip — the period discriminator's (k,1) convolutions as implicit GEMMs on the bf16 matrix cores with the
// fp32 operands carried as error-compensated bf16 pairs (gfx950, wave64, v_mfma_f32_32x32x16_bf16).
//
// Reference: model/discriminator.py:21-147 (PeriodDiscriminator: Conv2d (5,1) stride (3,1) / (1,1) on the folded signal,
// GELU after each), trainer/trainer.py:369-399 (its four passes per step), run in fp32 there.  csrc/split.hip explains why
// an fp32 product runs as  a b = a_hi b_hi + a_lo b_hi + a_hi b_lo  on this chip.  Round 2/3 ran each triple as THREE
// hipBLASLt GEMMs over materialised operands (im2col -> hi/lo columns, partial products, bias + GELU epilogue pass,
// [hi|lo|hi] gradient operand, col2im): ~7 ms of a 33 ms step were those operand / epilogue passes.  Here a layer is
//
//   forward : ONE launch.  A tile = rows of the previous layer's activation (hi, lo) gathered by (sequence, position, tap)
//             — im2col never exists —, B tile = the (tap, channel)-ordered weight rows (hi, lo); each A/B fragment is read
//             from LDS once for the three MFMAs of the triple, which accumulate into ONE fp32 accumulator; the epilogue
//             adds the bias, applies the exact-erf GELU and writes pre-activation, activation and the activation's bf16
//             pair (the next layer's A operand).
//   dgrad   : ONE launch.  dx[seq, h, c] = sum over the taps t with (h + pad - t) % stride == 0 of g[seq, (h+pad-t)/stride, :] W[:, t, c]:
//             the input rows are processed in `stride` residue classes, each of which sees a fixed tap subset, so no MFMA
//             works on structural zeros and col2im never exists (each dx row is written exactly once, no atomics).
//   wgrad   : dW[co, (t, c)] = sum over rows of g[row, co] x[row(t), c]: both operands have the contraction index as their
//             slow axis, so their LDS tiles are read with ds_read_b64_tr_b16 (hardware transpose read).
//
// Tiling of the NT kernel (forward, dgrad): 128 x 128 output tile per 256-thread workgroup (2 x 2 waves, 64 x 64 per wave =
// 2 x 2 accumulators of 32 x 32), K step 32; per K step a wave reads 16 fragments (ds_read_b128) for 24 MFMAs.  LDS: 4 operand
// tiles (A_hi, A_lo, B_hi, B_lo) of 128 rows x 64 B, double-buffered = 64 KB -> 2 workgroups per CU.  Rows of 64 B would put
// rows r and r + 4 on the same banks for a ds_read_b128 lane group; the 16-B chunk index is XOR-ed with (row >> 2) & 3 on both
// the store and the read, which makes the fragment reads conflict-free.  Global -> register -> LDS staging with the next tile's
// loads in flight during the current tile's MFMAs; one barrier per K step.
#include <algorithm>
#include <cstdlib>
#include <type_traits>

#include "common.h"

namespace vmasr {
namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) const void *cg_gptr;
typedef __attribute__((address_space(3))) void *cg_lptr;
__device__ uint4 cg_zero_page[4];   // 64 zero bytes: the DMA source of operand rows that do not exist   // a native vector: HIP's uint4 struct behind a ?: goes through scratch

constexpr int CG_BM = 128, CG_BN = 128, CG_BK = 32;
constexpr int CG_TILE = CG_BM * CG_BK * 2;            // bytes of one operand tile (8 KB)
constexpr int CG_STAGE = 4 * CG_TILE;                 // A_hi, A_lo, B_hi, B_lo
constexpr int CG_MAXP = 24;                           // (slot, residue class) problems per launch (CgParams must stay under the 4 KB of kernel arguments)

// One GEMM problem of a launch: C[crow(m), :] (+)= sum_j A[arow(m, j), :] . B[:, tap(j) CA + :]^T for m < M.
struct CgProb {
    const bf16_t *ah, *al, *bh, *bl;
    float *c0, *c1;            // c0: pre-activation (forward) / dx (dgrad); c1: activation fp32 (forward, may be null)
    bf16_t *ch, *cl;           // bf16 pair of the activation (forward, may be null)
    const float *bias;         // (NB) or null
    int M, Q;                  // rows, rows per sequence (m = seq Q + q)
    int HA;                    // positions per sequence on the A side
    int HC;                    // positions per sequence on the C side
    int hq_mul, hq_add;        // A position of tap j: q hq_mul + hq_add + j dstep, valid inside [0, HA)
    int crow_mul, crow_add;    // C row = seq HC + q crow_mul + crow_add
    int ntaps, tap0, tap_step, dstep;
    int tile_start, mtiles;    // first tile of the problem in the launch's tile list; its number of 128-row tiles
    int zero_rows;             // rows m in [M, zero_rows) are written as zeros (padding rows of the stacked C)
    // EPI 2 (input gradient with the activation's backward of the layer BELOW in the epilogue): `bias` holds that layer's
    // pre-activation (rows of C), `c1` its feature-matching sign map (int8, may be null); see conv_mfma_nt_kernel
    float fscale;              // weight of the sign term for this slot (times *CgParams::gtok)
    int fvalid;                // C rows below this index carry the sign term
    float *db;                 // EPI 2: (NB) column sums of g are ADDED here (the layer below's bias gradient), or null
};

struct CgParams {
    CgProb prob[CG_MAXP];
    int nprob, ntiles_n, total_tiles;
    int CA;                    // channels per tap on the A side (K extent per tap)
    int NB;                    // output columns
    int KB;                    // row length of B
    const float *gtok;         // EPI 2: device scalar, the upstream gradient of the feature-matching loss term (may be null)
};
static_assert(sizeof(CgParams) <= 4096, "CgParams is passed by value as the kernel's argument block");

// LDS byte offset of 16-byte chunk `chunk` of row `row` (64-byte rows).  The chunk index is XOR-ed with f(row >> 2), f(g) = (-g) & 3:
// conflict-free for the ds_read_b128 lane groups of BOTH fragment shapes (32x32x16: 32 rows x one chunk per half wave; 16x16x32:
// 16 rows x the four chunks, one per 16 lanes — the identity map f(g) = g is 2-way for the latter).
__device__ __forceinline__ int cg_swz(const int row) { return (0 - (row >> 2)) & 3; }
__device__ __forceinline__ int cg_off(const int row, const int chunk) { return row * 64 + ((chunk ^ cg_swz(row)) << 4); }

__device__ __forceinline__ float cg_gelu(float x) { return 0.5f * x * (1.f + erff(x * 0.70710678118654752f)); }
__device__ __forceinline__ float cg_gelu_grad(float x) {      // = gelu_grad_f of csrc/split.hip
    return 0.5f * (1.f + erff(x * 0.70710678118654752f)) + x * 0.3989422804014327f * __expf(-0.5f * x * x);
}

__device__ __forceinline__ f32x16 cg_mfma(const bf16x8 a, const bf16x8 b, const f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

// Tile configurations of the NT kernel: BM x BN output tile, WM x WN waves, each wave (BM / WM) x (BN / WN) = TI x TJ accumulators
// of 32 x 32.  128 x 128 / 4 waves: 64 KB of LDS, two workgroups per CU.  256 x 256 / 8 waves (128 x 64 per wave): 128 KB, one
// workgroup per CU — half the global -> LDS bytes and 3/4 of the LDS fragment reads per MFMA, and a K step lasts twice as long
// (48 MFMAs per wave), which is what covers the latency of the next tile's DMA; used whenever the output width allows.
// MF: MFMA shape, 32 = v_mfma_f32_32x32x16_bf16 (two K sub-steps of 16 per stage), 16 = v_mfma_f32_16x16x32_bf16 (one of 32).
// EPI: what the epilogue does with the fp32 tile.  0: + bias -> c0.  1 (forward): + bias -> c0 (pre-activation), GELU -> c1 and the bf16 pair
// ch / cl.  2 (input gradient of layer l + 1, finishing layer l's activation backward): t = tile (+ gtok fscale sign[row, col] for rows
// < fvalid: the feature-matching term that autograd would add to this gradient), g = t GELU'(pre_l[row, col]) -> c0 (fp32, optional) and
// the bf16 pair ch / cl (optional) — what gelu_bwd_split (csrc/split.hip) and masked_l1_bwd_add (csrc/featloss.hip) do in two more passes
// over the map (r 4 + 5 + 8, w 4 + 4 + 4 bytes per element -> r 5, w 4).
// OPS: 0 = operands are bf16 (hi, lo) pairs, three bf16 MFMAs per fragment pair (above); 1 = operands are FP32 rows, exact-f32 products on
// v_mfma_f32_32x32x2_f32 (MF must be 32) — the 32 -> 128 layer, whose forward at the pair's 16-17 bits moved d(loss)/d(wave) out of its
// gate (vm_asr_amd/discriminator.py).  Same LDS image and the same fragment addresses: a K step consumes 32 FLOATS of a row = 128 B,
// floats 0..15 in the tile the pair form calls A_hi / B_hi, floats 16..31 in A_lo / B_lo (pointers: lo = hi + 32 bf16-units); the
// 16-byte chunk a lane reads holds 4 consecutive k of its row, lane half lh takes chunk 2 ks + lh, and four 32x32x2 MFMAs contract
// (k, k + 4) pairs — the pairing is the same on the A and on the B side, so the sum over the K step is complete.  CA, KB count FLOATS here.
template <int BM, int BN, int WM, int WN, int EPI, int MF, int OPS = 0>
__global__ __launch_bounds__(64 * WM * WN, 2) void conv_mfma_nt_kernel(const CgParams P) {
    static_assert(OPS == 0 || MF == 32, "the f32 form uses the 32-row fragment layout");
    constexpr bool ACT = EPI == 1;
    constexpr int NT = 64 * WM * WN;                   // threads
    constexpr int WTM = BM / WM, WTN = BN / WN;        // per-wave tile
    constexpr int TI = WTM / MF, TJ = WTN / MF;
    constexpr int TILE_A = BM * 64, TILE_B = BN * 64;  // bytes of one operand tile (rows of 32 bf16)
    constexpr int STAGE = 2 * TILE_A + 2 * TILE_B;     // A_hi, A_lo, B_hi, B_lo
    constexpr int RPP = NT / 4;                        // rows staged per pass (4 threads per 64-byte row)
    constexpr int PA = BM / RPP, PB = (BN + RPP - 1) / RPP;
    constexpr bool B_PART = BN < RPP;                  // narrow B tile (BN = 32): only the waves that own its rows stage it
    static_assert(BM % RPP == 0 && (BN % RPP == 0 || (B_PART && BN % 16 == 0)) && BM % 128 == 0, "tile / thread geometry");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    int *crow_tab = reinterpret_cast<int *>(smem + 2 * STAGE);         // C row of each of the tile's BM rows (-1: none)

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN, lr = lane & 31, lh = lane >> 5;

    // ---- which tile: one per workgroup, or — launched with fewer workgroups than tiles (vmasr_conv_set_cu_limit: the two-stream
    // train step leaves CUs to the generator's kernels) — tiles blockIdx.x, blockIdx.x + gridDim.x, ... (gridDim.x % 8 == 0 keeps a
    // workgroup's tiles on its XCD's run of the remap)
#pragma unroll 1
    for (int tile = blockIdx.x; tile < P.total_tiles; tile += gridDim.x) {
    const int t = xcd_remap(tile, P.total_tiles);
    int pi = 0;
#pragma unroll 1
    for (int i = 1; i < P.nprob; ++i) pi = (t >= P.prob[i].tile_start) ? i : pi;
    const CgProb &pr = P.prob[pi];
    const int local = t - pr.tile_start;
    const int nt = local % P.ntiles_n, mt = local / P.ntiles_n;
    const int m0 = mt * BM, n0 = nt * BN;
    const int M = pr.M, Q = pr.Q, HA = pr.HA, CA = P.CA, NB = P.NB;
    constexpr int UPK = OPS ? 64 : 32;                 // bf16-units of a row that one K step consumes (f32 form: 32 floats = 64 units)
    const int LDA = OPS ? 2 * CA : CA;                 // row stride of A / per-tap stride inside a B row, in bf16-units
    const int KB = OPS ? 2 * P.KB : P.KB;

    if (tid < BM) {
        const int m = m0 + tid;
        int cr = -1;
        if (m < M) {
            const int seq = m / Q, q = m - seq * Q;
            cr = seq * pr.HC + q * pr.crow_mul + pr.crow_add;
        } else if (m < pr.zero_rows) {
            cr = m;                                                    // padding rows: the stacked tensor's row m itself
        }
        crow_tab[tid] = cr;
    }

    // ---- staging: LDS-DMA (global_load_lds_dwordx4).  One wave instruction fills 1 KB = 16 rows x 64 B in lane order, so the lane
    // that owns LDS unit (row, physical chunk c') fetches the row's LOGICAL chunk c' ^ ((row >> 2) & 3) — the swizzle lives on
    // the source address, the fragment reads apply the same XOR.  Per pass the workgroup fills RPP rows (wave w: rows 16w ..);
    // rows outside the data (padding taps, rows >= M) read a zero page instead (a DMA cannot be masked).
    const int srow = tid >> 2, chunk = (tid & 3) ^ cg_swz(srow);
    int a_base[PA], a_hq[PA];
    bool a_ok[PA];
#pragma unroll
    for (int i = 0; i < PA; ++i) {
        const int m = m0 + srow + RPP * i;
        a_ok[i] = m < M;
        const int mm = a_ok[i] ? m : 0;
        const int seq = mm / Q, q = mm - seq * Q;
        a_base[i] = seq * HA;
        a_hq[i] = q * pr.hq_mul + pr.hq_add;
    }
    const bf16_t *bsrc_h[PB], *bsrc_l[PB];
#pragma unroll
    for (int i = 0; i < PB; ++i) {
        const size_t o = (size_t)(n0 + ((B_PART && srow >= BN) ? 0 : srow + RPP * i)) * KB + chunk * 8;
        bsrc_h[i] = pr.bh + o;
        bsrc_l[i] = pr.bl + o;
    }
    const int cpk = CA / CG_BK;                                        // K steps per tap (32 elements of either form)
    const int nk = (m0 < M) ? pr.ntaps * cpk : 0;                      // all-padding tiles only write zeros
    const int wbase = wave * 16 * 64;                                  // this wave's 1 KB of each pass
    const bf16_t *zp = reinterpret_cast<const bf16_t *>(cg_zero_page);

#define CG_DMA(src, dst) __builtin_amdgcn_global_load_lds((cg_gptr)(src), (cg_lptr)(dst), 16, 0, 0)
    auto stage_tile = [&](const int kt, const int buf) {
        const int j = kt / cpk, c0 = (kt - j * cpk) * UPK;
        const int koff = (pr.tap0 + j * pr.tap_step) * LDA + c0;
        unsigned char *d = smem + buf * STAGE + wbase;
#pragma unroll
        for (int i = 0; i < PA; ++i) {
            const int pos = a_hq[i] + j * pr.dstep;
            const bool ok = a_ok[i] && pos >= 0 && pos < HA;
            const size_t o = (size_t)(a_base[i] + pos) * LDA + c0 + chunk * 8;
            CG_DMA(ok ? pr.ah + o : zp, d + i * RPP * 64);
            CG_DMA(ok ? (OPS ? pr.ah + o + 32 : pr.al + o) : zp, d + TILE_A + i * RPP * 64);
        }
#pragma unroll
        for (int i = 0; i < PB; ++i) {
            if (B_PART && srow >= BN) break;           // (wave-uniform: a wave stages 16 whole rows)
            CG_DMA(bsrc_h[i] + koff, d + 2 * TILE_A + i * RPP * 64);
            CG_DMA((OPS ? bsrc_h[i] + 32 : bsrc_l[i]) + koff, d + 2 * TILE_A + TILE_B + i * RPP * 64);
        }
    };

    using acc_t = typename std::conditional<MF == 32, f32x16, f32x4>::type;
    constexpr int NR = MF == 32 ? 16 : 4;              // accumulator registers per tile
    acc_t acc[TI][TJ];
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j)
#pragma unroll
            for (int r = 0; r < NR; ++r) acc[i][j][r] = 0.f;

    if (nk > 0) stage_tile(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) stage_tile(kt + 1, buf ^ 1);                  // lands during this tile's MFMAs (every wave left buf ^ 1 at the last barrier)
        const unsigned char *s = smem + buf * STAGE;
        if constexpr (OPS == 1) {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                f32x4 a0[TI], a1[TI], b0[TJ], b1[TJ];          // floats 0..15 / 16..31 of the K step: this lane's 4 consecutive k of each
#pragma unroll
                for (int i = 0; i < TI; ++i) {
                    const int oa = cg_off(wm * WTM + i * 32 + lr, ks * 2 + lh);
                    a0[i] = *reinterpret_cast<const f32x4 *>(s + oa);
                    a1[i] = *reinterpret_cast<const f32x4 *>(s + TILE_A + oa);
                }
#pragma unroll
                for (int j = 0; j < TJ; ++j) {
                    const int ob = cg_off(wn * WTN + j * 32 + lr, ks * 2 + lh);
                    b0[j] = *reinterpret_cast<const f32x4 *>(s + 2 * TILE_A + ob);
                    b1[j] = *reinterpret_cast<const f32x4 *>(s + 2 * TILE_A + TILE_B + ob);
                }
#pragma unroll
                for (int i = 0; i < TI; ++i)
#pragma unroll
                    for (int j = 0; j < TJ; ++j) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[i][e], b0[j][e], acc[i][j], 0, 0, 0);
#pragma unroll
                        for (int e = 0; e < 4; ++e) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[i][e], b1[j][e], acc[i][j], 0, 0, 0);
                    }
            }
        } else if constexpr (MF == 32) {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                bf16x8 ah[TI], al[TI], bh[TJ], bl[TJ];
#pragma unroll
                for (int i = 0; i < TI; ++i) {
                    const int oa = cg_off(wm * WTM + i * 32 + lr, ks * 2 + lh);
                    ah[i] = *reinterpret_cast<const bf16x8 *>(s + oa);
                    al[i] = *reinterpret_cast<const bf16x8 *>(s + TILE_A + oa);
                }
#pragma unroll
                for (int j = 0; j < TJ; ++j) {
                    const int ob = cg_off(wn * WTN + j * 32 + lr, ks * 2 + lh);
                    bh[j] = *reinterpret_cast<const bf16x8 *>(s + 2 * TILE_A + ob);
                    bl[j] = *reinterpret_cast<const bf16x8 *>(s + 2 * TILE_A + TILE_B + ob);
                }
#pragma unroll
                for (int i = 0; i < TI; ++i)
#pragma unroll
                    for (int j = 0; j < TJ; ++j) {
                        acc[i][j] = cg_mfma(al[i], bh[j], acc[i][j]);      // the two small products first, the large one last
                        acc[i][j] = cg_mfma(ah[i], bl[j], acc[i][j]);
                        acc[i][j] = cg_mfma(ah[i], bh[j], acc[i][j]);
                    }
            }
        } else {
            // 16x16x32: lane l holds row (l & 15), k = 8 (l >> 4) .. + 7 of a 16-row operand tile: chunk l >> 4 of the row's 64 bytes
            const int r16 = lane & 15, c16 = lane >> 4;
            bf16x8 bh[TJ], bl[TJ];
#pragma unroll
            for (int j = 0; j < TJ; ++j) {
                const int ob = cg_off(wn * WTN + j * 16 + r16, c16);
                bh[j] = *reinterpret_cast<const bf16x8 *>(s + 2 * TILE_A + ob);
                bl[j] = *reinterpret_cast<const bf16x8 *>(s + 2 * TILE_A + TILE_B + ob);
            }
#pragma unroll
            for (int i = 0; i < TI; ++i) {
                const int oa = cg_off(wm * WTM + i * 16 + r16, c16);
                const bf16x8 ah = *reinterpret_cast<const bf16x8 *>(s + oa);
                const bf16x8 al = *reinterpret_cast<const bf16x8 *>(s + TILE_A + oa);
#pragma unroll
                for (int j = 0; j < TJ; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh[j], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl[j], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh[j], acc[i][j], 0, 0, 0);
                }
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
#undef CG_DMA

    // ---- epilogue: the fp32 tile goes through LDS (the staging buffers are free now), 128 rows at a time, so that every thread
    // finishes 4 consecutive columns of a row: 16-byte stores, a wave covers whole 512-byte row segments.
    // Accumulator element r of a 32 x 32 tile is row (r & 3) + 8 (r >> 2) + 4 lh, column lr.
    float *ct = reinterpret_cast<float *>(smem);
    static_assert(128 * BN * 4 <= 2 * STAGE, "epilogue tile must fit the staging buffers");
    constexpr int C4 = BN / 4;                         // float4 per row
    constexpr int RPI = NT / C4;                       // rows per iteration of the store loop
    const int c4 = tid % C4;
    float4 bias4 = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 dbs = make_float4(0.f, 0.f, 0.f, 0.f);      // EPI 2: this thread's share of the column sums of g (its 4 columns, its rows)
    float gs = 0.f;
    if constexpr (EPI == 2) {
        if (pr.c1 && P.gtok) gs = P.gtok[0] * pr.fscale;
    } else {
        if (pr.bias) bias4 = *reinterpret_cast<const float4 *>(pr.bias + n0 + 4 * c4);
    }
#pragma unroll 1
    for (int h = 0; h < BM / 128; ++h) {
        if (h > 0) __syncthreads();
        if ((wm * WTM) / 128 == h) {
#pragma unroll
            for (int i = 0; i < TI; ++i)
#pragma unroll
                for (int j = 0; j < TJ; ++j)
#pragma unroll
                    for (int r = 0; r < NR; ++r) {
                        if constexpr (MF == 32)
                            ct[((wm * WTM) % 128 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh) * BN + wn * WTN + j * 32 + lr] = acc[i][j][r];
                        else      // 16 x 16 tile: element r of lane l is row 4 (l >> 4) + r, column l & 15
                            ct[((wm * WTM) % 128 + i * 16 + 4 * (lane >> 4) + r) * BN + wn * WTN + j * 16 + (lane & 15)] = acc[i][j][r];
                    }
        }
        __syncthreads();
        static_assert(128 % RPI == 0, "epilogue geometry");
#pragma unroll 4
        for (int it = 0; it < 128 / RPI; ++it) {
            const int row = it * RPI + tid / C4;
            const int cr = crow_tab[h * 128 + row];
            if (cr < 0) continue;
            const bool live = m0 + h * 128 + row < M;
            const size_t o = (size_t)cr * NB + n0 + 4 * c4;
            float4 v = *reinterpret_cast<const float4 *>(ct + row * BN + 4 * c4);
            if (live) { v.x += bias4.x; v.y += bias4.y; v.z += bias4.z; v.w += bias4.w; }
            else v = make_float4(0.f, 0.f, 0.f, 0.f);
            if constexpr (EPI == 2) {
                if (live) {
                    if (gs != 0.f && cr < pr.fvalid) {
                        const char4 c = *reinterpret_cast<const char4 *>(reinterpret_cast<const signed char *>(pr.c1) + o);
                        v.x = fmaf(gs, (float)c.x, v.x); v.y = fmaf(gs, (float)c.y, v.y); v.z = fmaf(gs, (float)c.z, v.z); v.w = fmaf(gs, (float)c.w, v.w);
                    }
                    const float4 p = *reinterpret_cast<const float4 *>(pr.bias + o);
                    v.x *= cg_gelu_grad(p.x); v.y *= cg_gelu_grad(p.y); v.z *= cg_gelu_grad(p.z); v.w *= cg_gelu_grad(p.w);
                    dbs.x += v.x; dbs.y += v.y; dbs.z += v.z; dbs.w += v.w;
                }
                if (pr.c0) *reinterpret_cast<float4 *>(pr.c0 + o) = v;
                if (pr.ch) {
                    union { uint2 raw; bf16_t e[4]; } hh, ll;
                    const float yy[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        hh.e[e] = (bf16_t)yy[e];
                        ll.e[e] = (bf16_t)(yy[e] - (float)hh.e[e]);
                    }
                    *reinterpret_cast<uint2 *>(pr.ch + o) = hh.raw;
                    *reinterpret_cast<uint2 *>(pr.cl + o) = ll.raw;
                }
                continue;
            }
            *reinterpret_cast<float4 *>(pr.c0 + o) = v;
            if constexpr (ACT) {
                float4 y = make_float4(0.f, 0.f, 0.f, 0.f);
                if (live) y = make_float4(cg_gelu(v.x), cg_gelu(v.y), cg_gelu(v.z), cg_gelu(v.w));
                if (pr.c1) *reinterpret_cast<float4 *>(pr.c1 + o) = y;
                if (pr.ch) {
                    union { uint2 raw; bf16_t e[4]; } hh, ll;
                    const float yy[4] = {y.x, y.y, y.z, y.w};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        hh.e[e] = (bf16_t)yy[e];
                        ll.e[e] = (bf16_t)(yy[e] - (float)hh.e[e]);
                    }
                    *reinterpret_cast<uint2 *>(pr.ch + o) = hh.raw;
                    *reinterpret_cast<uint2 *>(pr.cl + o) = ll.raw;
                }
            }
        }
    }
    if constexpr (EPI == 2) {
        if (pr.db && pr.ntaps > 0) {                   // fold the RPI row groups in LDS, then ONE atomic per column and tile
            __syncthreads();                           // (every thread has left the store loop: the fp32 tile can be overwritten)
            *reinterpret_cast<float4 *>(ct + (tid / C4) * BN + 4 * c4) = dbs;
            __syncthreads();
            if (tid < BN) {
                float t = 0.f;
#pragma unroll
                for (int q = 0; q < RPI; ++q) t += ct[q * BN + tid];
                atomicAdd(pr.db + n0 + tid, t);
            }
        }
    }
    __syncthreads();                                   // the next tile's row table and first stage overwrite this tile's LDS
    }
}

// ---- weight gradient: dW[co, t Cin + c] = sum_m g[m, co] x[xrow(m, t), c] ---------------------------------------------------
// 128 (co) x 128 (c of one tap) output tile, contraction over 32 rows per step.  Both operand tiles are stored as they come
// from memory ([row][128 elements], 256-B rows) and read TRANSPOSED with ds_read_b64_tr_b16: per 16-lane group the instruction
// takes a 4-row x 16-column block (lane 4q + p supplies the address of row q, columns 4p .. 4p+3) and hands lane i column i of
// the 4 rows — exactly 4 consecutive contraction indices of the lane's own output index, i.e. half an MFMA fragment.  The 16-byte
// chunk index of row r is XOR-ed with ((r & 3) << 2) | ((r >> 2) & 3): the four rows of a block then sit 64 B apart in the
// 256-B bank row (conflict-free transposed reads), and the staging stores stay 16-byte.
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;

constexpr int CW_BR = 32;                              // contraction rows per step

struct CwProb {
    const bf16_t *gh, *gl, *xh, *xl;
    float *dw;                 // (splits, Cout, k Cin)
    int M, Q, H;               // rows of g, g positions per sequence (H1), x positions per sequence
    int mchunk;                // contraction rows per split (multiple of 32)
};
struct CwParams {
    CwProb prob[8];
    int nslots, splits, tiles_co, tiles_kc, Cin, Cout, k, stride, pad;
};

// byte offset of 16-byte chunk `ch` of row `row` in an operand tile with ROWB-byte rows
template <int ROWB>
__device__ __forceinline__ int cw_off(const int row, const int ch) {
    if constexpr (ROWB == 64) return row * 64 + (ch << 4);     // 4 rows = one 256-B bank row: a transposed-read block is conflict-free as it lies
    else return row * ROWB + ((ch ^ (((row & 3) << 2) | ((row >> 2) & 3))) << 4);
}

// MF = 32: fragment of v_mfma_f32_32x32x16_bf16 for output index colbase + (lane & 31), contraction rows 16 ks + 8 (lane >> 5) ..;
// MF = 16: fragment of v_mfma_f32_16x16x32_bf16 for output index colbase + (lane & 15), contraction rows 8 (lane >> 4) .. (all 32).
template <int ROWB, int MF>
__device__ __forceinline__ bf16x8 cw_frag(const unsigned char *tile, const int ks, const int colbase, const int lane) {
    const int g = lane >> 4, q = (lane & 15) >> 2, p = lane & 3;
    const int ch = (MF == 32 ? ((colbase + 16 * (g & 1)) >> 3) : (colbase >> 3)) + (p >> 1);
    const int r0 = (MF == 32 ? ks * 16 + 8 * (g >> 1) : 8 * g) + q;
    const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4 *)(tile + cw_off<ROWB>(r0, ch) + 8 * (p & 1)));
    const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4 *)(tile + cw_off<ROWB>(r0 + 4, ch) + 8 * (p & 1)));
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}

// BCO x BKC output tile (output channels x (tap, channel) columns of ONE tap), WM x WN waves of (BCO / WM) x (BKC / WN) each.
// 128 x 128 / 4 waves: 64 KB LDS, two workgroups per CU; 256 x 256 / 8 waves: 128 KB, one per CU (half the staged bytes per MFMA).
template <int BCO, int BKC, int WM, int WN, int MF>
__global__ __launch_bounds__(64 * WM * WN, 2) void conv_mfma_wgrad_kernel(const CwParams P) {
    constexpr int NT = 64 * WM * WN;
    constexpr int WTM = BCO / WM, WTN = BKC / WN, TI = WTM / MF, TJ = WTN / MF;
    using acc_t = typename std::conditional<MF == 32, f32x16, f32x4>::type;
    constexpr int NR = MF == 32 ? 16 : 4;
    constexpr int RBG = BCO * 2, RBX = BKC * 2;                 // row bytes of the g / x tiles
    constexpr int TILE_G = CW_BR * RBG, TILE_X = CW_BR * RBX;
    constexpr int STAGE = 2 * TILE_G + 2 * TILE_X;              // g_hi, g_lo, x_hi, x_lo
    constexpr int CG = BCO / 8, CX = BKC / 8;                   // 16-byte chunks per row
    constexpr int PG = CW_BR * CG / NT, PX = (CW_BR * CX + NT - 1) / NT;   // staging passes (one chunk per thread and pass)
    constexpr bool MID_STORE = MF == 16 && BCO * BKC >= 65536;   // (256 x 256: 567 -> 500 us, 1 020 -> 915 us on the two wide layers; the small tiles measured no gain)
    constexpr bool X_PART = CW_BR * CX < NT;                    // narrow x tile (32 channels): only the first CW_BR * CX threads stage it
    static_assert(PG >= 1 && (CW_BR * CG) % NT == 0 && (X_PART || (CW_BR * CX) % NT == 0), "staging geometry");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN, lr = lane & 31, lh = lane >> 5;

    const int per_slot = P.splits * P.tiles_co * P.tiles_kc;
#pragma unroll 1
    for (int tile = blockIdx.x; tile < P.nslots * per_slot; tile += gridDim.x) {      // (persistent when launched with fewer workgroups than tiles)
    const int t = xcd_remap(tile, P.nslots * per_slot);
    const int slot = t / per_slot;
    int rem = t - slot * per_slot;
    const int split = rem / (P.tiles_co * P.tiles_kc);
    rem -= split * (P.tiles_co * P.tiles_kc);
    const int tco = rem / P.tiles_kc, tkc = rem - tco * P.tiles_kc;
    const CwProb &pr = P.prob[slot];
    const int Cin = P.Cin, Cout = P.Cout;
    const int co0 = tco * BCO, kc0 = tkc * BKC;
    const int tap = kc0 / Cin, c0 = kc0 - tap * Cin;
    const int mbeg = split * pr.mchunk, mend = min(pr.M, mbeg + pr.mchunk);
    const int nk = mend > mbeg ? (mend - mbeg + CW_BR - 1) / CW_BR : 0;

    // (register staging: an LDS-DMA version of this loop measured 20-25 % SLOWER, profiles/r04_convgemm_microbench_v1.log)
    // pass p of the g tile: chunk (tid + p NT) -> row (tid + p NT) / CG, chunk % CG; likewise for x with CX
    int g_row[PG], g_ch[PG], g_m[PG];
    int x_row[PX], x_ch[PX], x_m[PX], x_seq[PX], x_q[PX];
#pragma unroll
    for (int i = 0; i < PG; ++i) {
        const int idx = tid + i * NT;
        g_row[i] = idx / CG; g_ch[i] = idx % CG;
        g_m[i] = mbeg + g_row[i];
    }
#pragma unroll
    for (int i = 0; i < PX; ++i) {
        const int idx = tid + i * NT;
        x_row[i] = (idx / CX) % CW_BR; x_ch[i] = idx % CX;
        x_m[i] = mbeg + x_row[i];
        x_seq[i] = x_m[i] / pr.Q;
        x_q[i] = x_m[i] - x_seq[i] * pr.Q;
    }
    u32x4 rg_h[PG], rg_l[PG], rx_h[PX], rx_l[PX];
    bool g_ok[PG], x_ok[PX];
    const u32x4 z4 = {0u, 0u, 0u, 0u};

    acc_t acc[TI][TJ];
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j)
#pragma unroll
            for (int r = 0; r < NR; ++r) acc[i][j][r] = 0.f;

#define CW_LOAD()                                                                                              \
    do {                                                                                                       \
        _Pragma("unroll") for (int i = 0; i < PG; ++i) {                                                       \
            g_ok[i] = g_m[i] < mend;                                                                           \
            const size_t o_ = (size_t)(g_ok[i] ? g_m[i] : 0) * Cout + co0 + g_ch[i] * 8;                       \
            rg_h[i] = *reinterpret_cast<const u32x4 *>(pr.gh + o_);                                            \
            rg_l[i] = *reinterpret_cast<const u32x4 *>(pr.gl + o_);                                            \
            g_m[i] += CW_BR;                                                                                   \
        }                                                                                                      \
        _Pragma("unroll") for (int i = 0; i < PX; ++i) {                                                       \
            if (X_PART && tid >= CW_BR * CX) break;                                                            \
            const int p_ = x_q[i] * P.stride + tap - P.pad;                                                    \
            x_ok[i] = x_m[i] < mend && p_ >= 0 && p_ < pr.H;                                                   \
            const size_t o_ = (size_t)(x_ok[i] ? x_seq[i] * pr.H + p_ : 0) * Cin + c0 + x_ch[i] * 8;           \
            rx_h[i] = *reinterpret_cast<const u32x4 *>(pr.xh + o_);                                            \
            rx_l[i] = *reinterpret_cast<const u32x4 *>(pr.xl + o_);                                            \
            x_m[i] += CW_BR; x_q[i] += CW_BR;                                                                  \
            while (x_q[i] >= pr.Q) { x_q[i] -= pr.Q; ++x_seq[i]; }                                             \
        }                                                                                                      \
    } while (0)
#define CW_STORE(buf)                                                                                          \
    do {                                                                                                       \
        unsigned char *s_ = smem + (buf) * STAGE;                                                              \
        _Pragma("unroll") for (int i = 0; i < PG; ++i) {                                                       \
            const int o_ = cw_off<RBG>(g_row[i], g_ch[i]);                                                     \
            *reinterpret_cast<u32x4 *>(s_ + o_) = g_ok[i] ? rg_h[i] : z4;                                      \
            *reinterpret_cast<u32x4 *>(s_ + TILE_G + o_) = g_ok[i] ? rg_l[i] : z4;                             \
        }                                                                                                      \
        _Pragma("unroll") for (int i = 0; i < PX; ++i) {                                                       \
            if (X_PART && tid >= CW_BR * CX) break;                                                            \
            const int o_ = cw_off<RBX>(x_row[i], x_ch[i]);                                                     \
            *reinterpret_cast<u32x4 *>(s_ + 2 * TILE_G + o_) = x_ok[i] ? rx_h[i] : z4;                         \
            *reinterpret_cast<u32x4 *>(s_ + 2 * TILE_G + TILE_X + o_) = x_ok[i] ? rx_l[i] : z4;                \
        }                                                                                                      \
    } while (0)

    if (nk > 0) {
        CW_LOAD();
        CW_STORE(0);
    }
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        const bool more = kt + 1 < nk;
        if (more) CW_LOAD();
        const unsigned char *s = smem + buf * STAGE;
        if constexpr (MF == 32) {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                bf16x8 ah[TI], al[TI], bh[TJ], bl[TJ];
#pragma unroll
                for (int i = 0; i < TI; ++i) {
                    ah[i] = cw_frag<RBG, 32>(s, ks, wm * WTM + i * 32, lane);
                    al[i] = cw_frag<RBG, 32>(s + TILE_G, ks, wm * WTM + i * 32, lane);
                }
#pragma unroll
                for (int j = 0; j < TJ; ++j) {
                    bh[j] = cw_frag<RBX, 32>(s + 2 * TILE_G, ks, wn * WTN + j * 32, lane);
                    bl[j] = cw_frag<RBX, 32>(s + 2 * TILE_G + TILE_X, ks, wn * WTN + j * 32, lane);
                }
#pragma unroll
                for (int i = 0; i < TI; ++i)
#pragma unroll
                    for (int j = 0; j < TJ; ++j) {
                        acc[i][j] = cg_mfma(al[i], bh[j], acc[i][j]);
                        acc[i][j] = cg_mfma(ah[i], bl[j], acc[i][j]);
                        acc[i][j] = cg_mfma(ah[i], bh[j], acc[i][j]);
                    }
            }
        } else {
            bf16x8 bh[TJ], bl[TJ];
#pragma unroll
            for (int j = 0; j < TJ; ++j) {
                bh[j] = cw_frag<RBX, 16>(s + 2 * TILE_G, 0, wn * WTN + j * 16, lane);
                bl[j] = cw_frag<RBX, 16>(s + 2 * TILE_G + TILE_X, 0, wn * WTN + j * 16, lane);
            }
#pragma unroll
            for (int i = 0; i < TI; ++i) {
                // the next stage's registers go to LDS in the MIDDLE of the MFMA sequence (their loads were issued a half step ago;
                // nobody reads buf ^ 1 in this step): the ds_writes run under the second half's MFMAs instead of after them
                if (MID_STORE && i == TI / 2 && more) CW_STORE(buf ^ 1);
                const bf16x8 ah = cw_frag<RBG, 16>(s, 0, wm * WTM + i * 16, lane);
                const bf16x8 al = cw_frag<RBG, 16>(s + TILE_G, 0, wm * WTM + i * 16, lane);
#pragma unroll
                for (int j = 0; j < TJ; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh[j], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl[j], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh[j], acc[i][j], 0, 0, 0);
                }
            }
        }
        if (!MID_STORE && more) CW_STORE(buf ^ 1);
        __syncthreads();
    }
#undef CW_LOAD
#undef CW_STORE

    const int KC = P.k * Cin;
    float *dw = pr.dw + (size_t)split * Cout * KC;
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j) {
            const int col = kc0 + wn * WTN + j * MF + (MF == 32 ? lr : (lane & 15));
#pragma unroll
            for (int r = 0; r < NR; ++r) {
                const int row = co0 + wm * WTM + i * MF + (MF == 32 ? (r & 3) + 8 * (r >> 2) + 4 * lh : 4 * (lane >> 4) + r);
                dw[(size_t)row * KC + col] = acc[i][j][r];
            }
        }
    __syncthreads();                                   // (every wave is out of the last stage before the next tile's first store)
    }
}

}  // namespace
}  // namespace vmasr

using namespace vmasr;

namespace {

// CUs the convolution kernels may occupy (0: all).  The two-stream train step (trainer._backward_two) sets it while the generator's
// small kernels run beside them: a 256 x 256 tile holds a CU's LDS for ~300 us, and the dispatcher hands freed CUs to the running grid's
// next workgroup first, so without spare CUs the other stream's kernels start late.
int g_cg_cu_limit = 0;

int cg_grid(int tiles, size_t smem) {
    if (g_cg_cu_limit <= 0) return tiles;
    // Workgroups per CU by LDS; the limit is soft by 24 CUs where that saves a whole round of tiles (200 or 400 tiles under a limit of
    // 192: 200 workgroups -> 1 / 2 rounds instead of 2 / 3), and the grid is the SMALLEST that keeps the number of rounds (480 tiles:
    // 3 rounds either way -> 160 workgroups, not 192: the other stream gets the CUs that would idle in a ragged last round).
    const int per_cu = std::max(1, (int)((160 * 1024) / std::max<size_t>(smem, 1)));
    static const int slack = [] { const char *e = getenv("VMASR_CONV_CU_SLACK"); return e ? atoi(e) : 24; }();
    const int cap = std::min(256, g_cg_cu_limit + slack) * per_cu;
    const int rounds = (tiles + cap - 1) / cap;
    const int g = ((tiles + rounds - 1) / rounds + 7) / 8 * 8;
    return std::min(tiles, std::min(g, cap));
}

template <int BM, int BN, int WM, int WN, int MF, int OPS = 0>
int cg_launch_cfg(CgParams &P, int epi, hipStream_t st, int kid, double bytes) {
    int tiles = 0;
    P.ntiles_n = P.NB / BN;
    for (int i = 0; i < P.nprob; ++i) {
        P.prob[i].tile_start = tiles;
        P.prob[i].mtiles = (std::max(P.prob[i].M, P.prob[i].zero_rows) + BM - 1) / BM;
        tiles += P.prob[i].mtiles * P.ntiles_n;
    }
    P.total_tiles = tiles;
    if (tiles == 0) return 0;
    constexpr size_t smem = 2 * (2 * BM * 64 + 2 * BN * 64) + BM * sizeof(int);
    constexpr bool HAS_EPI2 = BN >= 128 && OPS == 0;   // (the 32-wide tile and the f32 form only serve the 32 -> 128 layer)
    static bool attr_done = false;
    if (!attr_done && smem > 65536) {      // > 64 KB of dynamic LDS needs the opt-in attribute
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&conv_mfma_nt_kernel<BM, BN, WM, WN, 1, MF, OPS>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&conv_mfma_nt_kernel<BM, BN, WM, WN, 0, MF, OPS>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        if constexpr (HAS_EPI2)
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&conv_mfma_nt_kernel<BM, BN, WM, WN, 2, MF, OPS>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        attr_done = true;
    }
    if (epi == 1) {
        VMASR_LAUNCH(kid, bytes, (conv_mfma_nt_kernel<BM, BN, WM, WN, 1, MF, OPS>), dim3(cg_grid(tiles, smem)), dim3(64 * WM * WN), smem, st, P);
    } else if (epi == 2) {
        if constexpr (HAS_EPI2) {
            VMASR_LAUNCH(kid, bytes, (conv_mfma_nt_kernel<BM, BN, WM, WN, 2, MF, OPS>), dim3(cg_grid(tiles, smem)), dim3(64 * WM * WN), smem, st, P);
        } else {
            set_error("conv_mfma: no fused activation backward for this tile");
            return VMASR_EINVAL;
        }
    } else {
        VMASR_LAUNCH(kid, bytes, (conv_mfma_nt_kernel<BM, BN, WM, WN, 0, MF, OPS>), dim3(cg_grid(tiles, smem)), dim3(64 * WM * WN), smem, st, P);
    }
    return check_launch("conv_mfma");
}

// 256 x 256 tiles when the output width allows and there are enough of them to fill the chip; VMASR_CONV_TILE=128 forces the small tile
int cg_launch(CgParams &P, int epi, hipStream_t st, int kid, double bytes) {
    static const int forced = [] { const char *e = getenv("VMASR_CONV_TILE"); return e ? atoi(e) : 0; }();
    // v_mfma_f32_16x16x32_bf16 by default: same cycles per FLOP as 32x32x16, but the chip holds a higher clock on it under load
    // (MI355X_MICROARCH.md, DVFS give-back item 7): 5-9 % less time on the 512 -> 1024 and 1024 -> 1024 layers, forward and dgrad
    // (profiles/r04_convgemm_microbench_v4.log); VMASR_CONV_MFMA=32 selects the 32x32x16 form
    static const int mf = [] { const char *e = getenv("VMASR_CONV_MFMA"); return e ? atoi(e) : 16; }();
    if (P.NB % 256 == 0 && forced != 128)
        return mf == 16 ? cg_launch_cfg<256, 256, 2, 4, 16>(P, epi, st, kid, bytes) : cg_launch_cfg<256, 256, 2, 4, 32>(P, epi, st, kid, bytes);
    // 256 x 128 (per wave 64 x 64): only for the 32-channel input side (K = 160: five K steps) — on the 128 -> 512 layer's dgrad it measured
    // SLOWER than 128 x 128 with two workgroups per CU (317 vs 279 us, profiles/r04_convgemm_microbench_v6_b4.log)
    if (P.NB % 128 == 0 && P.CA == 32 && forced != 128) return cg_launch_cfg<256, 128, 4, 2, 16>(P, epi, st, kid, bytes);
    if (P.NB == 32) return cg_launch_cfg<256, 32, 8, 1, 16>(P, epi, st, kid, bytes);                            // per wave 32 x 32 (the 32-channel side)
    return mf == 16 ? cg_launch_cfg<128, 128, 2, 2, 16>(P, epi, st, kid, bytes) : cg_launch_cfg<128, 128, 2, 2, 32>(P, epi, st, kid, bytes);
}

}  // namespace

VMASR_EXPORT void vmasr_conv_set_cu_limit(int32_t cus) { g_cg_cu_limit = cus < 0 ? 0 : cus; }
VMASR_EXPORT int32_t vmasr_conv_get_cu_limit(void) { return g_cg_cu_limit; }

VMASR_EXPORT int vmasr_conv_mfma_supported(int32_t Cin, int32_t Cout, int32_t k, int32_t stride) {
    // channel counts: multiples of 128, or exactly 32 on the input side (the 32 -> 128 layer: 256 x 32 / 128 x 32 tile configurations)
    const bool in_ok = Cin > 0 && (Cin % 128 == 0 || Cin == 32), out_ok = Cout > 0 && Cout % 128 == 0;
    return in_ok && out_ok && k >= 1 && k <= 8 && stride >= 1 && stride <= 3;
}

// Capability query of a whole stacked launch: the shape AND what the three launchers bound — slots per launch (forward 24, the input
// gradient 24 / (stride + 1) because every slot becomes `stride` residue-class problems plus a zero-fill one, the weight gradient 8)
// and the rows a launch addresses.  A caller dispatches on THIS, so that an MPD built with more periods, or a batch / segment beyond
// the row bound, takes its split-GEMM path instead of failing with VMASR_EINVAL in the middle of a backward pass.
VMASR_EXPORT int vmasr_conv_mfma_supported_launch(int32_t Cin, int32_t Cout, int32_t k, int32_t stride, int32_t n, int64_t rows) {
    if (!vmasr_conv_mfma_supported(Cin, Cout, k, stride)) return 0;
    if (n < 1 || n > 8 || n * (stride + 1) > CG_MAXP) return 0;
    const int64_t widest = std::max<int64_t>(std::max(Cin, Cout), (int64_t)k * std::max(Cin, Cout));
    return rows >= 1 && rows < (1LL << 31) / widest ? 1 : 0;
}

namespace {
// ops: 0 = bf16 (hi, lo) pairs (slots' ah / al, bh / bl), 1 = fp32 operands (ah = x fp32, bh = W fp32; al / bl unused): conv_mfma_nt_kernel OPS
int cg_fwd(const vmasr_cg_slot *slots, int32_t n, int32_t Cin, int32_t Cout, int32_t k, int32_t stride, int32_t pad, int64_t rows_out,
           int32_t act, vmasr_stream_t stream, int ops) {
    const char *what = ops ? "conv_f32_fwd" : "conv_mfma_fwd";
    VMASR_REQUIRE(slots && n >= 1 && n <= CG_MAXP, VMASR_EINVAL, "%s: 1..%d slots", what, CG_MAXP);
    VMASR_REQUIRE(vmasr_conv_mfma_supported(Cin, Cout, k, stride) && (!ops || Cin == 32), VMASR_EINVAL,
                  "%s: unsupported shape (Cin %d, Cout %d, k %d, stride %d)", what, Cin, Cout, k, stride);
    VMASR_REQUIRE(rows_out % 256 == 0, VMASR_EINVAL, "%s: rows_out must be a multiple of 256", what);
    CgParams P = {};
    P.nprob = n;
    P.CA = Cin; P.NB = Cout; P.KB = k * Cin;
    double bytes = 0;
    for (int i = 0; i < n; ++i) {
        const vmasr_cg_slot &s = slots[i];
        VMASR_REQUIRE(s.ah && s.bh && s.c0 && (ops || (s.al && s.bl)), VMASR_EINVAL, "%s: null tensor in slot %d", what, i);
        VMASR_REQUIRE(!ops || (aligned_to(s.ah, 16) && aligned_to(s.bh, 16)), VMASR_EALIGN, "%s: slot %d: operands must be 16-byte aligned", what, i);
        const int64_t H1 = ((int64_t)s.H + 2 * pad - k) / stride + 1;
        const int64_t M = s.nseq * H1;
        VMASR_REQUIRE(H1 >= 1 && M <= rows_out && rows_out < (1LL << 30) / std::max(Cout, k * Cin), VMASR_EINVAL,
                      "%s: slot %d: %lld rows do not fit rows_out %lld (or 32-bit offsets)", what, i, (long long)M, (long long)rows_out);
        CgProb &p = P.prob[i];
        p.ah = (const bf16_t *)s.ah; p.al = (const bf16_t *)s.al; p.bh = (const bf16_t *)s.bh; p.bl = (const bf16_t *)s.bl;
        p.c0 = s.c0; p.c1 = s.c1; p.ch = (bf16_t *)s.ch; p.cl = (bf16_t *)s.cl; p.bias = s.bias;
        p.M = (int)M; p.Q = (int)H1; p.HA = s.H; p.HC = (int)H1;
        p.hq_mul = stride; p.hq_add = -pad; p.crow_mul = 1; p.crow_add = 0;
        p.ntaps = k; p.tap0 = 0; p.tap_step = 1; p.dstep = 1;
        p.zero_rows = (int)rows_out;
        bytes += (double)s.nseq * s.H * Cin * 4 + (double)Cout * k * Cin * 4 + (double)M * Cout * (act ? 12 : 4);
    }
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (ops) {      // 128 x 128 / 4 waves, two workgroups per CU: the layer is as much output-write- as MFMA-bound
        VMASR_REQUIRE(Cout % 128 == 0, VMASR_EINVAL, "%s: Cout %% 128", what);
        static const int tile = [] { const char *e = getenv("VMASR_CONV_F32_TILE"); return e ? atoi(e) : 128; }();
        if (tile == 256) return cg_launch_cfg<256, 128, 4, 2, 32, 1>(P, act != 0 ? 1 : 0, st, VMASR_K_CONV_MFMA_FWD, bytes);
        return cg_launch_cfg<128, 128, 2, 2, 32, 1>(P, act != 0 ? 1 : 0, st, VMASR_K_CONV_MFMA_FWD, bytes);
    }
    return cg_launch(P, act != 0 ? 1 : 0, st, VMASR_K_CONV_MFMA_FWD, bytes);
}
}  // namespace

VMASR_EXPORT int vmasr_conv_mfma_fwd(const vmasr_cg_slot *slots, int32_t n, int32_t Cin, int32_t Cout, int32_t k, int32_t stride,
                                     int32_t pad, int64_t rows_out, int32_t act, vmasr_stream_t stream) {
    return cg_fwd(slots, n, Cin, Cout, k, stride, pad, rows_out, act, stream, 0);
}

// The same stacked convolution with FP32 operands and exact-f32 products (v_mfma_f32_32x32x2_f32): slots' ah = x (fp32 rows of Cin),
// bh = W (Cout, k Cin) fp32 in (tap, channel) order; al / bl are ignored.  Cin = 32 (the 32 -> 128 layer of model/discriminator.py:40-60).
VMASR_EXPORT int vmasr_conv_f32_fwd(const vmasr_cg_slot *slots, int32_t n, int32_t Cin, int32_t Cout, int32_t k, int32_t stride,
                                    int32_t pad, int64_t rows_out, int32_t act, vmasr_stream_t stream) {
    return cg_fwd(slots, n, Cin, Cout, k, stride, pad, rows_out, act, stream, 1);
}

namespace {

// The input gradient's problems (one per slot and residue class of the stride, plus one zero-fill problem per slot).  epi: the fused
// activation backward of the layer below (vmasr_conv_mfma_dgrad_gelu), or null.
int cg_dgrad(const vmasr_cg_slot *slots, const vmasr_cg_gelu_bwd *epi, const float *gtok, int32_t n, int32_t Cin, int32_t Cout, int32_t k,
             int32_t stride, int32_t pad, int64_t rows_in, vmasr_stream_t stream, int ops = 0) {
    const char *what = ops ? "conv_f32_dgrad" : epi ? "conv_mfma_dgrad_gelu" : "conv_mfma_dgrad";
    VMASR_REQUIRE(!ops || (!epi && Cin == 32), VMASR_EINVAL, "%s: the f32 form serves Cin = 32 without a fused activation backward", what);
    VMASR_REQUIRE(slots && n >= 1 && n * (stride + 1) <= CG_MAXP, VMASR_EINVAL, "%s: too many slots", what);
    VMASR_REQUIRE(vmasr_conv_mfma_supported(Cin, Cout, k, stride), VMASR_EINVAL, "%s: unsupported shape (Cin %d, Cout %d, k %d, stride %d)", what,
                  Cin, Cout, k, stride);
    VMASR_REQUIRE(!epi || Cin % 128 == 0, VMASR_EINVAL, "%s: the fused activation backward needs Cin %% 128 == 0", what);
    CgParams P = {};
    P.CA = Cout; P.NB = Cin; P.KB = k * Cout;
    P.gtok = gtok;
    double bytes = 0;
    int np = 0;
    for (int i = 0; i < n; ++i) {
        const vmasr_cg_slot &s = slots[i];            // here: A = g (nseq * H1, Cout), B = W^T (Cin, k * Cout), c0 = dx (rows_in, Cin), H = H_in
        VMASR_REQUIRE(s.ah && s.bh && (ops || (s.al && s.bl)) && (epi ? (s.c0 || (s.ch && s.cl)) && (!s.ch == !s.cl) : s.c0 != nullptr), VMASR_EINVAL,
                      "%s: null tensor in slot %d", what, i);
        VMASR_REQUIRE(!ops || (aligned_to(s.ah, 16) && aligned_to(s.bh, 16)), VMASR_EALIGN, "%s: slot %d: operands must be 16-byte aligned", what, i);
        VMASR_REQUIRE(!epi || (epi[i].pre && epi[i].valid >= 0 && (!epi[i].sgn || gtok)), VMASR_EINVAL,
                      "%s: slot %d: pre-activation missing (or a sign map without the loss gradient)", what, i);
        const int64_t H = s.H, H1 = (H + 2 * pad - k) / stride + 1;
        VMASR_REQUIRE(H1 >= 1 && s.nseq * H <= rows_in && rows_in < (1LL << 31) / std::max(Cin, k * Cout), VMASR_EINVAL,
                      "%s: slot %d: %lld rows do not fit rows_in %lld (or 32-bit offsets)", what, i, (long long)(s.nseq * H), (long long)rows_in);
        auto outputs = [&](CgProb &p) {
            p.c0 = s.c0; p.c1 = nullptr; p.ch = nullptr; p.cl = nullptr; p.bias = nullptr;
            if (epi) {
                p.ch = (bf16_t *)s.ch; p.cl = (bf16_t *)s.cl;
                p.bias = epi[i].pre;
                p.c1 = reinterpret_cast<float *>(const_cast<signed char *>(epi[i].sgn));
                p.fscale = epi[i].scale;
                p.fvalid = (int)std::min<int64_t>(epi[i].valid, rows_in);
                p.db = epi[i].db;
            }
        };
        for (int r = 0; r < stride; ++r) {
            const int r0 = ((r - pad) % stride + stride) % stride;          // input positions h = stride q + r0 have (h + pad) % stride == r
            const int64_t Q = H > r0 ? (H - r0 + stride - 1) / stride : 0;
            if (Q == 0) continue;
            CgProb &p = P.prob[np++];
            p.ah = (const bf16_t *)s.ah; p.al = (const bf16_t *)s.al; p.bh = (const bf16_t *)s.bh; p.bl = (const bf16_t *)s.bl;
            outputs(p);
            p.M = (int)(s.nseq * Q); p.Q = (int)Q; p.HA = (int)H1; p.HC = (int)H;
            p.hq_mul = 1; p.hq_add = (r0 + pad - r) / stride;               // g position of tap t = r + stride j: q + hq_add - j
            p.crow_mul = stride; p.crow_add = r0;
            p.ntaps = r < k ? (k - r + stride - 1) / stride : 0;
            p.tap0 = r; p.tap_step = stride; p.dstep = -1;
            p.zero_rows = 0;
        }
        if (s.nseq * H < rows_in) {                                         // zero rows below the slot's data: one pseudo sequence, no taps
            CgProb &p = P.prob[np++];
            p.ah = (const bf16_t *)s.ah; p.al = (const bf16_t *)s.al; p.bh = (const bf16_t *)s.bh; p.bl = (const bf16_t *)s.bl;
            outputs(p);
            p.M = (int)(rows_in - s.nseq * H); p.Q = p.M; p.HA = 1; p.HC = 0;
            p.hq_mul = 0; p.hq_add = 0; p.crow_mul = 1; p.crow_add = (int)(s.nseq * H);
            p.ntaps = 0; p.tap0 = 0; p.tap_step = 1; p.dstep = 0;
            p.zero_rows = 0;
        }
        bytes += (double)s.nseq * H1 * Cout * 4 + (double)Cout * k * Cin * 4 + (double)s.nseq * H * Cin * 4;
        if (epi) bytes += (double)s.nseq * H * Cin * ((s.c0 && s.ch ? 4.0 : 0.0) + 4.0 + (epi[i].sgn ? 1.0 : 0.0));
    }
    P.nprob = np;
    if (ops) return cg_launch_cfg<256, 32, 8, 1, 32, 1>(P, 0, static_cast<hipStream_t>(stream), VMASR_K_CONV_MFMA_DGRAD, bytes);
    return cg_launch(P, epi ? 2 : 0, static_cast<hipStream_t>(stream), VMASR_K_CONV_MFMA_DGRAD, bytes);
}

}  // namespace

VMASR_EXPORT int vmasr_conv_mfma_dgrad(const vmasr_cg_slot *slots, int32_t n, int32_t Cin, int32_t Cout, int32_t k, int32_t stride,
                                       int32_t pad, int64_t rows_in, vmasr_stream_t stream) {
    return cg_dgrad(slots, nullptr, nullptr, n, Cin, Cout, k, stride, pad, rows_in, stream);
}

// Input gradient with FP32 operands and exact-f32 products: slots' ah = g (fp32 rows of Cout), bh = W^T (Cin, k Cout) fp32 in (tap, output
// channel) order, c0 = dx (rows_in, Cin) fp32.  Cin = 32: no GEMM over a materialised column operand, no col2im.
VMASR_EXPORT int vmasr_conv_f32_dgrad(const vmasr_cg_slot *slots, int32_t n, int32_t Cin, int32_t Cout, int32_t k, int32_t stride,
                                      int32_t pad, int64_t rows_in, vmasr_stream_t stream) {
    return cg_dgrad(slots, nullptr, nullptr, n, Cin, Cout, k, stride, pad, rows_in, stream, 1);
}

VMASR_EXPORT int vmasr_conv_mfma_dgrad_gelu(const vmasr_cg_slot *slots, const vmasr_cg_gelu_bwd *epi, const float *gtok, int32_t n, int32_t Cin,
                                            int32_t Cout, int32_t k, int32_t stride, int32_t pad, int64_t rows_in, vmasr_stream_t stream) {
    VMASR_REQUIRE(epi, VMASR_EINVAL, "conv_mfma_dgrad_gelu: epi is required");
    return cg_dgrad(slots, epi, gtok, n, Cin, Cout, k, stride, pad, rows_in, stream);
}

VMASR_EXPORT int vmasr_conv_mfma_wgrad(const vmasr_cg_slot *slots, int32_t n, int32_t Cin, int32_t Cout, int32_t k, int32_t stride,
                                       int32_t pad, int32_t splits, vmasr_stream_t stream) {
    VMASR_REQUIRE(slots && n >= 1 && n <= 8, VMASR_EINVAL, "conv_mfma_wgrad: 1..8 slots");
    VMASR_REQUIRE(vmasr_conv_mfma_supported(Cin, Cout, k, stride) && splits >= 1 && splits <= 64, VMASR_EINVAL,
                  "conv_mfma_wgrad: unsupported shape (Cin %d, Cout %d, k %d, stride %d, splits %d)", Cin, Cout, k, stride, splits);
    CwParams P = {};
    static const int forced = [] { const char *e = getenv("VMASR_CONV_TILE"); return e ? atoi(e) : 0; }();
    static const int mf = [] { const char *e = getenv("VMASR_CONV_MFMA"); return e ? atoi(e) : 16; }();
    // 256 x 256 / 8 waves when both channel counts allow (with the 16x16x32 MFMA form: 597 -> 504-518 us on the 512 -> 1024 layer, 1 012 -> 945-960 us
    // on 1024 -> 1024, profiles/r04_convgemm_microbench_v5.log; with the 32x32x16 form it had measured no gain); VMASR_CONV_TILE=128 forces the small tile
    const bool big = Cout % 256 == 0 && Cin % 256 == 0 && forced != 128;
    const bool narrow = Cin == 32;                       // the 32 -> 128 layer: 128 x 32 tiles (one tap's 32 channels)
    const int T = big ? 256 : 128, TK = narrow ? 32 : T;
    P.nslots = n; P.splits = splits; P.tiles_co = Cout / T; P.tiles_kc = k * Cin / TK;
    P.Cin = Cin; P.Cout = Cout; P.k = k; P.stride = stride; P.pad = pad;
    double bytes = 0;
    for (int i = 0; i < n; ++i) {
        const vmasr_cg_slot &s = slots[i];            // ah/al = g pair, bh/bl = x pair, c0 = dW partials, H = input positions
        VMASR_REQUIRE(s.ah && s.al && s.bh && s.bl && s.c0, VMASR_EINVAL, "conv_mfma_wgrad: null tensor in slot %d", i);
        const int64_t H1 = ((int64_t)s.H + 2 * pad - k) / stride + 1, M = s.nseq * H1;
        VMASR_REQUIRE(H1 >= 1 && M < (1LL << 31) / std::max(Cout, Cin) && s.nseq * (int64_t)s.H < (1LL << 31) / Cin, VMASR_EINVAL,
                      "conv_mfma_wgrad: slot %d too large for 32-bit row offsets", i);
        CwProb &p = P.prob[i];
        p.gh = (const bf16_t *)s.ah; p.gl = (const bf16_t *)s.al; p.xh = (const bf16_t *)s.bh; p.xl = (const bf16_t *)s.bl;
        p.dw = s.c0; p.M = (int)M; p.Q = (int)H1; p.H = s.H;
        const int64_t per = (M + splits - 1) / splits;
        p.mchunk = (int)((per + CW_BR - 1) / CW_BR * CW_BR);
        bytes += (double)M * Cout * 4 + (double)s.nseq * s.H * Cin * 4 + (double)splits * Cout * k * Cin * 4;
    }
    const int tiles = n * splits * P.tiles_co * P.tiles_kc;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (big) {
        constexpr size_t smem = 2 * 4 * CW_BR * 512;
        static bool attr_done = false;
        if (!attr_done) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&conv_mfma_wgrad_kernel<256, 256, 2, 4, 16>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
            attr_done = true;
        }
        VMASR_LAUNCH(VMASR_K_CONV_MFMA_WGRAD, bytes, (conv_mfma_wgrad_kernel<256, 256, 2, 4, 16>), dim3(cg_grid(tiles, smem)), dim3(512), smem, st, P);
    } else if (narrow) {
        VMASR_LAUNCH(VMASR_K_CONV_MFMA_WGRAD, bytes, (conv_mfma_wgrad_kernel<128, 32, 4, 1, 16>), dim3(cg_grid(tiles, 2 * (2 * CW_BR * 256 + 2 * CW_BR * 64))), dim3(256), 2 * (2 * CW_BR * 256 + 2 * CW_BR * 64), st, P);
    } else if (mf == 32) {
        VMASR_LAUNCH(VMASR_K_CONV_MFMA_WGRAD, bytes, (conv_mfma_wgrad_kernel<128, 128, 2, 2, 32>), dim3(cg_grid(tiles, 2 * 4 * CW_BR * 256)), dim3(256), 2 * 4 * CW_BR * 256, st, P);
    } else {
        VMASR_LAUNCH(VMASR_K_CONV_MFMA_WGRAD, bytes, (conv_mfma_wgrad_kernel<128, 128, 2, 2, 16>), dim3(cg_grid(tiles, 2 * 4 * CW_BR * 256)), dim3(256), 2 * 4 * CW_BR * 256, st, P);
    }
    return check_launch("conv_mfma_wgrad");
}
