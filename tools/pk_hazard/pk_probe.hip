// pk_probe — is a packed-fp32 instruction pair bit-exact beside another stream's kernels?  (gfx950, round-6 determinism hunt)
//
// tests/test_determinism.py failed on the driver's box in round 5 because ONE partial sum of vmasr small_linear_bwd<bf16,bf16,1,4>
// differed between two evaluations — only in the two-stream step, only the LOW element of a v_pk_fma_f32 accumulator pair
// (profiles/r06_determinism_hunt.md).  This probe runs the instruction patterns of that kernel from inline asm against the same
// arithmetic done with scalar v_fma_f32 / v_add_f32, alone and beside MFMA / VALU / memory kernels on a second stream, and counts
// lanes whose results differ.   build: hipcc --offload-arch=gfx950 -O2 -o pk_probe pk_probe.hip     run: ./pk_probe [rounds]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef short s4v __attribute__((ext_vector_type(4)));

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)

__device__ __forceinline__ float seedf(unsigned h) {     // small exact-ish floats in (-1, 1)
    h ^= h >> 16; h *= 0x7feb352du; h ^= h >> 15; h *= 0x846ca68bu; h ^= h >> 16;
    return (float)((int)(h & 0xffff) - 32768) * (1.f / 32768.f);
}

// MODE 0: v_pk_fma_f32 acc, g, x, acc ; v_pk_add_f32 g, g, d      (the next instruction overwrites the FMA's source pair)
// MODE 1: the same with s_nop 0 between them
// MODE 2: v_pk_fma_f32 acc, g, x, acc ; v_pk_add_f32 t, g, d      (no overwrite; t folded in later)
// MODE 3: v_pk_add_f32 t, a, g ; v_pk_fma_f32 g, b, x, c          (a pair written right after it was read as an addend)
// MODE 4: MODE 0 with v_pk_mul/add replaced by scalar ops on the packed side too (control: must never differ)
template <int MODE>
__global__ __launch_bounds__(256) void probe(unsigned *bad, unsigned long long *worst, int iters, unsigned salt) {
    const unsigned t = blockIdx.x * blockDim.x + threadIdx.x;
    unsigned nbad = 0;
    for (int rep = 0; rep < iters; ++rep) {
        const unsigned h = (t * 9781u + rep * 6271u) ^ salt;
        f2 g = {seedf(h), seedf(h + 1)}, x = {seedf(h + 2), seedf(h + 3)}, d = {seedf(h + 4) * 0.125f, seedf(h + 5) * 0.125f}, acc = {0.f, 0.f};
        f2 tt = {0.f, 0.f};
        float rg0 = g.x, rg1 = g.y, ra0 = 0.f, ra1 = 0.f, rt0 = 0.f, rt1 = 0.f;
        const float x0 = x.x, d0 = d.x, d1 = d.y;
#pragma unroll 1
        for (int k = 0; k < 32; ++k) {
            if (MODE == 0)
                asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]\n\tv_pk_add_f32 %1, %1, %3" : "+v"(acc), "+v"(g) : "v"(x), "v"(d));
            else if (MODE == 1)
                asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]\n\ts_nop 0\n\tv_pk_add_f32 %1, %1, %3" : "+v"(acc), "+v"(g) : "v"(x), "v"(d));
            else if (MODE == 2) {
                asm volatile("v_pk_fma_f32 %0, %2, %3, %0 op_sel_hi:[1,0,1]\n\tv_pk_add_f32 %1, %2, %4" : "+v"(acc), "=v"(tt) : "v"(g), "v"(x), "v"(d));
                g = tt;
            } else if (MODE == 3) {
                // tt = d + g ; g = g*x0 + d   (g read as an addend, then overwritten by the next packed op)
                asm volatile("v_pk_add_f32 %0, %2, %1\n\tv_pk_fma_f32 %1, %1, %3, %2 op_sel_hi:[1,0,1]" : "=&v"(tt), "+v"(g) : "v"(d), "v"(x));
                acc += tt;
            } else {
                asm volatile("v_fma_f32 %0, %2, %4, %0\n\tv_fma_f32 %1, %3, %4, %1\n\tv_add_f32 %2, %2, %5\n\tv_add_f32 %3, %3, %6"
                             : "+v"(acc.x), "+v"(acc.y), "+v"(g.x), "+v"(g.y) : "v"(x0), "v"(d0), "v"(d1));
            }
            // reference: scalar VALU ops
            if (MODE == 3) {
                asm volatile("v_add_f32 %0, %4, %2\n\tv_add_f32 %1, %5, %3\n\tv_fma_f32 %2, %2, %6, %4\n\tv_fma_f32 %3, %3, %6, %5"
                             : "=&v"(rt0), "=&v"(rt1), "+v"(rg0), "+v"(rg1) : "v"(d0), "v"(d1), "v"(x0));
                ra0 += rt0; ra1 += rt1;
            } else {
                asm volatile("v_fma_f32 %0, %2, %4, %0\n\tv_fma_f32 %1, %3, %4, %1\n\tv_add_f32 %2, %2, %5\n\tv_add_f32 %3, %3, %6"
                             : "+v"(ra0), "+v"(ra1), "+v"(rg0), "+v"(rg1) : "v"(x0), "v"(d0), "v"(d1));
            }
        }
        const bool lo = __float_as_uint(acc.x) != __float_as_uint(ra0) || __float_as_uint(g.x) != __float_as_uint(rg0);
        const bool hi = __float_as_uint(acc.y) != __float_as_uint(ra1) || __float_as_uint(g.y) != __float_as_uint(rg1);
        if (lo) atomicAdd(bad + 0, 1u);
        if (hi) atomicAdd(bad + 1, 1u);
        nbad += lo || hi;
    }
    if (nbad) atomicAdd(worst, 1ull);
}


// MODE 10: the accumulation block of small_linear_bwd_kernel<bf16, bf16, 1, 4> VERBATIM (same registers, same interleaved SALU / 64-bit
// VALU instructions, as hipcc 7.2 emitted it: profiles/r06_determinism_hunt.md), fed from operands and read back; the reference does
// the same sixteen FMAs and twelve additions with scalar VALU instructions.
__global__ __launch_bounds__(256) void probe_verbatim(unsigned *bad, unsigned long long *worst, int iters, unsigned salt, int variant) {
    const unsigned t = blockIdx.x * blockDim.x + threadIdx.x;
    unsigned nbad = 0;
    for (int rep = 0; rep < iters; ++rep) {
        const unsigned h = (t * 9781u + rep * 6271u) ^ salt;
        float g[16], x[4], a[8], o[8], r[8];
#pragma unroll
        for (int i = 0; i < 16; ++i) g[i] = seedf(h + i);
#pragma unroll
        for (int i = 0; i < 4; ++i) x[i] = seedf(h + 16 + i);
#pragma unroll
        for (int i = 0; i < 8; ++i) a[i] = seedf(h + 20 + i);       // a[0..3] = dW accumulators (v30, v31, v26, v27), a[4..7] = db (v24, v25, v22, v23)
        if (variant == 0) {
        asm volatile(
            "v_mov_b32 v2, %[x0]\n\tv_mov_b32 v3, %[x1]\n\tv_mov_b32 v4, %[x2]\n\tv_mov_b32 v38, %[x3]\n\t"
            "v_mov_b32 v6, %[g0]\n\tv_mov_b32 v7, %[g1]\n\tv_mov_b32 v8, %[g2]\n\tv_mov_b32 v9, %[g3]\n\t"
            "v_mov_b32 v10, %[g4]\n\tv_mov_b32 v11, %[g5]\n\tv_mov_b32 v12, %[g6]\n\tv_mov_b32 v13, %[g7]\n\t"
            "v_mov_b32 v14, %[g8]\n\tv_mov_b32 v15, %[g9]\n\tv_mov_b32 v16, %[g10]\n\tv_mov_b32 v17, %[g11]\n\t"
            "v_mov_b32 v18, %[g12]\n\tv_mov_b32 v19, %[g13]\n\tv_mov_b32 v20, %[g14]\n\tv_mov_b32 v21, %[g15]\n\t"
            "v_mov_b32 v30, %[a0]\n\tv_mov_b32 v31, %[a1]\n\tv_mov_b32 v26, %[a2]\n\tv_mov_b32 v27, %[a3]\n\t"
            "v_mov_b32 v24, %[a4]\n\tv_mov_b32 v25, %[a5]\n\tv_mov_b32 v22, %[a6]\n\tv_mov_b32 v23, %[a7]\n\t"
            "s_mov_b64 s[16:17], 0\n\ts_mov_b64 s[18:19], 64\n\ts_mov_b64 s[20:21], 0x7fffffff\n\tv_mov_b32 v28, 0\n\tv_mov_b32 v29, 0\n\t"
            "v_mov_b32 v32, 0\n\tv_mov_b32 v33, 0\n\tv_mov_b32 v34, 0\n\tv_mov_b32 v35, 0\n\ts_mov_b32 s22, 0\n\ts_mov_b32 s23, 0\n\ts_mov_b32 s24, 8\n\ts_mov_b32 s25, 0\n\t"
            "s_nop 4\n\t"
            "s_or_b64 exec, exec, s[16:17]\n\t"
            "v_pk_fma_f32 v[30:31], v[6:7], v[2:3], v[30:31] op_sel_hi:[1,0,1]\n\t"
            "v_pk_add_f32 v[6:7], v[6:7], v[24:25]\n\t"
            "v_pk_fma_f32 v[30:31], v[10:11], v[2:3], v[30:31] op_sel:[0,1,0]\n\t"
            "v_pk_add_f32 v[6:7], v[10:11], v[6:7]\n\t"
            "s_add_u32 s22, s22, s24\n\t"
            "v_pk_add_f32 v[6:7], v[14:15], v[6:7]\n\t"
            "s_addc_u32 s23, s23, s25\n\t"
            "v_pk_add_f32 v[24:25], v[18:19], v[6:7]\n\t"
            "v_pk_fma_f32 v[6:7], v[8:9], v[2:3], v[26:27] op_sel_hi:[1,0,1]\n\t"
            "v_lshl_add_u64 v[28:29], v[28:29], 0, s[18:19]\n\t"
            "v_pk_fma_f32 v[2:3], v[12:13], v[2:3], v[6:7] op_sel:[0,1,0]\n\t"
            "s_add_u32 s22, s22, s24\n\t"
            "v_pk_fma_f32 v[2:3], v[16:17], v[4:5], v[2:3] op_sel_hi:[1,0,1]\n\t"
            "v_pk_fma_f32 v[30:31], v[14:15], v[4:5], v[30:31] op_sel_hi:[1,0,1]\n\t"
            "v_pk_fma_f32 v[26:27], v[20:21], v[38:39], v[2:3] op_sel_hi:[1,0,1]\n\t"
            "v_pk_add_f32 v[2:3], v[8:9], v[22:23]\n\t"
            "s_addc_u32 s23, s23, s25\n\t"
            "v_pk_add_f32 v[2:3], v[12:13], v[2:3]\n\t"
            "v_cmp_le_i64_e32 vcc, s[20:21], v[28:29]\n\t"
            "v_pk_add_f32 v[2:3], v[16:17], v[2:3]\n\t"
            "v_pk_fma_f32 v[30:31], v[18:19], v[38:39], v[30:31] op_sel_hi:[1,0,1]\n\t"
            "v_pk_add_f32 v[22:23], v[20:21], v[2:3]\n\t"
            "v_lshl_add_u64 v[32:33], v[32:33], 0, s[18:19]\n\t"
            "v_lshl_add_u64 v[34:35], v[34:35], 0, s[18:19]\n\t"
            "s_nop 4\n\t"
            "v_mov_b32 %[o0], v30\n\tv_mov_b32 %[o1], v31\n\tv_mov_b32 %[o2], v26\n\tv_mov_b32 %[o3], v27\n\t"
            "v_mov_b32 %[o4], v24\n\tv_mov_b32 %[o5], v25\n\tv_mov_b32 %[o6], v22\n\tv_mov_b32 %[o7], v23\n\t"
            : [o0] "=&v"(o[0]), [o1] "=&v"(o[1]), [o2] "=&v"(o[2]), [o3] "=&v"(o[3]), [o4] "=&v"(o[4]), [o5] "=&v"(o[5]), [o6] "=&v"(o[6]), [o7] "=&v"(o[7])
            : [x0] "v"(x[0]), [x1] "v"(x[1]), [x2] "v"(x[2]), [x3] "v"(x[3]),
              [g0] "v"(g[0]), [g1] "v"(g[1]), [g2] "v"(g[2]), [g3] "v"(g[3]), [g4] "v"(g[4]), [g5] "v"(g[5]), [g6] "v"(g[6]), [g7] "v"(g[7]),
              [g8] "v"(g[8]), [g9] "v"(g[9]), [g10] "v"(g[10]), [g11] "v"(g[11]), [g12] "v"(g[12]), [g13] "v"(g[13]), [g14] "v"(g[14]), [g15] "v"(g[15]),
              [a0] "v"(a[0]), [a1] "v"(a[1]), [a2] "v"(a[2]), [a3] "v"(a[3]), [a4] "v"(a[4]), [a5] "v"(a[5]), [a6] "v"(a[6]), [a7] "v"(a[7])
            : "v2", "v3", "v4", "v5", "v6", "v7", "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23",
              "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35", "v38", "v39", "s16", "s17", "s18", "s19", "s20", "s21", "s22",
              "s23", "s24", "s25", "vcc", "scc");
        } else {
            // the same arithmetic, packed, but every instruction followed by s_nop 1 (control for an issue-timing hazard)
            asm volatile(
            "v_mov_b32 v2, %[x0]\n\tv_mov_b32 v3, %[x1]\n\tv_mov_b32 v4, %[x2]\n\tv_mov_b32 v38, %[x3]\n\t"
            "v_mov_b32 v6, %[g0]\n\tv_mov_b32 v7, %[g1]\n\tv_mov_b32 v8, %[g2]\n\tv_mov_b32 v9, %[g3]\n\t"
            "v_mov_b32 v10, %[g4]\n\tv_mov_b32 v11, %[g5]\n\tv_mov_b32 v12, %[g6]\n\tv_mov_b32 v13, %[g7]\n\t"
            "v_mov_b32 v14, %[g8]\n\tv_mov_b32 v15, %[g9]\n\tv_mov_b32 v16, %[g10]\n\tv_mov_b32 v17, %[g11]\n\t"
            "v_mov_b32 v18, %[g12]\n\tv_mov_b32 v19, %[g13]\n\tv_mov_b32 v20, %[g14]\n\tv_mov_b32 v21, %[g15]\n\t"
            "v_mov_b32 v30, %[a0]\n\tv_mov_b32 v31, %[a1]\n\tv_mov_b32 v26, %[a2]\n\tv_mov_b32 v27, %[a3]\n\t"
            "v_mov_b32 v24, %[a4]\n\tv_mov_b32 v25, %[a5]\n\tv_mov_b32 v22, %[a6]\n\tv_mov_b32 v23, %[a7]\n\t"
            "s_nop 4\n\t"
            "v_pk_fma_f32 v[30:31], v[6:7], v[2:3], v[30:31] op_sel_hi:[1,0,1]\n\ts_nop 1\n\t"
            "v_pk_add_f32 v[6:7], v[6:7], v[24:25]\n\ts_nop 1\n\t"
            "v_pk_fma_f32 v[30:31], v[10:11], v[2:3], v[30:31] op_sel:[0,1,0]\n\ts_nop 1\n\t"
            "v_pk_add_f32 v[6:7], v[10:11], v[6:7]\n\ts_nop 1\n\t"
            "v_pk_add_f32 v[6:7], v[14:15], v[6:7]\n\ts_nop 1\n\t"
            "v_pk_add_f32 v[24:25], v[18:19], v[6:7]\n\ts_nop 1\n\t"
            "v_pk_fma_f32 v[6:7], v[8:9], v[2:3], v[26:27] op_sel_hi:[1,0,1]\n\ts_nop 1\n\t"
            "v_pk_fma_f32 v[2:3], v[12:13], v[2:3], v[6:7] op_sel:[0,1,0]\n\ts_nop 1\n\t"
            "v_pk_fma_f32 v[2:3], v[16:17], v[4:5], v[2:3] op_sel_hi:[1,0,1]\n\ts_nop 1\n\t"
            "v_pk_fma_f32 v[30:31], v[14:15], v[4:5], v[30:31] op_sel_hi:[1,0,1]\n\ts_nop 1\n\t"
            "v_pk_fma_f32 v[26:27], v[20:21], v[38:39], v[2:3] op_sel_hi:[1,0,1]\n\ts_nop 1\n\t"
            "v_pk_add_f32 v[2:3], v[8:9], v[22:23]\n\ts_nop 1\n\t"
            "v_pk_add_f32 v[2:3], v[12:13], v[2:3]\n\ts_nop 1\n\t"
            "v_pk_add_f32 v[2:3], v[16:17], v[2:3]\n\ts_nop 1\n\t"
            "v_pk_fma_f32 v[30:31], v[18:19], v[38:39], v[30:31] op_sel_hi:[1,0,1]\n\ts_nop 1\n\t"
            "v_pk_add_f32 v[22:23], v[20:21], v[2:3]\n\ts_nop 1\n\t"
            "s_nop 4\n\t"
            "v_mov_b32 %[o0], v30\n\tv_mov_b32 %[o1], v31\n\tv_mov_b32 %[o2], v26\n\tv_mov_b32 %[o3], v27\n\t"
            "v_mov_b32 %[o4], v24\n\tv_mov_b32 %[o5], v25\n\tv_mov_b32 %[o6], v22\n\tv_mov_b32 %[o7], v23\n\t"
            : [o0] "=&v"(o[0]), [o1] "=&v"(o[1]), [o2] "=&v"(o[2]), [o3] "=&v"(o[3]), [o4] "=&v"(o[4]), [o5] "=&v"(o[5]), [o6] "=&v"(o[6]), [o7] "=&v"(o[7])
            : [x0] "v"(x[0]), [x1] "v"(x[1]), [x2] "v"(x[2]), [x3] "v"(x[3]),
              [g0] "v"(g[0]), [g1] "v"(g[1]), [g2] "v"(g[2]), [g3] "v"(g[3]), [g4] "v"(g[4]), [g5] "v"(g[5]), [g6] "v"(g[6]), [g7] "v"(g[7]),
              [g8] "v"(g[8]), [g9] "v"(g[9]), [g10] "v"(g[10]), [g11] "v"(g[11]), [g12] "v"(g[12]), [g13] "v"(g[13]), [g14] "v"(g[14]), [g15] "v"(g[15]),
              [a0] "v"(a[0]), [a1] "v"(a[1]), [a2] "v"(a[2]), [a3] "v"(a[3]), [a4] "v"(a[4]), [a5] "v"(a[5]), [a6] "v"(a[6]), [a7] "v"(a[7])
            : "v2", "v3", "v4", "v5", "v6", "v7", "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23",
              "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35", "v38", "v39");
        }
        // reference: acc_o = fma(g3o, x3, fma(g2o, x2, fma(g1o, x1, fma(g0o, x0, a_o)))) — note the kernel's association per column:
        //   columns 0, 1: a += g0*x0; a += g1*x1; a += g2*x2; a += g3*x3        columns 2, 3: the same chain
        //   db columns 0, 1: ((g0 + db) + g1 -> g1 + (g0 + db)) ... as emitted: t = g0 + db; t = g1 + t; t = g2 + t; db = g3 + t
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            float acc = a[c];
#pragma unroll
            for (int k = 0; k < 4; ++k) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(g[4 * k + c]), "v"(x[k]));
            r[c] = acc;
            float tdb = a[4 + c];
#pragma unroll
            for (int k = 0; k < 4; ++k) asm volatile("v_add_f32 %0, %1, %0" : "+v"(tdb) : "v"(g[4 * k + c]));
            r[4 + c] = tdb;
        }
        bool lo = false, hi = false;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const bool d = __float_as_uint(o[c]) != __float_as_uint(r[c]);
            if (c & 1) hi |= d; else lo |= d;
        }
        if (lo) atomicAdd(bad + 0, 1u);
        if (hi) atomicAdd(bad + 1, 1u);
        nbad += lo || hi;
    }
    if (nbad) atomicAdd(worst, 1ull);
}

// co-runners -----------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void spin_mfma(float *out, int iters) {
    f16v c = {};
    s4v a = {(short)(0x3f80 + threadIdx.x), 0x3f80, 0x3f00, 0x3e80}, b = {0x3f80, 0x3f00, (short)(0x3e80 + threadIdx.x), 0x3f80};
    for (int i = 0; i < iters; ++i) {
        c = __builtin_amdgcn_mfma_f32_32x32x8bf16_1k(a, b, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x8bf16_1k(b, a, c, 0, 0, 0);
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = c[0] + c[7];
}
__global__ __launch_bounds__(256) void spin_valu(float *out, int iters) {
    f2 a = {1.0001f, 0.9999f}, b = {(float)threadIdx.x * 1e-6f, 0.5f};
    for (int i = 0; i < iters; ++i) {
        asm volatile("v_pk_fma_f32 %0, %0, %1, %1\n\tv_pk_mul_f32 %1, %1, %0" : "+v"(a), "+v"(b));
        a.x = a.x * 0.5f + 0.25f; b.y = b.y * 0.5f + 0.125f;
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a.x + b.y;
}
__global__ __launch_bounds__(256) void spin_lds(float *out, int iters) {
    __shared__ float s[4096];
    for (int i = threadIdx.x; i < 4096; i += 256) s[i] = (float)i;
    __syncthreads();
    float v = 0.f;
    for (int i = 0; i < iters; ++i) {
        v += s[(threadIdx.x * 17 + i * 33) & 4095];
        s[(threadIdx.x + i) & 4095] = v * 0.5f;
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = v;
}
__global__ __launch_bounds__(256) void spin_mem(float4 *buf, size_t n, int iters) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (int k = 0; k < iters; ++k) {
        float4 v = buf[i % n];
        v.x += 1.f;
        buf[(i + 7919) % n] = v;
        i += (size_t)gridDim.x * blockDim.x;
    }
}

template <int MODE>
void run_mode(const char *name, const char *co, hipStream_t s0, hipStream_t s1, int rounds, float *scratch, float4 *big, size_t nbig, unsigned *bad, unsigned long long *worst) {
    CK(hipMemsetAsync(bad, 0, 8, s0));
    CK(hipMemsetAsync(worst, 0, 8, s0));
    CK(hipStreamSynchronize(s0));
    for (int r = 0; r < rounds; ++r) {
        // keep the co-runner resident while the probe runs: a few long launches on the second stream
        if (co[0] == 'm' && co[1] == 'f') hipLaunchKernelGGL(spin_mfma, dim3(1024), dim3(256), 0, s1, scratch, 20000);
        else if (co[0] == 'v') hipLaunchKernelGGL(spin_valu, dim3(1024), dim3(256), 0, s1, scratch, 40000);
        else if (co[0] == 'l') hipLaunchKernelGGL(spin_lds, dim3(1024), dim3(256), 0, s1, scratch, 40000);
        else if (co[0] == 'm' && co[1] == 'e') hipLaunchKernelGGL(spin_mem, dim3(2048), dim3(256), 0, s1, big, nbig, 64);
        for (int k = 0; k < 8; ++k) hipLaunchKernelGGL(probe<MODE>, dim3(512), dim3(256), 0, s0, bad, worst, 64, (unsigned)(r * 8 + k));
    }
    CK(hipDeviceSynchronize());
    unsigned h[2];
    unsigned long long w;
    CK(hipMemcpy(h, bad, 8, hipMemcpyDeviceToHost));
    CK(hipMemcpy(&w, worst, 8, hipMemcpyDeviceToHost));
    printf("%-44s beside %-6s: lanes-with-wrong LOW %u  HIGH %u  (threads hit %llu) of %llu checks\n", name, co, h[0], h[1], w,
           (unsigned long long)rounds * 8 * 512 * 256 * 64);
    fflush(stdout);
}

void run_verbatim(int variant, const char *co, hipStream_t s0, hipStream_t s1, int rounds, float *scratch, float4 *big, size_t nbig, unsigned *bad, unsigned long long *worst) {
    CK(hipMemsetAsync(bad, 0, 8, s0));
    CK(hipMemsetAsync(worst, 0, 8, s0));
    CK(hipStreamSynchronize(s0));
    for (int r = 0; r < rounds; ++r) {
        if (co[0] == 'm' && co[1] == 'f') hipLaunchKernelGGL(spin_mfma, dim3(1024), dim3(256), 0, s1, scratch, 20000);
        else if (co[0] == 'v') hipLaunchKernelGGL(spin_valu, dim3(1024), dim3(256), 0, s1, scratch, 40000);
        else if (co[0] == 'l') hipLaunchKernelGGL(spin_lds, dim3(1024), dim3(256), 0, s1, scratch, 40000);
        else if (co[0] == 'm' && co[1] == 'e') hipLaunchKernelGGL(spin_mem, dim3(2048), dim3(256), 0, s1, big, nbig, 64);
        for (int k = 0; k < 8; ++k) hipLaunchKernelGGL(probe_verbatim, dim3(512), dim3(256), 0, s0, bad, worst, 256, (unsigned)(r * 8 + k), variant);
    }
    CK(hipDeviceSynchronize());
    unsigned h[2];
    unsigned long long w;
    CK(hipMemcpy(h, bad, 8, hipMemcpyDeviceToHost));
    CK(hipMemcpy(&w, worst, 8, hipMemcpyDeviceToHost));
    printf("%-44s beside %-6s: blocks-with-wrong LOW %u  HIGH %u  (threads hit %llu) of %llu block evaluations\n",
           variant == 0 ? "kernel's accumulation block, verbatim" : "the same block, s_nop 1 after every op", co, h[0], h[1], w, (unsigned long long)rounds * 8 * 512 * 256 * 256);
    fflush(stdout);
}

int main(int argc, char **argv) {
    const int rounds = argc > 1 ? atoi(argv[1]) : 40;
    hipStream_t s0, s1;
    CK(hipStreamCreate(&s0));
    CK(hipStreamCreate(&s1));
    float *scratch;
    float4 *big;
    unsigned *bad;
    unsigned long long *worst;
    const size_t nbig = (size_t)1 << 26;
    CK(hipMalloc(&scratch, 1024 * 256 * 4));
    CK(hipMalloc(&big, nbig * 16));
    CK(hipMalloc(&bad, 8));
    CK(hipMalloc(&worst, 8));
    const char *cos[] = {"none", "mfma", "valu", "lds", "mem"};
    for (const char *co : cos) {
        run_verbatim(0, co, s0, s1, rounds, scratch, big, nbig, bad, worst);
        run_verbatim(1, co, s0, s1, rounds, scratch, big, nbig, bad, worst);
    }
    if (argc > 2) return 0;
    for (const char *co : cos) {
        run_mode<0>("pk_fma ; pk_add overwrites its source", co, s0, s1, rounds, scratch, big, nbig, bad, worst);
        run_mode<1>("pk_fma ; s_nop 0 ; pk_add overwrites source", co, s0, s1, rounds, scratch, big, nbig, bad, worst);
        run_mode<2>("pk_fma ; pk_add into another pair", co, s0, s1, rounds, scratch, big, nbig, bad, worst);
        run_mode<3>("pk_add reads pair ; pk_fma overwrites it", co, s0, s1, rounds, scratch, big, nbig, bad, worst);
        run_mode<4>("scalar control", co, s0, s1, rounds, scratch, big, nbig, bad, worst);
    }
    return 0;
}
