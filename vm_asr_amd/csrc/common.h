// common.h — shared device/host helpers for libvmasr_hip (gfx950 only, wave64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>

#include "vmasr_hip.h"

#define VMASR_EXPORT extern "C" __attribute__((visibility("default")))

namespace vmasr {

// ---- host-side error plumbing -------------------------------------------------------
void set_error(const char *fmt, ...);

#define VMASR_REQUIRE(cond, code, ...)      \
    do {                                    \
        if (!(cond)) {                      \
            ::vmasr::set_error(__VA_ARGS__); \
            return (code);                  \
        }                                   \
    } while (0)

// ---- optional per-launch HIP-event timing (api.hip) ----------------------------------
extern bool g_prof_on;
void prof_begin(int kernel_id, hipStream_t st, double alg_bytes);
void prof_end(hipStream_t st);

// launch `kernel` and, when profiling is on, bracket it with events on its own stream
#define VMASR_LAUNCH(kid, bytes, kernel, grid, block, smem, st, ...)            \
    do {                                                                        \
        if (::vmasr::g_prof_on) ::vmasr::prof_begin((kid), (st), (double)(bytes)); \
        hipLaunchKernelGGL(kernel, grid, block, smem, st, __VA_ARGS__);         \
        if (::vmasr::g_prof_on) ::vmasr::prof_end((st));                        \
    } while (0)

inline int check_launch(const char *what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("%s: %s", what, hipGetErrorString(e));
        return (int)e;
    }
    return 0;
}

// ---- element types -------------------------------------------------------------------
using f16_t = _Float16;
using bf16_t = __bf16;

template <typename T> struct VecOf4;
template <> struct VecOf4<float> { using type = float4; };
template <> struct VecOf4<f16_t> { using type = uint2; };
template <> struct VecOf4<bf16_t> { using type = uint2; };

__device__ __forceinline__ float to_f32(float v) { return v; }
__device__ __forceinline__ float to_f32(f16_t v) { return (float)v; }
__device__ __forceinline__ float to_f32(bf16_t v) { return (float)v; }
template <typename T> __device__ __forceinline__ T from_f32(float v) { return (T)v; }

constexpr int kWave = 64;

// 4 consecutive elements starting at p[t] (t may run past `len`; out-of-range -> fill).
// VEC: p + t is 4-element aligned and the caller guarantees 16-B (fp32) / 8-B (16-bit)
// alignment of the row base, so a full quad is one dwordx4 / dwordx2 load.
template <typename T, bool VEC>
__device__ __forceinline__ void load4(const T *__restrict__ p, int t, int len, float (&v)[4],
                                      float fill = 0.f) {
    if (VEC && t + 3 < len) {
        if constexpr (sizeof(T) == 4) {
            const float4 q = *reinterpret_cast<const float4 *>(p + t);
            v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
        } else {
            union { uint2 raw; T e[4]; } q;
            q.raw = *reinterpret_cast<const uint2 *>(p + t);
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] = to_f32(q.e[i]);
        }
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = (t + i < len) ? to_f32(p[t + i]) : fill;
    }
}

// Same with a wave-uniform `full` (the whole wave's 256-element tile is in range and the rows are
// vector-aligned): the choice is a scalar branch, not a per-lane one, so full tiles are straight-line
// vector loads the scheduler can hoist and keep in flight.
template <typename T, bool VEC>
__device__ __forceinline__ void load4u(const T *__restrict__ p, int t, int len, float (&v)[4], bool full,
                                       float fill = 0.f) {
    if (VEC && full) {
        if constexpr (sizeof(T) == 4) {
            const float4 q = *reinterpret_cast<const float4 *>(p + t);
            v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
        } else {
            union { uint2 raw; T e[4]; } q;
            q.raw = *reinterpret_cast<const uint2 *>(p + t);
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] = to_f32(q.e[i]);
        }
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = (t + i < len) ? to_f32(p[t + i]) : fill;
    }
}

template <typename T, bool VEC>
__device__ __forceinline__ void store4u(T *__restrict__ p, int t, int len, const float (&v)[4], bool full) {
    if (VEC && full) {
        if constexpr (sizeof(T) == 4) {
            *reinterpret_cast<float4 *>(p + t) = make_float4(v[0], v[1], v[2], v[3]);
        } else {
            union { uint2 raw; T e[4]; } q;
#pragma unroll
            for (int i = 0; i < 4; ++i) q.e[i] = from_f32<T>(v[i]);
            *reinterpret_cast<uint2 *>(p + t) = q.raw;
        }
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (t + i < len) p[t + i] = from_f32<T>(v[i]);
    }
}

template <typename T, bool VEC>
__device__ __forceinline__ void store4(T *__restrict__ p, int t, int len, const float (&v)[4]) {
    if (VEC && t + 3 < len) {
        if constexpr (sizeof(T) == 4) {
            *reinterpret_cast<float4 *>(p + t) = make_float4(v[0], v[1], v[2], v[3]);
        } else {
            union { uint2 raw; T e[4]; } q;
#pragma unroll
            for (int i = 0; i < 4; ++i) q.e[i] = from_f32<T>(v[i]);
            *reinterpret_cast<uint2 *>(p + t) = q.raw;
        }
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (t + i < len) p[t + i] = from_f32<T>(v[i]);
    }
}

__device__ __forceinline__ float readlane_f(float v, int lane) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}

// wave64 all-reduce sum (result valid in every lane)
// Sum over the 64 lanes of a wave, returned in every lane.  DPP only (4 in-row butterfly steps, 2 row broadcasts, one
// v_readlane): `__shfl_xor` compiles to ds_bpermute_b32 on gfx9 — an LDS-pipe round trip per step, 6 per sum, and the
// fused SS2D backward does 56 sums per wave-tile (336 of its 4 000 instructions were bpermutes, each followed by a wait).
template <int CTRL, int ROW_MASK = 0xf>
__device__ __forceinline__ float dpp_add(float v) {   // v + (v moved by CTRL); lanes without a source / masked rows add 0
    return v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xf, false));
}

// Several wave sums at once (butterfly with halving): NV values per lane -> every lane l ends up with the wave total of
// value (l & (NV-1)).  Step k pairs the values (2j, 2j+1): lanes with bit k clear keep 2j and hand 2j+1 to their partner
// (lane ^ 2^k) and vice versa, so the number of live values halves while the lane distance doubles: NV - 1 + log2(64/NV)
// exchanges instead of 6 NV, and the exchanges of one step are independent of each other (no DPP wait states).
// lane ^ 1, ^ 2: quad_perm; ^ 4, ^ 8: row_shl / row_shr under complementary bank masks; ^ 16, ^ 32: ds_bpermute.
template <int CTRL_LO, int BANKS_LO, int CTRL_HI, int BANKS_HI>
__device__ __forceinline__ float dpp_xor_row(float send) {   // value of the lane at distance 4 (or 8) across the bank pattern
    int r = __builtin_amdgcn_update_dpp(0, __float_as_int(send), CTRL_LO, 0xf, BANKS_LO, false);
    r = __builtin_amdgcn_update_dpp(r, __float_as_int(send), CTRL_HI, 0xf, BANKS_HI, false);
    return __int_as_float(r);
}
__device__ __forceinline__ float xor1(float v) { return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xf, 0xf, false)); }
__device__ __forceinline__ float xor2(float v) { return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xf, 0xf, false)); }
__device__ __forceinline__ float xor4(float v) { return dpp_xor_row<0x104, 0x5, 0x114, 0xA>(v); }   // row_shl:4 banks 0,2 | row_shr:4 banks 1,3
__device__ __forceinline__ float xor8(float v) { return dpp_xor_row<0x108, 0x3, 0x118, 0xC>(v); }   // row_shl:8 banks 0,1 | row_shr:8 banks 2,3

// lane l (any l) returns the wave total of v[l & 3]
__device__ __forceinline__ float wave_sum4(const float (&v)[4], const int lane) {
    const bool b0 = lane & 1, b1 = lane & 2;
    const float w0 = (b0 ? v[1] : v[0]) + xor1(b0 ? v[0] : v[1]);
    const float w1 = (b0 ? v[3] : v[2]) + xor1(b0 ? v[2] : v[3]);
    float y = (b1 ? w1 : w0) + xor2(b1 ? w0 : w1);
    y += xor4(y);
    y += xor8(y);
    y += __shfl_xor(y, 16);
    y += __shfl_xor(y, 32);
    return y;
}

// lane l (any l) returns the wave total of v[l & 7]
__device__ __forceinline__ float wave_sum8(const float (&v)[8], const int lane) {
    const bool b0 = lane & 1, b1 = lane & 2, b2 = lane & 4;
    float w[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) w[j] = (b0 ? v[2 * j + 1] : v[2 * j]) + xor1(b0 ? v[2 * j] : v[2 * j + 1]);
    const float x0 = (b1 ? w[1] : w[0]) + xor2(b1 ? w[0] : w[1]);
    const float x1 = (b1 ? w[3] : w[2]) + xor2(b1 ? w[2] : w[3]);
    float y = (b2 ? x1 : x0) + xor4(b2 ? x0 : x1);
    y += xor8(y);
    y += __shfl_xor(y, 16);
    y += __shfl_xor(y, 32);
    return y;
}

__device__ __forceinline__ float wave_sum(float v) {
    v = dpp_add<0xB1>(v);           // quad_perm [1,0,3,2]: lane ^ 1
    v = dpp_add<0x4E>(v);           // quad_perm [2,3,0,1]: lane ^ 2
    v = dpp_add<0x141>(v);          // row_half_mirror: the other quad of each 8 lanes
    v = dpp_add<0x140>(v);          // row_mirror: the other half of each 16-lane row -> every lane holds its row's sum
    v = dpp_add<0x142, 0xa>(v);     // row_bcast:15 into rows 1 and 3
    v = dpp_add<0x143, 0xc>(v);     // row_bcast:31 into rows 2 and 3 -> lane 63 holds the total
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

// XCD-aware block remap (blocks b and b+8 share an XCD under round-robin dispatch): give
// each XCD a contiguous run of logical blocks so neighbours that share operand tiles hit
// the same L2.  Pure speed: any placement is correct.
__device__ __forceinline__ int xcd_remap(int bid, int nblocks) {
    const int x = bid % 8, q = nblocks / 8, r = nblocks % 8;   // XCD x runs blocks x, x + 8, ...: q (+1 if x < r) of them
    return x * q + (x < r ? x : r) + bid / 8;
}

inline bool aligned_to(const void *p, size_t a) { return (reinterpret_cast<uintptr_t>(p) % a) == 0; }

// ---- deterministic-reduction switch (SURVEY.md §5 debug aid; VMASR_DETERMINISTIC=1 / vmasr_set_deterministic) -----------------------
// The parameter-gradient sums of several kernels end in fp32 atomics (as the reference's backward does: cus/selective_scan_bwd_kernel.cuh:
// 218-219,262-271), so two runs differ at rounding level.  In deterministic mode the workgroups of such a launch take their closing
// atomics IN WORKGROUP ORDER: a ticket word per kernel id (api.hip: det_ticket) counts the workgroups that are done; workgroup k
// waits for ticket == k, performs its atomics (they execute at the memory side; `s_waitcnt vmcnt(0)` = performed), then passes the
// turn on; the last one resets the ticket.  The float additions to every address then happen in a fixed order -> bit-reproducible
// results.  Slow by design (the tails serialise, ~1 us per workgroup): a debug aid, off by default (ticket == nullptr: plain atomics).
// Safe: workgroups are dispatched in linear order, so every workgroup waits only for ones dispatched before it; the wait is bounded
// (a wait that runs out is counted: vmasr_det_timeouts).  ONE STREAM ONLY: the ticket is per kernel id, so the same kernel must not
// run on two streams at once in this mode (the trainer keeps its one-stream layout when VMASR_DETERMINISTIC=1).
unsigned *det_ticket(int kernel_id);          // nullptr unless deterministic mode is on (host side, api.hip)
void det_set(bool on);
bool det_get();

__device__ __forceinline__ unsigned det_me() { return blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z); }
// every thread of the workgroup calls both; between them the workgroup issues its global atomics
__device__ __forceinline__ void det_enter(unsigned *ticket) {
    if (ticket == nullptr) return;
    if (threadIdx.x == 0) {
        const unsigned me = det_me();
        unsigned spins = 0;
        while (__hip_atomic_load(ticket, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != me && ++spins < (1u << 26)) __builtin_amdgcn_s_sleep(4);
        if (spins >= (1u << 26)) atomicAdd(ticket + VMASR_K_COUNT, 1u);   // the wait ran out: counted, read by vmasr_det_timeouts()
    }
    __syncthreads();
}
__device__ __forceinline__ void det_leave(unsigned *ticket) {
    if (ticket == nullptr) return;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's atomics have been performed
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned me = det_me(), total = gridDim.x * gridDim.y * gridDim.z;
        __hip_atomic_store(ticket, me + 1 == total ? 0u : me + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}
// LDS float atomics of the waves of a workgroup in wave order (deterministic mode): `body` does the wave's ds_add's
#define VMASR_DET_WAVE_ORDER(ticket, nwaves, ...)                                   \
    do {                                                                            \
        if ((ticket) == nullptr) { __VA_ARGS__; }                                   \
        else {                                                                      \
            for (int w_ = 0; w_ < (nwaves); ++w_) {                                 \
                if ((int)(threadIdx.x >> 6) == w_) { __VA_ARGS__; }                 \
                __syncthreads();                                                    \
            }                                                                       \
        }                                                                           \
    } while (0)

}  // namespace vmasr
