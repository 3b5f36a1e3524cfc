// Cycle cost of the building blocks of csrc/sscan_n.hip on one CU-resident wave set (dev tool):
//   hipcc --offload-arch=gfx950 -O3 -I../../vm_asr_amd/csrc -I../../include probe.hip -o probe && ./probe
// Each kernel runs ITER iterations of one block on register data; time / (ITER * waves per SIMD) = issue cycles per block.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#define PROBE_INCLUDE
#include "../../vm_asr_amd/csrc/sscan_n_prims.h"
using namespace vmasr;

constexpr int ITER = 4096;

template <int WHICH>
__global__ __launch_bounds__(1024) void k(float *out, const float *in) {
    const int lane = threadIdx.x & 63;
    float dl[4] = {in[lane], in[lane + 64], in[lane + 128], in[lane + 192]};
    v2f A2 = {in[256] * -1.f, in[257] * -2.f};
    v2f acc = splat(0.f);
    Pair2 st{splat(1.f), splat(0.f)};
    for (int it = 0; it < ITER; ++it) {
        if constexpr (WHICH == 0) {          // decay of 4 items x 2 states
            v2f a[4];
            decay2x4<false>(dl, A2, a);
            acc += (a[0] + a[1]) + (a[2] + a[3]);
            dl[0] += 1e-7f;
        } else if constexpr (WHICH == 1) {   // forward scan of a pair
            Pair2 e, t;
            wave_scan_fwd2(Pair2{A2 * splat(0.999f), acc + splat(dl[0])}, e, t);
            acc += e.b + t.b * splat(1e-3f);
            A2 = e.a * splat(0.5f) + splat(0.4f);
        } else if constexpr (WHICH == 2) {   // reverse scan of a pair
            Pair2 e, t;
            wave_scan_rev2(Pair2{A2 * splat(0.999f), acc + splat(dl[0])}, lane, e, t);
            acc += e.b + t.b * splat(1e-3f);
            A2 = e.a * splat(0.5f) + splat(0.4f);
        } else if constexpr (WHICH == 3) {   // two wave sums
            const float s0 = wave_sum(acc.x + dl[0]), s1 = wave_sum(acc.y + dl[1]);
            acc += (v2f){s0, s1} * splat(1e-3f);
        } else if constexpr (WHICH == 4) {   // compose + apply of 4 items (the in-lane recurrences)
            v2f a[4] = {A2, A2 * splat(0.9f), A2 * splat(0.8f), A2 * splat(0.7f)}, b[4];
            for (int i = 0; i < 4; ++i) b[i] = splat(dl[i]) * acc;
            Pair2 agg{a[0], b[0]};
            for (int i = 1; i < 4; ++i) agg = then2(agg, Pair2{a[i], b[i]});
            v2f h = agg.b;
            for (int i = 0; i < 4; ++i) { h = fma2(a[i], h, b[i]); acc = fma2(h, A2, acc); }
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc.x + acc.y + A2.x + st.a.x;
}

template <int WHICH>
void run(const char *name, float *out, float *in) {
    for (int waves : {4, 16}) {   // waves per workgroup = 1 / 4 per SIMD; one workgroup per CU on 256 CUs
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(k<WHICH>, dim3(256), dim3(64 * waves), 0, 0, out, in);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<WHICH>, dim3(256), dim3(64 * waves), 0, 0, out, in);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        const double cyc = ms * 1e-3 * 2.4e9 / ITER / (waves / 4);
        printf("%-28s %2d waves/WG: %7.1f cycles per block per wave-slot (%.3f ms)\n", name, waves, cyc, ms);
    }
}

int main() {
    float *out, *in;
    hipMalloc(&out, 256 * 1024 * 4);
    hipMalloc(&in, 4096);
    std::vector<float> h(1024, 0.01f);
    hipMemcpy(in, h.data(), 4096, hipMemcpyHostToDevice);
    run<0>("decay 4 items x 2 states", out, in);
    run<1>("fwd scan pair", out, in);
    run<2>("rev scan pair", out, in);
    run<3>("2 wave sums", out, in);
    run<4>("compose + apply 4 items", out, in);
    return 0;
}
