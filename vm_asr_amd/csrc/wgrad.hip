// wgrad.hip — finish of the split-K weight-gradient GEMMs of MANY layers in one launch (gfx950).
//
// dW = gy^T x of a Linear with 10^4 .. 10^6 rows and a (32 x 5) .. (256 x 72) weight is ONE output tile; the host side runs it
// as a batched GEMM over S row slabs (vm_asr_amd/linear.py: weight_grad) and then needs sum over the slabs, the bias gradient
// (the extra "ones" column of the operand) split off, and both as CONTIGUOUS tensors (autograd clones a strided gradient before
// adopting it as .grad).  That was sum + 2 strided copies per GEMM — 6 launches of 3-5 us per fused Mlp block backward, 160 per
// generator step.  Here every weight gradient of a backward pass is finished by ONE launch at the end of the pass (the host
// queues (partials, dW, dbias) per GEMM: vm_asr_amd/wgrad.py, flushed by autograd's end-of-pass callback like the LayerNorm
// reductions of ln.hip).  Deterministic: fixed summation order over the slabs.
//
// Reference lines this serves: the parameter gradients of nn.Linear in model/vmamba.py:483-509 (Mlp), :855,881 (in_proj / out_proj).
#include <algorithm>

#include "common.h"

namespace vmasr {
namespace {

constexpr int kMaxWgItems = 40;
struct WgItem {
    const float *parts;   // (S, N, ld) fp32
    float *dw;            // (N, K)
    float *db;            // (N) = column K of the summed partials, or null
    float *e1, *e2;       // (N) = columns K + 1, K + 2 (the deep SS2D core's dA_log, dD behind dW_dt | d dt_bias), or null
    int S, N, K, ld;
};
struct WgTable {
    WgItem it[kMaxWgItems];
};

// grid (chunks, items): thread = one output element (n, k), k in [0, K] (k == K: the bias column)
__global__ __launch_bounds__(256) void wgrad_finish_multi_kernel(const WgTable t) {
    const WgItem &w = t.it[blockIdx.y];
    const int cols = w.K + (w.db ? 1 : 0) + (w.e1 ? 1 : 0) + (w.e2 ? 1 : 0);
    const long total = (long)w.N * cols;
    const size_t sstride = (size_t)w.N * w.ld;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int n = (int)(i / cols), k = (int)(i - (long)n * cols);
        const float *p = w.parts + (size_t)n * w.ld + k;
        float s = 0.f;
        for (int j = 0; j < w.S; ++j) s += p[(size_t)j * sstride];
        if (k < w.K) w.dw[(size_t)n * w.K + k] = s;
        else if (k == w.K) w.db[n] = s;
        else if (k == w.K + 1) w.e1[n] = s;
        else w.e2[n] = s;
    }
}

}  // namespace
}  // namespace vmasr

using namespace vmasr;

VMASR_EXPORT int vmasr_wgrad_finish_multi(const float *const *parts, float *const *dws, float *const *dbs, float *const *e1s, float *const *e2s, const int32_t *Ss,
                                          const int32_t *Ns, const int32_t *Ks, const int32_t *lds, int32_t n, vmasr_stream_t stream) {
    VMASR_REQUIRE(parts && dws && dbs && Ss && Ns && Ks && lds && n > 0, VMASR_EINVAL, "wgrad_finish_multi: null argument");
    hipStream_t st = static_cast<hipStream_t>(stream);
    for (int base = 0; base < n; base += kMaxWgItems) {
        const int m = std::min(kMaxWgItems, n - base);
        WgTable t{};
        long maxel = 0;
        double bytes = 0;
        for (int i = 0; i < m; ++i) {
            const int j = base + i;
            float *e1 = e1s ? e1s[j] : nullptr, *e2 = e2s ? e2s[j] : nullptr;
            VMASR_REQUIRE((!e1 || dbs[j]) && (!e2 || e1), VMASR_EINVAL, "wgrad_finish_multi: item %d: extra columns need the ones before them", j);
            VMASR_REQUIRE(parts[j] && dws[j] && Ss[j] > 0 && Ns[j] > 0 && Ks[j] > 0 && lds[j] >= Ks[j] + (dbs[j] ? 1 : 0) + (e1 ? 1 : 0) + (e2 ? 1 : 0), VMASR_EINVAL,
                          "wgrad_finish_multi: bad item %d", j);
            t.it[i] = WgItem{parts[j], dws[j], dbs[j], e1, e2, Ss[j], Ns[j], Ks[j], lds[j]};
            maxel = std::max(maxel, (long)Ns[j] * (Ks[j] + 3));
            bytes += (double)Ss[j] * Ns[j] * lds[j] * 4 + (double)Ns[j] * (Ks[j] + 1) * 4;
        }
        const int gx = (int)std::min<long>(std::max<long>((maxel + 255) / 256, 1), 64);
        VMASR_LAUNCH(VMASR_K_WGRAD_FINISH, bytes, wgrad_finish_multi_kernel, dim3(gx, m), dim3(256), 0, st, t);
    }
    return check_launch("wgrad_finish_multi");
}
