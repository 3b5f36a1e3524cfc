// adamw.hip — the optimiser step of both models as ONE launch, gfx950.
//
// Reference: torch.optim.AdamW as the reference configures it (utils/optimizer.py:16-50: AdamW, eps / betas / weight
// decay from the yaml, 1-D parameters and biases without decay), stepped once per model per training step
// (trainer/trainer.py:150-156, 396-399, 436-438).  PyTorch's fused multi-tensor AdamW walks the ~460 parameter tensors
// in 54 launches at ≈1.2 TB/s (1.0 ms per step for 44 M parameters); the update is a pure stream — read p, g, m, v, write
// p, m, v: 28 B per parameter — so one launch over a chunk table runs it at HBM speed, and the bf16 shadow copy of the
// weights the autocast region reads (trainer._make_shadows) is written in the same pass instead of by a second
// multi-tensor copy.
//
// Arithmetic (fp32, as torch's _fused_adamw non-amsgrad path):
//     p -= lr * wd * p;  m += (1 - b1) (g - m);  v = b2 v + (1 - b2) g g;
//     p -= (lr / (1 - b1^t)) * m / (sqrt(v) / sqrt(1 - b2^t) + eps)
// lr and t are read from DEVICE memory at run time (the schedule keeps working under graph replay).
#include "common.h"

namespace vmasr {
namespace {

constexpr int kAdamChunk = 4096;   // elements per workgroup (256 threads x 4 float4)

__global__ __launch_bounds__(256) void adamw_kernel(const vmasr_adamw_item *__restrict__ items, const int2 *__restrict__ chunks,
                                                    const float *__restrict__ lr_p, const float *__restrict__ step_p, const float b1,
                                                    const float b2, const float eps) {
    const int2 ck = chunks[blockIdx.x];                       // (item, chunk index within the item)
    const vmasr_adamw_item it = items[ck.x];
    const long base = (long)ck.y * kAdamChunk;
    const long n = it.n - base < kAdamChunk ? it.n - base : kAdamChunk;
    const float lr = lr_p[0], t = step_p[0];
    const float bc1 = 1.f - powf(b1, t), bc2 = 1.f - powf(b2, t);
    const float step_size = lr / bc1, inv_bc2_sqrt = 1.f / sqrtf(bc2), decay = 1.f - lr * it.weight_decay;
    float *p = it.p + base, *m = it.m + base, *v = it.v + base;
    const float *g = it.g + base;
    bf16_t *lp = it.lp ? static_cast<bf16_t *>(it.lp) + base : nullptr;
    bf16_t *lpt = static_cast<bf16_t *>(it.lpt);            // transposed shadow (2-D weights): scattered 2-byte stores, few MB in all
    const int rows = it.rows, cols = it.cols;
    auto put_t = [&](const long i, const float P) {        // i: index within this chunk
        const long gi = base + i;
        const int r = (int)(gi / cols), c = (int)(gi - (long)r * cols);
        lpt[(size_t)c * rows + r] = (bf16_t)P;
    };
    auto upd = [&](float &pp, float gg, float &mm, float &vv) {
        pp *= decay;
        mm = fmaf(1.f - b1, gg - mm, mm);
        vv = fmaf(b2, vv, (1.f - b2) * gg * gg);
        pp -= step_size * mm / (sqrtf(vv) * inv_bc2_sqrt + eps);
    };
    if (it.vec) {   // every pointer of the item 16-byte aligned (bf16 shadow: 8) and n % 4 == 0 is not needed: tail below
        const long n4 = n / 4;
        for (long i = threadIdx.x; i < n4; i += 256) {
            float4 P = reinterpret_cast<float4 *>(p)[i], M = reinterpret_cast<float4 *>(m)[i], V = reinterpret_cast<float4 *>(v)[i];
            const float4 G = reinterpret_cast<const float4 *>(g)[i];
            upd(P.x, G.x, M.x, V.x); upd(P.y, G.y, M.y, V.y); upd(P.z, G.z, M.z, V.z); upd(P.w, G.w, M.w, V.w);
            reinterpret_cast<float4 *>(p)[i] = P;
            reinterpret_cast<float4 *>(m)[i] = M;
            reinterpret_cast<float4 *>(v)[i] = V;
            if (lp) {
                union { uint2 raw; bf16_t b[4]; } o;
                o.b[0] = (bf16_t)P.x; o.b[1] = (bf16_t)P.y; o.b[2] = (bf16_t)P.z; o.b[3] = (bf16_t)P.w;
                reinterpret_cast<uint2 *>(lp)[i] = o.raw;
            }
            if (lpt) { put_t(4 * i, P.x); put_t(4 * i + 1, P.y); put_t(4 * i + 2, P.z); put_t(4 * i + 3, P.w); }
        }
        for (long i = n4 * 4 + threadIdx.x; i < n; i += 256) {
            float P = p[i], M = m[i], V = v[i];
            upd(P, g[i], M, V);
            p[i] = P; m[i] = M; v[i] = V;
            if (lp) lp[i] = (bf16_t)P;
            if (lpt) put_t(i, P);
        }
    } else {
        for (long i = threadIdx.x; i < n; i += 256) {
            float P = p[i], M = m[i], V = v[i];
            upd(P, g[i], M, V);
            p[i] = P; m[i] = M; v[i] = V;
            if (lp) lp[i] = (bf16_t)P;
            if (lpt) put_t(i, P);
        }
    }
}

}  // namespace
}  // namespace vmasr

using namespace vmasr;

VMASR_EXPORT int32_t vmasr_adamw_chunk(void) { return kAdamChunk; }

VMASR_EXPORT int vmasr_adamw_step(const vmasr_adamw_item *items, const int32_t *chunks, int32_t nchunks, int64_t total_elems,
                                  const float *lr, const float *step, float beta1, float beta2, float eps, vmasr_stream_t stream) {
    VMASR_REQUIRE(items && chunks && lr && step, VMASR_EINVAL, "adamw_step: null argument");
    VMASR_REQUIRE(nchunks > 0 && beta1 >= 0.f && beta1 < 1.f && beta2 >= 0.f && beta2 < 1.f && eps >= 0.f, VMASR_EINVAL,
                  "adamw_step: bad hyper-parameters");
    hipStream_t st = static_cast<hipStream_t>(stream);
    VMASR_LAUNCH(VMASR_K_ADAMW, 28.0 * (double)total_elems, adamw_kernel, dim3(nchunks), dim3(256), 0, st, items,
                 reinterpret_cast<const int2 *>(chunks), lr, step, beta1, beta2, eps);
    return check_launch("adamw_step");
}
