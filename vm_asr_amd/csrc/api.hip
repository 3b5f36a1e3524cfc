// api.hip — ABI version, thread-local error message and the optional HIP-event profiler.
#include "common.h"

#include <mutex>
#include <vector>

namespace vmasr {
namespace {
thread_local char g_err[512] = "";

struct Rec {
    int kid;
    hipEvent_t e0, e1;
    double bytes;
};
std::mutex g_mu;
std::vector<Rec> g_recs;
std::vector<hipEvent_t> g_pool;
thread_local Rec g_open{-1, nullptr, nullptr, 0.0};

hipEvent_t take_event() {
    if (!g_pool.empty()) {
        hipEvent_t e = g_pool.back();
        g_pool.pop_back();
        return e;
    }
    hipEvent_t e = nullptr;
    (void)hipEventCreate(&e);
    return e;
}

const char *kNames[VMASR_K_COUNT] = {
    "sscan_fwd", "sscan_fwd_agg", "sscan_fwd_carry", "sscan_fwd_apply", "sscan_bwd", "sscan_bwd_agg",
    "sscan_bwd_carry", "sscan_bwd_apply", "cross_scan", "cross_merge", "dwconv_silu_fwd", "dwconv_silu_bwd_a",
    "dwconv_silu_bwd_b", "stft", "istft_frames", "istft_ola", "istft_bwd", "layer_norm_fwd", "layer_norm_bwd",
    "layer_norm_bwd_reduce", "small_linear_fwd", "small_linear_bwd", "small_linear_reduce",
    "xproj_fwd", "xproj_bwd_a", "xproj_bwd_b", "spectral_power_iter", "im2col_kx1", "col2im_kx1", "split_bf16", "bias_gelu",
    "ss2d_transpose", "ss2d_fwd_agg", "ss2d_carry", "ss2d_fwd_apply", "ss2d_merge", "ss2d_bwd_agg", "ss2d_bwd_apply",
    "ss2d_pre", "ln_gate", "stack_rows", "feat_l1", "adamw", "conv_post", "mlp_fwd", "mlp_bwd", "inproj_fwd", "inproj_bwd",
    "ss2d_deep_xproj", "ss2d_deep_fwd", "ss2d_deep_bwd", "ss2d_deep_xbwd", "outproj_fwd", "outproj_bwd", "stft_loss",
    "conv_mfma_fwd", "conv_mfma_dgrad", "conv_mfma_wgrad", "wgrad_finish", "skinny_linear"};
}  // namespace

bool g_prof_on = false;
static bool g_det_on = false;
// Deterministic mode: per device, VMASR_K_COUNT ticket words + VMASR_K_COUNT timeout counters (common.h: det_enter).  Allocated for
// EVERY visible device when the mode is switched on (vmasr_set_deterministic) — never lazily inside a launcher, where a stream
// capture could be active and a second GPU of the process would be handed the first one's pointer.
constexpr int kMaxDev = 16;
static unsigned *g_det_buf[kMaxDev] = {};

unsigned *det_ticket(int kid) {
    if (!g_det_on || kid < 0 || kid >= VMASR_K_COUNT) return nullptr;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDev || g_det_buf[dev] == nullptr) {
        set_error("deterministic mode: no ticket buffer for device %d (vmasr_set_deterministic(1) must run before the first launch)", dev);
        return nullptr;
    }
    return g_det_buf[dev] + kid;
}
static int det_prepare() {
    int n = 0, cur = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return 0;   // no GPU (CPU-side tests): nothing to allocate
    (void)hipGetDevice(&cur);
    int rc = 0;
    for (int d = 0; d < n && d < kMaxDev; ++d) {
        if (g_det_buf[d]) continue;
        if (hipSetDevice(d) != hipSuccess || hipMalloc(reinterpret_cast<void **>(&g_det_buf[d]), 2 * VMASR_K_COUNT * sizeof(unsigned)) != hipSuccess ||
            hipMemset(g_det_buf[d], 0, 2 * VMASR_K_COUNT * sizeof(unsigned)) != hipSuccess || hipDeviceSynchronize() != hipSuccess) {
            g_det_buf[d] = nullptr;
            set_error("deterministic mode: cannot allocate the ticket buffer of device %d (switch the mode on outside a stream capture)", d);
            rc = VMASR_EINVAL;
        }
    }
    (void)hipSetDevice(cur);
    return rc;
}
void det_set(bool on) { g_det_on = on; }
bool det_get() { return g_det_on; }

void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

void prof_begin(int kid, hipStream_t st, double bytes) {
    std::lock_guard<std::mutex> lk(g_mu);
    g_open = Rec{kid, take_event(), take_event(), bytes};
    (void)hipEventRecord(g_open.e0, st);
}

void prof_end(hipStream_t st) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (g_open.kid < 0) return;
    (void)hipEventRecord(g_open.e1, st);
    g_recs.push_back(g_open);
    g_open.kid = -1;
}
}  // namespace vmasr

using namespace vmasr;

VMASR_EXPORT int vmasr_abi_version(void) { return VMASR_ABI_VERSION; }
VMASR_EXPORT const char *vmasr_last_error(void) { return g_err; }

VMASR_EXPORT void vmasr_prof_enable(int on) { g_prof_on = on != 0; }

VMASR_EXPORT void vmasr_set_deterministic(int on) {
    if (on && vmasr::det_prepare() != 0) return;   // (vmasr_last_error says why; the mode stays as it was)
    vmasr::det_set(on != 0);
}
// number of workgroups (all devices, all kernels) whose wait for their turn ran out since the mode was switched on: must be 0 —
// a non-zero count means an ordering stalled (e.g. the same kernel on two streams) and the sums of that launch are not reproducible
VMASR_EXPORT int64_t vmasr_det_timeouts(void) {
    int64_t total = 0;
    int cur = 0;
    (void)hipGetDevice(&cur);
    for (int d = 0; d < vmasr::kMaxDev; ++d) {
        if (!vmasr::g_det_buf[d]) continue;
        unsigned host[VMASR_K_COUNT];
        if (hipSetDevice(d) != hipSuccess || hipDeviceSynchronize() != hipSuccess ||
            hipMemcpy(host, vmasr::g_det_buf[d] + VMASR_K_COUNT, sizeof(host), hipMemcpyDeviceToHost) != hipSuccess) { total = -1; break; }
        for (unsigned v : host) total += v;
    }
    (void)hipSetDevice(cur);
    return total;
}
VMASR_EXPORT int vmasr_get_deterministic(void) { return vmasr::det_get() ? 1 : 0; }

VMASR_EXPORT void vmasr_prof_reset(void) {
    std::lock_guard<std::mutex> lk(g_mu);
    for (Rec &r : g_recs) {
        g_pool.push_back(r.e0);
        g_pool.push_back(r.e1);
    }
    g_recs.clear();
}

VMASR_EXPORT const char *vmasr_prof_name(int kid) { return (kid >= 0 && kid < VMASR_K_COUNT) ? kNames[kid] : ""; }

VMASR_EXPORT int vmasr_prof_collect(int kid, int64_t *launches, double *total_ms, double *alg_bytes) {
    std::lock_guard<std::mutex> lk(g_mu);
    int64_t n = 0;
    double ms = 0.0, by = 0.0;
    for (Rec &r : g_recs) {
        if (r.kid != kid) continue;
        hipError_t e = hipEventSynchronize(r.e1);
        if (e != hipSuccess) return (int)e;
        float t = 0.f;
        e = hipEventElapsedTime(&t, r.e0, r.e1);
        if (e != hipSuccess) return (int)e;
        ++n;
        ms += t;
        by += r.bytes;
    }
    if (launches) *launches = n;
    if (total_ms) *total_ms = ms;
    if (alg_bytes) *alg_bytes = by;
    return 0;
}

// Per call shape: the launches of kernel `kid` grouped by their algorithmic byte count (a kernel's call shapes differ in it),
// up to `max_groups` groups in order of first appearance.  Returns the number of groups (negative: HIP error).
VMASR_EXPORT int vmasr_prof_collect_shapes(int kid, int max_groups, double *group_bytes, int64_t *group_launches, double *group_ms) {
    std::lock_guard<std::mutex> lk(g_mu);
    int ng = 0;
    for (Rec &r : g_recs) {
        if (r.kid != kid) continue;
        hipError_t e = hipEventSynchronize(r.e1);
        if (e != hipSuccess) return -(int)e;
        float t = 0.f;
        e = hipEventElapsedTime(&t, r.e0, r.e1);
        if (e != hipSuccess) return -(int)e;
        int g = 0;
        while (g < ng && group_bytes[g] != r.bytes) ++g;
        if (g == ng) {
            if (ng == max_groups) continue;
            group_bytes[ng] = r.bytes; group_launches[ng] = 0; group_ms[ng] = 0.0;
            ++ng;
        }
        ++group_launches[g];
        group_ms[g] += t;
    }
    return ng;
}


// Debug aid: the device's constant-rate clock (100 MHz) written to *dst when the stream reaches this point.  A kernel, so it can be captured
// into a HIP graph (timing events recorded in a capture cannot be read after a replay on ROCm 7.2): tools/phase_probe.py marks the phases of the
// two-stream step with it.
namespace {
__global__ void mark_time_kernel(unsigned long long *dst) { *dst = wall_clock64(); }
}  // namespace

VMASR_EXPORT int vmasr_mark_time(uint64_t *dst, vmasr_stream_t stream) {
    VMASR_REQUIRE(dst, VMASR_EINVAL, "mark_time: null destination");
    hipLaunchKernelGGL(mark_time_kernel, dim3(1), dim3(1), 0, static_cast<hipStream_t>(stream), reinterpret_cast<unsigned long long *>(dst));
    return vmasr::check_launch("mark_time");
}
