// api.hip — ABI version + thread-local error message for libvmasr_hip.
#include "common.h"

namespace vmasr {
namespace {
thread_local char g_err[512] = "";
}
void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
}  // namespace vmasr

VMASR_EXPORT int vmasr_abi_version(void) { return VMASR_ABI_VERSION; }
VMASR_EXPORT const char *vmasr_last_error(void) { return vmasr::g_err; }
