// Error reporting + HIP-event helpers of the C-ABI (include/mmx_relevancy.h).
#include <stdarg.h>

#include "mmx_common.h"

namespace mmx {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int hip_fail(hipError_t e, const char* what) {
    set_error("%s: %s (hipError %d)", what, hipGetErrorString(e), static_cast<int>(e));
    return MMX_EHIP - static_cast<int>(e);
}

}  // namespace mmx

extern "C" int mmx_abi_version(void) { return MMX_ABI_VERSION; }
extern "C" const char* mmx_last_error(void) { return mmx::g_err; }

extern "C" int mmx_event_create(void** event_out) {
    if (!event_out) { mmx::set_error("mmx_event_create: null"); return MMX_EINVAL; }
    hipEvent_t ev;
    hipError_t e = hipEventCreate(&ev);
    if (e != hipSuccess) return mmx::hip_fail(e, "hipEventCreate");
    *event_out = ev;
    return MMX_OK;
}
extern "C" int mmx_event_destroy(void* event) {
    hipError_t e = hipEventDestroy(static_cast<hipEvent_t>(event));
    return e == hipSuccess ? MMX_OK : mmx::hip_fail(e, "hipEventDestroy");
}
extern "C" int mmx_event_record(void* event, void* stream) {
    hipError_t e = hipEventRecord(static_cast<hipEvent_t>(event), static_cast<hipStream_t>(stream));
    return e == hipSuccess ? MMX_OK : mmx::hip_fail(e, "hipEventRecord");
}
extern "C" int mmx_event_elapsed_ms(void* start, void* stop, float* ms_out) {
    if (!ms_out) { mmx::set_error("mmx_event_elapsed_ms: null"); return MMX_EINVAL; }
    hipError_t e = hipEventSynchronize(static_cast<hipEvent_t>(stop));
    if (e != hipSuccess) return mmx::hip_fail(e, "hipEventSynchronize");
    e = hipEventElapsedTime(ms_out, static_cast<hipEvent_t>(start), static_cast<hipEvent_t>(stop));
    return e == hipSuccess ? MMX_OK : mmx::hip_fail(e, "hipEventElapsedTime");
}
