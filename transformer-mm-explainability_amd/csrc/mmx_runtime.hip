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

// Compute units of the current device (256 on MI355X), read once per device: grids that fill the chip "once" size themselves by it.
int device_cu_count() {
    static int cached[16] = {0};
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 256;
    if (dev >= 0 && dev < 16 && cached[dev]) return cached[dev];
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) return 256;
    if (dev >= 0 && dev < 16) cached[dev] = cus;
    return cus;
}

__global__ __launch_bounds__(256) void zero_words_kernel(unsigned* __restrict__ dst, size_t words) {
    const size_t stride = static_cast<size_t>(gridDim.x) * blockDim.x;
    for (size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < words; i += stride) dst[i] = 0u;
}

__global__ __launch_bounds__(256) void zero_vec_kernel(uint4* __restrict__ dst, size_t vecs) {
    const size_t stride = static_cast<size_t>(gridDim.x) * blockDim.x;
    for (size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < vecs; i += stride)
        dst[i] = make_uint4(0u, 0u, 0u, 0u);
}

// one workgroup per (row, batch): zeros with a 1 on the diagonal
__global__ __launch_bounds__(256) void identity_kernel(float* __restrict__ R, int N) {
    const int r = blockIdx.x;
    float* row = R + (static_cast<size_t>(blockIdx.y) * N + r) * N;
    for (int c = threadIdx.x; c < N; c += blockDim.x) row[c] = c == r ? 1.0f : 0.0f;
}

int zero_async(void* dst, size_t bytes, hipStream_t s) {
    if (!bytes) return MMX_OK;
    if (bytes % 4 || !dst) { set_error("zero_async: %zu bytes at %p", bytes, dst); return MMX_EINVAL; }
    if (bytes % 16 == 0 && reinterpret_cast<uintptr_t>(dst) % 16 == 0) {
        const size_t vecs = bytes / 16;
        const unsigned grid = static_cast<unsigned>(vecs / 256 + 1 < 4096 ? vecs / 256 + 1 : 4096);
        zero_vec_kernel<<<grid, 256, 0, s>>>(static_cast<uint4*>(dst), vecs);
    } else {
        const size_t words = bytes / 4;
        const unsigned grid = static_cast<unsigned>(words / 256 + 1 < 4096 ? words / 256 + 1 : 4096);
        zero_words_kernel<<<grid, 256, 0, s>>>(static_cast<unsigned*>(dst), words);
    }
    MMX_LAUNCH_CHECK("zero_async");
    return MMX_OK;
}

int identity_async(float* R, int batch, int N, hipStream_t s) {
    if (batch <= 0 || N <= 0) return MMX_OK;
    identity_kernel<<<dim3(N, batch), 256, 0, s>>>(R, N);
    MMX_LAUNCH_CHECK("identity_async");
    return MMX_OK;
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
