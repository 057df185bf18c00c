// libnns_hip.so: error state, version and device query (include/nns.h).
#include "nns_common.h"

namespace nns {
static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
}  // namespace nns

NNS_API const char* nns_last_error(void) { return nns::g_err; }

NNS_API int nns_version(void) { return NNS_VERSION_MAJOR * 1000 + NNS_VERSION_MINOR; }

NNS_API int nns_device_info(char* name_host, int cap, int* cu_count_host, size_t* hbm_bytes_host) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return nns::fail(NNS_ERR_LAUNCH, "hipGetDevice: %s", hipGetErrorString(e));
    hipDeviceProp_t prop;
    e = hipGetDeviceProperties(&prop, dev);
    if (e != hipSuccess) return nns::fail(NNS_ERR_LAUNCH, "hipGetDeviceProperties: %s", hipGetErrorString(e));
    if (name_host && cap > 0) {
        snprintf(name_host, (size_t)cap, "%s (%s)", prop.name, prop.gcnArchName);
    }
    if (cu_count_host) *cu_count_host = prop.multiProcessorCount;
    if (hbm_bytes_host) *hbm_bytes_host = prop.totalGlobalMem;
    return NNS_OK;
}
