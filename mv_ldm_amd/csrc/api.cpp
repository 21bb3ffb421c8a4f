// Error reporting + device info for libmvldm_hip.so.
#include <cstdarg>
#include <cstdio>
#include <cstring>

#include "common.h"

namespace mvldm {
static thread_local char g_err[512] = "";
int set_error(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}
}  // namespace mvldm

extern "C" int mvldm_abi_version(void) { return MVLDM_ABI_VERSION; }
extern "C" const char* mvldm_last_error(void) { return mvldm::g_err; }
extern "C" int mvldm_build_flags(void) {
#ifdef MVLDM_EXPERIMENTS
    return 1;
#else
    return 0;
#endif
}

extern "C" int mvldm_device_info(int* cu_count, size_t* hbm_bytes, char* arch, int arch_len) {
    int dev = 0;
    MVLDM_CHECK_HIP(hipGetDevice(&dev));
    hipDeviceProp_t prop;
    MVLDM_CHECK_HIP(hipGetDeviceProperties(&prop, dev));
    if (cu_count) *cu_count = prop.multiProcessorCount;
    if (hbm_bytes) *hbm_bytes = prop.totalGlobalMem;
    if (arch && arch_len > 0) {
        strncpy(arch, prop.gcnArchName, arch_len - 1);
        arch[arch_len - 1] = 0;
    }
    return MVLDM_OK;
}
