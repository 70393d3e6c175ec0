// Library-level entry points: version, thread-local error string, device info.
#include "mh_common.h"

#include <string.h>

static thread_local char g_err[512] = "";

void mh_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int mh_num_cus() {
    static int cus = 0;
    if (cus == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess)
            cus = prop.multiProcessorCount;
        if (cus <= 0) cus = 256;
    }
    return cus;
}

extern "C" {

int32_t mh_version(void) { return 100; }

const char* mh_last_error(void) { return g_err; }

int32_t mh_device_info(int32_t* cu_count, int64_t* hbm_bytes) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) {
        mh_set_error("mh_device_info: no HIP device");
        return MH_ERR_LAUNCH;
    }
    if (cu_count) *cu_count = prop.multiProcessorCount;
    if (hbm_bytes) *hbm_bytes = (int64_t)prop.totalGlobalMem;
    return MH_OK;
}

}  // extern "C"
