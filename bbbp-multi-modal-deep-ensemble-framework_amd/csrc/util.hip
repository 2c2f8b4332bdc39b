// Host-side helpers shared by the C-ABI entry points.
#include "common.h"
#include "bbbp_hip.h"
#include <stdarg.h>

static thread_local char g_err[512] = "";

void bbbp_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* bbbp_last_error(void) { return g_err; }

int bbbp_num_cus() {
    static int cached = 0;
    if (cached == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess)
            cached = prop.multiProcessorCount;
        if (cached <= 0) cached = 256;
    }
    return cached;
}

extern "C" int bbbp_abi_version(void) { return 1; }
