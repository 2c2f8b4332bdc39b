// Host-side helpers shared by the C-ABI entry points.
#include "common.h"
#include "bbbp_hip.h"
#include <stdarg.h>

static thread_local char g_err[512] = "";
int g_bbbp_reserved_cus = 0;
size_t g_bbbp_small_lds_pad = 0;
int g_bbbp_wino_side_cus = 0;
const unsigned long long* g_bbbp_seed_base = nullptr;

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

// Experiment / tuning knob: reserve CUs for side-stream kernels (see common.h).  Returns the previous reservation.
extern "C" int bbbp_set_partition(int reserved_cus, size_t small_lds_pad) {
    int prev = g_bbbp_reserved_cus;
    g_bbbp_reserved_cus = reserved_cus;
    g_bbbp_small_lds_pad = small_lds_pad;
    return prev;
}
