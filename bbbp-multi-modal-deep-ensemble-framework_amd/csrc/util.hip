// Host-side helpers shared by the C-ABI entry points.
#include "common.h"
#include "bbbp_hip.h"
#include <stdarg.h>
#include <stdlib.h>
#include <mutex>
#include <map>
#include <utility>

static thread_local char g_err[512] = "";
// Per-call scopes of the engine (common.h): set by the host thread that enqueues a forward / backward call and read by the launchers
// that thread reaches -- thread-local, so that two host threads driving two models do not see each other's scopes.
thread_local int g_bbbp_reserved_cus = 0;
thread_local size_t g_bbbp_small_lds_pad = 0;
thread_local int g_bbbp_wino_side_cus = 0;
thread_local int g_bbbp_conv1_fwd_f32 = 0;
thread_local int g_bbbp_conv1_fwd_per_cu = 0;
thread_local int g_bbbp_conv_wgrad_beside_encoder = 0;
thread_local int g_bbbp_conv2_fwd_pipe = 0;
thread_local const unsigned long long* g_bbbp_seed_base = nullptr;

// More than 64 KB of dynamic LDS needs hipFuncAttributeMaxDynamicSharedMemorySize per (kernel, DEVICE): code objects are loaded
// per device, so a process-wide flag is not enough when one process touches two GPUs.  Callers pass sizes that vary at run time
// (attention: grows with the batch; conv: the CU-reservation pad), so the LARGEST size granted so far is remembered and the attribute
// is raised again whenever a launch needs more -- a first call at B = 320 must not pin the kernel below what B = 512 needs.
int bbbp_ensure_dyn_lds(const void* kernel, size_t bytes) {
    static std::mutex mu;
    static std::map<std::pair<int, const void*>, size_t> granted;
    int dev = 0;
    BBBP_CHECK_HIP(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lock(mu);
    const auto key = std::make_pair(dev, kernel);
    const auto it = granted.find(key);
    if (it != granted.end() && it->second >= bytes) return BBBP_OK;
    hipError_t e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e != hipSuccess) {
        bbbp_set_error("hipFuncSetAttribute(%zu B LDS) failed: %s", bytes, hipGetErrorString(e));
        return BBBP_ERR_HIP;
    }
    granted[key] = bytes;
    return BBBP_OK;
}

void bbbp_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* bbbp_last_error(void) { return g_err; }

// CUs kept out of every persistent grid (they are all sized from bbbp_num_cus()): room for RCCL's ring kernels beside the conv work-groups
// in a multi-GPU run.  Initial value BBBP_COMM_CUS (default 0); bench.py's N-rank diagnostic pass measures 0 against 8 and keeps the
// faster one (bbbp_set_comm_cus).  Never measured on a multi-GPU node before that pass runs there.
static int g_comm_cus = -1;
static int comm_cus() {
    if (g_comm_cus < 0) { const char* e = getenv("BBBP_COMM_CUS"); const int v = e ? atoi(e) : 0; g_comm_cus = v < 0 ? 0 : v; }
    return g_comm_cus;
}
extern "C" int bbbp_set_comm_cus(int n) { const int prev = comm_cus(); g_comm_cus = n < 0 ? 0 : n; return prev; }

int bbbp_num_cus() {
    static int cached[64] = {0};                  // per device: the hardware's count (benign race: every thread computes the same value)
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
    if (cached[dev] == 0) {
        hipDeviceProp_t prop;
        int n = (hipGetDeviceProperties(&prop, dev) == hipSuccess) ? prop.multiProcessorCount : 0;
        cached[dev] = n > 0 ? n : 256;
    }
    const int n = cached[dev], c = comm_cus();
    return (c > 0 && n - c >= 64) ? n - c : n;
}

extern "C" int bbbp_abi_version(void) { return 1; }

// Experiment / tuning knob: reserve CUs for side-stream kernels (see common.h).  Returns the previous reservation.
extern "C" int bbbp_set_partition(int reserved_cus, size_t small_lds_pad) {
    int prev = g_bbbp_reserved_cus;
    g_bbbp_reserved_cus = reserved_cus;
    g_bbbp_small_lds_pad = small_lds_pad;
    return prev;
}

// Op-level dropout streams keyed from device memory (round 4): every seeded entry point (bbbp_dropout, bbbp_softmax_*, bbbp_layernorm_*,
// bbbp_linear* with output dropout, bbbp_attention_*) mixes *base into its `seed` argument when a base is set for the CALLING THREAD:
// effective seed = *base * 0x9E3779B97F4A7C15 + seed.  A training step captured into a HIP graph thereby draws new masks on every replay
// (the caller bumps *base before each one) although the recorded `seed` arguments are frozen.  NULL (the default) restores seed-only streams.
extern "C" int bbbp_set_seed_base(const void* base_dev) {
    g_bbbp_seed_base = static_cast<const unsigned long long*>(base_dev);
    return BBBP_OK;
}

// The conv2-family weight gradient has two structured-sparse forms (conv_b3.hip): 8 waves (fastest alone) and 4 waves (one wave per SIMD: the
// form to run while ANOTHER branch's small kernels share the GPU).  bbbp_mixed_backward picks by itself; a caller that composes the model
// op by op and overlaps its branches on two streams says so for the calling thread around its bbbp_conv3x3_relu_pool_bwd_weight call.
// Returns the previous setting.
extern "C" int bbbp_set_conv_wgrad_beside_encoder(int on) {
    const int prev = g_bbbp_conv_wgrad_beside_encoder;
    g_bbbp_conv_wgrad_beside_encoder = on ? 1 : 0;
    return prev;
}

// Forward of the 32 -> 64 / 64 -> 128 stages on 64 x 64 maps: 1 selects, for the calling thread, the software-pipelined kernel that runs ONE
// work-group per CU (conv_b3.hip: conv_b3p_kernel) -- 3 % slower alone, but beside an encoder chain the step is 2.8 % faster
// (bbbp_mixed_forward sets it by itself for training plans with an encoder).  Same arithmetic in the same order: bit-identical outputs
// and decisions.  Returns the previous setting.  BBBP_C2_PIPE=0 / 1 overrides every caller.
extern "C" int bbbp_set_conv2_fwd_pipe(int on) {
    const int prev = g_bbbp_conv2_fwd_pipe;
    g_bbbp_conv2_fwd_pipe = on ? 1 : 0;
    return prev;
}
