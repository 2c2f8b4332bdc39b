// Shared helpers for the gfx950 kernels of the BBBP hot path.  CDNA4 only: wave = 64 lanes.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

#define BBBP_OK 0
#define BBBP_ERR_ARG 1
#define BBBP_ERR_HIP 2
#define BBBP_ERR_WORKSPACE 3

// last-error string, one per host thread (never throws across the C ABI)
void bbbp_set_error(const char* fmt, ...);

#define BBBP_CHECK_ARG(cond, ...)                 \
    do {                                          \
        if (!(cond)) {                            \
            bbbp_set_error(__VA_ARGS__);          \
            return BBBP_ERR_ARG;                  \
        }                                         \
    } while (0)

#define BBBP_CHECK_HIP(expr)                                                            \
    do {                                                                                \
        hipError_t _e = (expr);                                                         \
        if (_e != hipSuccess) {                                                         \
            bbbp_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
            return BBBP_ERR_HIP;                                                        \
        }                                                                               \
    } while (0)

#define BBBP_CHECK_LAUNCH() BBBP_CHECK_HIP(hipGetLastError())

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }
static inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

int bbbp_num_cus();   // cached multiProcessorCount of the current device
int bbbp_ensure_dyn_lds(const void* kernel, size_t bytes);   // hipFuncAttributeMaxDynamicSharedMemorySize once per (kernel, device)

// CU partitioning for the two-branch overlap (engine.hip).  A persistent conv work-group takes >= 120 KB of a CU's
// 160 KB LDS, so exactly one fits per CU; with `reserved_cus` > 0 the conv grids shrink to (CUs - reserved) and the
// small side-stream kernels request `small_lds_pad` bytes (> 40 KB) so that they can ONLY land on the CUs the conv
// grids left free.  Measured without it: a 5 us kernel sharing CUs with a conv kernel takes 35-85 us.
extern thread_local int g_bbbp_reserved_cus;
extern thread_local size_t g_bbbp_small_lds_pad;
// engine scope: keep the first conv stage's forward on the f32 kernel for this call even when bit 6 of the conv mask selects the split-bf16
// form.  Set while a training step's encoder chain runs beside the image branch: the split-bf16 kernel is faster alone (0.19 vs 0.26 ms
// at B = 512) but holds 2 x 248 registers per lane slot on every SIMD, and the forward pass of that step is bound by the encoder's
// latency chain, which then finds no wave slots (measured: conv1 0.31 -> 0.20 ms in-step, encoder forward 1.21 -> 1.33, step 2.61 -> 2.70)
extern thread_local int g_bbbp_conv1_fwd_f32;
// engine scope: work-groups per CU for the split-bf16 conv1 forward (0 = the kernel's own default).  The software-pipelined form keeps the
// matrix pipe busy with ONE wave per SIMD, so beside an encoder chain the engine asks for one work-group per CU (half the register file and
// 100 KB of LDS stay free for the chain's kernels)
extern thread_local int g_bbbp_conv1_fwd_per_cu;
// conv2's weight gradient on the structured-sparse MFMA (conv_b3.hip: conv_b3_wgrad_sp_kernel) has an 8-wave form (fastest alone: two waves
// of 256 registers per SIMD) and a 4-wave form that leaves ~200 registers per lane slot to the fingerprint branch's kernels; the engine
// asks for the latter while an encoder chain runs beside the image branch (0 = no preference: 8 waves)
extern thread_local int g_bbbp_conv_wgrad_beside_encoder;
// conv_b3.hip, forward of the 64 x 64-map stages: 1 = the software-pipelined one-work-group-per-CU kernel (what the engine asks for while an encoder
// chain runs beside the image branch: slower alone, but it leaves the chain three quarters of every SIMD), 0 = two work-groups per CU
extern thread_local int g_bbbp_conv2_fwd_pipe;
extern thread_local int g_bbbp_wino_side_cus;      // CUs the Winograd conv grids leave free while the engine overlaps its branches
// head.hip: fused fusion-block + regression-head forward (two launches); `partial`: ceil(B/16) * 2 * 256 floats
int bbbp_head_forward_fused(hipStream_t st, const float* comb, const float* const* fw1, const float* const* fb1,
                            const float* const* fw2, const float* const* fb2, const float* w0, const float* b0, const float* gamma,
                            const float* beta, float* running_mean, float* running_var, const float* w3, const float* b3,
                            const float* w5, const float* b5, const float* w7, const float* b7, float* hid, float* attn, float* fused,
                            float* h, float* hb, float* bn_mean, float* bn_rstd, float* h2, float* h3, float* out, float* partial,
                            int B, int training, int concat);
// head.hip: the input-gradient chain of the head and fusion block backward (two launches); `partial` as above
int bbbp_head_backward_fused(hipStream_t st, const float* dout, const float* comb, const float* hid, const float* attn, const float* h,
                             const float* h2, const float* h3, const float* bn_mean, const float* bn_rstd, const float* gamma,
                             const float* const* fw1, const float* const* fw2, const float* w0, const float* w3, const float* w5,
                             const float* w7, float* dh3, float* dh2, float* dhb, float* dh, float* dlogit, float* dpre, float* dcomb,
                             float* dgamma, float* dbeta, float* partial, int B, int training);
// encoder.hip: the row-local stretches of an encoder layer as single launches (see the file header)
struct bbbp_enc_row_fwd_args {
    const float* ctx; const float* xin;
    const float *wo, *bo, *g1, *be1, *w1, *b1, *w2, *b2, *g2, *be2;
    const float *wn, *bn; float* outn; int nn, ldn, actn;
    float *z1, *y1, *hff, *z2, *y2, *mean1, *rstd1, *mean2, *rstd2;
    int B, F, DFF; float p; uint64_t seed1, seed2, seed3;
};
struct bbbp_enc_row_bwd_args {
    const float* dqkv_up; const float* win_up; const float* dz1_up; float* dyout;
    const float *z2, *mean2, *rstd2, *g2, *w2, *hff, *w1, *z1, *mean1, *rstd1, *g1, *wo;
    float *dz2, *dff, *dhff, *dy1, *dz1, *dsa, *dctx;
    int B, F, DFF; float p; uint64_t seed1, seed3;
};
bool bbbp_enc_rows_supported(int F, int nhead, int dff);
int bbbp_enc_row_fwd(hipStream_t st, const bbbp_enc_row_fwd_args* a);
int bbbp_enc_row_bwd(hipStream_t st, const bbbp_enc_row_bwd_args* a);
int bbbp_ln_param_grad_multi(hipStream_t st, int n, const float* const* dy, const float* const* z, const float* const* mean,
                             const float* const* rstd, float* const* dgamma, float* const* dbeta, int rows, int cols);
// attention.hip: fused (flash-style) self-attention for many heads of head_dim 8 / 16, one work-group per head
bool bbbp_attn_small_supported(int B, int nhead, int head_dim);
bool bbbp_attn_wide_supported(int B, int nhead, int head_dim);      // one wide head (161..176 columns): opt-in, see attention.hip
// `keep` (optional, bbbp_attn_small_keep_bytes): the forward pass leaves its dropout decisions there and the backward pass reads them
// back instead of drawing the Philox blocks again; NULL: backward recomputes them (same stream, same result)
size_t bbbp_attn_small_keep_bytes(int B, int nhead, int head_dim);
int bbbp_attn_small_fwd(hipStream_t st, const float* qkv, float* ctx, float* lse, int B, int F, int nhead, float scale, float p, uint64_t seed,
                        uint8_t* keep = nullptr);
int bbbp_attn_small_bwd(hipStream_t st, const float* qkv, const float* ctx, const float* lse, const float* dctx, float* dqkv, int B, int F,
                        int nhead, float scale, float p, uint64_t seed, const uint8_t* keep = nullptr);
constexpr size_t BBBP_CONV_MIN_LDS = 120 * 1024;
// conv_wino.hip: Winograd F(2x2,3x3) form of the 32 -> 64 @ 64x64 stage; workspace = 16*32*64 floats of transformed filters
int bbbp_wino_conv2_fwd(hipStream_t st, const float* x, const float* w, const float* bias, float* y, uint8_t* mask, int B,
                        float* workspace);
int bbbp_wino_conv2_dgrad(hipStream_t st, const float* gy, const uint8_t* gmask, const float* w, float* dx, int B, float* workspace);
int bbbp_wino_last_clock(unsigned long long* shader_cycles, unsigned long long* ticks_100mhz);
// conv_b3.hip: the same stage as a direct implicit GEMM on the bf16 matrix pipe with every float32 operand split into three bf16
// pieces (six products per float32 product, float32 accumulate); workspace = bbbp_b3_workspace_bytes() of pre-split filters
size_t bbbp_b3_workspace_bytes();
// round 4: the same kernels for the two large stages of the wide / deep variant (64 -> 128 @ 64 x 64, 128 -> 256 @ 32 x 32)
bool bbbp_b3_conv_supported(int cin, int cout, int hw);
size_t bbbp_b3_workspace_bytes(int cin, int cout);
int bbbp_b3_conv_fwd(hipStream_t st, const float* x, const float* w, const float* bias, float* y, uint8_t* mask, int B, int cin, int cout, void* workspace);
int bbbp_b3_conv_dgrad(hipStream_t st, const float* gy, const uint8_t* gmask, const float* w, float* dx, int B, int cin, int cout, void* workspace);
int bbbp_b3_conv2_fwd(hipStream_t st, const float* x, const float* w, const float* bias, float* y, uint8_t* mask, int B, void* workspace);
int bbbp_b3_conv2_dgrad(hipStream_t st, const float* gy, const uint8_t* gmask, const float* w, float* dx, int B, void* workspace);
// form: 0 dense split-bf16, 1 structured-sparse MFMA (8 waves, or 4 beside an encoder chain), 2 structured-sparse, 4 waves
// (cin_total, cout_total, groups: the stage's channel counts and the work-groups per (32 ci, 64 co) block pair; grid = pairs * groups;
//  map: 64 x 64 maps, or 32 x 32 for the sparse forms)
int bbbp_b3_conv2_wgrad(hipStream_t st, const float* x, const float* gy, const uint8_t* mask, float* slab, float* bslab, int B, int grid, int form,
                        int cin_total = 32, int cout_total = 64, int groups = 0, int map = 64);
int bbbp_b3_last_clock(unsigned long long* shader_cycles, unsigned long long* ticks_100mhz);
// conv_b3c1.hip: forward of the first stage (3 -> 32 @ 128x128) in the same arithmetic, channel-innermost LDS strip, no operand assembly
size_t bbbp_b3_conv1_fwd_workspace_bytes();
int bbbp_b3_conv1_fwd(hipStream_t st, const float* x, const float* w, const float* bias, float* y, uint8_t* mask, int B, void* workspace);
int bbbp_b3_conv1_wgrad(hipStream_t st, const float* x, const float* gy, const uint8_t* mask, float* slab, float* bslab, int B, int grid);
int bbbp_wino_last_phases(unsigned long long* phases4);     // BBBP_WINO_PROBE=1 builds of the kernel only
// rowops.hip: a slice of the optimizer step deferred to a side stream (bbbp_adamw_step_deferred).  Entry points that read parameters
// order themselves behind it: bbbp_param_wait(stream, ptr) waits when ptr lies in the slice (null: unconditionally)
int bbbp_param_wait(hipStream_t st, const void* ptr);
bool bbbp_param_pending_elsewhere(const void* ptr);
// gemm.hip: LayerNorm absorbed by the consuming Linear (bbbp_layernorm_linear_fwd) -- is that the faster form for this product?
bool bbbp_layernorm_linear_preferred(int M, int N, int K);
// fold.hip: out_proj folded into the value projection of a one-head encoder layer (W' = Wo Wv, b' = Wo bv) and the gradients unfolded
bool bbbp_outproj_fold_supported(int F, int nhead, int layers);
size_t bbbp_outproj_fold_floats(int F, int which);          // 0 wf [3F][F], 1 bf [3F], 2 wvt [F + 1][F], 3 tdw [F][F], 4 tdb [F]
int bbbp_outproj_fold(hipStream_t st, int layers, int F, const float* const* win, const float* const* bin, const float* const* wo,
                      const float* const* bo, float* const* wf, float* const* bf, float* const* wvt);
int bbbp_outproj_unfold(hipStream_t st, int F, int B, const float* tdw, const float* tdb, const float* wo, const float* wvt, const float* dz,
                        int lddz, float* g_outw, float* g_outb, float* g_inw_v, float* g_inb_v);

// Work-groups are dealt to the 8 XCDs round-robin by id and every XCD has its own L2.  Persistent kernels whose consecutive
// work items share input rows (conv strips and their halos) map group w of n to logical index (w % 8) * (n / 8) + w / 8: the
// groups of one XCD then hold consecutive items at the same time and the shared rows are L2 hits instead of second HBM reads.
__device__ __forceinline__ int xcd_adjacent(int w, int n) { return (n & 7) ? w : (w & 7) * (n >> 3) + (w >> 3); }

// exact f32 MFMA: D[32x32] += A[32x2] * B[2x32]; lane l holds A[l&31][l>>5], B[l>>5][l&31];
// D: col = l&31, row = (r&3) + 8*(r>>2) + 4*(l>>5) for register r of 16.
__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ int mfma_row(int r, int lane) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }

// Split-bf16 arithmetic (conv_b3.hip, gemm.hip): a float is the exact sum of three bf16 pieces hi + mid + lo (3 x 8 significand bits),
// and a product of two floats is, to float32 accuracy, the sum of the six piece products hi*hi, hi*mid, mid*hi, hi*lo, lo*hi, mid*mid --
// six v_mfma_f32_32x32x16_bf16 (f32 accumulate) in place of eight v_mfma_f32_32x32x2_f32, at 2.67x the matrix rate.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2v __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));      // clang vectors stay in registers; arrays of HIP's uint4 struct were demoted to scratch
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
// hi / mid / lo of two floats, packed pairwise (a in the low half).  The residuals are formed with packed-f32 subtractions.
// Range: |x| > 3.3895e38 (the largest bf16, 0x7F7F) rounds to +-inf in the `hi` piece and the residuals become -+inf / NaN: such an
// element poisons its products like an overflow would.  Irrelevant on this path (standardised inputs, weights of order 1, gradients
// far below 1e38) -- stated so that nobody re-uses the split on unscaled data without a clamp.  NaN stays NaN, +-inf stays +-inf in
// `hi` (v_cvt_pk_bf16_f32) with NaN residuals.
__device__ __forceinline__ void split2(float a, float b, uint32_t& hi, uint32_t& mid, uint32_t& lo) {
    f32x2v v = {a, b};
    bf16x2 h = __builtin_convertvector(v, bf16x2);
    hi = __builtin_bit_cast(uint32_t, h);
    f32x2v hf = {__builtin_bit_cast(float, hi << 16), __builtin_bit_cast(float, hi & 0xffff0000u)};
    v = v - hf;
    bf16x2 m = __builtin_convertvector(v, bf16x2);
    mid = __builtin_bit_cast(uint32_t, m);
    f32x2v mf = {__builtin_bit_cast(float, mid << 16), __builtin_bit_cast(float, mid & 0xffff0000u)};
    v = v - mf;
    bf16x2 l = __builtin_convertvector(v, bf16x2);
    lo = __builtin_bit_cast(uint32_t, l);
}

// Philox-4x32-10 counter RNG for dropout masks (recomputed, never stored)
__device__ __forceinline__ uint4 philox4(uint64_t seed, uint64_t ctr) {
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
    uint32_t c0 = (uint32_t)ctr, c1 = (uint32_t)(ctr >> 32), c2 = 0x9E3779B9u, c3 = 0xBB67AE85u;
#pragma unroll
    for (int i = 0; i < 10; ++i) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    return make_uint4(c0, c1, c2, c3);
}
// The engine keeps the per-call dropout seed in DEVICE memory (a slot of the forward workspace, written by a one-thread
// kernel) so that the enqueued work does not depend on it and can be replayed as a HIP graph; kernels then receive the
// slot's address in `base` and a per-site salt in `seed`.  The op-level C entry points pass base = nullptr.
extern thread_local const unsigned long long* g_bbbp_seed_base;
__device__ __forceinline__ uint64_t effective_seed(uint64_t seed, const unsigned long long* base) {
    return base ? (uint64_t)(*base) * 0x9E3779B97F4A7C15ull + seed : seed;
}
// keep-scale for element `idx` of dropout stream `seed`: 0 or 1/(1-p)
__device__ __forceinline__ float dropout_scale(uint64_t seed, uint64_t idx, float p, float inv_keep) {
    uint4 r = philox4(seed, idx >> 2);
    uint32_t v = (idx & 3) == 0 ? r.x : (idx & 3) == 1 ? r.y : (idx & 3) == 2 ? r.z : r.w;
    // uniform in [0,1): top 24 bits
    float u = (float)(v >> 8) * (1.0f / 16777216.0f);
    return u >= p ? inv_keep : 0.0f;
}

// Small latency-bound kernels run beside persistent MFMA-bound conv work-groups (engine.hip, two streams); raising
// their wave priority lets them win issue arbitration against the older conv waves on the same SIMD.
#ifndef BBBP_HIGH_PRIO
#define BBBP_HIGH_PRIO() __builtin_amdgcn_s_setprio(3)
#endif
