// Winograd F(2x2, 3x3) form of the flagship CNN's second conv stage (32 -> 64 channels on 64x64 maps, fused
// ReLU + 2x2 max-pool forward; 64 -> 32 channels data gradient), exact-f32 MFMA (v_mfma_f32_32x32x2_f32), gfx950.
//
// Same operation as conv.hip's direct kernel (reference: `nn.Conv2d(32, 64, 3, 1, 1) -> ReLU -> MaxPool2d(2, 2)`,
// Models/multi_input_data_regression_opt_transformer_cnn_20250113.py:87-89) in 2.25x fewer
// multiplies, still all in float32:
//     Y = A^T [ (G g G^T) (.) (B^T d B) ] A        per 2x2 output tile, d = its 4x4 input patch, g = one 3x3 filter
// A 2x2 output tile IS one pooling window, so a lane finishes one pooled output from its own 16 accumulators.
// Sixteen GEMMs (one per transformed position) share the operands' origin:
//     M[pos][co][tile] = sum_ci U[pos][co][ci] * V[pos][ci][tile]
//   U = G g G^T is made once per call by a prep kernel and stays in LDS for the whole (persistent) work-group;
//   V = B^T d B is computed by the lane that owns (tile, ci) from the LDS strip: 8 ds_read_b64 + 32 VALU adds feed 16 MFMAs.
// A work-group = 4 waves, one per SIMD (the 16 x 16 accumulator registers of a wave leave no room for a second one);
// a wave owns one row of 32 tiles x 32 output channels, the work-group 8 output rows of one image; output channels
// beyond 32 are a second work item over the same strip (forward: 2 blocks), which a different work-group computes.
// A patch that is flat gives V = 0 except at position (1,1), so the four outputs of its window are bit-equal and the
// pool keeps the first maximum exactly like the direct form (the white background of the depictions stays tie-stable).
#include "common.h"
#include "bbbp_hip.h"
#include <stdlib.h>

namespace {

constexpr int W_FWD = 0;
constexpr int W_DGRAD = 1;
constexpr int IMG = 64;                          // side of the conv's full-resolution maps
constexpr int ROWS = 10;                         // strip: 8 output rows + halo
constexpr int LDW = 72;
constexpr int PLANE = ROWS * LDW;
constexpr int XOFF = 5;                          // image column x sits at index x + XOFF: patch pairs are 8-byte aligned

struct WinoParams {
    const float* x;         // FWD: input [B][32][64][64];  DGRAD: pooled gradient [B][64][32][32]
    const uint8_t* xmask;   // DGRAD: its mask
    const float* u;         // transformed filters [block][pos 16][k][32]
    const float* bias;      // FWD: [64]
    float* y;               // FWD: pooled [B][64][32][32];  DGRAD: [B][32][64][64]
    uint8_t* ymask;         // FWD
    int B;
    int xcd_pairing;        // forward: pair the two channel blocks of a strip on one XCD
};

template <int MODE>
struct WinoCfg {
    static constexpr int KIN = MODE == W_FWD ? 32 : 64;      // reduction channels
    static constexpr int MOUT = MODE == W_FWD ? 64 : 32;     // produced channels
    static constexpr int NCB = MOUT / 32;
    static constexpr int CC = MODE == W_FWD ? 8 : 4;         // channels per LDS stage
    static constexpr int NCH = KIN / CC;
    static constexpr int XST = CC * PLANE;
    static constexpr int UF = 16 * KIN * 32;
    static constexpr size_t LDS_BYTES = (size_t)(UF + 2 * XST) * sizeof(float);
};

__device__ unsigned long long g_wino_clock[2];
// phase breakdown of work-group 0 / wave 0 (PROBE instantiation, selected by BBBP_WINO_PROBE=1; printed by tools/bench_conv2.py): shader cycles spent in
// [0] accumulator init, [1] k-steps (LDS reads, transforms, MFMAs), [2] stage hand-over (LDS writes + barrier), [3] output transform + stores
__device__ unsigned long long g_wino_phase[4];

template <int MODE, bool PROBE>
__device__ __forceinline__ void wino_conv_body(const WinoParams& p) {
    using C = WinoCfg<MODE>;
    constexpr int KIN = C::KIN, MOUT = C::MOUT, NCB = C::NCB, CC = C::CC, NCH = C::NCH, XST = C::XST, UF = C::UF;
    const unsigned long long clk0 = __builtin_readcyclecounter(), wall0 = wall_clock64();
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Us = smem;
    float* Xs = smem + UF;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, j = lane & 31, kh = lane >> 5;
    // Work-groups are dealt to the 8 XCDs round-robin by id, and each XCD has its own L2.  The two channel blocks of a strip
    // (forward) therefore go to ids w and w + 8: same XCD, resident together, so the second read of the strip is an L2 hit
    // (ids w and w + 1 would fetch it from HBM into two L2s).  Needs a grid that is a multiple of 16; else plain pairing.
    const bool xcd_pairs = NCB == 2 && (gridDim.x & 15) == 0 && p.xcd_pairing;
    const int cb = NCB == 1 ? 0 : xcd_pairs ? (blockIdx.x >> 3) & 1 : blockIdx.x % NCB;
    const int nstrips = p.B * (IMG / 8);
    const int stride = gridDim.x / NCB;
    // ... and consecutive strips (shared halo rows) to the groups of one XCD
    const int first = NCB == 1 ? xcd_adjacent(blockIdx.x, gridDim.x)
                    : xcd_pairs ? ((stride & 7) ? (blockIdx.x & 7) + 8 * (blockIdx.x >> 4) : (blockIdx.x & 7) * (stride >> 3) + (blockIdx.x >> 4))
                    : blockIdx.x / NCB;

    for (int i = t * 4; i < UF; i += 1024)
        *reinterpret_cast<float4*>(Us + i) = *reinterpret_cast<const float4*>(p.u + (size_t)cb * UF + i);
    for (int i = t * 4; i < 2 * XST; i += 1024) *reinterpret_cast<float4*>(Xs + i) = make_float4(0.f, 0.f, 0.f, 0.f);
    __syncthreads();

    // ---- stage loader, register staged; every load is unconditional (clamped), validity is applied at the LDS write ----
    constexpr int QW = MODE == W_FWD ? IMG / 4 : IMG / 8;
    constexpr int ITEMS = CC * ROWS * QW;
    constexpr int NIT = (ITEMS + 255) / 256;
    f32x4 rg[NIT];
    uint32_t rm[NIT];
    uint32_t okbits = 0;
    auto load_stage = [&](int strip, int chunk) __attribute__((always_inline)) {
        const int b = strip >> 3, h0 = (strip & 7) * 8;
        okbits = 0;
#pragma unroll
        for (int i = 0; i < NIT; ++i) {
            const int raw = t + i * 256;
            const int idx = raw < ITEMS ? raw : ITEMS - 1;
            const int q = idx % QW, row = (idx / QW) % ROWS, ci = idx / (QW * ROWS);
            const int c = chunk * CC + ci, hh = h0 - 1 + row;
            const bool ok = raw < ITEMS && hh >= 0 && hh < IMG;
            okbits |= (ok ? 1u : 0u) << i;
            const int hc = min(max(hh, 0), IMG - 1);
            if (MODE == W_FWD) {
                rg[i] = *reinterpret_cast<const f32x4*>(p.x + (((long)b * KIN + c) * IMG + hc) * IMG + q * 4);
            } else {
                const long off = (((long)b * KIN + c) * (IMG / 2) + (hc >> 1)) * (IMG / 2) + q * 4;
                rg[i] = *reinterpret_cast<const f32x4*>(p.x + off);
                rm[i] = *reinterpret_cast<const uint32_t*>(p.xmask + off);
            }
        }
    };
    auto store_stage = [&](int strip, int buf) __attribute__((always_inline)) {
        const int h0 = (strip & 7) * 8;
        float* xs = Xs + buf * XST;
#pragma unroll
        for (int i = 0; i < NIT; ++i) {
            const int idx = t + i * 256;
            if (idx >= ITEMS) continue;
            const int q = idx % QW, row = (idx / QW) % ROWS, ci = idx / (QW * ROWS);
            const bool ok = (okbits >> i) & 1u;
            if (MODE == W_FWD) {
                float* d = xs + ci * PLANE + row * LDW + XOFF + q * 4;          // odd index: b32, b64, b32
                const f32x4 v = rg[i];
                d[0] = ok ? v.x : 0.f;
                *reinterpret_cast<float2*>(d + 1) = make_float2(ok ? v.y : 0.f, ok ? v.z : 0.f);
                d[3] = ok ? v.w : 0.f;
            } else {
                // expand 4 pooled gradients to the 8 full-resolution columns of image row hh through the mask
                const uint32_t pr = (uint32_t)(((h0 - 1 + row) & 1) * 2);
                const uint32_t m = ok ? rm[i] : 0x04040404u;
                const float g0 = rg[i].x, g1 = rg[i].y, g2 = rg[i].z, g3 = rg[i].w;
                const uint32_t m0 = m & 0xff, m1 = (m >> 8) & 0xff, m2 = (m >> 16) & 0xff, m3 = m >> 24;
                float* d = xs + ci * PLANE + row * LDW + XOFF + q * 8;
                d[0] = m0 == pr ? g0 : 0.f;
                *reinterpret_cast<float2*>(d + 1) = make_float2(m0 == pr + 1 ? g0 : 0.f, m1 == pr ? g1 : 0.f);
                *reinterpret_cast<float2*>(d + 3) = make_float2(m1 == pr + 1 ? g1 : 0.f, m2 == pr ? g2 : 0.f);
                *reinterpret_cast<float2*>(d + 5) = make_float2(m2 == pr + 1 ? g2 : 0.f, m3 == pr ? g3 : 0.f);
                d[7] = m3 == pr + 1 ? g3 : 0.f;
            }
        }
    };

    f32x16 acc[16];
    unsigned long long ph[4] = {0, 0, 0, 0}, tprev = PROBE ? __builtin_readcyclecounter() : 0;
    auto mark = [&](int k) __attribute__((always_inline)) {
        if (PROBE) { const unsigned long long now = __builtin_readcyclecounter(); ph[k] += now - tprev; tprev = now; }
    };
    int strip = first;
    if (strip < nstrips) {
        load_stage(strip, 0);
        store_stage(strip, 0);
    }
    __syncthreads();
    int buf = 0;
    mark(2);
    for (; strip < nstrips; strip += stride) {
        const int b = strip >> 3, h0 = (strip & 7) * 8;
#pragma unroll
        for (int pos = 0; pos < 16; ++pos)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[pos][r] = 0.f;
        if (MODE == W_FWD) {
            // a constant added to position (1,1) lands on all four outputs of the tile: the bias rides in the accumulator
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[5][r] = p.bias[cb * 32 + mfma_row(r, lane)];
        }
        mark(0);
        for (int chunk = 0; chunk < NCH; ++chunk) {
            int nstrip = strip, nchunk = chunk + 1;
            if (nchunk == NCH) { nchunk = 0; nstrip = strip + stride; }
            const bool have_next = nstrip < nstrips;
            if (have_next) load_stage(nstrip, nchunk);

            const float* xb = Xs + buf * XST + kh * PLANE + (2 * wave) * LDW + 2 * j + (XOFF - 1);
            const float* ub = Us + (chunk * CC + kh) * 32 + j;
            constexpr int KS = CC / 2;
            float a[2][16], v[2][16];
            auto ld = [&](int ks, float* av, float* vv) __attribute__((always_inline)) {
                // the patch first: its transform is the first consumer, the A operands are not needed before the next k-step
                f32x2 d[4][2];
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int cp = 0; cp < 2; ++cp)
                        d[r][cp] = *reinterpret_cast<const f32x2*>(xb + 2 * ks * PLANE + r * LDW + 2 * cp);
#pragma unroll
                for (int pos = 0; pos < 16; ++pos) av[pos] = ub[(pos * KIN + 2 * ks) * 32];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    // row r of B^T d as two column pairs (packed f32 adds), then (B^T d) B: columns 0 and 3 are one packed op
                    f32x2 p0, p1;
                    if (r == 0) { p0 = d[0][0] - d[2][0]; p1 = d[0][1] - d[2][1]; }
                    else if (r == 1) { p0 = d[1][0] + d[2][0]; p1 = d[1][1] + d[2][1]; }
                    else if (r == 2) { p0 = d[2][0] - d[1][0]; p1 = d[2][1] - d[1][1]; }
                    else { p0 = d[1][0] - d[3][0]; p1 = d[1][1] - d[3][1]; }
                    const f32x2 q = p0 - p1;
                    vv[r * 4 + 0] = q.x;
                    vv[r * 4 + 1] = p0.y + p1.x;
                    vv[r * 4 + 2] = p1.x - p0.y;
                    vv[r * 4 + 3] = q.y;
                }
            };
            ld(0, a[0], v[0]);
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                __builtin_amdgcn_sched_barrier(0);
                if (ks + 1 < KS) ld(ks + 1, a[(ks + 1) & 1], v[(ks + 1) & 1]);
#pragma unroll
                for (int pos = 0; pos < 16; ++pos) acc[pos] = mfma32(a[ks & 1][pos], v[ks & 1][pos], acc[pos]);
                if (ks + 1 < KS) {
                    // next step's LDS reads first, then this step's sixteen MFMAs back to back, then the next step's transform as
                    // one clump: with one wave per SIMD nothing hides a VALU op's latency, and the same ops spread between the MFMAs
                    // cost the k-step more (1496 vs 1411 cycles, forward 0.443 -> 0.429 ms)
                    __builtin_amdgcn_sched_group_barrier(0x100, 24, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, 16, 0);
                    __builtin_amdgcn_sched_group_barrier(0x002, 64, 0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            mark(1);
            if (have_next) store_stage(nstrip, buf ^ 1);
            __syncthreads();
            buf ^= 1;
            mark(2);
        }
        // ---- output transform A^T m A per accumulator register (one output channel x one tile) ----
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float s0[4], s1[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                s0[c] = acc[c][r] + acc[4 + c][r] + acc[8 + c][r];
                s1[c] = acc[4 + c][r] - acc[8 + c][r] - acc[12 + c][r];
            }
            const float v00 = s0[0] + s0[1] + s0[2], v01 = s0[1] - s0[2] - s0[3];
            const float v10 = s1[0] + s1[1] + s1[2], v11 = s1[1] - s1[2] - s1[3];
            const int co = cb * 32 + mfma_row(r, lane);
            if (MODE == W_FWD) {
                // PyTorch max-pool keeps the FIRST maximum in (h, w) scan order
                float m = v00; int am = 0;
                if (v01 > m) { m = v01; am = 1; }
                if (v10 > m) { m = v10; am = 2; }
                if (v11 > m) { m = v11; am = 3; }
                const long o = (((long)b * MOUT + co) * (IMG / 2) + (h0 >> 1) + wave) * (IMG / 2) + j;
                p.y[o] = m > 0.f ? m : 0.f;
                if (p.ymask) p.ymask[o] = m > 0.f ? (uint8_t)am : (uint8_t)4;
            } else {
                float* o = p.y + (((long)b * MOUT + co) * IMG + h0 + 2 * wave) * IMG + 2 * j;
                *reinterpret_cast<float2*>(o) = make_float2(v00, v01);
                *reinterpret_cast<float2*>(o + IMG) = make_float2(v10, v11);
            }
        }
        mark(3);
    }
    if (blockIdx.x == 0 && t == 0) {
        g_wino_clock[0] = __builtin_readcyclecounter() - clk0;
        g_wino_clock[1] = wall_clock64() - wall0;
        if (PROBE)
            for (int k = 0; k < 4; ++k) g_wino_phase[k] = ph[k];
    }
}

template <int MODE>
__global__ __launch_bounds__(256) void wino_conv_kernel(WinoParams p) { wino_conv_body<MODE, false>(p); }
template <int MODE>
__global__ __launch_bounds__(256) void wino_conv_probe_kernel(WinoParams p) { wino_conv_body<MODE, true>(p); }

// U[block][pos][k][32] = G g G^T, G = [[1,0,0],[.5,.5,.5],[.5,-.5,.5],[0,0,1]], from the reference layout w[64][32][3][3].
//  FWD:   k = input channel, produced channel = block * 32 + m, g = w[co][ci]
//  DGRAD: k = conv output channel, produced channel m = conv input channel, g = w[k][m] rotated by 180 degrees
__global__ void wino_prep_kernel(const float* w, float* u, int mode) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= 64 * 32 * 16) return;
    const int kin = mode == W_FWD ? 32 : 64;
    const int m = idx % 32, k = (idx / 32) % kin, pos = (idx / (32 * kin)) % 16, blk = idx / (32 * kin * 16);
    float g[3][3];
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int c = 0; c < 3; ++c)
            g[a][c] = mode == W_FWD ? w[((blk * 32 + m) * 32 + k) * 9 + a * 3 + c] : w[(k * 32 + m) * 9 + (2 - a) * 3 + (2 - c)];
    const int r = pos >> 2, c = pos & 3;
    // row r of G applied on the left, row c of G on the right
    float gr[3];
#pragma unroll
    for (int q = 0; q < 3; ++q)
        gr[q] = r == 0 ? g[0][q] : r == 3 ? g[2][q] : r == 1 ? 0.5f * (g[0][q] + g[1][q] + g[2][q]) : 0.5f * (g[0][q] - g[1][q] + g[2][q]);
    u[idx] = c == 0 ? gr[0] : c == 3 ? gr[2] : c == 1 ? 0.5f * (gr[0] + gr[1] + gr[2]) : 0.5f * (gr[0] - gr[1] + gr[2]);
}

template <int MODE>
int launch_wino(const WinoParams& p, hipStream_t st) {
    using C = WinoCfg<MODE>;
    static const int cu_cap = [] {
        const char* e = getenv(MODE == W_FWD ? "BBBP_WINO_CUS_FWD" : "BBBP_WINO_CUS_DGRAD");
        if (!e) e = getenv("BBBP_WINO_CUS");
        return e ? atoi(e) : 0;
    }();
    static const int probe = [] { const char* e = getenv("BBBP_WINO_PROBE"); return e ? atoi(e) : 0; }();
    auto kernel = probe ? wino_conv_probe_kernel<MODE> : wino_conv_kernel<MODE>;
    { int rc_ = bbbp_ensure_dyn_lds(reinterpret_cast<const void*>(kernel), (size_t)C::LDS_BYTES); if (rc_) return rc_; }
    const int nwork = p.B * (IMG / 8) * C::NCB;
    int grid = bbbp_num_cus();
    if (cu_cap > 0 && cu_cap < grid) grid = cu_cap;
    else if (cu_cap == 0 && g_bbbp_wino_side_cus > 0 && g_bbbp_wino_side_cus < grid) grid -= g_bbbp_wino_side_cus;
    grid -= grid % C::NCB;
    if (grid > nwork) grid = nwork;
    if (grid < C::NCB) grid = C::NCB;
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(256), C::LDS_BYTES, st, p);
    BBBP_CHECK_LAUNCH();
    return BBBP_OK;
}

}  // namespace

// conv.hip dispatches the 32 -> 64 @ 64x64 stage here when the Winograd form is selected (bbbp_set_conv_winograd).
// workspace: 16 * 32 * 64 floats of transformed filters.
int bbbp_wino_conv2_fwd(hipStream_t st, const float* x, const float* w, const float* bias, float* y, uint8_t* mask, int B,
                        float* workspace) {
    hipLaunchKernelGGL(wino_prep_kernel, dim3(128), dim3(256), 0, st, w, workspace, W_FWD);
    BBBP_CHECK_LAUNCH();
    static const int pairing = [] { const char* e = getenv("BBBP_WINO_XCD_PAIRS"); return e ? atoi(e) : 1; }();
    WinoParams p{x, nullptr, workspace, bias, y, mask, B, pairing};
    return launch_wino<W_FWD>(p, st);
}

int bbbp_wino_conv2_dgrad(hipStream_t st, const float* gy, const uint8_t* gmask, const float* w, float* dx, int B, float* workspace) {
    hipLaunchKernelGGL(wino_prep_kernel, dim3(128), dim3(256), 0, st, w, workspace, W_DGRAD);
    BBBP_CHECK_LAUNCH();
    WinoParams p{gy, gmask, workspace, nullptr, dx, nullptr, B, 0};
    return launch_wino<W_DGRAD>(p, st);
}

int bbbp_wino_last_phases(unsigned long long* phases4) {
    BBBP_CHECK_HIP(hipMemcpyFromSymbol(phases4, HIP_SYMBOL(g_wino_phase), 4 * sizeof(unsigned long long)));
    return BBBP_OK;
}

int bbbp_wino_last_clock(unsigned long long* shader_cycles, unsigned long long* ticks_100mhz) {
    unsigned long long h[2] = {0, 0};
    BBBP_CHECK_HIP(hipMemcpyFromSymbol(h, HIP_SYMBOL(g_wino_clock), sizeof(h)));
    *shader_cycles = h[0]; *ticks_100mhz = h[1];
    return BBBP_OK;
}
