// The flagship CNN's second conv stage (32 -> 64 channels on 64x64 maps: `nn.Conv2d(32, 64, 3, 1, 1) -> ReLU -> MaxPool2d(2, 2)`,
// Models/multi_input_data_regression_opt_transformer_cnn_20250113.py:87-89) as a direct implicit GEMM whose
// float32 products run on the BF16 matrix pipe: every operand x is split ONCE, when it is staged into LDS, into three bfloat16
// pieces  x = hi + mid + lo  (hi = bf16(x), mid = bf16(x - hi), lo = bf16(x - hi - mid): 24 significand bits, the subtractions
// are exact), and a product a*b becomes the six bf16 products  hi*hi + hi*mid + mid*hi + hi*lo + lo*hi + mid*mid  accumulated in
// float32 inside v_mfma_f32_32x32x16_bf16 (the three dropped terms are <= 2^-24 |a b|: below the rounding of the float32 product).
// gfx950's f32-input MFMA runs at 1/16 of the bf16 rate and blocks the SIMD's vector issue for its whole 64 cycles; six bf16
// MFMAs do the work of eight of them in 6 x 32 cycles and hold the vector issue for 8 of every 32, so the same float32 arithmetic
// (to ~1 ulp per product) gets 2.67x the matrix rate and leaves the VALU to the stage loader and to the other stream's kernels.
//
// GEMM view (forward): D[co][pixel] = sum_{tap, ci} W[co][ci][tap] X[ci][pixel + tap]; a k-step is one tap x 16 channels.  LDS
// holds the input strip channel-INNERMOST per pixel ([plane][row][px][16 ch] bf16), so a lane's B fragment -- 8 consecutive
// channels of its pixel -- is one aligned ds_read_b128 and a tap only moves the pixel index; the weights arrive pre-split from a
// prep kernel in exactly the A-fragment order ([tap][plane][k-half][m 32][8]).  A work-group = 4 waves (one per SIMD) owns a strip
// of 4 output rows x 64 pixels x 32 output channels; a wave owns 2 rows x 32 columns (two accumulator tiles: the 2x2 pool partners
// are its own two tiles and the neighbouring lane), channels stream through a double-buffered stage of 16.
// The data gradient is the same kernel: the loader expands the pooled gradient through the arg-max mask, the prep kernel flips the
// filter taps and swaps the channel roles, the epilogue stores full-resolution rows.
#include "common.h"
#include "bbbp_hip.h"
#include <stdlib.h>

namespace {

constexpr int B3_FWD = 0, B3_DGRAD = 1;
constexpr int IMG = 64;
constexpr int R = 4, ROWS = R + 2;            // output rows per strip, staged rows (halo)
constexpr int PXW = IMG + 2;                  // staged pixels per row: index 0 is x = -1, index 65 is x = 64 (always zero)
constexpr int CH = 16;                        // channels per stage = K of one MFMA
constexpr int XPLANE = ROWS * PXW * CH;       // bf16 elements of one plane of one stage
constexpr int XBUF = 3 * XPLANE;
constexpr int WSTAGE = 9 * 3 * 2 * 32 * 8;    // bf16 elements: [tap][plane][k-half][m][8]
[[maybe_unused]] constexpr size_t LDS_BYTES = (size_t)(XBUF + WSTAGE) * 2;      // 65.7 KB (flagship geometry): two work-groups per CU

struct B3Params {
    const float* x;         // FWD: input [B][32][64][64];  DGRAD: pooled gradient [B][64][32][32]
    const uint8_t* xmask;   // DGRAD: its arg-max / ReLU mask
    const uint16_t* wp;     // pre-split filters: [m-block][chunk][WSTAGE]
    const float* bias;      // FWD: [64]
    float* y;               // FWD: pooled [B][64][32][32];  DGRAD: [B][32][64][64]
    uint8_t* ymask;         // FWD
    int B;
    int prio;               // wave priority the kernel raises itself to (0: leave it): see conv_bwd_prio()
};

// filters -> [m-block][chunk][tap][plane][k-half][m 32][8] bf16.
//   FWD  : m = output channel co (2 blocks of 32), k = input channel ci (2 chunks of 16), tap = kh * 3 + kw
//   DGRAD: m = input channel ci (1 block), k = output channel co (4 chunks of 16), tap' = the flipped tap
__global__ void b3_prep_kernel(const float* __restrict__ w, uint16_t* __restrict__ wp, int mode, int cin, int cout) {
    // FWD  : m-blocks = cout / 32, chunks = cin / 16;   DGRAD: m-blocks = cin / 32, chunks = cout / 16
    const int nmb = (mode == B3_FWD ? cout : cin) / 32, nch = (mode == B3_FWD ? cin : cout) / 16;
    const int total = nmb * nch * 9 * 2 * 32 * 4;                 // (mb, chunk, tap, half, m, pair-of-k)
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
        int r = idx;
        const int pr = r % 4; r /= 4;
        const int m = r % 32; r /= 32;
        const int h = r % 2; r /= 2;
        const int tap = r % 9; r /= 9;
        const int chunk = r % nch; r /= nch;
        const int mb = r;
        float v[2];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int k = chunk * 16 + 8 * h + 2 * pr + e;
            if (mode == B3_FWD) v[e] = w[((long)(mb * 32 + m) * cin + k) * 9 + tap];          // W[co][ci][tap]
            else v[e] = w[((long)k * cin + mb * 32 + m) * 9 + (8 - tap)];                      // W[co = k][ci = mb * 32 + m][flipped tap]
        }
        uint32_t hi, mid, lo;
        split2(v[0], v[1], hi, mid, lo);
        uint32_t* dst = reinterpret_cast<uint32_t*>(wp + (size_t)(mb * nch + chunk) * WSTAGE);
        const int base = ((tap * 3 * 2 + h) * 32 + m) * 4 + pr;       // plane stride: 2 * 32 * 4 words
        dst[base] = hi; dst[base + 2 * 32 * 4] = mid; dst[base + 2 * 2 * 32 * 4] = lo;
    }
}

// The stages this form serves: the flagship's second stage (32 -> 64 @ 64 x 64) and, since round 4, the two large stages of the wide / deep
// variant (Models/..._opt_20250107_network.py:129-138: 64 -> 128 @ 64 x 64 and 128 -> 256 @ 32 x 32).  A work-group is always 4 waves of
// 2 rows x 32 columns: 4 rows x 64 pixels on the 64-wide maps, 8 rows x 32 pixels on the 32-wide ones.
template <int CIN_, int COUT_, int IMGS_>
struct B3Geom { static constexpr int CIN = CIN_, COUT = COUT_, IMGS = IMGS_; };
using GeomFlagship = B3Geom<32, 64, 64>;

// phase breakdown of work-group 0 / wave 0 (BBBP_B3_PROBE=1 selects the stamping instantiation; tools/bench_conv2.py prints it): shader
// cycles in [0] global-load issue, [1] MFMA block of a stage, [2] split + LDS writes + barriers, [3] epilogue
__device__ unsigned long long g_b3_phase[4];
// shader cycles and 100 MHz wall ticks work-group 0 of the last forward / data-gradient launch spent (bbbp_conv_last_clock): their
// ratio is the clock the chip sustains under this kernel -- ~1.9 GHz for bf16 MFMA loops on real data, not the 2.4 GHz of the spec
__device__ unsigned long long g_b3_clock[2];

template <int MODE, bool PROBE, class G>
__device__ __forceinline__ void conv_b3_body(const B3Params& p) {
    // geometry of this instantiation (the names shadow the flagship constants above, which the weight-gradient kernels keep using)
    constexpr int IMG = G::IMGS, R = 256 / IMG, ROWS = R + 2, PXW = IMG + 2, XPLANE = ROWS * PXW * CH, XBUF = 3 * XPLANE;
    constexpr int KIN = MODE == B3_FWD ? G::CIN : G::COUT;   // reduction channels
    constexpr int NCHUNK = KIN / CH;
    constexpr int NMB = (MODE == B3_FWD ? G::COUT : G::CIN) / 32;      // blocks of 32 produced channels
    constexpr int NOUT = NMB * 32;                           // produced channels in all
    constexpr int SRC_PLANE = MODE == B3_FWD ? IMG * IMG : (IMG / 2) * (IMG / 2);     // floats per source channel plane
    extern __shared__ __attribute__((aligned(16))) uint16_t smem[];
    uint16_t* Xs = smem;                                     // [plane 3][row][px][16]
    uint16_t* Ws = smem + XBUF;                              // [WSTAGE]
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, r = lane & 31, h = lane >> 5;
    const unsigned long long clk_begin = __builtin_readcyclecounter(), wall_begin = wall_clock64();
    if (p.prio >= 3) __builtin_amdgcn_s_setprio(3); else if (p.prio == 2) __builtin_amdgcn_s_setprio(2); else if (p.prio == 1) __builtin_amdgcn_s_setprio(1);

    // the produced-channel block is fixed per work-group; blocks of one strip sit on one XCD (ids w and w + 8: common.h)
    const bool pairs = NMB == 2 && (gridDim.x & 15) == 0;
    const int mb = NMB == 1 ? 0 : pairs ? (blockIdx.x >> 3) & 1 : blockIdx.x % NMB;
    const int nstrips = p.B * (IMG / R);
    const int stride = gridDim.x / NMB;
    const int first = NMB == 1 ? xcd_adjacent(blockIdx.x, gridDim.x)
                    : pairs ? ((stride & 7) ? (blockIdx.x & 7) + 8 * (blockIdx.x >> 4) : (blockIdx.x & 7) * (stride >> 3) + (blockIdx.x >> 4))
                    : blockIdx.x / NMB;

    for (int i = t * 8; i < XBUF; i += 256 * 8) *reinterpret_cast<u32x4*>(Xs + i) = u32x4{0, 0, 0, 0};

    // ---- stage loader: item = (channel group of 8, row, pixel); 8 dword loads -> split -> three 16-byte LDS writes.
    //      Everything that does not depend on the strip is computed once: a stage costs one uniform base + 32-bit offsets. ----
    constexpr int ITEMS = 2 * ROWS * IMG, NIT = (ITEMS + 255) / 256;       // (32-wide maps: 640 items, the last round half empty)
    constexpr bool RAGGED = ITEMS % 256 != 0;
    constexpr int WPIECES = WSTAGE * 2 / 16, WIT = (WPIECES + 255) / 256;       // 16-byte pieces of a filter stage per thread
    // item i of thread t: idx = t + 256 i -> pixel px = t % 64 (the same for every item), row (t / 64 + 4 i) % 6, channel group idx / 384.
    // RECOMPUTE (round 4, measured, OFF): the flagship's data-gradient kernel stands at 166 + 62 = 228 registers, two work-groups per CU =
    // 2 x 232 per SIMD, which leaves 48 -- not one 64-register wave of the encoder chain can be placed on ANY SIMD while conv2's data gradient
    // runs: the chain stalls for the kernel's whole 0.4 ms in every step (profiles/r04_stall_outliers.txt: exactly one chain launch and one leaf
    // launch of ~400 us per step, both beside conv_b3_kernel<1>: the "stall outliers" of VERDICT round 3).  Recomputing these twelve index
    // registers per stage from an opaque copy of t brings the kernel to 154 + 62 = 216 and the chain runs beside it (its backward 1.30 -> 1.13-1.19
    // ms on the device timeline) -- but conv2's data gradient then takes 0.46-0.49 ms instead of 0.40, the weight gradient before it 0.39
    // instead of 0.37, and the STEP gets slower: 2.453-2.474 -> 2.487-2.496 ms (profiles/r04_dgrad_registers.txt).  Time-slicing this one
    // kernel against the chain beats co-running them; the stall is the better schedule, so the twelve registers stay.
    constexpr bool RECOMPUTE = false;
    // EARLY_LOADS (round 4; the schedule of the MFMA block below): the data gradient's kernel gains 9 % alone (0.39 -> 0.355 ms at B = 512; its
    // 48 loads per stage -- gradients + mask bytes -- were the longest issue phase), the forward kernel LOSES 4 % at B = 4096 (3.15 -> 3.29 ms:
    // 24 loads, and 52 more live registers through its two unrolled chunk copies) -- measured A/B on one box, profiles/r04_conv2_early_loads.txt.
    // -DB3_EARLY_LOADS=0 / 1 forces it off / on for both.
#ifdef B3_EARLY_LOADS
    constexpr bool EARLY_LOADS = B3_EARLY_LOADS != 0;
#else
    constexpr bool EARLY_LOADS = MODE == B3_DGRAD;
#endif
    int goff[NIT], loff[NIT], irow[NIT], ipar[NIT];
    auto item = [&](int tt, int i, int& g, int& l, int& rw, int& pr) __attribute__((always_inline)) {
        const int idx = RAGGED ? min(tt + i * 256, ITEMS - 1) : tt + i * 256;        // (clamped: a ragged item loads valid memory and is never stored)
        const int px = idx % IMG, row = (idx / IMG) % ROWS, cg = idx / (IMG * ROWS);
        g = cg * 8 * SRC_PLANE + (MODE == B3_FWD ? px : px >> 1);
        l = (row * PXW + px + 1) * CH + cg * 8;
        rw = row; pr = px & 1;
    };
    if (!RECOMPUTE) {
#pragma unroll
        for (int i = 0; i < NIT; ++i) item(t, i, goff[i], loff[i], irow[i], ipar[i]);
    }
    auto opaque_t = [&]() __attribute__((always_inline)) { int tt = t; asm volatile("" : "+v"(tt)); return tt; };
    float xr[NIT][8];
    uint32_t mr[NIT][2];                                     // DGRAD: the eight mask bytes of an item
    u32x4 wr[WIT];
    uint32_t okbits = 0;
    // a stage's loads come in NIT + 1 pieces -- item i of every thread (8 dword loads; DGRAD: + 8 mask bytes), then the filter stage --
    // so that the MFMA block can place one piece per tap (below)
    const char* ld_xb = nullptr; const uint8_t* ld_mb = nullptr; int ld_h0 = 0; const u32x4* ld_w = nullptr;
    auto load_begin = [&](int strip, int chunk) __attribute__((always_inline)) {
        const int b = strip / (IMG / R);
        ld_h0 = (strip % (IMG / R)) * R;
        ld_xb = reinterpret_cast<const char*>(p.x + ((long)b * KIN + chunk * CH) * SRC_PLANE);
        ld_mb = MODE == B3_DGRAD ? p.xmask + ((long)b * KIN + chunk * CH) * SRC_PLANE : nullptr;
        ld_w = reinterpret_cast<const u32x4*>(p.wp + (size_t)(mb * NCHUNK + chunk) * WSTAGE);
        okbits = 0;
    };
    auto load_item = [&](int i) __attribute__((always_inline)) {
        int gi = 0, li = 0, ri = 0, pi = 0;
        if (RECOMPUTE) item(opaque_t(), i, gi, li, ri, pi); else { gi = goff[i]; ri = irow[i]; }
        const int yr = ld_h0 - 1 + ri;
        const int yy = min(max(yr, 0), IMG - 1);
        okbits |= (yr >= 0 && yr < IMG ? 1u : 0u) << i;
        const unsigned o = gi + (MODE == B3_FWD ? yy * IMG : (yy >> 1) * (IMG / 2));
#pragma unroll
        for (int j = 0; j < 8; ++j) xr[i][j] = *reinterpret_cast<const float*>(ld_xb + (size_t)(4u * (o + j * SRC_PLANE)));
        if (MODE == B3_DGRAD) {
            uint32_t m0 = 0, m1 = 0;
#pragma unroll
            for (int j = 0; j < 4; ++j) { m0 |= (uint32_t)ld_mb[o + j * SRC_PLANE] << (8 * j); m1 |= (uint32_t)ld_mb[o + (j + 4) * SRC_PLANE] << (8 * j); }
            mr[i][0] = m0; mr[i][1] = m1;
        }
    };
    auto load_filters = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < WIT; ++i) wr[i] = ld_w[min(t + i * 256, WPIECES - 1)];
    };
    auto load_stage = [&](int strip, int chunk) __attribute__((always_inline)) {
        load_begin(strip, chunk);
#pragma unroll
        for (int i = 0; i < NIT; ++i) load_item(i);
        load_filters();
    };
    auto store_stage = [&](int strip) __attribute__((always_inline)) {
        const int h0 = (strip % (IMG / R)) * R;
        const int ts = RECOMPUTE ? opaque_t() : t;
#pragma unroll
        for (int i = 0; i < NIT; ++i) {
            const bool ok = (okbits >> i) & 1u;
            int gi = 0, li = 0, ri = 0, pi = 0;
            if (RECOMPUTE) item(ts, i, gi, li, ri, pi); else { li = loff[i]; ri = irow[i]; pi = ipar[i]; }
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                if (MODE == B3_FWD) v[j] = ok ? xr[i][j] : 0.f;
                else {
                    const uint32_t want = (uint32_t)(((h0 - 1 + ri) & 1) * 2 + pi);
                    v[j] = (ok && ((mr[i][j >> 2] >> (8 * (j & 3))) & 0xff) == want) ? xr[i][j] : 0.f;
                }
            }
            uint32_t hi[4], mid[4], lo[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) split2(v[2 * j], v[2 * j + 1], hi[j], mid[j], lo[j]);
            uint16_t* d = Xs + li;
            if (!RAGGED || t + i * 256 < ITEMS) {
                *reinterpret_cast<u32x4*>(d) = u32x4{hi[0], hi[1], hi[2], hi[3]};
                *reinterpret_cast<u32x4*>(d + XPLANE) = u32x4{mid[0], mid[1], mid[2], mid[3]};
                *reinterpret_cast<u32x4*>(d + 2 * XPLANE) = u32x4{lo[0], lo[1], lo[2], lo[3]};
            }
        }
        u32x4* wd = reinterpret_cast<u32x4*>(Ws);
#pragma unroll
        for (int i = 0; i < WIT; ++i)
            if (t + i * 256 < WPIECES) wd[t + i * 256] = wr[i];
    };

    // this wave's two accumulator tiles: rows y0, y0 + 1 of the strip, columns cb .. cb + 31
    constexpr int WPR = IMG / 32;                            // waves per row pair
    const int y0 = (wave / WPR) * 2, cb = (wave % WPR) * 32;
    unsigned long long ph[4] = {0, 0, 0, 0}, tprev = PROBE ? __builtin_readcyclecounter() : 0;
    auto mark = [&](int k) __attribute__((always_inline)) {
        if (PROBE) { const unsigned long long now = __builtin_readcyclecounter(); ph[k] += now - tprev; tprev = now; }
    };
    int strip = first;
    __syncthreads();                                         // the zero fill (halo columns stay zero for the kernel's life)
    if (strip < nstrips) { load_stage(strip, 0); store_stage(strip); }
    mark(2);
    const uint16_t* xs = Xs + 8 * h;
    const uint16_t* ws = Ws + (h * 32 + r) * 8;
    for (; strip < nstrips; strip += stride) {
        const int b = strip / (IMG / R), h0 = (strip % (IMG / R)) * R;
        f32x16 acc[2];
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const float bv = MODE == B3_FWD ? p.bias[mb * 32 + mfma_row(q, lane)] : 0.f;
            acc[0][q] = bv; acc[1][q] = bv;
        }
        for (int chunk = 0; chunk < NCHUNK; ++chunk) {
            int nstrip = strip, nchunk = chunk + 1;
            if (nchunk == NCHUNK) { nchunk = 0; nstrip = strip + stride; }
            const bool have_next = nstrip < nstrips;
            __syncthreads();                                 // this stage's LDS image is complete
            // the next stage's global loads are in flight under this stage's MFMAs; its split and LDS writes follow the block.
            // ONE LDS image per work-group (66 KB): two work-groups share a CU, and while one splits and stores the other
            // one's waves keep the matrix pipe busy (a bf16 MFMA holds the vector issue for only 8 of its 32 cycles)
            // Round 4: the next stage's loads are issued UNCONDITIONALLY (the last stage of a work-group re-reads its own strip and never
            // stores it) and piece by piece INSIDE the MFMA block: item i in tap i, the filter stage in tap FTAP -- each tap is one
            // scheduling region (sched_barrier) whose pattern puts the loads into the MFMA gaps.  Left to itself the compiler placed all
            // of them behind the 89th of 108 MFMAs (chunk 0: their latency then lay open in front of the split) or in a block of their
            // own before the MFMAs (chunk 1: ~30 issue slots nothing overlapped).
            constexpr int FTAP = 6;
            if (EARLY_LOADS) load_begin(have_next ? nstrip : strip, have_next ? nchunk : chunk);
            else if (have_next) load_stage(nstrip, nchunk);
            mark(0);
            // fragments of tap t + 1 are fetched while the MFMAs of tap t run
            bf16x8 a[2][3], bq[2][2][3];
            auto fetch = [&](int tap, int slot) __attribute__((always_inline)) {
                const int dy = tap / 3 - 1, dx = tap % 3 - 1;
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) a[slot][pl] = *reinterpret_cast<const bf16x8*>(ws + (tap * 3 + pl) * (2 * 32 * 8));
#pragma unroll
                for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                    for (int pl = 0; pl < 3; ++pl)
                        bq[slot][nt][pl] = *reinterpret_cast<const bf16x8*>(xs + pl * XPLANE + ((y0 + nt + dy + 1) * PXW + (cb + r + dx + 1)) * CH);
            };
            fetch(0, 0);
            if (EARLY_LOADS) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const int sl = tap & 1;
                if (tap + 1 < 9) fetch(tap + 1, sl ^ 1);
                if (EARLY_LOADS && tap < NIT) load_item(tap);
                if (EARLY_LOADS && tap == FTAP) load_filters();
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) {
                    // small terms first, the leading product last
                    acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[sl][1], bq[sl][nt][1], acc[nt], 0, 0, 0);
                    acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[sl][2], bq[sl][nt][0], acc[nt], 0, 0, 0);
                    acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[sl][0], bq[sl][nt][2], acc[nt], 0, 0, 0);
                    acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[sl][1], bq[sl][nt][0], acc[nt], 0, 0, 0);
                    acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[sl][0], bq[sl][nt][1], acc[nt], 0, 0, 0);
                    acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[sl][0], bq[sl][nt][0], acc[nt], 0, 0, 0);
                }
                // issue order of this tap: one LDS read of the next tap's fragments in each of the first nine MFMA gaps, and this tap's
                // piece of the next stage's loads (VPG per gap, behind the vector adds of their offsets)
                constexpr int VPG = MODE == B3_DGRAD ? 2 : 1;
                const bool piece = EARLY_LOADS && (tap < NIT || tap == FTAP);
                if (tap + 1 < 9) {
#pragma unroll
                    for (int k = 0; k < 9; ++k) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                        if (piece) {
                            __builtin_amdgcn_sched_group_barrier(0x002, 2 * VPG, 0);
                            __builtin_amdgcn_sched_group_barrier(0x020, VPG, 0);
                        }
                    }
                    __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
                } else {
                    __builtin_amdgcn_sched_group_barrier(0x008, 12, 0);
                }
                if (EARLY_LOADS) __builtin_amdgcn_sched_barrier(0);
            }
            mark(1);
            __syncthreads();                                 // every wave is done reading this stage
            if (have_next) store_stage(nstrip);
            mark(2);
        }
        // ---- epilogue ----
        const int x = cb + r;
        if (MODE == B3_FWD) {
            // ReLU + 2x2 max-pool + arg-max mask (PyTorch scan order, first maximum wins; 4 = ReLU inactive).  The window of pixel
            // pair (x even, x + 1) x rows (y0, y0 + 1): the even lane finishes even accumulator registers, the odd lane odd ones.
            const int odd = lane & 1;
            const int ph2 = (h0 + y0) >> 1, pw = x >> 1;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int mine = 2 * k + odd;
                // each lane sends the register its partner finishes, receives the partner's copy of the one it finishes
                const float s0 = odd ? acc[0][2 * k] : acc[0][2 * k + 1], s1 = odd ? acc[1][2 * k] : acc[1][2 * k + 1];
                const float r0 = __shfl_xor(s0, 1), r1 = __shfl_xor(s1, 1);
                const float m0 = odd ? acc[0][2 * k + 1] : acc[0][2 * k], m1 = odd ? acc[1][2 * k + 1] : acc[1][2 * k];
                // window in scan order: (y0, xe), (y0, xe + 1), (y0 + 1, xe), (y0 + 1, xe + 1); this lane is xe + odd
                const float v0 = odd ? r0 : m0, v1 = odd ? m0 : r0, v2 = odd ? r1 : m1, v3 = odd ? m1 : r1;
                float best = v0; int arg = 0;
                if (v1 > best) { best = v1; arg = 1; }
                if (v2 > best) { best = v2; arg = 2; }
                if (v3 > best) { best = v3; arg = 3; }
                const bool act = best > 0.f;
                const int co = mb * 32 + mfma_row(mine, lane);
                const long o = (((long)b * NOUT + co) * (IMG / 2) + ph2) * (IMG / 2) + pw;
                p.y[o] = act ? best : 0.f;
                if (p.ymask) p.ymask[o] = act ? (uint8_t)arg : (uint8_t)4;        // null: a forward-only plan keeps no decisions
            }
        } else {
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    const int ci = mb * 32 + mfma_row(q, lane);
                    p.y[(((long)b * NOUT + ci) * IMG + (h0 + y0 + nt)) * IMG + x] = acc[nt][q];
                }
        }
        mark(3);
    }
    if (PROBE && blockIdx.x == 0 && t == 0) {
#pragma unroll
        for (int k = 0; k < 4; ++k) g_b3_phase[k] = ph[k];
    }
    if (blockIdx.x == 0 && t == 0) { g_b3_clock[0] = __builtin_readcyclecounter() - clk_begin; g_b3_clock[1] = wall_clock64() - wall_begin; }
}

// The data gradient (loads inside its MFMA block: more live registers) is held to two waves per SIMD = two work-groups per CU by the launch
// bound (232 registers; unbounded the compiler took 314 and halved the occupancy).  The forward kernel keeps NO bound: it settles at
// 187 + 32, which leaves every SIMD the 64 registers of one encoder-chain wave beside two of its own -- under the bound it grew to 254 and
// shut the chain out for the whole forward conv (the chain is the forward half's critical path).
template <int MODE, class G = GeomFlagship>
__global__ __launch_bounds__(256, (MODE == B3_DGRAD ? 2 : 1)) void conv_b3_kernel(B3Params p) { conv_b3_body<MODE, false, G>(p); }
template <int MODE>
__global__ __launch_bounds__(256) void conv_b3_probe_kernel(B3Params p) { conv_b3_body<MODE, true, GeomFlagship>(p); }

// ------------------------------------------------------------------------------------------------------------------------------
// Round 4: the FORWARD kernel software-pipelined inside a wave, ONE work-group per CU (g_bbbp_conv2_fwd_pipe: training plans beside an
// encoder chain; BBBP_C2_PIPE=0 / 1 overrides).
// conv_b3_kernel's waves spend ~40 % of their time in the MFMA block and the rest issuing loads, splitting, writing LDS, waiting at two
// barriers per stage and in the epilogue; a second work-group on the CU fills those gaps (matrix pipe ~60 % busy).  Here a stage's LDS
// image is DOUBLE-buffered (2 x 65.7 KB: one work-group per CU) and every wave carries, inside the 108 MFMAs of stage q:
//   taps 0-2  the split + LDS writes of stage q + 1's three items into the other buffer (loaded during stage q - 1),
//   tap  3    its filter stage's LDS writes,
//   taps 4-6  the global loads of stage q + 2's items, tap 7 its filter loads (same registers, free again after tap 3),
//   taps 0-7  (first chunk of a strip) the pooling epilogue of the PREVIOUS strip from a copy of its accumulators,
// one barrier per stage.  Each tap is one scheduling region; the pattern puts one LDS read, a few vector instructions and one memory
// operation behind every MFMA.  Loads and writes are unconditional: past the last stage they re-read / re-write data nobody uses.
// ------------------------------------------------------------------------------------------------------------------------------
// (MODE = B3_DGRAD, round 4, experimental -- BBBP_C2_DGRAD_PIPE=1: the data gradient in the same form: the loader expands the pooled gradient
// through the decisions, the epilogue stores two full-resolution rows of 32 pixels x 32 channels per wave.)
template <int MODE, class G, bool MASK>
__global__ __launch_bounds__(256) void conv_b3p_kernel(B3Params p) {
    constexpr int IMG = G::IMGS, R = 256 / IMG, ROWS = R + 2, PXW = IMG + 2, XPLANE = ROWS * PXW * CH, XBUF = 3 * XPLANE;
    constexpr int KIN = MODE == B3_FWD ? G::CIN : G::COUT, NCHUNK = KIN / CH, NMB = (MODE == B3_FWD ? G::COUT : G::CIN) / 32, NOUT = NMB * 32;
    constexpr int SRC_PLANE = MODE == B3_FWD ? IMG * IMG : (IMG / 2) * (IMG / 2);
    static_assert(NCHUNK % 2 == 0, "the buffer parity of a stage is its chunk's parity");
    constexpr int STAGE = XBUF + WSTAGE;                     // bf16 elements of one buffer
    static_assert(IMG == 64, "the pipelined form is written for the 64 x 64 maps");
    extern __shared__ __attribute__((aligned(16))) uint16_t smem[];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, r = lane & 31, h = lane >> 5;
    const bool pairs = NMB == 2 && (gridDim.x & 15) == 0;
    const int mb = NMB == 1 ? 0 : pairs ? (blockIdx.x >> 3) & 1 : blockIdx.x % NMB;
    const int nstrips = p.B * (IMG / R);
    const int stride = gridDim.x / NMB;
    const int first = NMB == 1 ? xcd_adjacent(blockIdx.x, gridDim.x)
                    : pairs ? ((stride & 7) ? (blockIdx.x & 7) + 8 * (blockIdx.x >> 4) : (blockIdx.x & 7) * (stride >> 3) + (blockIdx.x >> 4))
                    : blockIdx.x / NMB;
    for (int i = t * 8; i < 2 * STAGE; i += 256 * 8) *reinterpret_cast<u32x4*>(smem + i) = u32x4{0, 0, 0, 0};

    constexpr int ITEMS = 2 * ROWS * IMG, NIT = ITEMS / 256;
    static_assert(ITEMS % 256 == 0 && NIT == 3, "three items per thread and stage");
    constexpr int WPIECES = WSTAGE * 2 / 16, WIT = (WPIECES + 255) / 256;
    int goff[NIT], loff[NIT], irow[NIT];
    const int ipar = t & 1;                                  // (px & 1: the same for the three items of a thread)
#pragma unroll
    for (int i = 0; i < NIT; ++i) {
        const int idx = t + i * 256, px = idx % IMG, row = (idx / IMG) % ROWS, cg = idx / (IMG * ROWS);
        goff[i] = cg * 8 * SRC_PLANE + (MODE == B3_FWD ? px : px >> 1);
        loff[i] = (row * PXW + px + 1) * CH + cg * 8;
        irow[i] = row;
    }
    float xr[NIT][8];
    uint32_t mr[NIT][2];                                     // DGRAD: the eight decision bytes of an item
    u32x4 wr[WIT];
    uint32_t okbits = 0, oknext = 0;
    const char* ld_xb = nullptr; const uint8_t* ld_mb = nullptr; int ld_h0 = 0, st_h0 = 0; const u32x4* ld_w = nullptr;
    auto load_begin = [&](int strip, int chunk) __attribute__((always_inline)) {
        const int b = strip / (IMG / R);
        ld_h0 = (strip % (IMG / R)) * R;
        ld_xb = reinterpret_cast<const char*>(p.x + ((long)b * KIN + chunk * CH) * SRC_PLANE);
        ld_mb = MODE == B3_DGRAD ? p.xmask + ((long)b * KIN + chunk * CH) * SRC_PLANE : nullptr;
        ld_w = reinterpret_cast<const u32x4*>(p.wp + (size_t)(mb * NCHUNK + chunk) * WSTAGE);
        oknext = 0;
    };
    auto load_item = [&](int i) __attribute__((always_inline)) {
        const int yr = ld_h0 - 1 + irow[i];
        const int yy = min(max(yr, 0), IMG - 1);
        oknext |= (yr >= 0 && yr < IMG ? 1u : 0u) << i;
        const unsigned o = goff[i] + (MODE == B3_FWD ? yy * IMG : (yy >> 1) * (IMG / 2));
#pragma unroll
        for (int j = 0; j < 8; ++j) xr[i][j] = *reinterpret_cast<const float*>(ld_xb + (size_t)(4u * (o + j * SRC_PLANE)));
        if (MODE == B3_DGRAD) {
            uint32_t m0 = 0, m1 = 0;
#pragma unroll
            for (int j = 0; j < 4; ++j) { m0 |= (uint32_t)ld_mb[o + j * SRC_PLANE] << (8 * j); m1 |= (uint32_t)ld_mb[o + (j + 4) * SRC_PLANE] << (8 * j); }
            mr[i][0] = m0; mr[i][1] = m1;
        }
    };
    auto load_filters = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < WIT; ++i) wr[i] = ld_w[min(t + i * 256, WPIECES - 1)];
    };
    auto store_item = [&](uint16_t* Xd, int i) __attribute__((always_inline)) {
        const bool ok = (okbits >> i) & 1u;
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (MODE == B3_FWD) v[j] = ok ? xr[i][j] : 0.f;
            else {
                const uint32_t want = (uint32_t)(((st_h0 - 1 + irow[i]) & 1) * 2 + ipar);
                v[j] = (ok && ((mr[i][j >> 2] >> (8 * (j & 3))) & 0xff) == want) ? xr[i][j] : 0.f;
            }
        }
        uint32_t hi[4], mid[4], lo[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) split2(v[2 * j], v[2 * j + 1], hi[j], mid[j], lo[j]);
        uint16_t* d = Xd + loff[i];
        *reinterpret_cast<u32x4*>(d) = u32x4{hi[0], hi[1], hi[2], hi[3]};
        *reinterpret_cast<u32x4*>(d + XPLANE) = u32x4{mid[0], mid[1], mid[2], mid[3]};
        *reinterpret_cast<u32x4*>(d + 2 * XPLANE) = u32x4{lo[0], lo[1], lo[2], lo[3]};
    };
    auto store_filters = [&](uint16_t* Wd) __attribute__((always_inline)) {
        u32x4* wd = reinterpret_cast<u32x4*>(Wd);
#pragma unroll
        for (int i = 0; i < WIT; ++i) wd[min(t + i * 256, WPIECES - 1)] = wr[i];      // (clamped like the load: the tail lanes rewrite the last piece with itself)
    };
    // stage counter of this work-group: stage q = (strip first + (q / NCHUNK) * stride, chunk q % NCHUNK); clamped to the last one
    const int nmine = first < nstrips ? (nstrips - first + stride - 1) / stride : 0;
    const int nq = nmine * NCHUNK;
    auto strip_of = [&](int q) { return first + (min(q, nq - 1) / NCHUNK) * stride; };
    auto chunk_of = [&](int q) { return min(q, nq - 1) % NCHUNK; };
    if (nq == 0) return;

    constexpr int WPR = IMG / 32;
    const int y0 = (wave / WPR) * 2, cb = (wave % WPR) * 32;
    const int odd = lane & 1, x = cb + r;
    __syncthreads();                                         // the zero fill of both buffers
    // prologue: stage 0 into buffer 0, stage 1's loads in flight
    load_begin(strip_of(0), chunk_of(0));
#pragma unroll
    for (int i = 0; i < NIT; ++i) load_item(i);
    load_filters();
    okbits = oknext; st_h0 = ld_h0;
#pragma unroll
    for (int i = 0; i < NIT; ++i) store_item(smem, i);
    store_filters(smem + XBUF);
    load_begin(strip_of(1), chunk_of(1));
#pragma unroll
    for (int i = 0; i < NIT; ++i) load_item(i);
    load_filters();

    f32x16 acc[2], eacc[2];
#pragma unroll
    for (int q = 0; q < 16; ++q) { eacc[0][q] = 0.f; eacc[1][q] = 0.f; }
    int eb = strip_of(0) / (IMG / R), eh0 = (strip_of(0) % (IMG / R)) * R;          // the strip eacc belongs to (first pass: the first strip, rewritten later)
    float bv[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) bv[q] = MODE == B3_FWD ? p.bias[mb * 32 + mfma_row(q, lane)] : 0.f;
    // one output of the pooling epilogue (k = 0 .. 7) of the strip in eacc
    auto epilogue_piece = [&](int k) __attribute__((always_inline)) {
        if (MODE == B3_DGRAD) {
            // two accumulator registers of both row tiles: full-resolution rows eh0 + y0 (+ 1), channel mfma_row(q), pixel x (128-byte rows)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const int q = 2 * k + e;
                    const int ci = mb * 32 + mfma_row(q, lane);
                    p.y[(((long)eb * NOUT + ci) * IMG + (eh0 + y0 + nt)) * IMG + x] = eacc[nt][q];
                }
            return;
        }
        const int mine = 2 * k + odd;
        // (opaque copies: the compiler otherwise rewrites "odd ? v[2k] : v[2k + 1]" into v[(2k + 1) ^ odd], a per-lane register-array index
        // that it lowers to a chain of 15 compares and selects per access -- 780 of this kernel's first version's 1120 vector instructions)
        float e00 = eacc[0][2 * k], e01 = eacc[0][2 * k + 1], e10 = eacc[1][2 * k], e11 = eacc[1][2 * k + 1];
        asm volatile("" : "+v"(e00), "+v"(e01), "+v"(e10), "+v"(e11));
        const float s0 = odd ? e00 : e01, s1 = odd ? e10 : e11;
        const float r0 = __shfl_xor(s0, 1), r1 = __shfl_xor(s1, 1);
        const float m0 = odd ? e01 : e00, m1 = odd ? e11 : e10;
        const float v0 = odd ? r0 : m0, v1 = odd ? m0 : r0, v2 = odd ? r1 : m1, v3 = odd ? m1 : r1;
        float best = v0; int arg = 0;
        if (v1 > best) { best = v1; arg = 1; }
        if (v2 > best) { best = v2; arg = 2; }
        if (v3 > best) { best = v3; arg = 3; }
        const bool act = best > 0.f;
        const int co = mb * 32 + mfma_row(mine, lane);
        const long o = (((long)eb * NOUT + co) * (IMG / 2) + ((eh0 + y0) >> 1)) * (IMG / 2) + (x >> 1);
        p.y[o] = act ? best : 0.f;
        if (MASK) p.ymask[o] = act ? (uint8_t)arg : (uint8_t)4;          // (a template parameter: a branch here would cut the tap's scheduling region)
    };

    for (int q0 = 0; q0 < nq; q0 += NCHUNK) {
#pragma unroll
        for (int chunk = 0; chunk < NCHUNK; ++chunk) {
            const int q = q0 + chunk;
            uint16_t* cur = smem + ((q & 1) ? STAGE : 0);
            uint16_t* nxt = smem + ((q & 1) ? 0 : STAGE);
            const uint16_t* xs = cur + 8 * h;
            const uint16_t* ws = cur + XBUF + (h * 32 + r) * 8;
            __syncthreads();                                 // buffer `cur` is complete; everybody is done reading `nxt` (stage q - 1)
            if (chunk == 0) {
#pragma unroll
                for (int i = 0; i < 16; ++i) { acc[0][i] = bv[i]; acc[1][i] = bv[i]; }
            }
            okbits = oknext; st_h0 = ld_h0;                   // validity / strip row of the rows held in xr (stage q + 1)
            bf16x8 a[2][3], bq[2][2][3];
            auto fetch = [&](int tap, int slot) __attribute__((always_inline)) {
                const int dy = tap / 3 - 1, dx = tap % 3 - 1;
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) a[slot][pl] = *reinterpret_cast<const bf16x8*>(ws + (tap * 3 + pl) * (2 * 32 * 8));
#if defined(B3P_ABLATE_DX)
                // ABLATION (wrong results): the dx = +-1 taps reuse the dx = 0 fragments -- what the kernel would cost without 36 of its 81 LDS reads
                if (dx != 0) {
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                        for (int pl = 0; pl < 3; ++pl) bq[slot][nt][pl] = bq[slot ^ 1][nt][pl];
                    return;
                }
#endif
#pragma unroll
                for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                    for (int pl = 0; pl < 3; ++pl)
                        bq[slot][nt][pl] = *reinterpret_cast<const bf16x8*>(xs + pl * XPLANE + ((y0 + nt + dy + 1) * PXW + (cb + r + dx + 1)) * CH);
            };
            fetch(0, 0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const int sl = tap & 1;
                if (tap + 1 < 9) fetch(tap + 1, sl ^ 1);
                if (tap < NIT) store_item(nxt, tap);                              // stage q + 1 -> the other buffer
                if (tap == NIT) { store_filters(nxt + XBUF); load_begin(strip_of(q + 2), chunk_of(q + 2)); }
                if (tap > NIT && tap <= 2 * NIT) load_item(tap - NIT - 1);        // stage q + 2 -> the registers stage q + 1 just left
                if (tap == 2 * NIT + 1) load_filters();
                if (chunk == 0 && tap < 8) epilogue_piece(tap);                   // the previous strip's pooled outputs
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) {
                    acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[sl][1], bq[sl][nt][1], acc[nt], 0, 0, 0);
                    acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[sl][2], bq[sl][nt][0], acc[nt], 0, 0, 0);
                    acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[sl][0], bq[sl][nt][2], acc[nt], 0, 0, 0);
                    acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[sl][1], bq[sl][nt][0], acc[nt], 0, 0, 0);
                    acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[sl][0], bq[sl][nt][1], acc[nt], 0, 0, 0);
                    acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[sl][0], bq[sl][nt][0], acc[nt], 0, 0, 0);
                }
                // per MFMA gap: one LDS read of the next tap's fragments, up to six vector instructions, one memory operation
#pragma unroll
                for (int k = 0; k < 12; ++k) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    if (tap + 1 < 9 && k < 9) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x002, 6, 0);
                    __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            if (chunk == NCHUNK - 1) {
                // hand the strip's accumulators to the epilogue that runs inside the next strip's first stage
#pragma unroll
                for (int i = 0; i < 16; ++i) { eacc[0][i] = acc[0][i]; eacc[1][i] = acc[1][i]; }
                const int sp = strip_of(q);
                eb = sp / (IMG / R); eh0 = (sp % (IMG / R)) * R;
            }
        }
    }
    // the last strip's epilogue
#pragma unroll
    for (int k = 0; k < 8; ++k) epilogue_piece(k);
}

// ------------------------------------------------------------------------------------------------------------------------------
// Weight gradient of the same stage: dW[co][ci][tap] = sum_{b, y, x} dY[b][co][y][x] X[b][ci][y + dy][x + dx] with dY the pooled
// gradient expanded through the arg-max mask; db[co] = sum dY.  GEMM view: M = co (2 tiles of 32), N = (tap, ci) (9 tiles of 32),
// K = pixels, 16 per MFMA.  Both operands sit in LDS pixel-INNERMOST, split in three bf16 planes: a lane's A fragment is 8
// consecutive pixels of its output channel (one aligned ds_read_b128); its B fragment is 8 consecutive pixels of its input channel
// shifted by the tap: the row shift is an address, the column shift dx = -1 / +1 is 2 bytes, so the three dx fragments of a row come
// from ONE aligned 16-byte read plus the two neighbouring words, assembled with five v_alignbit_b32.
// A work-group = 4 waves = (m-tile, k-group) pairs; a stage is one pooled row = 2 full-resolution rows x 64 pixels of one image
// (8 k-steps, 4 per k-group); a wave keeps the 9 tap tiles of its m-tile (144 accumulator registers) for the whole kernel.  The two
// k-groups are summed through LDS at the end, each work-group writes one partial slab [64][288] (+ 64 bias partials) and
// conv.hip's fixed-order slab reduce finishes (bit-reproducible, no float atomics) -- the same slab contract as conv_wgrad32_kernel.
constexpr int DYCO = 2 * IMG + 8;              // bf16 per output channel of a stage: 2 rows x 64 px (+ 16 B: bank spread)
constexpr int DYPLANE = 64 * DYCO;
constexpr int XROW = IMG + 16;                 // 8 px of zero halo on both sides: pixel x sits at index x + 8 (16-byte aligned groups)
constexpr int XCI = 4 * XROW + 8;              // 4 rows per input channel (+ 16 B)
constexpr int WXPLANE = 32 * XCI;
constexpr size_t WG_LDS_BYTES = (size_t)(3 * DYPLANE + 3 * WXPLANE) * 2;

struct B3WgradParams {
    const float* x;         // [B][32][64][64]
    const float* gy;        // pooled gradient [B][64][32][32]
    const uint8_t* mask;    // [B][64][32][32]
    float* slab;            // [grid][64][288]
    float* bslab;           // [grid][64]
    int B;
    int prio;
    // round 4: the kernel handles ONE (32-input-channel, 64-output-channel) block pair of a stage with cin_total / cout_total channels on
    // 64 x 64 maps (the wide / deep variant's 64 -> 128 stage: four pairs); blockIdx.x = pair * groups + g, like conv_wgrad32_kernel
    int cin_total, cout_total, groups;
};

__global__ __launch_bounds__(256) void conv_b3_wgrad_kernel(B3WgradParams p) {
    extern __shared__ __attribute__((aligned(16))) uint16_t smem[];
    uint16_t* DYs = smem;                       // [plane][co][row 2][px]
    uint16_t* XsW = smem + 3 * DYPLANE;         // [plane][ci][row 4][XROW]
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, r = lane & 31, h = lane >> 5;
    const int mt = wave & 1, kg = wave >> 1;
    const int nstrips = p.B * (IMG / 2);
    const int pair = blockIdx.x / p.groups, grp = blockIdx.x % p.groups, ncib = p.cin_total / 32;
    // this pair's channel blocks as uniform base pointers + per-image strides (scalar registers: the loaders' vector arithmetic is unchanged)
    const float* const gyb = p.gy + (long)(pair / ncib) * 64 * (IMG / 2) * (IMG / 2);
    const uint8_t* const mkb = p.mask + (long)(pair / ncib) * 64 * (IMG / 2) * (IMG / 2);
    const float* const xbk = p.x + (long)(pair % ncib) * 32 * IMG * IMG;
    const long gy_img = (long)p.cout_total * (IMG / 2) * (IMG / 2), x_img = (long)p.cin_total * IMG * IMG;
    for (int i = t * 8; i < 3 * DYPLANE + 3 * WXPLANE; i += 256 * 8) *reinterpret_cast<u32x4*>(smem + i) = u32x4{0, 0, 0, 0};

    // ---- loader: dY items (co, quad of 4 pooled px) x 2 per thread; X items (ci, row, 8 px) x 4 per thread ----
    f32x4 gq[2]; uint32_t mq[2]; f32x4 xq[4][2];
    float bsum[2] = {0.f, 0.f};
    uint32_t okx = 0;
    auto load_stage = [&](int strip) __attribute__((always_inline)) {
        const int b = strip / (IMG / 2), ph = strip % (IMG / 2);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int idx = t + i * 256, q = idx & 7, co = idx >> 3;
            const long off = (long)b * gy_img + (co * (IMG / 2) + ph) * (IMG / 2) + q * 4;
            gq[i] = *reinterpret_cast<const f32x4*>(gyb + off);
            mq[i] = *reinterpret_cast<const uint32_t*>(mkb + off);
        }
        okx = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int idx = t + i * 256, q = idx & 7, row = (idx >> 3) & 3, ci = idx >> 5;
            const int yr = 2 * ph - 1 + row;
            okx |= (yr >= 0 && yr < IMG ? 1u : 0u) << i;
            const float* src = xbk + (long)b * x_img + (ci * IMG + min(max(yr, 0), IMG - 1)) * IMG + q * 8;
            xq[i][0] = *reinterpret_cast<const f32x4*>(src);
            xq[i][1] = *reinterpret_cast<const f32x4*>(src + 4);
        }
    };
    auto store_stage = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int idx = t + i * 256, q = idx & 7, co = idx >> 3;
            float v0[8], v1[8];                 // full-resolution rows 2 ph and 2 ph + 1, pixels 8 q .. 8 q + 7
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const uint32_t m = (mq[i] >> (8 * e)) & 0xff;
                const float g = gq[i][e];
                bsum[i] += m < 4 ? g : 0.f;
                v0[2 * e] = m == 0 ? g : 0.f; v0[2 * e + 1] = m == 1 ? g : 0.f;
                v1[2 * e] = m == 2 ? g : 0.f; v1[2 * e + 1] = m == 3 ? g : 0.f;
            }
            uint32_t hi[4], mid[4], lo[4];
            uint16_t* d = DYs + co * DYCO + q * 8;
#pragma unroll
            for (int j = 0; j < 4; ++j) split2(v0[2 * j], v0[2 * j + 1], hi[j], mid[j], lo[j]);
            *reinterpret_cast<u32x4*>(d) = u32x4{hi[0], hi[1], hi[2], hi[3]};
            *reinterpret_cast<u32x4*>(d + DYPLANE) = u32x4{mid[0], mid[1], mid[2], mid[3]};
            *reinterpret_cast<u32x4*>(d + 2 * DYPLANE) = u32x4{lo[0], lo[1], lo[2], lo[3]};
#pragma unroll
            for (int j = 0; j < 4; ++j) split2(v1[2 * j], v1[2 * j + 1], hi[j], mid[j], lo[j]);
            *reinterpret_cast<u32x4*>(d + IMG) = u32x4{hi[0], hi[1], hi[2], hi[3]};
            *reinterpret_cast<u32x4*>(d + IMG + DYPLANE) = u32x4{mid[0], mid[1], mid[2], mid[3]};
            *reinterpret_cast<u32x4*>(d + IMG + 2 * DYPLANE) = u32x4{lo[0], lo[1], lo[2], lo[3]};
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int idx = t + i * 256, q = idx & 7, row = (idx >> 3) & 3, ci = idx >> 5;
            const bool ok = (okx >> i) & 1u;
            uint32_t hi[4], mid[4], lo[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float a = j < 2 ? xq[i][0][2 * j] : xq[i][1][2 * j - 4], c = j < 2 ? xq[i][0][2 * j + 1] : xq[i][1][2 * j - 3];
                split2(ok ? a : 0.f, ok ? c : 0.f, hi[j], mid[j], lo[j]);
            }
            uint16_t* d = XsW + ci * XCI + row * XROW + 8 + q * 8;
            *reinterpret_cast<u32x4*>(d) = u32x4{hi[0], hi[1], hi[2], hi[3]};
            *reinterpret_cast<u32x4*>(d + WXPLANE) = u32x4{mid[0], mid[1], mid[2], mid[3]};
            *reinterpret_cast<u32x4*>(d + 2 * WXPLANE) = u32x4{lo[0], lo[1], lo[2], lo[3]};
        }
    };

    f32x16 acc[9];
#pragma unroll
    for (int i = 0; i < 9; ++i)
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[i][q] = 0.f;

    int strip = xcd_adjacent(grp, p.groups);
    __syncthreads();
    if (strip < nstrips) { load_stage(strip); store_stage(); }
    const uint16_t* abase = DYs + (mt * 32 + r) * DYCO + 8 * h;
    const uint16_t* bbase = XsW + r * XCI + 8 + 8 * h;
    for (; strip < nstrips; strip += p.groups) {
        const int nstrip = strip + p.groups;
        __syncthreads();                                     // this stage's LDS image is complete
        if (nstrip < nstrips) load_stage(nstrip);
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) {                     // this k-group's four k-steps: row rr, pixels x0 .. x0 + 15
            const int ks = kg * 4 + s4, rr = ks >> 2, x0 = (ks & 3) * 16;
            bf16x8 a[3];
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) a[pl] = *reinterpret_cast<const bf16x8*>(abase + pl * DYPLANE + rr * IMG + x0);
#pragma unroll
            for (int dyi = 0; dyi < 3; ++dyi) {
                bf16x8 bf[3][3];                             // [dx][plane]
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) {
                    const uint16_t* src = bbase + pl * WXPLANE + (rr + dyi) * XROW + x0;
                    const u32x4 w = *reinterpret_cast<const u32x4*>(src);
                    const uint32_t wm = *reinterpret_cast<const uint32_t*>(src - 2), wp = *reinterpret_cast<const uint32_t*>(src + 8);
                    const uint32_t s01 = __builtin_amdgcn_alignbit(w[1], w[0], 16), s12 = __builtin_amdgcn_alignbit(w[2], w[1], 16),
                                   s23 = __builtin_amdgcn_alignbit(w[3], w[2], 16);
                    const u32x4 left = {__builtin_amdgcn_alignbit(w[0], wm, 16), s01, s12, s23};            // pixels x - 1 ..
                    const u32x4 right = {s01, s12, s23, __builtin_amdgcn_alignbit(wp, w[3], 16)};           // pixels x + 1 ..
                    bf[0][pl] = __builtin_bit_cast(bf16x8, left);
                    bf[1][pl] = __builtin_bit_cast(bf16x8, w);
                    bf[2][pl] = __builtin_bit_cast(bf16x8, right);
                }
#pragma unroll
                for (int dxi = 0; dxi < 3; ++dxi) {
                    const int tap = dyi * 3 + dxi;
                    acc[tap] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], bf[dxi][1], acc[tap], 0, 0, 0);
                    acc[tap] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], bf[dxi][0], acc[tap], 0, 0, 0);
                    acc[tap] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], bf[dxi][2], acc[tap], 0, 0, 0);
                    acc[tap] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], bf[dxi][0], acc[tap], 0, 0, 0);
                    acc[tap] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], bf[dxi][1], acc[tap], 0, 0, 0);
                    acc[tap] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], bf[dxi][0], acc[tap], 0, 0, 0);
                }
            }
        }
        __syncthreads();                                     // every wave is done reading this stage
        if (nstrip < nstrips) store_stage();
    }
    __syncthreads();
    // ---- the two k-groups through LDS (the staging buffers are free now), tap by tap; k-group 0 adds and keeps ----
    float* red = reinterpret_cast<float*>(smem);             // [mt 2][16][64] floats per tap pass
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
        if (kg == 1) {
#pragma unroll
            for (int q = 0; q < 16; ++q) red[(mt * 16 + q) * 64 + lane] = acc[tap][q];
        }
        __syncthreads();
        if (kg == 0) {
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[tap][q] += red[(mt * 16 + q) * 64 + lane];
        }
        __syncthreads();
    }
    if (kg == 0) {
        float* sdst = p.slab + (long)blockIdx.x * 64 * 288;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap)
#pragma unroll
            for (int q = 0; q < 16; ++q) sdst[(mt * 32 + mfma_row(q, lane)) * 288 + tap * 32 + r] = acc[tap][q];
    }
    // bias-gradient partials: the 8 threads of one channel (consecutive lanes) reduce by shuffles
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        float v = bsum[i];
        v += __shfl_xor(v, 1); v += __shfl_xor(v, 2); v += __shfl_xor(v, 4);
        const int idx = t + i * 256;
        if ((idx & 7) == 0) p.bslab[(long)blockIdx.x * 64 + (idx >> 3)] = v;
    }
}

// ------------------------------------------------------------------------------------------------------------------------------
// The same weight gradient on the 2:4 STRUCTURED-SPARSE matrix instruction (round 3; conv-form bit 7, part of the default mask 252; without
// it the dense kernel above runs).  The A operand -- the pooled gradient expanded through the arg-max mask -- is sparse by construction: a 2 x 2 pooling window
// passes its gradient to ONE of its four pixels, so along a full-resolution row every pair of adjacent pixels holds at most one nonzero,
// i.e. at most two in any aligned group of four: exactly the pattern v_smfmac_f32_32x32x32_bf16 wants.  Its compressed A operand (the
// two kept values of every group + a 2-bit position each) IS the pooled gradient: value = g[pooled px] if the window's maximum sits in
// this row, else 0; position = (pooled px & 1) * 2 + (mask & 1).  Nothing is expanded: the operand is a quarter of the dense one in LDS
// and in LDS reads, and one instruction contracts 32 pixels for the cost of 16 -- half the matrix-pipe time for the same products (the
// six split-bf16 piece products per float32 product are kept: float32 accuracy, exact zeros skipped).
// Operand layout of the instruction, measured (tools/micro/smfmac_probe.hip; profiles/r03_smfmac_probe.txt): lane (row i = l & 31,
// kb = l >> 5) of A holds 8 compressed elements; elements (0,1) and (2,3) pair with B rows (pixels) 8 kb + 0..3 and 8 kb + 4..7 as held by
// the B lanes of wave half 0, elements (4,5) / (6,7) with the same positions of wave half 1; a B lane (column j, half h) holds 16
// consecutive k; the 2-bit positions sit at bits [2 e + 1 : 2 e] of the index register's low half (ABID = 0).  With B half h holding
// pixels x0 + 16 h .. + 15, A lane (co, kb) therefore holds pooled pixels P0 + 4 kb .. + 3 and P0 + 8 + 4 kb .. + 3 (P0 = x0 / 2): two
// quads of four, stored adjacent in LDS (one ds_read_b128 per plane).
// Work-group: NW waves = (m-tile 2) x (k-group NW / 2); a stage (one pooled row, as before) is four k-blocks of 32 pixels (upper / lower
// row x left / right half).  NW = 8: one k-block per wave, two waves per SIMD -- fastest alone (0.236 ms at B = 512 against 0.405 dense), but
// its 2 x 256 registers per SIMD lane leave nothing for the fingerprint branch's kernels beside it (whole step 2.59 -> 2.67 ms).  NW = 4
// (what the engine asks for while an encoder chain runs beside the image branch, common.h: g_bbbp_conv_wgrad_beside_encoder;
// BBBP_C2_WGRAD_SPARSE_WAVES overrides): two k-blocks per wave, one wave of 297 registers per SIMD: 0.26 ms alone, 0.33 in the step
// (dense: 0.39 / 0.54), whole step 2.59 -> 2.52 ms; the two-branch model without an encoder takes the 8-wave form (1.126 -> 1.046 ms).
constexpr int ASP_CO = 32 + 8;                  // pooled pixels of one output channel and row (+ 16 B: bank spread), quads in slot order
constexpr int ASP_PLANE = 64 * ASP_CO;
constexpr int ASP_RR = 3 * ASP_PLANE;
constexpr size_t WGS_LDS_BYTES = (size_t)(3 * WXPLANE + 2 * ASP_RR) * 2 + 64 * 8;
typedef __bf16 bf16x16 __attribute__((ext_vector_type(16)));
typedef uint32_t u32x8 __attribute__((ext_vector_type(8)));

// IW = 64: a stage is one pooled row (k-blocks = window row x left / right half).  IW = 32 (round 4: the wide / deep variant's 128 -> 256 stage): a
// stage is TWO pooled rows of 16 -- the same 32 pooled pixels per channel, contiguous in memory -- and six input rows of 32 pixels; k-blocks =
// window row x pooled row, each the full width.  Everything downstream of the LDS image (fragments, positions, MFMAs) is the same code.
template <int NW, int IW = 64>
__global__ __launch_bounds__(64 * NW) void conv_b3_wgrad_sp_kernel(B3WgradParams p) {
    static_assert(IW == 64 || IW == 32, "64 x 64 or 32 x 32 maps");
    constexpr int XR = IW == 64 ? 4 : 6;                    // staged input rows
    constexpr int XROWW = IW + 16, XCIW = XR * XROWW + 8;   // (IW = 64: XROW, XCI)
    constexpr int QPR = IW / 8;                             // 8-pixel groups per input row
    constexpr int NXI = 32 * XR * QPR;                      // X items per stage (1024 / 768)
    static_assert(XCIW <= XCI, "LDS image of the 32-pixel form fits the 64-pixel allocation");
    if (p.prio >= 3) __builtin_amdgcn_s_setprio(3); else if (p.prio == 2) __builtin_amdgcn_s_setprio(2); else if (p.prio == 1) __builtin_amdgcn_s_setprio(1);
    constexpr int NTH = 64 * NW, NA = 512 / NTH, NX = (NXI + NTH - 1) / NTH, NKB = 8 / NW;          // items per thread, k-blocks per wave
    extern __shared__ __attribute__((aligned(16))) uint16_t smem[];
    uint16_t* XsW = smem;                        // [plane][ci][row 4][XROW]
    uint16_t* Asp = smem + 3 * WXPLANE;          // [row of the window 2][plane][co][slot 8][4 pooled px]
    uint8_t* Aidx = reinterpret_cast<uint8_t*>(smem + 3 * WXPLANE + 2 * ASP_RR);      // [co][slot 8]: four 2-bit positions per quad
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, r = lane & 31, h = lane >> 5;
    const int mt = wave & 1, kq = wave >> 1;                 // k-group: NW = 8: one k-block (row kq >> 1, half kq & 1); NW = 4: row kq, both halves
    const int rr = NW == 8 ? kq >> 1 : kq;
    const int nstrips = p.B * (IW == 64 ? 32 : 8);
    const int pair = blockIdx.x / p.groups, grp = blockIdx.x % p.groups, ncib = p.cin_total / 32;
    // this pair's channel blocks as uniform base pointers + per-image strides (scalar registers: the loaders' vector arithmetic is unchanged)
    const float* const gyb = p.gy + (long)(pair / ncib) * 64 * (IW / 2) * (IW / 2);
    const uint8_t* const mkb = p.mask + (long)(pair / ncib) * 64 * (IW / 2) * (IW / 2);
    const float* const xbk = p.x + (long)(pair % ncib) * 32 * IW * IW;
    const long gy_img = (long)p.cout_total * (IW / 2) * (IW / 2), x_img = (long)p.cin_total * IW * IW;
    for (int i = t * 8; i < 3 * WXPLANE + 2 * ASP_RR + 64 * 4; i += NTH * 8) *reinterpret_cast<u32x4*>(smem + i) = u32x4{0, 0, 0, 0};

    // ---- loader: NA dY items (co, quad of 4 pooled px) and NX X items (ci, row, 8 px) per thread ----
    f32x4 gq[NA]; uint32_t mq[NA]; f32x4 xq[NX][2];
    float bsum[NA];
#pragma unroll
    for (int i = 0; i < NA; ++i) bsum[i] = 0.f;
    uint32_t okx = 0;
    auto load_stage = [&](int strip) __attribute__((always_inline)) {
        constexpr int SPI = IW == 64 ? 32 : 8;             // stages per image; a stage's 32 pooled pixels are contiguous in both forms
        const int b = strip / SPI, ph = strip % SPI;
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int idx = t + i * NTH, a_q = idx & 7, a_co = idx >> 3;
            const long off = (long)b * gy_img + a_co * (IW / 2) * (IW / 2) + ph * 32 + a_q * 4;
            gq[i] = *reinterpret_cast<const f32x4*>(gyb + off);
            mq[i] = *reinterpret_cast<const uint32_t*>(mkb + off);
        }
        okx = 0;
#pragma unroll
        for (int i = 0; i < NX; ++i) {
            const int idx = min(t + i * NTH, NXI - 1), q = idx % QPR, row = (idx / QPR) % XR, ci = idx / (QPR * XR);
            const int yr = (IW == 64 ? 2 : 4) * ph - 1 + row;
            okx |= (yr >= 0 && yr < IW ? 1u : 0u) << i;
            const float* src = xbk + (long)b * x_img + (ci * IW + min(max(yr, 0), IW - 1)) * IW + q * 8;
            xq[i][0] = *reinterpret_cast<const f32x4*>(src);
            xq[i][1] = *reinterpret_cast<const f32x4*>(src + 4);
        }
    };
    auto store_stage = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int idx = t + i * NTH, a_q = idx & 7, a_co = idx >> 3;
            const int a_slot = (a_q >> 2) * 4 + (a_q & 1) * 2 + ((a_q >> 1) & 1);       // quads q and q + 2 of a 16-pixel half end up adjacent
            uint32_t pc[2][3];
            split2(gq[i][0], gq[i][1], pc[0][0], pc[0][1], pc[0][2]);
            split2(gq[i][2], gq[i][3], pc[1][0], pc[1][1], pc[1][2]);
            uint32_t m[4], pos = 0;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                m[e] = (mq[i] >> (8 * e)) & 0xffu;
                bsum[i] += m[e] < 4u ? gq[i][e] : 0.f;
                pos |= (((uint32_t)(e & 1) << 1) | (m[e] & 1u)) << (2 * e);        // even element: 0 / 1, odd element: 2 / 3 of its group of four
            }
            Aidx[a_co * 8 + a_slot] = (uint8_t)pos;
#pragma unroll
            for (int row = 0; row < 2; ++row) {
                // keep a pooled pixel's pieces only in the row its maximum came from (mask 4 = ReLU inactive: in neither)
                const uint32_t k01 = ((m[0] >> 1) == (uint32_t)row ? 0xffffu : 0u) | ((m[1] >> 1) == (uint32_t)row ? 0xffff0000u : 0u);
                const uint32_t k23 = ((m[2] >> 1) == (uint32_t)row ? 0xffffu : 0u) | ((m[3] >> 1) == (uint32_t)row ? 0xffff0000u : 0u);
                uint16_t* d = Asp + row * ASP_RR + a_co * ASP_CO + a_slot * 4;
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) *reinterpret_cast<u32x2*>(d + pl * ASP_PLANE) = u32x2{pc[0][pl] & k01, pc[1][pl] & k23};
            }
        }
#pragma unroll
        for (int i = 0; i < NX; ++i) {
            if (NXI % NTH != 0 && t + i * NTH >= NXI) continue;
            const int idx = t + i * NTH, q = idx % QPR, row = (idx / QPR) % XR, ci = idx / (QPR * XR);
            const bool ok = (okx >> i) & 1u;
            uint32_t hi[4], mid[4], lo[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float a = j < 2 ? xq[i][0][2 * j] : xq[i][1][2 * j - 4], c = j < 2 ? xq[i][0][2 * j + 1] : xq[i][1][2 * j - 3];
                split2(ok ? a : 0.f, ok ? c : 0.f, hi[j], mid[j], lo[j]);
            }
            uint16_t* d = XsW + ci * XCIW + row * XROWW + 8 + q * 8;
            *reinterpret_cast<u32x4*>(d) = u32x4{hi[0], hi[1], hi[2], hi[3]};
            *reinterpret_cast<u32x4*>(d + WXPLANE) = u32x4{mid[0], mid[1], mid[2], mid[3]};
            *reinterpret_cast<u32x4*>(d + 2 * WXPLANE) = u32x4{lo[0], lo[1], lo[2], lo[3]};
        }
    };

    f32x16 acc[9];
#pragma unroll
    for (int i = 0; i < 9; ++i)
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[i][q] = 0.f;

    int strip = xcd_adjacent(grp, p.groups);
    __syncthreads();
    if (strip < nstrips) { load_stage(strip); store_stage(); }
    const int co = mt * 32 + r;
    for (; strip < nstrips; strip += p.groups) {
        const int nstrip = strip + p.groups;
        __syncthreads();                                     // this stage's LDS image is complete
        if (nstrip < nstrips) load_stage(nstrip);
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb) {
        const int xb = NW == 8 ? (kq & 1) : kb;
        const uint16_t* abase = Asp + rr * ASP_RR + co * ASP_CO + (4 * xb + 2 * h) * 4;
        // IW = 64: k-block xb = the left / right 32 pixels of rows rr ..; IW = 32: xb = the pooled row, all 32 pixels of rows 2 xb + rr ..
        const uint16_t* bbase = XsW + r * XCIW + 8 + (IW == 64 ? 32 * xb : 2 * xb * XROWW) + 16 * h;
        bf16x8 a[3];
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) a[pl] = *reinterpret_cast<const bf16x8*>(abase + pl * ASP_PLANE);
        const int idx = *reinterpret_cast<const uint16_t*>(Aidx + co * 8 + 4 * xb + 2 * h);
#pragma unroll
        for (int dyi = 0; dyi < 3; ++dyi) {
            // 16 pixels of input channel r in row rr + dyi, three bf16 planes: words w[0..7] + the two neighbouring words
            u32x8 w[3]; uint32_t wm[3], wp[3];
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) {
                const uint16_t* src = bbase + pl * WXPLANE + (rr + dyi) * XROWW;
                const u32x4 w0 = *reinterpret_cast<const u32x4*>(src), w1 = *reinterpret_cast<const u32x4*>(src + 8);
                w[pl] = u32x8{w0[0], w0[1], w0[2], w0[3], w1[0], w1[1], w1[2], w1[3]};
                wm[pl] = *reinterpret_cast<const uint32_t*>(src - 2);
                wp[pl] = *reinterpret_cast<const uint32_t*>(src + 16);
            }
#pragma unroll
            for (int dxi = 0; dxi < 3; ++dxi) {
                bf16x16 bf[3];
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) {
                    u32x8 f;
                    if (dxi == 1) f = w[pl];
                    else {
#pragma unroll
                        for (int j = 0; j < 8; ++j) {
                            // dx = -1: pixels x - 1 ..: (previous word's high half, this word's low half); dx = +1: (this high, next low)
                            const uint32_t lo_w = dxi == 0 ? (j == 0 ? wm[pl] : w[pl][j - 1]) : w[pl][j];
                            const uint32_t hi_w = dxi == 0 ? w[pl][j] : (j == 7 ? wp[pl] : w[pl][j + 1]);
                            f[j] = __builtin_amdgcn_alignbit(hi_w, lo_w, 16);
                        }
                    }
                    bf[pl] = __builtin_bit_cast(bf16x16, f);
                }
                const int tap = dyi * 3 + dxi;
                acc[tap] = __builtin_amdgcn_smfmac_f32_32x32x32_bf16(a[1], bf[1], acc[tap], idx, 0, 0);
                acc[tap] = __builtin_amdgcn_smfmac_f32_32x32x32_bf16(a[2], bf[0], acc[tap], idx, 0, 0);
                acc[tap] = __builtin_amdgcn_smfmac_f32_32x32x32_bf16(a[0], bf[2], acc[tap], idx, 0, 0);
                acc[tap] = __builtin_amdgcn_smfmac_f32_32x32x32_bf16(a[1], bf[0], acc[tap], idx, 0, 0);
                acc[tap] = __builtin_amdgcn_smfmac_f32_32x32x32_bf16(a[0], bf[1], acc[tap], idx, 0, 0);
                acc[tap] = __builtin_amdgcn_smfmac_f32_32x32x32_bf16(a[0], bf[0], acc[tap], idx, 0, 0);
            }
        }
        }
        __syncthreads();                                     // every wave is done reading this stage
        if (nstrip < nstrips) store_stage();
    }
    __syncthreads();
    // ---- the k-groups through LDS (the staging buffers are free now), tap by tap, in k-group order; k-group 0 adds and keeps ----
    float* red = reinterpret_cast<float*>(smem);             // [mt 2][16][64] floats per pass
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
        for (int g = 1; g < NW / 2; ++g) {
            if (kq == g) {
#pragma unroll
                for (int q = 0; q < 16; ++q) red[(mt * 16 + q) * 64 + lane] = acc[tap][q];
            }
            __syncthreads();
            if (kq == 0) {
#pragma unroll
                for (int q = 0; q < 16; ++q) acc[tap][q] += red[(mt * 16 + q) * 64 + lane];
            }
            __syncthreads();
        }
    }
    if (kq == 0) {
        float* sdst = p.slab + (long)blockIdx.x * 64 * 288;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap)
#pragma unroll
            for (int q = 0; q < 16; ++q) sdst[(mt * 32 + mfma_row(q, lane)) * 288 + tap * 32 + r] = acc[tap][q];
    }
    // bias-gradient partials: the 8 threads of one channel (consecutive lanes) reduce by shuffles
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        float v = bsum[i];
        v += __shfl_xor(v, 1); v += __shfl_xor(v, 2); v += __shfl_xor(v, 4);
        const int idx = t + i * NTH;
        if ((idx & 7) == 0) p.bslab[(long)blockIdx.x * 64 + (idx >> 3)] = v;
    }
}

// ------------------------------------------------------------------------------------------------------------------------------
// Weight gradient of the FIRST conv stage (3 -> 32 channels on 128x128 images, R:85-87): M = co (32), N = (ci, kh, kw) = 27 of 32
// columns, K = pixels.  435 MB of compulsory traffic against 14.5 GFLOP: the kernel is HBM-bound (~0.09 ms), what the split-bf16 form
// buys here is not matrix rate but (i) a bf16 MFMA stream that leaves the vector ALU to the loader and to the other branch's kernels
// (the f32 form took 0.21 ms alone and 0.35 ms beside the encoder's backward chain) and (ii) room for two work-groups per CU.
// Column n of the B operand has its OWN (ci, kh, kw): every lane reads an aligned 16-byte group of its input row plus the two
// neighbouring words and picks the kw shift with v_alignbit_b32 (shift 0 or 16) on per-lane selected word pairs.
// A stage is one pooled row (2 full-resolution rows x 128 pixels = 16 k-steps, 4 per wave); global loads run TWO stages ahead in two
// register sets.  Same slab contract as conv_wgrad3_kernel: slab[g][32][32] (column j = ci * 9 + kh * 3 + kw), bslab[g][32].
constexpr int IMG1 = 128;
constexpr int DY1CO = 2 * IMG1 + 8;             // bf16 per output channel of a stage (+16 B)
constexpr int DY1PLANE = 32 * DY1CO;
constexpr int X1ROW = IMG1 + 16;                // pixel x at index x + 8
constexpr int X1CI = 4 * X1ROW + 8;
constexpr int X1PLANE = 3 * X1CI;
constexpr size_t WG3_LDS_BYTES = (size_t)(3 * DY1PLANE + 3 * X1PLANE) * 2;

struct B3Wgrad3Params {
    const float* x;         // [B][3][128][128]
    const float* gy;        // pooled gradient [B][32][64][64]
    const uint8_t* mask;    // [B][32][64][64]
    float* slab;            // [grid][32][32]
    float* bslab;           // [grid][32]
    int B;
    int prio;
};

__global__ __launch_bounds__(256) void conv_b3_wgrad3_kernel(B3Wgrad3Params p) {
    if (p.prio >= 3) __builtin_amdgcn_s_setprio(3); else if (p.prio == 2) __builtin_amdgcn_s_setprio(2); else if (p.prio == 1) __builtin_amdgcn_s_setprio(1);
    extern __shared__ __attribute__((aligned(16))) uint16_t smem[];
    uint16_t* DYs = smem;                        // [plane][co 32][row 2][128]
    uint16_t* X1s = smem + 3 * DY1PLANE;         // [plane][ci 3][row 4][X1ROW]
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, r = lane & 31, h = lane >> 5;
    const int nstrips = p.B * (IMG1 / 2);
    for (int i = t * 8; i < 3 * DY1PLANE + 3 * X1PLANE; i += 256 * 8) *reinterpret_cast<u32x4*>(smem + i) = u32x4{0, 0, 0, 0};

    // ---- loader: dY items (co, quad of 4 pooled px) x 2 per thread; X items (ci, row, 8 px): 192, threads 0..191 ----
    f32x4 gq[2][2]; uint32_t mq[2][2]; f32x4 xq[2][2];        // [register set][item]
    uint32_t okx[2] = {0, 0};
    float bsum[2] = {0.f, 0.f};
    const int xi_q = t & 15, xi_row = (t >> 4) & 3, xi_ci = min(t >> 6, 2);
    auto load_stage = [&](int strip, int set) __attribute__((always_inline)) {
        const int b = strip / (IMG1 / 2), ph = strip % (IMG1 / 2);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int idx = t + i * 256, q = idx & 15, co = idx >> 4;
            const long off = (((long)b * 32 + co) * (IMG1 / 2) + ph) * (IMG1 / 2) + q * 4;
            gq[set][i] = *reinterpret_cast<const f32x4*>(p.gy + off);
            mq[set][i] = *reinterpret_cast<const uint32_t*>(p.mask + off);
        }
        const int yr = 2 * ph - 1 + xi_row;
        okx[set] = (yr >= 0 && yr < IMG1 && t < 192) ? 1u : 0u;
        const float* src = p.x + (((long)b * 3 + xi_ci) * IMG1 + min(max(yr, 0), IMG1 - 1)) * IMG1 + xi_q * 8;
        xq[set][0] = *reinterpret_cast<const f32x4*>(src);
        xq[set][1] = *reinterpret_cast<const f32x4*>(src + 4);
    };
    auto store_stage = [&](int set) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int idx = t + i * 256, q = idx & 15, co = idx >> 4;
            float v0[8], v1[8];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const uint32_t m = (mq[set][i] >> (8 * e)) & 0xff;
                const float g = gq[set][i][e];
                bsum[i] += m < 4 ? g : 0.f;
                v0[2 * e] = m == 0 ? g : 0.f; v0[2 * e + 1] = m == 1 ? g : 0.f;
                v1[2 * e] = m == 2 ? g : 0.f; v1[2 * e + 1] = m == 3 ? g : 0.f;
            }
            uint32_t hi[4], mid[4], lo[4];
            uint16_t* d = DYs + co * DY1CO + q * 8;
#pragma unroll
            for (int j = 0; j < 4; ++j) split2(v0[2 * j], v0[2 * j + 1], hi[j], mid[j], lo[j]);
            *reinterpret_cast<u32x4*>(d) = u32x4{hi[0], hi[1], hi[2], hi[3]};
            *reinterpret_cast<u32x4*>(d + DY1PLANE) = u32x4{mid[0], mid[1], mid[2], mid[3]};
            *reinterpret_cast<u32x4*>(d + 2 * DY1PLANE) = u32x4{lo[0], lo[1], lo[2], lo[3]};
#pragma unroll
            for (int j = 0; j < 4; ++j) split2(v1[2 * j], v1[2 * j + 1], hi[j], mid[j], lo[j]);
            *reinterpret_cast<u32x4*>(d + IMG1) = u32x4{hi[0], hi[1], hi[2], hi[3]};
            *reinterpret_cast<u32x4*>(d + IMG1 + DY1PLANE) = u32x4{mid[0], mid[1], mid[2], mid[3]};
            *reinterpret_cast<u32x4*>(d + IMG1 + 2 * DY1PLANE) = u32x4{lo[0], lo[1], lo[2], lo[3]};
        }
        if (t < 192) {
            const bool ok = okx[set] != 0;
            uint32_t hi[4], mid[4], lo[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float a = j < 2 ? xq[set][0][2 * j] : xq[set][1][2 * j - 4], c = j < 2 ? xq[set][0][2 * j + 1] : xq[set][1][2 * j - 3];
                split2(ok ? a : 0.f, ok ? c : 0.f, hi[j], mid[j], lo[j]);
            }
            uint16_t* d = X1s + xi_ci * X1CI + xi_row * X1ROW + 8 + xi_q * 8;
            *reinterpret_cast<u32x4*>(d) = u32x4{hi[0], hi[1], hi[2], hi[3]};
            *reinterpret_cast<u32x4*>(d + X1PLANE) = u32x4{mid[0], mid[1], mid[2], mid[3]};
            *reinterpret_cast<u32x4*>(d + 2 * X1PLANE) = u32x4{lo[0], lo[1], lo[2], lo[3]};
        }
    };

    f32x16 acc;
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[q] = 0.f;
    // column r <-> (ci, kh, kw) = (r / 9, (r % 9) / 3, r % 3) for r < 27; the other five columns compute finite garbage nobody reads
    const int jj = r < 27 ? r : 0;
    const int ci = jj / 9, kh = (jj % 9) / 3, kw = jj % 3;
    const uint16_t* abase = DYs + r * DY1CO + 8 * h;
    const uint16_t* bbase = X1s + ci * X1CI + kh * X1ROW + 8 + 8 * h;
    const uint32_t shift = kw == 1 ? 0u : 16u;

    const int stride = gridDim.x;
    int strip = xcd_adjacent(blockIdx.x, gridDim.x);
    __syncthreads();
    if (strip < nstrips) { load_stage(strip, 0); store_stage(0); }
    if (strip + stride < nstrips) load_stage(strip + stride, 1);
    // the loop is unrolled by two so that the register-set index is a compile-time constant: stage s computes from LDS, the global
    // loads of stage s + 2 go into the set that stage s's data left, stage s + 1's data (loaded a whole stage ago) is split and stored
    for (; strip < nstrips; strip += 2 * stride) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int cur = strip + u * stride;
            if (cur >= nstrips) break;
            __syncthreads();                                 // LDS image of stage `cur` is complete
            if (cur + 2 * stride < nstrips) load_stage(cur + 2 * stride, u);
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) {
                const int ks = wave * 4 + s4, rr = ks >> 3, x0 = (ks & 7) * 16;
                bf16x8 a[3], bf[3];
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) {
                    a[pl] = *reinterpret_cast<const bf16x8*>(abase + pl * DY1PLANE + rr * IMG1 + x0);
                    const uint16_t* src = bbase + pl * X1PLANE + rr * X1ROW + x0;
                    const u32x4 w = *reinterpret_cast<const u32x4*>(src);
                    const uint32_t wm = *reinterpret_cast<const uint32_t*>(src - 2), wp = *reinterpret_cast<const uint32_t*>(src + 8);
                    // kw = 0: pixels x - 1 ..; kw = 1: x ..; kw = 2: x + 1 ..   (alignbit(hi, lo, s) = (lo >> s) | (hi << (32 - s)); s = 0 gives lo)
                    const uint32_t l0 = kw == 0 ? wm : w[0], l1 = kw == 0 ? w[0] : w[1], l2 = kw == 0 ? w[1] : w[2], l3 = kw == 0 ? w[2] : w[3];
                    const uint32_t u0 = kw == 0 ? w[0] : w[1], u1 = kw == 0 ? w[1] : w[2], u2 = kw == 0 ? w[2] : w[3], u3 = kw == 0 ? w[3] : wp;
                    const uint32_t c0 = kw == 1 ? w[0] : l0, c1 = kw == 1 ? w[1] : l1, c2 = kw == 1 ? w[2] : l2, c3 = kw == 1 ? w[3] : l3;
                    const u32x4 f = {__builtin_amdgcn_alignbit(u0, c0, shift), __builtin_amdgcn_alignbit(u1, c1, shift),
                                     __builtin_amdgcn_alignbit(u2, c2, shift), __builtin_amdgcn_alignbit(u3, c3, shift)};
                    bf[pl] = __builtin_bit_cast(bf16x8, f);
                }
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], bf[1], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], bf[0], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], bf[2], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], bf[0], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], bf[1], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], bf[0], acc, 0, 0, 0);
            }
            __syncthreads();                                 // every wave is done reading this stage
            if (cur + stride < nstrips) store_stage(u ^ 1);
        }
    }
    __syncthreads();
    // ---- the four waves' partial tiles through LDS in wave order; wave 0 adds and writes the slab ----
    float* red = reinterpret_cast<float*>(smem);
    for (int g = 1; g < 4; ++g) {
        if (wave == g) {
#pragma unroll
            for (int q = 0; q < 16; ++q) red[q * 64 + lane] = acc[q];
        }
        __syncthreads();
        if (wave == 0) {
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[q] += red[q * 64 + lane];
        }
        __syncthreads();
    }
    if (wave == 0) {
        float* sdst = p.slab + (long)blockIdx.x * 1024;
#pragma unroll
        for (int q = 0; q < 16; ++q) sdst[mfma_row(q, lane) * 32 + r] = acc[q];
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        float v = bsum[i];
        v += __shfl_xor(v, 1); v += __shfl_xor(v, 2); v += __shfl_xor(v, 4); v += __shfl_xor(v, 8);
        const int idx = t + i * 256;
        if ((idx & 15) == 0) p.bslab[(long)blockIdx.x * 32 + (idx >> 4)] = v;
    }
}

template <int MODE, class G>
int launch_b3(const B3Params& p, hipStream_t st) {
    static const int probe = [] { const char* e = getenv("BBBP_B3_PROBE"); return e ? atoi(e) : 0; }();
    constexpr bool flagship = G::CIN == 32 && G::COUT == 64 && G::IMGS == 64;
    if constexpr (G::IMGS == 64) {
        static const int pipe_env = [] { const char* e = getenv("BBBP_C2_PIPE"); return e ? (atoi(e) != 0 ? 1 : 0) : -1; }();
        static const int dpipe_env = [] { const char* e = getenv("BBBP_C2_DGRAD_PIPE"); return e ? (atoi(e) != 0 ? 1 : 0) : 0; }();
        const int pipe = MODE == B3_FWD ? (pipe_env >= 0 ? pipe_env : g_bbbp_conv2_fwd_pipe) : dpipe_env;
        if (pipe && !probe) {
            auto pk = MODE == B3_FWD ? (p.ymask ? conv_b3p_kernel<MODE, G, true> : conv_b3p_kernel<MODE, G, false>) : conv_b3p_kernel<MODE, G, false>;
            constexpr size_t plds = (size_t)2 * (3 * (256 / G::IMGS + 2) * (G::IMGS + 2) * CH + WSTAGE) * 2;
            { int rc_ = bbbp_ensure_dyn_lds(reinterpret_cast<const void*>(pk), plds); if (rc_) return rc_; }
            constexpr int NMBP = (MODE == B3_FWD ? G::COUT : G::CIN) / 32;
            const int nworkp = p.B * (G::IMGS / (256 / G::IMGS)) * NMBP;
            int gridp = bbbp_num_cus();
            if (gridp >= 8 * NMBP) gridp -= gridp % (8 * NMBP);
            if (gridp > nworkp) gridp = nworkp - nworkp % NMBP;
            if (gridp < NMBP) gridp = NMBP;
            hipLaunchKernelGGL(pk, dim3(gridp), dim3(256), plds, st, p);
            BBBP_CHECK_LAUNCH();
            return BBBP_OK;
        }
    }
    auto kernel = conv_b3_kernel<MODE, G>;
    if constexpr (flagship) { if (probe) kernel = conv_b3_probe_kernel<MODE>; }
    constexpr int IMGL = G::IMGS, RL = 256 / IMGL;
    constexpr size_t lds = (size_t)(3 * (RL + 2) * (IMGL + 2) * CH + WSTAGE) * 2;
    { int rc_ = bbbp_ensure_dyn_lds(reinterpret_cast<const void*>(kernel), lds); if (rc_) return rc_; }
    constexpr int NMB = (MODE == B3_FWD ? G::COUT : G::CIN) / 32;
    const int nwork = p.B * (IMGL / RL) * NMB;
    static const int per_cu = [] { const char* e = getenv("BBBP_B3_PER_CU"); const int v = e ? atoi(e) : 2; return v < 1 ? 1 : (v > 2 ? 2 : v); }();
    int grid = bbbp_num_cus() * per_cu;
    if (grid >= 8 * NMB) grid -= grid % (8 * NMB);
    if (grid > nwork) grid = nwork - nwork % NMB;
    if (grid < NMB) grid = NMB;
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(256), lds, st, p);
    BBBP_CHECK_LAUNCH();
    return BBBP_OK;
}

}  // namespace

// conv.hip dispatches the stages this form serves here when the split-bf16 form is selected (bbbp_set_conv_winograd bits 2 / 3).
// workspace: (cin / 16) * (cout / 32) [forward] or (cout / 16) * (cin / 32) [data gradient] stages of WSTAGE pre-split bf16 filters.
bool bbbp_b3_conv_supported(int cin, int cout, int hw) {
    return (cin == 32 && cout == 64 && hw == 64) || (cin == 64 && cout == 128 && hw == 64) || (cin == 128 && cout == 256 && hw == 32);
}
size_t bbbp_b3_workspace_bytes(int cin, int cout) { return (size_t)(cin / 16) * (cout / 32) * WSTAGE * 2 > (size_t)(cout / 16) * (cin / 32) * WSTAGE * 2
                                                             ? (size_t)(cin / 16) * (cout / 32) * WSTAGE * 2 : (size_t)(cout / 16) * (cin / 32) * WSTAGE * 2; }
size_t bbbp_b3_workspace_bytes() { return bbbp_b3_workspace_bytes(32, 64); }

// Wave priority of the BACKWARD conv kernels (BBBP_CONV_BWD_PRIO, default 0).  The encoder chain's small kernels raise themselves to 3 so that
// they win issue arbitration against the older conv waves; since round 4 the data-gradient kernel leaves the chain a wave slot per SIMD
// (<= 224 registers), the image branch is the longer pole of the backward pass, and equal priority (3: the older conv waves then go first)
// hands the arbitration back to it.
static int conv_bwd_prio() {
    static const int v = [] { const char* e = getenv("BBBP_CONV_BWD_PRIO"); const int x = e ? atoi(e) : 0; return x < 0 ? 0 : (x > 3 ? 3 : x); }();
    return v;
}

int bbbp_b3_conv_fwd(hipStream_t st, const float* x, const float* w, const float* bias, float* y, uint8_t* mask, int B, int cin, int cout, void* workspace) {
    hipLaunchKernelGGL(b3_prep_kernel, dim3(cin * cout >= 64 * 128 ? 288 : 72), dim3(256), 0, st, w, static_cast<uint16_t*>(workspace), B3_FWD, cin, cout);
    BBBP_CHECK_LAUNCH();
    B3Params p{x, nullptr, static_cast<const uint16_t*>(workspace), bias, y, mask, B, 0};
    if (cin == 32) return launch_b3<B3_FWD, GeomFlagship>(p, st);
    if (cin == 64) return launch_b3<B3_FWD, B3Geom<64, 128, 64>>(p, st);
    return launch_b3<B3_FWD, B3Geom<128, 256, 32>>(p, st);
}
int bbbp_b3_conv2_fwd(hipStream_t st, const float* x, const float* w, const float* bias, float* y, uint8_t* mask, int B, void* workspace) {
    return bbbp_b3_conv_fwd(st, x, w, bias, y, mask, B, 32, 64, workspace);
}

int bbbp_b3_conv_dgrad(hipStream_t st, const float* gy, const uint8_t* gmask, const float* w, float* dx, int B, int cin, int cout, void* workspace) {
    hipLaunchKernelGGL(b3_prep_kernel, dim3(cin * cout >= 64 * 128 ? 288 : 72), dim3(256), 0, st, w, static_cast<uint16_t*>(workspace), B3_DGRAD, cin, cout);
    BBBP_CHECK_LAUNCH();
    B3Params p{gy, gmask, static_cast<const uint16_t*>(workspace), nullptr, dx, nullptr, B, conv_bwd_prio()};
    if (cin == 32) return launch_b3<B3_DGRAD, GeomFlagship>(p, st);
    if (cin == 64) return launch_b3<B3_DGRAD, B3Geom<64, 128, 64>>(p, st);
    return launch_b3<B3_DGRAD, B3Geom<128, 256, 32>>(p, st);
}
int bbbp_b3_conv2_dgrad(hipStream_t st, const float* gy, const uint8_t* gmask, const float* w, float* dx, int B, void* workspace) {
    return bbbp_b3_conv_dgrad(st, gy, gmask, w, dx, B, 32, 64, workspace);
}

extern "C" int bbbp_conv_b3_phases(unsigned long long* phases4) {
    BBBP_CHECK_ARG(phases4, "conv_b3_phases: null pointer");
    BBBP_CHECK_HIP(hipMemcpyFromSymbol(phases4, HIP_SYMBOL(g_b3_phase), 4 * sizeof(unsigned long long)));
    return BBBP_OK;
}

// grid work-groups, each writes slab[g][64][288] and bslab[g][64] (conv.hip: conv_wgrad32_reduce_kernel finishes)
int bbbp_b3_conv2_wgrad(hipStream_t st, const float* x, const float* gy, const uint8_t* mask, float* slab, float* bslab, int B, int grid, int form,
                        int cin_total, int cout_total, int groups, int map) {
    B3WgradParams p{x, gy, mask, slab, bslab, B, conv_bwd_prio(), cin_total, cout_total, groups > 0 ? groups : grid};
    BBBP_CHECK_ARG(map == 64 || (map == 32 && form != 0), "b3 weight gradient: 64 x 64 maps (any form) or 32 x 32 maps (structured-sparse forms), got %d / form %d", map, form);
    const bool sparse = form != 0;
    static const int waves_env = [] { const char* e = getenv("BBBP_C2_WGRAD_SPARSE_WAVES"); return e ? atoi(e) : 0; }();
    const int waves = waves_env ? waves_env : ((form == 2 || g_bbbp_conv_wgrad_beside_encoder) ? 4 : 8);
    if (map == 32) {
        if (waves == 8) {
            { int rc_ = bbbp_ensure_dyn_lds(reinterpret_cast<const void*>(conv_b3_wgrad_sp_kernel<8, 32>), WGS_LDS_BYTES); if (rc_) return rc_; }
            hipLaunchKernelGGL((conv_b3_wgrad_sp_kernel<8, 32>), dim3(grid), dim3(512), WGS_LDS_BYTES, st, p);
        } else {
            { int rc_ = bbbp_ensure_dyn_lds(reinterpret_cast<const void*>(conv_b3_wgrad_sp_kernel<4, 32>), WGS_LDS_BYTES); if (rc_) return rc_; }
            hipLaunchKernelGGL((conv_b3_wgrad_sp_kernel<4, 32>), dim3(grid), dim3(256), WGS_LDS_BYTES, st, p);
        }
        BBBP_CHECK_LAUNCH();
        return BBBP_OK;
    }
    if (sparse && waves == 8) {
        { int rc_ = bbbp_ensure_dyn_lds(reinterpret_cast<const void*>(conv_b3_wgrad_sp_kernel<8>), WGS_LDS_BYTES); if (rc_) return rc_; }
        hipLaunchKernelGGL(conv_b3_wgrad_sp_kernel<8>, dim3(grid), dim3(512), WGS_LDS_BYTES, st, p);
        BBBP_CHECK_LAUNCH();
        return BBBP_OK;
    }
    if (sparse) {
        { int rc_ = bbbp_ensure_dyn_lds(reinterpret_cast<const void*>(conv_b3_wgrad_sp_kernel<4>), WGS_LDS_BYTES); if (rc_) return rc_; }
        hipLaunchKernelGGL(conv_b3_wgrad_sp_kernel<4>, dim3(grid), dim3(256), WGS_LDS_BYTES, st, p);
        BBBP_CHECK_LAUNCH();
        return BBBP_OK;
    }
    { int rc_ = bbbp_ensure_dyn_lds(reinterpret_cast<const void*>(conv_b3_wgrad_kernel), (size_t)WG_LDS_BYTES); if (rc_) return rc_; }
    hipLaunchKernelGGL(conv_b3_wgrad_kernel, dim3(grid), dim3(256), WG_LDS_BYTES, st, p);
    BBBP_CHECK_LAUNCH();
    return BBBP_OK;
}

// 3 -> 32 @ 128x128 weight gradient: grid work-groups, each writes slab[g][32][32] and bslab[g][32] (conv.hip: conv_wgrad3_reduce_kernel)
int bbbp_b3_conv1_wgrad(hipStream_t st, const float* x, const float* gy, const uint8_t* mask, float* slab, float* bslab, int B, int grid) {
    { int rc_ = bbbp_ensure_dyn_lds(reinterpret_cast<const void*>(conv_b3_wgrad3_kernel), (size_t)WG3_LDS_BYTES); if (rc_) return rc_; }
    B3Wgrad3Params p{x, gy, mask, slab, bslab, B, conv_bwd_prio()};
    hipLaunchKernelGGL(conv_b3_wgrad3_kernel, dim3(grid), dim3(256), WG3_LDS_BYTES, st, p);
    BBBP_CHECK_LAUNCH();
    return BBBP_OK;
}

int bbbp_b3_last_clock(unsigned long long* shader_cycles, unsigned long long* ticks_100mhz) {
    unsigned long long h[2] = {0, 0};
    BBBP_CHECK_HIP(hipMemcpyFromSymbol(h, HIP_SYMBOL(g_b3_clock), sizeof(h)));
    *shader_cycles = h[0]; *ticks_100mhz = h[1];
    return BBBP_OK;
}
