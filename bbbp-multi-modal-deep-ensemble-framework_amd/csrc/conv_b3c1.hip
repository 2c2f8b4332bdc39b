// The flagship CNN's FIRST conv stage (`nn.Conv2d(3, 32, 3, 1, 1) -> ReLU -> MaxPool2d(2, 2)` on 128x128 images,
// Models/multi_input_data_regression_opt_transformer_cnn_20250113.py:85-87), forward, on the BF16 matrix pipe with
// float32 operands split into three bf16 pieces (common.h: split2; six v_mfma_f32_32x32x16_bf16 per float32 product block).
//
// The stage is HBM-bound by its output (101 MB in, 268 MB of pooled activations + 67 MB of masks out at B = 512: floor ~0.07 ms), but
// the f32 form (conv.hip: K = 27 walked flat on v_mfma_f32_32x32x2_f32, 14 x 64 matrix cycles per 32-pixel tile, vector issue blocked
// throughout) is matrix-issue-bound at 0.23 ms.  Round 2's split-bf16 attempt ordered K as (ci, kh, 4 pixel slots) and paid 72
// v_alignbit + 72 selects per 36 MFMAs to assemble operands per lane; it lost.  This kernel needs NO operand assembly:
//   * the input strip sits in LDS pixel-major with the CHANNEL innermost, padded 3 -> 4: [plane][row][px][c0 c1 c2 0] bf16, 8 bytes
//     per pixel and plane.  K is ordered (tap, channel-of-4): 9 taps x 4 = 36, padded to 48 = three k-steps of 16 (taps 9..11 and the
//     fourth channel carry zero WEIGHTS, so whatever finite value the B operand holds there is multiplied by zero);
//   * a lane's B fragment (8 consecutive k of its pixel) is therefore TWO TAPS = two aligned ds_read_b64 of the pixels at those taps --
//     an im2col row that is never materialised: the address arithmetic is one per-lane offset per (k-step, tap);
//   * the A fragments (32 output channels x 48 k x 3 planes, pre-split by a prep kernel) live in 36 registers for the kernel's life;
//   * 18 MFMAs of 32 cycles per 32-pixel tile instead of 14 of 64; the vector ALU is left to the loader (one split per input element).
//   * the pooling epilogue is what bounds this stage once the MFMAs are cheap (67 M pooled outputs + masks: with conv_b3.hip's lane-pair
//     exchange -- ~40 vector instructions per lane and pooled value, half of them selects between "my" and "my partner's" register --
//     the epilogue ALONE ran 0.195 ms).  So the MFMA column index is mapped to EVERY OTHER pixel: a wave owns 2 rows x 64 columns as four
//     accumulator tiles (row 0 even x, row 0 odd x, row 1 even x, row 1 odd x), and lane n holds all four pre-activations of pooling
//     window n in its own registers: max3 + three compares per pooled value, no cross-lane traffic, no selects, every lane busy, and
//     a store instruction writes 128-byte runs.  The LDS strip keeps even and odd pixels of a row in separate runs so that the
//     every-other-pixel fragments are contiguous across lanes (conflict-free).
// A work-group = 4 waves owns a strip of 4 output rows x 128 pixels: wave w computes row pair w / 2, columns 64 (w % 2) .. + 63.
// The 6-row input stage is 18.7 KB and double-buffered: one barrier per strip.
#include "common.h"
#include "bbbp_hip.h"
#include <stdlib.h>
#include <type_traits>

namespace {

constexpr int W1 = 128;                        // image side
constexpr int R1 = 4, ROWS1 = R1 + 2;          // output rows per strip, staged rows (halo)
constexpr int PXW1 = W1 + 2;                   // staged pixels per row: index 0 is x = -1, index 129 is x = 128 (always zero)
constexpr int XPL1 = ROWS1 * PXW1 * 4;         // bf16 elements of one plane of one stage
constexpr int XBUF1 = 3 * XPL1;                // one stage: three planes
constexpr int OUT_WAVE_BYTES = 32 * 32 * 4 + 32 * 32;     // per-wave output staging: [32 co][32 pooled px] float + the same in mask bytes
constexpr size_t LDS1_BYTES = (size_t)2 * XBUF1 * 2 + 4 * OUT_WAVE_BYTES;       // 37.4 KB of input stages + 20 KB
constexpr int WFRAG_WORDS = 3 * 3 * 64 * 4;    // [k-step][plane][lane][4 words]

struct C1Params {
    const float* x;           // [B][3][128][128]
    const uint32_t* wfrag;    // pre-split filters in A-fragment order
    const float* bias;        // [32]
    float* y;                 // pooled [B][32][64][64]
    uint8_t* ymask;
    int B;
    int exp;                  // experiment bits (BBBP_C1_EXP): 1 no output stores, 2 no MFMAs, 4 no stage loads
};

// filters W[co 32][ci 3][tap 9] -> A fragments: k = 16 s + 8 h + e  <->  tap = k / 4, channel = k % 4 (taps >= 9 / channel 3: zero)
// `bias` (pipelined kernel only): rides as one more row of the GEMM -- k = 4 * 4 + 3, the padding channel of the CENTRE tap, whose B
// operand that kernel keeps at 1.0 for every pixel inside the image -- so the accumulators leave the MFMAs with the bias added.
__global__ void c1_prep_kernel(const float* __restrict__ w, uint32_t* __restrict__ wf, const float* __restrict__ bias) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;           // (s, lane, pair j)
    if (idx >= 3 * 64 * 4) return;
    const int j = idx & 3, lane = (idx >> 2) & 63, s = idx >> 8;
    const int m = lane & 31, h = lane >> 5;
    float v[2];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        const int k = 16 * s + 8 * h + 2 * j + e, tap = k >> 2, ci = k & 3;
        v[e] = (tap < 9 && ci < 3) ? w[(m * 3 + ci) * 9 + tap] : (bias && tap == 4 && ci == 3) ? bias[m] : 0.f;
    }
    uint32_t hi, mid, lo;
    split2(v[0], v[1], hi, mid, lo);
    wf[((s * 3 + 0) * 64 + lane) * 4 + j] = hi;
    wf[((s * 3 + 1) * 64 + lane) * 4 + j] = mid;
    wf[((s * 3 + 2) * 64 + lane) * 4 + j] = lo;
}

template <bool MASK>      // MASK = false: a forward-only plan, no pooling decisions are kept (p.ymask is null)
__global__ __launch_bounds__(256, 2) void conv1_b3_fwd_kernel(C1Params p) {
    extern __shared__ __attribute__((aligned(16))) uint16_t smem[];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, r = lane & 31, h = lane >> 5;
    const int nstrips = p.B * (W1 / R1);
    const int stride = gridDim.x;

    // the filters: nine fragments per lane, for the kernel's life
    bf16x8 a[3][3];
#pragma unroll
    for (int s = 0; s < 3; ++s)
#pragma unroll
        for (int pl = 0; pl < 3; ++pl)
            a[s][pl] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(p.wfrag + ((s * 3 + pl) * 64 + lane) * 4));
    // the bias is added to the pooled maximum (max commutes with adding one constant to the four candidates), not carried through
    // the accumulation: 16 registers and 64 moves per strip less

    for (int i = t * 8; i < 2 * XBUF1; i += 256 * 8) *reinterpret_cast<u32x4*>(smem + i) = u32x4{0, 0, 0, 0};

    // ---- stage loader: item = (row, pixel): three dword loads (one per channel plane), one split, three 8-byte LDS writes.
    //      LDS row: [65 even-x slots: x = 0, 2, .., 128][65 odd-x slots: x = -1, 1, .., 127]; x = 128 and x = -1 stay zero ----
    constexpr int NIT = ROWS1 * W1 / 256;                    // 3
    int goff[NIT], loff[NIT], irow[NIT];
#pragma unroll
    for (int i = 0; i < NIT; ++i) {
        const int idx = t + i * 256;
        const int px = idx % W1, row = idx / W1, par = px & 1;
        goff[i] = px; loff[i] = (row * PXW1 + par * 65 + (px >> 1) + par) * 4; irow[i] = row;
    }
    float xr[NIT][3];
    uint32_t okbits = 0;
    auto load_stage = [&](int strip) __attribute__((always_inline)) {
        const int b = strip / (W1 / R1), h0 = (strip % (W1 / R1)) * R1;
        const float* xb = p.x + (long)b * 3 * W1 * W1;
        okbits = 0;
#pragma unroll
        for (int i = 0; i < NIT; ++i) {
            const int yr = h0 - 1 + irow[i];
            const int yy = min(max(yr, 0), W1 - 1);
            okbits |= (yr >= 0 && yr < W1 ? 1u : 0u) << i;
            const int o = yy * W1 + goff[i];
#pragma unroll
            for (int c = 0; c < 3; ++c) xr[i][c] = xb[o + c * W1 * W1];
        }
    };
    auto store_stage = [&](uint16_t* Xs) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < NIT; ++i) {
            const bool ok = (okbits >> i) & 1u;
            uint32_t h01, m01, l01, h2, m2, l2;
            split2(ok ? xr[i][0] : 0.f, ok ? xr[i][1] : 0.f, h01, m01, l01);
            split2(ok ? xr[i][2] : 0.f, 0.f, h2, m2, l2);
            uint16_t* d = Xs + loff[i];
            *reinterpret_cast<u32x2*>(d) = u32x2{h01, h2};
            *reinterpret_cast<u32x2*>(d + XPL1) = u32x2{m01, m2};
            *reinterpret_cast<u32x2*>(d + 2 * XPL1) = u32x2{l01, l2};
        }
    };

    // this wave: row pair rp of the strip, columns 64 ch .. 64 ch + 63; lane column r <-> pixels x = 64 ch + 2 r (even tile) and + 1 (odd)
    const int rp = wave >> 1, ch = wave & 1;
    // per-lane fragment offsets (bf16 elements from the row's start).  k-step s, half h covers taps 4 s + 2 h and 4 s + 2 h + 1 (a tap
    // beyond 8 reads tap 8's pixel: its weights are zero).  Pixel x + dx of the even tile (x = 64 ch + 2 r): dx = 0 -> even slot
    // 32 ch + r; dx = -1 -> odd slot 32 ch + r; dx = +1 -> odd slot 32 ch + r + 1.  Odd tile (x + 1): dx = 0 -> odd slot 32 ch + r + 1;
    // dx = -1 -> even slot 32 ch + r; dx = +1 -> even slot 32 ch + r + 1.
    int toff[2][3][2];
#pragma unroll
    for (int s = 0; s < 3; ++s)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int tap = min(4 * s + 2 * h + j, 8), dy1 = tap / 3, dx = tap % 3 - 1;
            const int ev = dx == 0 ? 32 * ch + r : 65 + 32 * ch + r + (dx > 0 ? 1 : 0);
            const int od = dx == 0 ? 65 + 32 * ch + r + 1 : 32 * ch + r + (dx > 0 ? 1 : 0);
            toff[0][s][j] = (dy1 * PXW1 + ev) * 4;
            toff[1][s][j] = (dy1 * PXW1 + od) * 4;
        }

    int strip = xcd_adjacent(blockIdx.x, gridDim.x);
    __syncthreads();                                                 // the zero fill (halo slots stay zero for the kernel's life)
    if (strip < nstrips) { load_stage(strip); store_stage(smem); }
    __syncthreads();
    int cur = 0;
    for (; strip < nstrips; strip += stride) {
        const int b = strip / (W1 / R1), h0 = (strip % (W1 / R1)) * R1;
        const int nstrip = strip + stride;
        const bool have_next = nstrip < nstrips;
        if (have_next && !(p.exp & 4)) load_stage(nstrip);           // global loads in flight under this strip's MFMAs
        const uint16_t* Xs = smem + cur * XBUF1 + (2 * rp) * PXW1 * 4;   // staged row of output row 2 rp, tap row 0
        f32x16 acc[4];                                               // [row * 2 + parity]
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[nt][q] = 0.f;
        // twelve (tile, k-step) blocks of six MFMAs; the fragments of block i + 1 are fetched while block i runs
        bf16x8 bq[2][3];
        auto fetch = [&](int blk, int slot) __attribute__((always_inline)) {
            const int nt = blk / 3, s = blk % 3;
            const uint16_t* base = Xs + (nt >> 1) * PXW1 * 4;
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) {
                const u32x2 lo = *reinterpret_cast<const u32x2*>(base + pl * XPL1 + toff[nt & 1][s][0]);
                const u32x2 hi = *reinterpret_cast<const u32x2*>(base + pl * XPL1 + toff[nt & 1][s][1]);
                bq[slot][pl] = __builtin_bit_cast(bf16x8, u32x4{lo[0], lo[1], hi[0], hi[1]});
            }
        };
        if (!(p.exp & 2)) {
        fetch(0, 0);
#pragma unroll
        for (int blk = 0; blk < 12; ++blk) {
            const int sl = blk & 1, nt = blk / 3, s = blk % 3;
            if (blk + 1 < 12) fetch(blk + 1, sl ^ 1);
            // small terms first, the leading product last
            acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[s][1], bq[sl][1], acc[nt], 0, 0, 0);
            acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[s][2], bq[sl][0], acc[nt], 0, 0, 0);
            acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[s][0], bq[sl][2], acc[nt], 0, 0, 0);
            acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[s][1], bq[sl][0], acc[nt], 0, 0, 0);
            acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[s][0], bq[sl][1], acc[nt], 0, 0, 0);
            acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[s][0], bq[sl][0], acc[nt], 0, 0, 0);
        }
        }
        // ---- epilogue: ReLU + 2x2 max-pool + arg-max mask (PyTorch scan order (y0, xe), (y0, xo), (y1, xe), (y1, xo), first maximum wins;
        //      4 = ReLU inactive).  The four candidates of lane r's window are registers q of its own four tiles.  The wave's
        //      [32 co][32 pooled px] results pass through 5 KB of wave-private LDS so that they leave as 16-byte stores (4 + 1 store
        //      instructions per lane instead of 16 dword + 16 byte stores). ----
        {
            float* of = reinterpret_cast<float*>(reinterpret_cast<char*>(smem) + (size_t)2 * XBUF1 * 2 + wave * OUT_WAVE_BYTES);
            uint8_t* om = reinterpret_cast<uint8_t*>(of + 32 * 32);
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int co = (q & 3) + 8 * (q >> 2) + 4 * h;
                const float v0 = acc[0][q], v1 = acc[1][q], v2 = acc[2][q], v3 = acc[3][q];
                const float m = __builtin_fmaxf(__builtin_fmaxf(v0, v1), __builtin_fmaxf(v2, v3));
                // uniform addresses: the two candidates sit in scalar registers, the lane's half picks one
                const float bias0 = p.bias[(q & 3) + 8 * (q >> 2)], bias1 = p.bias[(q & 3) + 8 * (q >> 2) + 4];
                const float best = m + (h ? bias1 : bias0);
                const bool act = best > 0.f;
                of[co * 32 + r] = act ? best : 0.f;
                if (MASK) { const int arg = v0 == m ? 0 : v1 == m ? 1 : v2 == m ? 2 : 3; om[co * 32 + r] = act ? (uint8_t)arg : (uint8_t)4; }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            const int ph2 = (h0 >> 1) + rp;
            if (!(p.exp & 1)) {
                float* yb = p.y + ((long)b * 32 * (W1 / 2) + ph2) * (W1 / 2) + 32 * ch;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int i = lane + 64 * j, co = i >> 3, quad = i & 7;
                    *reinterpret_cast<f32x4*>(yb + (long)co * (W1 / 2) * (W1 / 2) + 4 * quad) = *reinterpret_cast<const f32x4*>(of + i * 4);
                }
                if (MASK) {
                    uint8_t* mbp = p.ymask + ((long)b * 32 * (W1 / 2) + ph2) * (W1 / 2) + 32 * ch;
                    *reinterpret_cast<u32x4*>(mbp + (long)(lane >> 1) * (W1 / 2) * (W1 / 2) + 16 * (lane & 1)) = *reinterpret_cast<const u32x4*>(om + lane * 16);
                }
            }
            __builtin_amdgcn_wave_barrier();     // the staging area is rewritten by the next strip
        }
        if (have_next && !(p.exp & 4)) store_stage(smem + (cur ^ 1) * XBUF1);
        __syncthreads();          // the next stage is complete, and every wave is done with this one (overwritten one strip later)
        cur ^= 1;
    }
}


// ------------------------------------------------------------------------------------------------------------------------------
// The same stage, SOFTWARE-PIPELINED inside every wave (round 4).  In the kernel above a wave runs its phases one after the other
// (issue the next stage's loads, 72 MFMAs, ~300 vector instructions of pooling epilogue, the split of the next stage, a barrier), and
// the ablation of round 4 (profiles/r04_c1_ablate.txt, B = 4096: 1.47 ms; without stores 1.19, without MFMAs 0.89, without stage loads
// 1.04, without all three 0.43) shows the phases ADD: the second work-group of a CU runs in lockstep with the first and hides
// nothing.  A bf16 MFMA holds its SIMD's vector issue for only 8 of its 32 cycles, so one wave can carry ~5 vector instructions in
// every MFMA gap for free.  Here ONE instruction stream interleaves, at a skew of half a strip:
//   first half of strip i  -- the 36 MFMAs of its row-0 tiles (even / odd pixels)  ||  the second half of strip i - 1's pooling epilogue
//                             (row 1 against the kept row-0 maxima, ReLU, decision byte, staging, 16-byte stores);
//   second half of strip i -- the 36 MFMAs of its row-1 tiles  ||  the first half of its own epilogue (row-0 maxima of the tiles just
//                             finished: 16 + 16 kept registers) and the split + LDS writes of strip i + 1's input stage.
// A tile's accumulators are consumed in the half after they are produced and rewritten in the half after that, so ONE set of 64 accumulator
// registers serves (a full-strip skew needs two sets and spilled: 320 live registers).  One wave per SIMD keeps the matrix pipe busy by
// itself, so ONE 4-wave work-group per CU is enough -- which leaves half of every SIMD's register file and 100 KB of LDS to the encoder
// chain's kernels on the other stream: the form a training step can run beside its chain.
// The bias rides as one more row of the GEMM (c1_prep_kernel: k = 19, the padding channel of the centre tap, whose B operand this kernel
// keeps at 1.0 for pixels inside the image): no bias registers, no add.  Same LDS image, filter fragments, tie rule as above.
// ------------------------------------------------------------------------------------------------------------------------------
#define C1P_SGB(mask, n) __builtin_amdgcn_sched_group_barrier(mask, n, 0)
// max(a, b) as ONE instruction: fmaxf quiets signalling NaNs first (a v_max_f32 x, x, x per operand that is not known canonical -- every
// accumulator register), v_med3_f32 with +inf does not.  For a NaN operand it returns the other one or +inf instead of fmaxf's "the
// non-NaN operand": inputs with NaNs are outside this path's contract either way (PyTorch would propagate the NaN).
__device__ __forceinline__ float fmax_raw(float a, float b) { return __builtin_amdgcn_fmed3f(a, b, __builtin_huge_valf()); }

template <bool MASK>
__global__ __launch_bounds__(256, 2) void conv1_b3p_fwd_kernel(C1Params p) {
    extern __shared__ __attribute__((aligned(16))) uint16_t smem[];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, r = lane & 31, h = lane >> 5;
    const int nstrips = p.B * (W1 / R1);
    const int stride = gridDim.x;
    const int first = xcd_adjacent(blockIdx.x, gridDim.x);
    if (first >= nstrips) return;                                    // uniform: the whole work-group leaves
    const int n = (nstrips - first + stride - 1) / stride;           // strips of this work-group
    const int last_strip = first + (n - 1) * stride;

    bf16x8 a[3][3];
#pragma unroll
    for (int s = 0; s < 3; ++s)
#pragma unroll
        for (int pl = 0; pl < 3; ++pl)
            a[s][pl] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(p.wfrag + ((s * 3 + pl) * 64 + lane) * 4));

    for (int i = t * 8; i < 2 * XBUF1; i += 256 * 8) *reinterpret_cast<u32x4*>(smem + i) = u32x4{0, 0, 0, 0};

    // ---- stage loader: item i of thread t = pixel (row t / 128 + 2 i, x = t % 128) ----
    constexpr int NIT = ROWS1 * W1 / 256;                            // 3
    const int px = t & (W1 - 1), row0 = t >> 7, par = px & 1;
    const int loff0 = (row0 * PXW1 + par * 65 + (px >> 1) + par) * 4;
    float xr[NIT][3];
    uint32_t okbits = 0;
    auto load_stage = [&](int strip) __attribute__((always_inline)) {
        const int b = strip / (W1 / R1), h0 = (strip % (W1 / R1)) * R1;
        // uniform 64-bit bases (one per channel plane) + ONE 32-bit per-lane byte offset per item: no 64-bit vector address arithmetic
        const char* xb = reinterpret_cast<const char*>(p.x + (long)b * 3 * W1 * W1);
        okbits = 0;
#pragma unroll
        for (int i = 0; i < NIT; ++i) {
            const int yr = h0 - 1 + row0 + 2 * i;
            const int yy = min(max(yr, 0), W1 - 1);
            okbits |= (yr >= 0 && yr < W1 ? 1u : 0u) << i;
            const unsigned o = 4u * (unsigned)(yy * W1 + px);
#pragma unroll
            for (int c = 0; c < 3; ++c) xr[i][c] = *reinterpret_cast<const float*>(xb + (size_t)c * (W1 * W1 * 4) + (size_t)o);
        }
    };
    auto store_item = [&](uint16_t* Xs, int i) __attribute__((always_inline)) {
        const bool ok = (okbits >> i) & 1u;
        uint32_t h01, m01, l01, h2, m2, l2;
        split2(ok ? xr[i][0] : 0.f, ok ? xr[i][1] : 0.f, h01, m01, l01);
        split2(ok ? xr[i][2] : 0.f, ok ? 1.f : 0.f, h2, m2, l2);     // the padding channel carries 1.0: the bias row of the filters
        uint16_t* d = Xs + loff0 + i * (2 * PXW1 * 4);
        *reinterpret_cast<u32x2*>(d) = u32x2{h01, h2};
        *reinterpret_cast<u32x2*>(d + XPL1) = u32x2{m01, m2};
        *reinterpret_cast<u32x2*>(d + 2 * XPL1) = u32x2{l01, l2};
    };

    const int rp = wave >> 1, ch = wave & 1;
    int toff[2][3][2];
#pragma unroll
    for (int s = 0; s < 3; ++s)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int tap = min(4 * s + 2 * h + j, 8), dy1 = tap / 3, dx = tap % 3 - 1;
            const int ev = dx == 0 ? 32 * ch + r : 65 + 32 * ch + r + (dx > 0 ? 1 : 0);
            const int od = dx == 0 ? 65 + 32 * ch + r + 1 : 32 * ch + r + (dx > 0 ? 1 : 0);
            toff[0][s][j] = (dy1 * PXW1 + ev) * 4;
            toff[1][s][j] = (dy1 * PXW1 + od) * 4;
        }
    float* const of = reinterpret_cast<float*>(reinterpret_cast<char*>(smem) + (size_t)2 * XBUF1 * 2 + wave * OUT_WAVE_BYTES);
    uint8_t* const om = reinterpret_cast<uint8_t*>(of + 32 * 32);

    __syncthreads();                                                 // the zero fill
    load_stage(first);
#pragma unroll
    for (int i = 0; i < NIT; ++i) store_item(smem, i);
    __syncthreads();

    f32x16 acc[4];                                                   // [row * 2 + pixel parity]
    float m01[16];                                                   // max over row 0 of the window, per accumulator register
    uint32_t a01[MASK ? 16 : 1];                                     // its position (0 / 1)

    // One pipeline step.  MMA: the MFMAs of strip `strip` (LDS stage `cur`); EPI: the second epilogue half of the strip before it.
    auto step = [&](auto MMAc, auto EPIc, int strip, int cur) __attribute__((always_inline)) {
        constexpr bool MMA = decltype(MMAc)::value, EPI = decltype(EPIc)::value;
        const int pstrip = MMA ? strip - stride : strip;             // the strip whose results leave in this step
        const int pb = pstrip / (W1 / R1), ph0 = (pstrip % (W1 / R1)) * R1;
        if (MMA) load_stage(min(strip + stride, last_strip));       // unconditional (clamped): a branch here would split the schedule
        const uint16_t* Xs = smem + cur * XBUF1 + (2 * rp) * PXW1 * 4;
        uint16_t* Xn = smem + (cur ^ 1) * XBUF1;
        bf16x8 bq[2][3];
        auto fetch = [&](int blk, int slot) __attribute__((always_inline)) {
            const int nt = blk / 3, s = blk % 3;
            const uint16_t* base = Xs + (nt >> 1) * PXW1 * 4;
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) {
                const u32x2 lo = *reinterpret_cast<const u32x2*>(base + pl * XPL1 + toff[nt & 1][s][0]);
                const u32x2 hi = *reinterpret_cast<const u32x2*>(base + pl * XPL1 + toff[nt & 1][s][1]);
                bq[slot][pl] = __builtin_bit_cast(bf16x8, u32x4{lo[0], lo[1], hi[0], hi[1]});
            }
        };
        // first epilogue half (row 0 of the window: tiles 0, 1), PyTorch scan order: the later pixel only if strictly larger
        auto epi_a = [&](int q) __attribute__((always_inline)) {
            const float v0 = acc[0][q], v1 = acc[1][q];
            m01[q] = fmax_raw(v0, v1);
            if (MASK) a01[q] = v1 > v0 ? 1u : 0u;
        };
        // second half (row 1: tiles 2, 3): the lower row only if strictly larger; ReLU; 4 = inactive
        auto epi_b = [&](int q) __attribute__((always_inline)) {
            const int co = (q & 3) + 8 * (q >> 2) + 4 * h;
            const float v2 = acc[2][q], v3 = acc[3][q];
            const float m23 = fmax_raw(v2, v3);
            const float best = fmax_raw(m01[q], m23);                // bias included (filter row k = 19)
            of[co * 32 + r] = fmax_raw(best, 0.f);
            if (MASK) {
                const uint32_t a23 = v3 > v2 ? 3u : 2u;
                const uint32_t arg = m23 > m01[q] ? a23 : a01[q];
                om[co * 32 + r] = best > 0.f ? (uint8_t)arg : (uint8_t)4;
            }
        };
        auto epi_out = [&]() __attribute__((always_inline)) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            const int ph2 = (ph0 >> 1) + rp;
            float* yb = p.y + ((long)pb * 32 * (W1 / 2) + ph2) * (W1 / 2) + 32 * ch;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int i = lane + 64 * j, co = i >> 3, quad = i & 7;
                *reinterpret_cast<f32x4*>(yb + (long)co * (W1 / 2) * (W1 / 2) + 4 * quad) = *reinterpret_cast<const f32x4*>(of + i * 4);
            }
            if (MASK) {
                uint8_t* mbp = p.ymask + ((long)pb * 32 * (W1 / 2) + ph2) * (W1 / 2) + 32 * ch;
                *reinterpret_cast<u32x4*>(mbp + (long)(lane >> 1) * (W1 / 2) * (W1 / 2) + 16 * (lane & 1)) = *reinterpret_cast<const u32x4*>(om + lane * 16);
            }
            __builtin_amdgcn_wave_barrier();
        };
        constexpr int QB[6] = {0, 4, 7, 10, 13, 16};                  // epilogue B: registers [QB[blk], QB[blk + 1]) ride in block blk = 0..4
        constexpr int QA[4] = {0, 6, 11, 16};                         // epilogue A: blocks 6..8
        if (MMA) {
            fetch(0, 0);
#pragma unroll
            for (int blk = 0; blk < 12; ++blk) {
                const int sl = blk & 1, nt = blk / 3, s = blk % 3;
                if (blk + 1 < 12) fetch(blk + 1, sl ^ 1);
                f32x16 c0;
                if (s == 0) {
#pragma unroll
                    for (int q = 0; q < 16; ++q) c0[q] = 0.f;         // folds into the instruction's inline zero
                } else c0 = acc[nt];
                // small terms first, the leading product last
                c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[s][1], bq[sl][1], c0, 0, 0, 0);
                c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[s][2], bq[sl][0], c0, 0, 0, 0);
                c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[s][0], bq[sl][2], c0, 0, 0, 0);
                c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[s][1], bq[sl][0], c0, 0, 0, 0);
                c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[s][0], bq[sl][1], c0, 0, 0, 0);
                acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[s][0], bq[sl][0], c0, 0, 0, 0);
                if (EPI && blk < 5) {
#pragma unroll
                    for (int q = QB[blk]; q < QB[blk + 1]; ++q) epi_b(q);
                }
                if (EPI && blk == 5) epi_out();
                if (blk >= 6 && blk < 9) {
#pragma unroll
                    for (int q = QA[blk - 6]; q < QA[blk - 5]; ++q) epi_a(q);
                }
                if (blk >= 9) store_item(Xn, blk - 9);
                // issue order of the block: behind every MFMA up to two LDS accesses, six vector instructions and one global access
#pragma unroll
                for (int k = 0; k < 6; ++k) {
                    C1P_SGB(0x008, 1);
                    C1P_SGB(0x080, 2);
                    C1P_SGB(0x002, 6);
                    C1P_SGB(0x010, 1);
                }
                __builtin_amdgcn_sched_barrier(0);                   // nothing is hoisted across blocks (register pressure)
            }
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");      // LDS only: the epilogue's global stores stay in flight
        } else if (EPI) {
#pragma unroll
            for (int q = 0; q < 16; ++q) epi_b(q);
            epi_out();
        }
    };
    using T = std::true_type; using Fa = std::false_type;
    step(T{}, Fa{}, first, 0);
    for (int i = 1; i < n; ++i) step(T{}, T{}, first + i * stride, i & 1);
    step(Fa{}, T{}, last_strip, 0);
}

}  // namespace

// workspace: WFRAG_WORDS uint32 of pre-split filter fragments (9 KB)
size_t bbbp_b3_conv1_fwd_workspace_bytes() { return (size_t)WFRAG_WORDS * sizeof(uint32_t); }

int bbbp_b3_conv1_fwd(hipStream_t st, const float* x, const float* w, const float* bias, float* y, uint8_t* mask, int B, void* workspace) {
    uint32_t* wf = static_cast<uint32_t*>(workspace);
    static const int pipe = [] { const char* e = getenv("BBBP_C1_PIPE"); return e ? atoi(e) : 1; }();     // round 4: the software-pipelined form (0: round 3's phase-by-phase kernel)
    hipLaunchKernelGGL(c1_prep_kernel, dim3(3), dim3(256), 0, st, w, wf, pipe ? bias : nullptr);
    BBBP_CHECK_LAUNCH();
    static const int exp_bits = [] { const char* e = getenv("BBBP_C1_EXP"); return e ? atoi(e) : 0; }();
    C1Params p{x, wf, bias, y, mask, B, exp_bits};
    static const int per_cu_env = [] { const char* e = getenv("BBBP_C1_PER_CU"); const int v = e ? atoi(e) : 2; return v < 1 ? 1 : (v > 4 ? 4 : v); }();
    const int per_cu = g_bbbp_conv1_fwd_per_cu > 0 ? g_bbbp_conv1_fwd_per_cu : per_cu_env;
    const int nstrips = B * (W1 / R1);
    int grid = bbbp_num_cus() * per_cu;
    if (grid >= 8) grid -= grid % 8;
    if (grid > nstrips) grid = nstrips;
    if (grid < 1) grid = 1;
    if (pipe) {
        if (mask) hipLaunchKernelGGL(conv1_b3p_fwd_kernel<true>, dim3(grid), dim3(256), LDS1_BYTES, st, p);
        else hipLaunchKernelGGL(conv1_b3p_fwd_kernel<false>, dim3(grid), dim3(256), LDS1_BYTES, st, p);
    } else if (mask) hipLaunchKernelGGL(conv1_b3_fwd_kernel<true>, dim3(grid), dim3(256), LDS1_BYTES, st, p);
    else hipLaunchKernelGGL(conv1_b3_fwd_kernel<false>, dim3(grid), dim3(256), LDS1_BYTES, st, p);
    BBBP_CHECK_LAUNCH();
    return BBBP_OK;
}
