// fp32 GEMM on the exact-f32 MFMA (v_mfma_f32_32x32x2_f32) for gfx950, with fused epilogues.
//
//   C[M,N] = act(alpha * op(A) * op(B) + bias[n]) (+ R[M,N])          row-major everywhere
//
// Replaces the ATen/oneDNN calls behind nn.Linear / F.linear and the attention matmuls of
// nn.TransformerEncoderLayer on the reference's hot path (a3, a4, a8, a9, a10 and
// their autograd counterparts a12).  Three storage layouts cover forward and backward:
//   NT: A[M][K], B[N][K]   y = x W^T (Linear forward), S = Q K^T, dP = dO V^T
//   NN: A[M][K], B[K][N]   dx = dy W, O = P V, dQ = dS K
//   TN: A[K][M], B[K][N]   dW = dy^T x, dV = P^T dO, dK = dS^T Q
// Both LDS tiles are k-major ([BK][BM], [BK][BN]) so that the MFMA operand fetch
// (lane l: A[i=l&31][k=l>>5], B[k=l>>5][j=l&31]) is a conflict-free ds_read_b32 of 32 consecutive
// floats per half-wave.  m-major global tiles are transposed on the way into LDS (odd row stride:
// at most 2 lanes per bank, which ds_write_b32 absorbs).  Arbitrary M, N, K and leading dimensions
// (the MACCS width is the prime 167): 16-byte loads when alignment allows, predicated scalar loads
// otherwise, zero fill out of range.  Split-K writes raw partial slabs that a second kernel sums
// in a fixed order (bit-reproducible; no float atomics).
// Launches below ~1.2 GFLOP (every GEMM of the F = 167 encoder) take a second, latency-oriented kernel further down
// (gemm_direct_kernel): no LDS staging, 16x16 wave tiles on v_mfma_f32_16x16x4_f32, K slices reduced inside the
// work-group.  bbbp_gemm_f32 picks the path; results of both agree to rounding (different summation order).
#include "common.h"
#include "bbbp_hip.h"
#include <mutex>
#include <unordered_map>

#define TRY_RC(expr) do { int _rc = (expr); if (_rc) return _rc; } while (0)

typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));      // gfx950 global loads need only 4-byte alignment
typedef float f32x2u __attribute__((ext_vector_type(2), aligned(4)));

namespace {

// K depth of one LDS stage.  The 64x64 tile keeps 17 KB of LDS and ~32 VGPRs on purpose: the encoder's small
// GEMMs run on a second stream BESIDE the persistent conv kernels (engine.hip), and a work-group only co-resides
// with a 140 KB / 224-VGPR conv work-group if it is this small.  Measured: deeper stages (64) bought < 5 %
// because these launches sit on the ~5 us per-kernel latency floor, not on MFMA issue.
constexpr int bk_of(int tile) { return tile == 64 ? 16 : 32; }

struct GemmParams {
    const float* A; const float* B; float* C;
    const float* bias; const float* R;
    int M, N, K;
    int lda, ldb, ldc, ldr;
    long sA, sB, sC, sR;     // batch strides (elements)
    float alpha;
    int act;                 // 0 none, 1 relu, 2 tanh
    int splits, kchunk;      // split-K: K range per split (multiple of BK)
    float* slab;             // [batch][split][M][N] when splits > 1
    unsigned* arrivals;      // split-bf16 kernel only: one counter per output tile; the LAST K range to arrive sums the tile's slabs itself
    int vecA, vecB;          // 16-byte global loads allowed
    int short_k;             // 128 x 128 tiles with 16-deep stages
    const float* gate; int ldg; long sG; float gate_scale;      // optional: result *= gate > 0 ? gate_scale : 0
    int gate_after;          // gate applied after the residual add
};

// Four consecutive elements starting at p, `valid` (0..4) of them inside the matrix; the rest read as 0.
// The loads are UNCONDITIONAL: out-of-range lanes read a clamped in-range address (or `safe`, the matrix base) and
// are zeroed by selects.  A load under a per-lane `if` makes hipcc wait vmcnt(0) at the join, which serialised
// every K stage behind a full memory round trip.  VEC: one 16-byte load (the host guarantees valid is 0 or 4).
template <bool VEC>
__device__ __forceinline__ float4 ld4(const float* p, const float* safe, int valid) {
    const float* q = valid > 0 ? p : safe;
    if (VEC) {          // gfx950 global loads need only 4-byte alignment: one dwordx4 also for odd parameter offsets / strides
        typedef float f32x4_any __attribute__((ext_vector_type(4), aligned(4)));
        const f32x4_any v = *reinterpret_cast<const f32x4_any*>(q);
        return make_float4(v.x, v.y, v.z, v.w);
    }
    const int m = max(valid, 1) - 1;
    return make_float4(q[0], q[min(1, m)], q[min(2, m)], q[min(3, m)]);
}
// ... and the zeroing select, applied when the staged registers are written to LDS (NOT right after the load: a
// consumer next to the load would put the wait back in front of the MFMA block)
__device__ __forceinline__ float4 mask4(float4 v, int valid) {
    return make_float4(valid > 0 ? v.x : 0.f, valid > 1 ? v.y : 0.f, valid > 2 ? v.z : 0.f, valid > 3 ? v.w : 0.f);
}

__device__ __forceinline__ float apply_act(float v, int act) {
    if (act == 1) return v > 0.f ? v : 0.f;
    if (act == 2) return tanhf(v);
    return v;
}

// LAYOUT 0: NT, 1: NN, 2: TN;  VEC: 16-byte global loads on both operands
// BKT: k-depth of a stage (0 = bk_of(BM)).  16 instead of 32 halves the LDS of the 128 x 128 tile to 33 KB, so four
// work-groups share a CU instead of two: for a short K (the image FC's input gradient, K = 128 = 4 stages of 32) the tile's
// prologue, epilogue and exposed load latency then overlap with three other groups' MFMAs instead of one.
template <int BM, int BN, int LAYOUT, bool VEC, int BKT = 0>
__global__ __launch_bounds__(256) void gemm_f32_kernel(GemmParams p) {
    BBBP_HIGH_PRIO();
    constexpr int BK = BKT ? BKT : bk_of(BM);
    constexpr bool A_KMAJ = (LAYOUT == 2);
    constexpr bool B_KMAJ = (LAYOUT != 0);
    // k-major global tiles land with 16-B stores (row stride % 4 == 0); m-major tiles are transposed with
    // ds_write_b32 at an odd row stride: lanes (k-quad, row) then hit every bank at most twice (free on gfx950)
    constexpr int LDAS = BM + (A_KMAJ ? 4 : 1);
    constexpr int LDBS = BN + (B_KMAJ ? 4 : 1);
    constexpr int QK = BK / 4;                    // 16-B quads along k per m-major row
    constexpr int RPP = 256 / QK;                 // m-major rows covered per pass
    constexpr int WM = BM / 2, WN = BN / 2;      // 2 x 2 waves
    constexpr int TM = WM / 32, TN = WN / 32;
    constexpr int NA = BM * BK / 256 / 4;         // float4 per thread per operand tile
    constexpr int NB = BN * BK / 256 / 4;
    extern __shared__ __attribute__((aligned(16))) float smem[];     // 2 * BK * (LDAS + LDBS) floats
    float* As = smem;
    float* Bs = smem + 2 * BK * LDAS;

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int batch = blockIdx.z / p.splits, split = blockIdx.z % p.splits;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    const int kbeg = split * p.kchunk;
    const int kend = min(p.K, kbeg + p.kchunk);
    const float* A = p.A + (long)batch * p.sA;
    const float* B = p.B + (long)batch * p.sB;

    float4 ra[NA], rb[NB];
    int va[NA], vb[NB];          // valid element counts of the staged quads

    auto load_tiles = [&](int k0) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            if (A_KMAJ) {       // global [K][M]: quads along m
                constexpr int QPR = BM / 4;
                int q = t % QPR, kr = t / QPR + i * (256 / QPR);
                int k = k0 + kr, m = m0 + q * 4;
                int valid = (k < kend) ? min(4, p.M - m) : 0;
                ra[i] = ld4<VEC>(A + (long)k * p.lda + m, A, valid); va[i] = valid;
            } else {            // global [M][K]: quads along k
                int row = t / QK + i * RPP, kq = t % QK;
                int m = m0 + row, k = k0 + kq * 4;
                int valid = (m < p.M) ? min(4, kend - k) : 0;
                ra[i] = ld4<VEC>(A + (long)m * p.lda + k, A, valid); va[i] = valid;
            }
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            if (B_KMAJ) {       // global [K][N]
                constexpr int QPR = BN / 4;
                int q = t % QPR, kr = t / QPR + i * (256 / QPR);
                int k = k0 + kr, n = n0 + q * 4;
                int valid = (k < kend) ? min(4, p.N - n) : 0;
                rb[i] = ld4<VEC>(B + (long)k * p.ldb + n, B, valid); vb[i] = valid;
            } else {            // global [N][K]
                int row = t / QK + i * RPP, kq = t % QK;
                int n = n0 + row, k = k0 + kq * 4;
                int valid = (n < p.N) ? min(4, kend - k) : 0;
                rb[i] = ld4<VEC>(B + (long)n * p.ldb + k, B, valid); vb[i] = valid;
            }
        }
    };
    auto store_tiles = [&](int buf) __attribute__((always_inline)) {
        float* as = As + buf * BK * LDAS;
        float* bs = Bs + buf * BK * LDBS;
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const float4 v = mask4(ra[i], va[i]);
            if (A_KMAJ) {
                constexpr int QPR = BM / 4;
                int q = t % QPR, kr = t / QPR + i * (256 / QPR);
                *reinterpret_cast<float4*>(as + kr * LDAS + q * 4) = v;
            } else {
                int row = t / QK + i * RPP, kq = t % QK;
                as[(kq * 4 + 0) * LDAS + row] = v.x;
                as[(kq * 4 + 1) * LDAS + row] = v.y;
                as[(kq * 4 + 2) * LDAS + row] = v.z;
                as[(kq * 4 + 3) * LDAS + row] = v.w;
            }
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const float4 v = mask4(rb[i], vb[i]);
            if (B_KMAJ) {
                constexpr int QPR = BN / 4;
                int q = t % QPR, kr = t / QPR + i * (256 / QPR);
                *reinterpret_cast<float4*>(bs + kr * LDBS + q * 4) = v;
            } else {
                int row = t / QK + i * RPP, kq = t % QK;
                bs[(kq * 4 + 0) * LDBS + row] = v.x;
                bs[(kq * 4 + 1) * LDBS + row] = v.y;
                bs[(kq * 4 + 2) * LDBS + row] = v.z;
                bs[(kq * 4 + 3) * LDBS + row] = v.w;
            }
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nt = (kend - kbeg + BK - 1) / BK;
    if (nt > 0) {
        load_tiles(kbeg);
        store_tiles(0);
    }
    __syncthreads();
    const int aoff = (lane >> 5) * LDAS + wm * WM + (lane & 31);
    const int boff = (lane >> 5) * LDBS + wn * WN + (lane & 31);
    for (int it = 0; it < nt; ++it) {
        const int buf = it & 1;
        if (it + 1 < nt) load_tiles(kbeg + (it + 1) * BK);
        const float* as = As + buf * BK * LDAS + aoff;
        const float* bs = Bs + buf * BK * LDBS + boff;
        // Fetch the operands of 8 k-steps at once, then issue their MFMAs: with one wave per SIMD nothing else hides the
        // LDS latency, and read -> wait -> MFMA per k-step cost ~250 cycles per step (measured by ablation,
        // tools/exp_gemm_dbg.py: the K loop took 0.8 us per 16-deep stage even with loads, stores and MFMAs removed).
        constexpr int G = 8;
#pragma unroll
        for (int k0 = 0; k0 < BK / 2; k0 += G) {
            float a[G][TM], b[G][TN];
#pragma unroll
            for (int g = 0; g < G; ++g) {
#pragma unroll
                for (int i = 0; i < TM; ++i) a[g][i] = as[(k0 + g) * 2 * LDAS + i * 32];
#pragma unroll
                for (int j = 0; j < TN; ++j) b[g][j] = bs[(k0 + g) * 2 * LDBS + j * 32];
            }
            __builtin_amdgcn_sched_barrier(0);      // keep hipcc from sinking the reads back next to their MFMAs
#pragma unroll
            for (int g = 0; g < G; ++g)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) acc[i][j] = mfma32(a[g][i], b[g][j], acc[i][j]);
        }
        if (it + 1 < nt) store_tiles(buf ^ 1);
        __syncthreads();
    }

    // epilogue
    if (p.splits > 1) {
        float* S = p.slab + ((long)batch * p.splits + split) * (long)p.M * p.N;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                int n = n0 + wn * WN + j * 32 + (lane & 31);
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    int m = m0 + wm * WM + i * 32 + mfma_row(r, lane);
                    if (m < p.M && n < p.N) S[(long)m * p.N + n] = acc[i][j][r];
                }
            }
        return;
    }
    float* C = p.C + (long)batch * p.sC;
    const float* R = p.R ? p.R + (long)batch * p.sR : nullptr;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            int n = n0 + wn * WN + j * 32 + (lane & 31);
            float bv = (p.bias && n < p.N) ? p.bias[n] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                int m = m0 + wm * WM + i * 32 + mfma_row(r, lane);
                if (m < p.M && n < p.N) {
                    float v = apply_act(p.alpha * acc[i][j][r] + bv, p.act);
                    const float gsel = p.gate ? (p.gate[(long)batch * p.sG + (long)m * p.ldg + n] > 0.f ? p.gate_scale : 0.f) : 1.f;
                    if (!p.gate_after) v *= gsel;
                    if (R) v += R[(long)m * p.ldr + n];
                    if (p.gate_after) v *= gsel;
                    C[(long)m * p.ldc + n] = v;
                }
            }
        }
}

// ---------------------------------------------------------------------------------------------------------------------
// The LARGE products (128 x 128 tiles: the F = 2048 encoder's projections / FFN and their gradients, the 65536-wide image FC) on the
// bf16 matrix pipe with split operands (common.h: split2): every float32 operand element is cut into three bf16 pieces on its way
// into LDS and a k-step of 16 is six v_mfma_f32_32x32x16_bf16 per 32 x 32 tile -- float32 accuracy (the dropped piece products are
// below 2^-24 of the product) at 2.67x the rate of v_mfma_f32_32x32x2_f32.  Same GemmParams, grid, split-K slabs and epilogue as
// gemm_f32_kernel<128, 128>.
//   * LDS: [operand 2][plane 3][row 128][32 k + 8] bf16 = 60 KB, ONE stage, so two work-groups share a CU;
//   * the global loads of stage s + 2 are issued at the top of stage s (two float32 register sets), the split of stage s + 1 into its
//     bf16 pieces is INTERLEAVED with the MFMAs of stage s (the bf16 MFMA leaves the vector ALU free for 24 of its 32 cycles) and waits
//     in registers; between two barriers only the 24 ds_write_b64 remain;
//   * both tiles are row-major in k: the operand of lane (r, h) -- row r, k = 8h .. 8h+7 -- is one ds_read_b128, conflict-free at the
//     80-byte row stride; a k-major global tile ([K][M]) is transposed in registers: a thread loads a 4 (k) x 4 (m) block with four
//     16-byte loads and writes four 8-byte k-quads per plane.  The rows a half-wave writes are 4 apart (80 * 4 bytes = 16 banks): no
//     bank conflicts on the way in either;
//   * a K that is not a multiple of 32 (round 3: the MACCS width 167 at screening batch sizes -- linear1, the in / out projections and
//     Q K^T of a 4096-row forward) ends in ONE partial stage whose loads are element-wise with clamped addresses and zero fill
//     (B3Loader::load_tail); every other stage keeps the select-free 16-byte path.  A k-major operand still needs extent % 4 == 0.
constexpr int B3_BK = 32;
constexpr int B3_LD = B3_BK + 8;
constexpr int B3_PLANE = 128 * B3_LD;
constexpr size_t B3_LDS = (size_t)6 * B3_PLANE * sizeof(uint16_t);

template <bool KMAJ>
struct B3Loader {
    const float* ptr[4];     // this thread's four 16-byte loads of the next stage to fetch
    const float* X0;         // a valid address for clamped loads of the tail stage
    long step;               // pointer advance per stage
    int kpos, kend;          // k of this thread's first element in the next stage to fetch; end of this work-group's K range
    int soff[4];             // LDS offsets (bf16 elements) of its four k-quads
    f32x4u raw[2][4];
    u32x2 pk[4][3];
    // thread (kq = t & 7, rq = t >> 3): k-contiguous operand: rows row(rq) + 32 i, k = 4 kq ..+3;  k-major: k = 4 kq + j, rows 4 rq ..+3
    __device__ __forceinline__ void init(const float* X, int ld, int extent, int row0, int kbeg, int kend_, int t) {
        const int kq = t & 7, rq = t >> 3;
        X0 = X; kpos = kbeg + 4 * kq; kend = kend_;
        if constexpr (KMAJ) {
            const int col = min(row0 + 4 * rq, extent - 4);
#pragma unroll
            for (int j = 0; j < 4; ++j) { ptr[j] = X + (long)(kbeg + 4 * kq + j) * ld + col; soff[j] = (4 * rq + j) * B3_LD + 4 * kq; }
            step = (long)B3_BK * ld;
        } else {
            const int a = rq & 3, b = rq >> 2;
            const int row = (b >> 2) * 16 + (b & 3) + 4 * a;          // the four rows of a half-wave are 4 apart
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                ptr[i] = X + (long)min(row0 + row + 32 * i, extent - 1) * ld + kbeg + 4 * kq;
                soff[i] = (row + 32 * i) * B3_LD + 4 * kq;
            }
            step = B3_BK;
        }
    }
    __device__ __forceinline__ void load(int set) {
#pragma unroll
        for (int i = 0; i < 4; ++i) { raw[set][i] = *reinterpret_cast<const f32x4u*>(ptr[i]); ptr[i] += step; }
        kpos += B3_BK;
    }
    // the last, partial stage of a K range that is not a multiple of 32: k >= kend contributes zeros
    __device__ __forceinline__ void load_tail(int set) {
        if constexpr (KMAJ) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const bool ok = kpos + j < kend;
                const f32x4u v = *reinterpret_cast<const f32x4u*>(ok ? ptr[j] : X0);
                raw[set][j] = ok ? v : f32x4u{0.f, 0.f, 0.f, 0.f};
            }
        } else {
            const int nv = kend - kpos;                      // valid elements of this thread's quad (<= 0: none)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float* q = nv > 0 ? ptr[i] : X0;
                f32x4u v;
#pragma unroll
                for (int e = 0; e < 4; ++e) { const float x = q[e < nv ? e : 0]; v[e] = e < nv ? x : 0.f; }
                raw[set][i] = v;
            }
        }
        kpos += B3_BK;
    }
    __device__ __forceinline__ void split(int set) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            uint32_t h0, m0, l0, h1, m1, l1;
            if constexpr (KMAJ) {
                split2(raw[set][0][i], raw[set][1][i], h0, m0, l0);
                split2(raw[set][2][i], raw[set][3][i], h1, m1, l1);
            } else {
                split2(raw[set][i][0], raw[set][i][1], h0, m0, l0);
                split2(raw[set][i][2], raw[set][i][3], h1, m1, l1);
            }
            pk[i][0] = u32x2{h0, h1}; pk[i][1] = u32x2{m0, m1}; pk[i][2] = u32x2{l0, l1};
        }
    }
    __device__ __forceinline__ void store(uint16_t* dst) const {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) *reinterpret_cast<u32x2*>(dst + soff[i] + pl * B3_PLANE) = pk[i][pl];
    }
};

// phase breakdown of work-group 0 / wave 0 of the last probed launch (BBBP_GEMM_B3_PROBE=1; tools/bench_gemm_forms.py prints it): shader
// cycles in [0] global-load issue, [1] LDS reads + MFMA block + split, [2] barrier after it, [3] LDS writes, [4] barrier after them
__device__ unsigned long long g_gemm_b3_phase[7];      // [5] all shader cycles of the wave, [6] the same span in 100 MHz wall ticks

template <int LAYOUT, bool PROBE>
__device__ __forceinline__ void gemm_b3_body(const GemmParams& p) {
    BBBP_HIGH_PRIO();
    unsigned long long ph[5] = {0, 0, 0, 0, 0}, c0 = 0;
    const unsigned long long cyc_begin = PROBE ? __builtin_readcyclecounter() : 0, wall_begin = PROBE ? wall_clock64() : 0;
    constexpr bool A_KMAJ = (LAYOUT == 2), B_KMAJ = (LAYOUT != 0);
    extern __shared__ __attribute__((aligned(16))) uint16_t smem16[];
    uint16_t* As = smem16;
    uint16_t* Bs = smem16 + 3 * B3_PLANE;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, r = lane & 31, h = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;
    const int batch = blockIdx.z / p.splits, split = blockIdx.z % p.splits;
    // (an XCD-aware order of the row tiles of one column block -- id % 8 = XCD, row tile the slot's fastest digit -- was measured on the
    // image FC's input gradient, 4 x 512 tiles, K = 128: 103.6 vs 103.2 us, the re-read column blocks come from the Infinity Cache anyway)
    const int m0 = blockIdx.y * 128, n0 = blockIdx.x * 128;
    const int kbeg = split * p.kchunk;
    const int kend = min(p.K, kbeg + p.kchunk);
    const int nt = (kend - kbeg + B3_BK - 1) / B3_BK;     // kchunk % 32 == 0: only the last K range can end in a partial stage
    const int tail_stage = ((kend - kbeg) % B3_BK) ? nt - 1 : -1;

    B3Loader<A_KMAJ> la;
    B3Loader<B_KMAJ> lb;
    la.init(p.A + (long)batch * p.sA, p.lda, p.M, m0, kbeg, kend, t);
    lb.init(p.B + (long)batch * p.sB, p.ldb, p.N, n0, kbeg, kend, t);
    auto load_stage = [&](int stage, int set) __attribute__((always_inline)) {
        if (stage == tail_stage) { la.load_tail(set); lb.load_tail(set); }       // wave-uniform
        else { la.load(set); lb.load(set); }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.f;

    if (nt > 0) {
        load_stage(0, 0);
        if (nt > 1) load_stage(1, 1);
        la.split(0); lb.split(0);
        la.store(As); lb.store(Bs);
    }
    __syncthreads();
    const uint16_t* afrag = As + (wm * 64 + r) * B3_LD + 8 * h;
    const uint16_t* bfrag = Bs + (wn * 64 + r) * B3_LD + 8 * h;
    for (int it = 0; it < nt; it += 2) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int cur = it + u;
            if (cur >= nt) break;
            if (PROBE) c0 = __builtin_readcyclecounter();
            // float32 set u held stage `cur` (split one stage ago): refill it with stage cur + 2
            if (cur + 2 < nt) load_stage(cur + 2, u);
            if (PROBE) { const unsigned long long c = __builtin_readcyclecounter(); ph[0] += c - c0; c0 = c; }
#pragma unroll
            for (int kk = 0; kk < B3_BK / 16; ++kk) {
                bf16x8 a[2][3], b[2][3];
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int pl = 0; pl < 3; ++pl) {
                        a[i][pl] = *reinterpret_cast<const bf16x8*>(afrag + pl * B3_PLANE + i * 32 * B3_LD + kk * 16);
                        b[i][pl] = *reinterpret_cast<const bf16x8*>(bfrag + pl * B3_PLANE + i * 32 * B3_LD + kk * 16);
                    }
                // smallest piece products first; the four tiles interleaved so that consecutive MFMAs are independent
                constexpr int PA[6] = {1, 2, 0, 1, 0, 0}, PB[6] = {1, 0, 2, 0, 1, 0};
#pragma unroll
                for (int q = 0; q < 6; ++q)
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int j = 0; j < 2; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][PA[q]], b[j][PB[q]], acc[i][j], 0, 0, 0);
            }
            // stage cur + 1 (float32 set u ^ 1, loaded a whole stage ago) -> bf16 pieces, in registers; unconditional so that it stays in
            // the MFMAs' basic block (after the last stage it chews on stale registers)
            la.split(u ^ 1); lb.split(u ^ 1);
#pragma unroll
            for (int g = 0; g < 48; ++g) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);      // one MFMA ...
                __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);      // ... three VALU instructions of the split under it
            }
            if (PROBE) { const unsigned long long c = __builtin_readcyclecounter(); ph[1] += c - c0; c0 = c; }
            __syncthreads();                         // every wave is done reading this stage
            if (PROBE) { const unsigned long long c = __builtin_readcyclecounter(); ph[2] += c - c0; c0 = c; }
            if (cur + 1 < nt) { la.store(As); lb.store(Bs); }
            if (PROBE) { const unsigned long long c = __builtin_readcyclecounter(); ph[3] += c - c0; c0 = c; }
            __syncthreads();
            if (PROBE) { const unsigned long long c = __builtin_readcyclecounter(); ph[4] += c - c0; c0 = c; }
        }
    }
    if (PROBE && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && t == 0) {
#pragma unroll
        for (int i = 0; i < 5; ++i) g_gemm_b3_phase[i] = ph[i];
        g_gemm_b3_phase[5] = __builtin_readcyclecounter() - cyc_begin;
        g_gemm_b3_phase[6] = wall_clock64() - wall_begin;
    }

    if (p.splits > 1) {
        float* S = p.slab + ((long)batch * p.splits + split) * (long)p.M * p.N;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int n = n0 + wn * 64 + j * 32 + r;
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    const int m = m0 + wm * 64 + i * 32 + mfma_row(q, lane);
                    if (m < p.M && n < p.N) {
                        // with the in-kernel reduction the slab goes straight to the coherence point (agent-scope relaxed store = sc1 write-through)
                        if (p.arrivals) __hip_atomic_store(&S[(long)m * p.N + n], acc[i][j][q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        else S[(long)m * p.N + n] = acc[i][j][q];
                    }
                }
            }
        if (!p.arrivals) return;                     // the reduce kernel follows
        // The K range that arrives LAST at this tile sums the tile's slabs -- in split order, from memory, its own included, so the
        // result does not depend on who was last and equals gemm_splitk_reduce_kernel's bit for bit -- and applies the epilogue: no
        // second launch (round 3; at F = 2048 the 42 reduce launches of a step are 0.6 ms).  OPT-IN: measured slower.  The slabs of the other K ranges were
        // written through other XCDs' L2s: every slab store and load of this path is an agent-scope relaxed atomic (sc1: written through
        // to / read from the coherence point), ordered by s_waitcnt vmcnt(0) before the arrival counter is bumped.  NOT by
        // __threadfence(): an agent-scope release / acquire is a write-back / invalidate of the whole L2 of the XCD, which every
        // kernel running beside this one pays for (measured: F = 2048 step 8.4 -> 12.5 ms with fences; 9.85 ms with the sc1 accesses,
        // whose 4-byte write-through stores and uncached loads cost more than the 15 us launch they replace).
        __shared__ int s_last;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (t == 0) {
            unsigned* cnt = p.arrivals + ((long)batch * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
            const unsigned old = __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            s_last = (old == (unsigned)p.splits - 1u) ? 1 : 0;
            if (s_last) __hip_atomic_store(cnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // ready for the next launch on this stream
        }
        __syncthreads();
        if (!s_last) return;
        const long mn = (long)p.M * p.N;
        const float* S0 = p.slab + (long)batch * p.splits * mn;
        float* Cf = p.C + (long)batch * p.sC;
        const float* Rf = p.R ? p.R + (long)batch * p.sR : nullptr;
        const int tw = min(128, p.N - n0), th = min(128, p.M - m0);
        for (int e = t; e < th * 128; e += 256) {
            const int lm = e >> 7, ln = e & 127;
            if (ln >= tw) continue;
            const int m = m0 + lm, n = n0 + ln;
            const float* sp = S0 + (long)m * p.N + n;
            float sum = 0.f;
            int k = 0;
            for (; k + 8 <= p.splits; k += 8) {
                float v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = __hip_atomic_load(sp + (long)(k + u) * mn, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
                for (int u = 0; u < 8; ++u) sum += v[u];
            }
            for (; k < p.splits; ++k) sum += __hip_atomic_load(sp + (long)k * mn, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            float v = apply_act(p.alpha * sum + (p.bias ? p.bias[n] : 0.f), p.act);
            const float gsel = p.gate ? (p.gate[(long)batch * p.sG + (long)m * p.ldg + n] > 0.f ? p.gate_scale : 0.f) : 1.f;
            if (!p.gate_after) v *= gsel;
            if (Rf) v += Rf[(long)m * p.ldr + n];
            if (p.gate_after) v *= gsel;
            Cf[(long)m * p.ldc + n] = v;
        }
        return;
    }
    float* C = p.C + (long)batch * p.sC;
    const float* R = p.R ? p.R + (long)batch * p.sR : nullptr;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int n = n0 + wn * 64 + j * 32 + r;
            const float bv = (p.bias && n < p.N) ? p.bias[n] : 0.f;
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int m = m0 + wm * 64 + i * 32 + mfma_row(q, lane);
                if (m < p.M && n < p.N) {
                    float v = apply_act(p.alpha * acc[i][j][q] + bv, p.act);
                    const float gsel = p.gate ? (p.gate[(long)batch * p.sG + (long)m * p.ldg + n] > 0.f ? p.gate_scale : 0.f) : 1.f;
                    if (!p.gate_after) v *= gsel;
                    if (R) v += R[(long)m * p.ldr + n];
                    if (p.gate_after) v *= gsel;
                    C[(long)m * p.ldc + n] = v;
                }
            }
        }
}

template <int LAYOUT>
__global__ __launch_bounds__(256, 2) void gemm_b3_kernel(GemmParams p) { gemm_b3_body<LAYOUT, false>(p); }
template <int LAYOUT>
__global__ __launch_bounds__(256, 2) void gemm_b3_probe_kernel(GemmParams p) { gemm_b3_body<LAYOUT, true>(p); }

// ---------------------------------------------------------------------------------------------------------------------
// 64 x 64 tiles of the same split-bf16 form (round 3): products with a deep K whose output has too few 128 x 128 tiles to fill the chip
// -- the F = 2048 encoder's in_proj at B = 512: 512 x 6144 outputs are 192 tiles of 128 x 128 (one pass on 192 of 256 CUs) -- run here as 768
// work-groups that each walk the WHOLE K: no slabs, no second launch.  Four waves, each one 32 x 32 accumulator tile (two accumulators: even / odd piece products, so consecutive
// MFMAs are independent); LDS [operand 2][plane 3][row 64][32 k + 8] bf16 = 30 KB, up to three work-groups per CU.  Per 16-deep k-step a
// wave issues 6 ds_read_b128 for 6 MFMAs (the 128 x 128 tile: 12 for 24), so this tile is LDS-bound at about two thirds of the matrix
// pipe -- still ahead of split-K plus reduce for these shapes.  Same loader idiom, pipeline (loads two stages ahead, the split of the next stage
// under the MFMAs of this one, two barriers per stage), K tail and epilogue as gemm_b3_kernel.
constexpr int B3S_T = 64;
#ifndef BBBP_B3S_DEPTH_NT
#define BBBP_B3S_DEPTH_NT 2
#endif
#ifndef BBBP_B3S_DEPTH_NN
#define BBBP_B3S_DEPTH_NN 2
#endif
constexpr int B3S_DEPTH_NT = BBBP_B3S_DEPTH_NT, B3S_DEPTH_NN = BBBP_B3S_DEPTH_NN;
constexpr int B3S_PLANE = B3S_T * B3_LD;
constexpr size_t B3S_LDS = (size_t)6 * B3S_PLANE * sizeof(uint16_t);

template <bool KMAJ, int SETS>
struct B3SLoader {
    static constexpr int NL = KMAJ ? 4 : 2;      // 16-byte loads per thread and stage (k-major: threads 0..127 only)
    const float* ptr[NL];
    const float* X0;
    long step;
    int kpos, kend;
    int soff[NL];
    bool live;
    f32x4u raw[SETS][NL];
    u32x2 pk[NL][3];
    __device__ __forceinline__ void init(const float* X, int ld, int extent, int row0, int kbeg, int kend_, int t) {
        const int kq = t & 7, rq = t >> 3;
        X0 = X; kpos = kbeg + 4 * kq; kend = kend_;
        if constexpr (KMAJ) {
            live = rq < 16;                                           // 16 column quads x 8 k quads = 128 threads
            const int cq = rq & 15;
            const int col = min(row0 + 4 * cq, extent - 4);
#pragma unroll
            for (int j = 0; j < 4; ++j) { ptr[j] = X + (long)(kbeg + 4 * kq + j) * ld + col; soff[j] = (4 * cq + j) * B3_LD + 4 * kq; }
            step = (long)B3_BK * ld;
        } else {
            live = true;
            const int a = rq & 3, b = rq >> 2;
            const int row = (b >> 2) * 16 + (b & 3) + 4 * a;          // 0..31; the four rows of a half-wave are 4 apart
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                ptr[i] = X + (long)min(row0 + row + 32 * i, extent - 1) * ld + kbeg + 4 * kq;
                soff[i] = (row + 32 * i) * B3_LD + 4 * kq;
            }
            step = B3_BK;
        }
    }
    __device__ __forceinline__ void load(int set) {
#pragma unroll
        for (int i = 0; i < NL; ++i) { raw[set][i] = *reinterpret_cast<const f32x4u*>(ptr[i]); ptr[i] += step; }
        kpos += B3_BK;
    }
    __device__ __forceinline__ void load_tail(int set) {
        if constexpr (KMAJ) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const bool ok = kpos + j < kend;
                const f32x4u v = *reinterpret_cast<const f32x4u*>(ok ? ptr[j] : X0);
                raw[set][j] = ok ? v : f32x4u{0.f, 0.f, 0.f, 0.f};
            }
        } else {
            const int nv = kend - kpos;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const float* q = nv > 0 ? ptr[i] : X0;
                f32x4u v;
#pragma unroll
                for (int e = 0; e < 4; ++e) { const float x = q[e < nv ? e : 0]; v[e] = e < nv ? x : 0.f; }
                raw[set][i] = v;
            }
        }
        kpos += B3_BK;
    }
    __device__ __forceinline__ void split(int set) {
        if constexpr (KMAJ) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                uint32_t h0, m0, l0, h1, m1, l1;
                split2(raw[set][0][i], raw[set][1][i], h0, m0, l0);
                split2(raw[set][2][i], raw[set][3][i], h1, m1, l1);
                pk[i][0] = u32x2{h0, h1}; pk[i][1] = u32x2{m0, m1}; pk[i][2] = u32x2{l0, l1};
            }
        } else {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                uint32_t h0, m0, l0, h1, m1, l1;
                split2(raw[set][i][0], raw[set][i][1], h0, m0, l0);
                split2(raw[set][i][2], raw[set][i][3], h1, m1, l1);
                pk[i][0] = u32x2{h0, h1}; pk[i][1] = u32x2{m0, m1}; pk[i][2] = u32x2{l0, l1};
            }
        }
    }
    __device__ __forceinline__ void store(uint16_t* dst) const {
        if (!live) return;
#pragma unroll
        for (int i = 0; i < NL; ++i)
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) *reinterpret_cast<u32x2*>(dst + soff[i] + pl * B3S_PLANE) = pk[i][pl];
    }
};

// D: stages of global loads in flight (float32 register sets; A/B builds: -DBBBP_B3S_DEPTH_NT=n).  Measured (config 4, B = 512): with ONE work-group
// per CU (512 x 2048 outputs: 256 tiles) a stage takes ~2200 cycles for 384 cycles of MFMAs -- one wave per SIMD serialises its loads'
// split, LDS writes, two barriers and LDS reads -- 67 us against 48 for the 128-tile split-K plan + reduce, and 6 stages of loads in
// flight change nothing (69 us); with THREE per CU (512 x 6144: 768 tiles) the groups cover each other: 114 us against 130.  The plan
// therefore only takes this tile when it puts at least 2.5 work-groups on every CU.
template <int LAYOUT, int D>
__global__ __launch_bounds__(256, (D <= 2 ? 3 : 2)) void gemm_b3s_kernel(GemmParams p) {
    BBBP_HIGH_PRIO();
    static_assert(LAYOUT == 0 || LAYOUT == 1, "NT and NN only: the weight gradients (TN) have 256+ tiles of 128 x 128");
    constexpr bool B_KMAJ = (LAYOUT != 0);
    extern __shared__ __attribute__((aligned(16))) uint16_t smem16[];
    uint16_t* As = smem16;
    uint16_t* Bs = smem16 + 3 * B3S_PLANE;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, r = lane & 31, h = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;
    const int batch = blockIdx.z;
    const int m0 = blockIdx.y * B3S_T, n0 = blockIdx.x * B3S_T;
    const int nt = (p.K + B3_BK - 1) / B3_BK;
    const int tail_stage = (p.K % B3_BK) ? nt - 1 : -1;

    B3SLoader<false, D> la;
    B3SLoader<B_KMAJ, D> lb;
    la.init(p.A + (long)batch * p.sA, p.lda, p.M, m0, 0, p.K, t);
    lb.init(p.B + (long)batch * p.sB, p.ldb, p.N, n0, 0, p.K, t);
    auto load_stage = [&](int stage, int set) __attribute__((always_inline)) {
        if (stage == tail_stage) { la.load_tail(set); lb.load_tail(set); }
        else { la.load(set); lb.load(set); }
    };
    f32x16 acc0, acc1;
#pragma unroll
    for (int q = 0; q < 16; ++q) { acc0[q] = 0.f; acc1[q] = 0.f; }
    if (nt > 0) {
#pragma unroll
        for (int d = 0; d < D; ++d)
            if (d < nt) load_stage(d, d);
        la.split(0); lb.split(0);
        la.store(As); lb.store(Bs);
    }
    __syncthreads();
    const uint16_t* afrag = As + (wm * 32 + r) * B3_LD + 8 * h;
    const uint16_t* bfrag = Bs + (wn * 32 + r) * B3_LD + 8 * h;
    for (int it = 0; it < nt; it += D) {
#pragma unroll
        for (int u = 0; u < D; ++u) {
            const int cur = it + u;
            if (cur >= nt) break;
            // float32 set u held stage `cur` (split one stage ago): refill it with stage cur + D
            if (cur + D < nt) load_stage(cur + D, u);
#pragma unroll
            for (int kk = 0; kk < B3_BK / 16; ++kk) {
                bf16x8 a[3], b[3];
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) {
                    a[pl] = *reinterpret_cast<const bf16x8*>(afrag + pl * B3S_PLANE + kk * 16);
                    b[pl] = *reinterpret_cast<const bf16x8*>(bfrag + pl * B3S_PLANE + kk * 16);
                }
                // smallest piece products first, alternating accumulators
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[1], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], b[0], acc1, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[2], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[0], acc1, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[1], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[0], acc1, 0, 0, 0);
            }
            // stage cur + 1 (loaded D - 1 stages ago) -> bf16 pieces, in registers
            la.split((u + 1) % D); lb.split((u + 1) % D);
#pragma unroll
            for (int g = 0; g < 12; ++g) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);      // one MFMA ...
                __builtin_amdgcn_sched_group_barrier(0x002, 6, 0);      // ... six VALU instructions of the split under it
            }
            __syncthreads();
            if (cur + 1 < nt) { la.store(As); lb.store(Bs); }
            __syncthreads();
        }
    }
    float* C = p.C + (long)batch * p.sC;
    const float* R = p.R ? p.R + (long)batch * p.sR : nullptr;
    const int n = n0 + wn * 32 + r;
    const float bv = (p.bias && n < p.N) ? p.bias[n] : 0.f;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        const int m = m0 + wm * 32 + mfma_row(q, lane);
        if (m < p.M && n < p.N) {
            float v = apply_act(p.alpha * (acc0[q] + acc1[q]) + bv, p.act);
            const float gsel = p.gate ? (p.gate[(long)batch * p.sG + (long)m * p.ldg + n] > 0.f ? p.gate_scale : 0.f) : 1.f;
            if (!p.gate_after) v *= gsel;
            if (R) v += R[(long)m * p.ldr + n];
            if (p.gate_after) v *= gsel;
            C[(long)m * p.ldc + n] = v;
        }
    }
}

// sums the split-K slabs in split order and applies the epilogue
__global__ __launch_bounds__(256) void gemm_splitk_reduce_kernel(GemmParams p) {
    BBBP_HIGH_PRIO();
    const long mn = (long)p.M * p.N;
    const int batch = blockIdx.y;
    const float* S = p.slab + (long)batch * p.splits * mn;
    float* C = p.C + (long)batch * p.sC;
    const float* R = p.R ? p.R + (long)batch * p.sR : nullptr;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < mn; idx += (long)gridDim.x * 256) {
        // 8 independent loads in flight, summed in slab order (the order, hence the result, does not change)
        float s = 0.f;
        int k = 0;
        for (; k + 8 <= p.splits; k += 8) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = S[(long)(k + u) * mn + idx];
#pragma unroll
            for (int u = 0; u < 8; ++u) s += v[u];
        }
        for (; k < p.splits; ++k) s += S[(long)k * mn + idx];
        int m = (int)(idx / p.N), n = (int)(idx % p.N);
        float v = apply_act(p.alpha * s + (p.bias ? p.bias[n] : 0.f), p.act);
        const float gsel = p.gate ? (p.gate[(long)batch * p.sG + (long)m * p.ldg + n] > 0.f ? p.gate_scale : 0.f) : 1.f;
        if (!p.gate_after) v *= gsel;
        if (R) v += R[(long)m * p.ldr + n];
        if (p.gate_after) v *= gsel;
        C[(long)m * p.ldc + n] = v;
    }
}


// the same sums, four consecutive columns per lane (N, every leading dimension and every base a multiple of 4 floats: the F = 2048
// encoder's 42 reduce launches per step; 16 K waves of one element each become 4 K waves of one 16-byte access per slab).  Element by
// element the slabs are added in the same order as above: bit-identical.
__global__ __launch_bounds__(256) void gemm_splitk_reduce4_kernel(GemmParams p) {
    BBBP_HIGH_PRIO();
    const long mn = (long)p.M * p.N, mn4 = mn >> 2;
    const int batch = blockIdx.y;
    const float* S = p.slab + (long)batch * p.splits * mn;
    float* C = p.C + (long)batch * p.sC;
    const float* R = p.R ? p.R + (long)batch * p.sR : nullptr;
    for (long q = (long)blockIdx.x * 256 + threadIdx.x; q < mn4; q += (long)gridDim.x * 256) {
        const long idx = q << 2;
        f32x4 s = {0.f, 0.f, 0.f, 0.f};
        int k = 0;
        for (; k + 8 <= p.splits; k += 8) {
            f32x4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const f32x4*>(S + (long)(k + u) * mn + idx);
#pragma unroll
            for (int u = 0; u < 8; ++u) s += v[u];
        }
        for (; k < p.splits; ++k) s += *reinterpret_cast<const f32x4*>(S + (long)k * mn + idx);
        const int m = (int)(idx / p.N), n = (int)(idx - (long)m * p.N);
        const f32x4 bv = p.bias ? *reinterpret_cast<const f32x4*>(p.bias + n) : f32x4{0.f, 0.f, 0.f, 0.f};
        f32x4 gt = {1.f, 1.f, 1.f, 1.f};
        if (p.gate) gt = *reinterpret_cast<const f32x4*>(p.gate + (long)batch * p.sG + (long)m * p.ldg + n);
        f32x4 rv = {0.f, 0.f, 0.f, 0.f};
        if (R) rv = *reinterpret_cast<const f32x4*>(R + (long)m * p.ldr + n);
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float v = apply_act(p.alpha * s[e] + bv[e], p.act);
            const float gsel = p.gate ? (gt[e] > 0.f ? p.gate_scale : 0.f) : 1.f;
            if (!p.gate_after) v *= gsel;
            if (R) v += rv[e];
            if (p.gate_after) v *= gsel;
            o[e] = v;
        }
        *reinterpret_cast<f32x4*>(C + (long)m * p.ldc + n) = o;
    }
}


// ---------------------------------------------------------------------------------------------------------------------
// Latency path for the SMALL GEMMs (the F = 167 encoder: M = 512, N, K in {167, 501, 512, 2048}).
// Those launches are not bound by MFMA issue or HBM but by the serial chain inside one work-group of the tiled
// kernel above: global load -> LDS store -> barrier -> LDS read -> 8 dependent MFMAs, 11 times for K = 167, ~1750
// cycles per 16-deep stage with one wave per SIMD and 24 of 256 CUs busy (cycle stamps: tools/micro/).  Here every
// wave fetches its operands STRAIGHT into the MFMA register layout -- no LDS staging, no barrier in the K loop --
// and the output is cut into 16x16 (or 32x32) wave tiles so that hundreds of waves share the work:
//   * v_mfma_f32_16x16x4_f32 (lane l holds A[i = l & 15][k = l >> 4], B[k = l >> 4][j = l & 15]; 32-cycle issue),
//     one 16-deep K chunk = 4 MFMAs per accumulator;
//   * a k-contiguous operand ([M][K] / [N][K]) is read with ONE 16-byte load per lane and chunk: lane (i, kq) takes
//     k = 16c + 4kq .. +3 of its row and feeds element j of that quad to MFMA j.  MFMA j therefore sums
//     k = 16c + 4kq + j over kq -- a permutation of K, applied to both operands alike, which a dot product does not
//     see.  gfx950 global loads need only 4-byte alignment, so the prime leading dimension 167 costs nothing;
//   * a k-major operand ([K][M] / [K][N]) is read as T consecutive columns of row k (T = sub-tiles per wave), which
//     permutes the wave's output columns instead (column = base + T * (l & 15) + u);
//   * a 4-deep register ring keeps 4 chunks of loads in flight per wave;
//   * the LAST chunk (K tail, and the last k row of a k-major operand, where a vector may overrun the buffer) takes
//     a per-element path with clamped addresses and zero fill.  Rows / columns past M / N are clamped to valid
//     addresses and never stored: a column of D only depends on its own column of B, so garbage stays there;
//   * deep K is split over the `ks` waves of the work-group (chunk c -> wave c % ks) and summed through LDS in a
//     fixed order by wave 0 -- one launch, no slab traffic, bit-reproducible.

struct DirectParams {
    const float* A; const float* B; float* C;
    const float* bias; const float* R;
    int M, N, K;
    int lda, ldb, ldc, ldr;
    long sA, sB, sC, sR;
    float alpha;
    int act;
    const float* gate; int ldg; long sG; float gate_scale; int gate_after;
    int wsm, wsn, ks;        // waves of a work-group: wsm x wsn output tiles, each computed by ks K-slices
    int batch, gx, gy;       // grid extent of THIS problem (a grouped launch covers the largest)
    float* asum;             // LAYOUT 2 only: asum[m] = sum_k A[k][m] through a virtual all-ones column N of B
    float drop_p, drop_inv_keep; unsigned long long drop_seed; const unsigned long long* seed_base;      // output dropout (after act)
};

template <int T>
__device__ __forceinline__ void ldv(const float* p, float (&out)[T]) {
    if constexpr (T == 1) out[0] = *p;
    else if constexpr (T == 2) { f32x2u v = *reinterpret_cast<const f32x2u*>(p); out[0] = v[0]; out[1] = v[1]; }
    else { f32x4u v = *reinterpret_cast<const f32x4u*>(p); out[0] = v[0]; out[1] = v[1]; out[2] = v[2]; out[3] = v[3]; }
}

// One operand of a wave: T sub-tiles of 16 rows (k-contiguous storage) or 16 T-wide column groups (k-major storage).
template <int T, bool KMAJ>
struct DirectOperand {
    const float* X;          // matrix base (batch applied)
    const float* ptr[KMAJ ? 1 : T];
    int ld, extent, base, q, kq;

    __device__ __forceinline__ void init(const float* X_, int ld_, int extent_, int base_, int lane) {
        X = X_; ld = ld_; extent = extent_; base = base_; q = lane & 15; kq = lane >> 4;
        if constexpr (KMAJ) {
            const int col = base + T * q;
            ptr[0] = X + (col < extent ? col : 0) + (long)(4 * kq) * ld;
        } else {
#pragma unroll
            for (int u = 0; u < T; ++u) ptr[u] = X + (long)min(base + 16 * u + q, extent - 1) * ld + 4 * kq;
        }
    }
    // full chunk c (all 16 k inside K, and not the last k row): r[j][u]
    __device__ __forceinline__ void fetch(int c, float (&r)[4][T]) const {
        if constexpr (KMAJ) {
#pragma unroll
            for (int j = 0; j < 4; ++j) ldv<T>(ptr[0] + (long)(16 * c + j) * ld, r[j]);
        } else {
#pragma unroll
            for (int u = 0; u < T; ++u) {
                f32x4u v = *reinterpret_cast<const f32x4u*>(ptr[u] + 16 * c);
#pragma unroll
                for (int j = 0; j < 4; ++j) r[j][u] = v[j];
            }
        }
    }
    // last chunk: element-wise, clamped addresses, zero fill
    __device__ __forceinline__ void fetch_tail(int c, int K, float (&r)[4][T]) const {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int k = 16 * c + 4 * kq + j;
#pragma unroll
            for (int u = 0; u < T; ++u) {
                if constexpr (KMAJ) {
                    const int col = base + T * q + u;
                    const bool ok = k < K && col < extent;
                    const float v = X[ok ? (long)k * ld + col : 0];
                    r[j][u] = ok ? v : 0.f;
                } else {
                    const bool ok = k < K;
                    const float v = (ptr[u] - 4 * kq)[ok ? k : 0];
                    r[j][u] = ok ? v : 0.f;
                }
            }
        }
    }
    // matrix index of (sub-tile u, MFMA index i in 0..15)
    __device__ __forceinline__ int index(int u, int i) const { return KMAJ ? base + T * i + u : base + 16 * u + i; }
    // k-major operand with a virtual all-ones column at index `extent`: overwrite what the clamped load fetched for it
    __device__ __forceinline__ void ones(float (&r)[4][T], int c, int K) const {
        if constexpr (KMAJ) {
#pragma unroll
            for (int u = 0; u < T; ++u)
                if (base + T * q + u == extent) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) r[j][u] = (16 * c + 4 * kq + j) < K ? 1.f : 0.f;
                }
        }
    }
};

template <int LAYOUT, int TM, int TN>
__device__ __forceinline__ void gemm_direct_body(const DirectParams& p, int batch, float* red) {
    constexpr bool A_KMAJ = (LAYOUT == 2), B_KMAJ = (LAYOUT != 0);
#ifndef BBBP_DIRECT_DEPTH
#define BBBP_DIRECT_DEPTH 4
#endif
    constexpr int D = BBBP_DIRECT_DEPTH;         // chunks in flight per wave (A/B builds: -DBBBP_DIRECT_DEPTH=n, tools/build_variant.sh)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int ks = wave % p.ks, sp = wave / p.ks;
    const int wm = sp / p.wsn, wn = sp % p.wsn;
    DirectOperand<TM, A_KMAJ> opa;
    DirectOperand<TN, B_KMAJ> opb;
    opa.init(p.A + (long)batch * p.sA, p.lda, p.M, (blockIdx.y * p.wsm + wm) * 16 * TM, lane);
    opb.init(p.B + (long)batch * p.sB, p.ldb, p.N, (blockIdx.x * p.wsn + wn) * 16 * TN, lane);

    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    auto mfma_chunk = [&](const float (&a)[4][TM], const float (&b)[4][TN]) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int um = 0; um < TM; ++um)
#pragma unroll
                for (int un = 0; un < TN; ++un)
                    acc[um][un] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j][um], b[j][un], acc[um][un], 0, 0, 0);
    };

    const int nch = (p.K + 15) / 16;
    // the wave whose columns include the virtual ones column (wave-uniform; never taken by the other layouts)
    bool has_ones = false;
    if constexpr (LAYOUT == 2) has_ones = p.asum != nullptr && opb.base <= p.N && p.N < opb.base + 16 * TN;
    // this wave's chunks: c = ks + i * p.ks; all but the globally last one take the vector path
    const int nfast = (nch - 1 > ks) ? (nch - 1 - ks + p.ks - 1) / p.ks : 0;
    if (nfast > 0) {
        float ra[D][4][TM], rb[D][4][TN];
#pragma unroll
        for (int d = 0; d < D; ++d) {
            const int c = ks + min(d, nfast - 1) * p.ks;
            opa.fetch(c, ra[d]); opb.fetch(c, rb[d]);
        }
        for (int i0 = 0; i0 < nfast; i0 += D) {
#pragma unroll
            for (int d = 0; d < D; ++d) {
                if (i0 + d < nfast) {
                    if (has_ones) opb.ones(rb[d], 0, p.K);          // a full chunk: every k < K
                    mfma_chunk(ra[d], rb[d]);
                }
                const int c = ks + min(i0 + d + D, nfast - 1) * p.ks;
                opa.fetch(c, ra[d]); opb.fetch(c, rb[d]);
            }
        }
    }
    if (nch > 0 && (nch - 1) % p.ks == ks) {
        float ta[4][TM], tb[4][TN];
        opa.fetch_tail(nch - 1, p.K, ta); opb.fetch_tail(nch - 1, p.K, tb);
        if (has_ones) opb.ones(tb, nch - 1, p.K);
        mfma_chunk(ta, tb);
    }

    if (p.ks > 1) {          // K slices -> LDS, summed in slice order by slice 0
        float* mine = red + (long)wave * (TM * TN * 4 * 64) + lane;
        if (ks != 0) {
#pragma unroll
            for (int um = 0; um < TM; ++um)
#pragma unroll
                for (int un = 0; un < TN; ++un)
#pragma unroll
                    for (int r = 0; r < 4; ++r) mine[((um * TN + un) * 4 + r) * 64] = acc[um][un][r];
        }
        __syncthreads();
        if (ks != 0) return;
        for (int s = 1; s < p.ks; ++s) {
            const float* other = mine + (long)s * (TM * TN * 4 * 64);
#pragma unroll
            for (int um = 0; um < TM; ++um)
#pragma unroll
                for (int un = 0; un < TN; ++un)
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[um][un][r] += other[((um * TN + un) * 4 + r) * 64];
        }
    }

    float* C = p.C + (long)batch * p.sC;
    const float* R = p.R ? p.R + (long)batch * p.sR : nullptr;
    const int q = lane & 15, kq = lane >> 4;
#pragma unroll
    for (int un = 0; un < TN; ++un) {
        const int n = opb.index(un, q);
        const float bv = (p.bias && n < p.N) ? p.bias[n] : 0.f;
#pragma unroll
        for (int um = 0; um < TM; ++um)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = opa.index(um, 4 * kq + r);
                if constexpr (LAYOUT == 2) {
                    if (p.asum && n == p.N && m < p.M) p.asum[(long)batch * p.M + m] = acc[um][un][r];
                }
                if (m < p.M && n < p.N) {
                    float v = apply_act(p.alpha * acc[um][un][r] + bv, p.act);
                    if (p.drop_p > 0.f)
                        v *= dropout_scale(effective_seed(p.drop_seed, p.seed_base), ((uint64_t)batch * p.M + m) * p.N + n, p.drop_p, p.drop_inv_keep);
                    const float gsel = p.gate ? (p.gate[(long)batch * p.sG + (long)m * p.ldg + n] > 0.f ? p.gate_scale : 0.f) : 1.f;
                    if (!p.gate_after) v *= gsel;
                    if (R) v += R[(long)m * p.ldr + n];
                    if (p.gate_after) v *= gsel;
                    C[(long)m * p.ldc + n] = v;
                }
            }
    }
}

template <int LAYOUT, int TM, int TN>
__global__ __launch_bounds__(1024) void gemm_direct_kernel(DirectParams p) {
    BBBP_HIGH_PRIO();
    extern __shared__ float red[];               // [wave][TM * TN * 4][64] when ks > 1
    gemm_direct_body<LAYOUT, TM, TN>(p, blockIdx.z, red);
}

// Linear + dropout + residual + LayerNorm as ONE launch for a narrow output (N <= 256: the encoder's out_proj -> norm1 and linear2 ->
// norm2 at F = 167, R:75-78): a work-group owns 16 rows and ALL their columns, one 16 x 16 wave tile per wave (same operand path as
// gemm_direct_kernel), so the row statistics are two LDS exchanges away from the accumulators.
//   z = dropout(x W^T + b) + residual   (written: backward needs it)      y = (z - mean) * rstd * gamma + beta
// The dropout draws element m * N + n of `seed`, exactly what bbbp_layernorm_fwd draws for the same tensor.
struct LnFuse {
    const float* gamma; const float* beta;
    float* y; int ldy;
    float* mean; float* rstd;
    float eps;
};

__global__ __launch_bounds__(1024, 2) void gemm_direct_ln_kernel(DirectParams p, LnFuse f) {
    BBBP_HIGH_PRIO();
    __shared__ float part[2][16][16];                     // [pass][wave][row]
    constexpr int D = 4;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const int q = lane & 15, kq = lane >> 4;
    DirectOperand<1, false> opa, opb;
    opa.init(p.A, p.lda, p.M, blockIdx.y * 16, lane);
    opb.init(p.B, p.ldb, p.N, wave * 16, lane);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    auto mfma_chunk = [&](const float (&a)[4][1], const float (&b)[4][1]) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j][0], b[j][0], acc, 0, 0, 0);
    };
    const int nch = (p.K + 15) / 16;
    const int nfast = nch - 1;
    if (nfast > 0) {
        float ra[D][4][1], rb[D][4][1];
#pragma unroll
        for (int d = 0; d < D; ++d) { const int c = min(d, nfast - 1); opa.fetch(c, ra[d]); opb.fetch(c, rb[d]); }
        for (int i0 = 0; i0 < nfast; i0 += D) {
#pragma unroll
            for (int d = 0; d < D; ++d) {
                if (i0 + d < nfast) mfma_chunk(ra[d], rb[d]);
                const int c = min(i0 + d + D, nfast - 1);
                opa.fetch(c, ra[d]); opb.fetch(c, rb[d]);
            }
        }
    }
    if (nch > 0) {
        float ta[4][1], tb[4][1];
        opa.fetch_tail(nch - 1, p.K, ta); opb.fetch_tail(nch - 1, p.K, tb);
        mfma_chunk(ta, tb);
    }
    // lane (q, kq): column n, rows m0 + 4 kq + r
    const int n = wave * 16 + q, m0 = blockIdx.y * 16 + 4 * kq;
    const bool coln = n < p.N;
    const float bv = (p.bias && coln) ? p.bias[n] : 0.f;
    const uint64_t seed = effective_seed(p.drop_seed, p.seed_base);
    float z[4], s[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int m = min(m0 + r, p.M - 1);
        float v = p.alpha * acc[r] + bv;
        if (p.drop_p > 0.f) v *= dropout_scale(seed, (uint64_t)m * p.N + min(n, p.N - 1), p.drop_p, p.drop_inv_keep);
        if (p.R && coln) v += p.R[(long)m * p.ldr + n];
        z[r] = coln ? v : 0.f;
        if (coln && m0 + r < p.M) p.C[(long)m * p.ldc + n] = v;
        s[r] = z[r];
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) { s[r] += __shfl_xor(s[r], 1); s[r] += __shfl_xor(s[r], 2); s[r] += __shfl_xor(s[r], 4); s[r] += __shfl_xor(s[r], 8); }
    if (q == 0) {
#pragma unroll
        for (int r = 0; r < 4; ++r) part[0][wave][4 * kq + r] = s[r];
    }
    __syncthreads();
    float mean[4], d2[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        float t = 0.f;
        for (int w = 0; w < nw; ++w) t += part[0][w][4 * kq + r];
        mean[r] = t / p.N;
        const float d = coln ? z[r] - mean[r] : 0.f;
        d2[r] = d * d;
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) { d2[r] += __shfl_xor(d2[r], 1); d2[r] += __shfl_xor(d2[r], 2); d2[r] += __shfl_xor(d2[r], 4); d2[r] += __shfl_xor(d2[r], 8); }
    if (q == 0) {
#pragma unroll
        for (int r = 0; r < 4; ++r) part[1][wave][4 * kq + r] = d2[r];
    }
    __syncthreads();
    const float gam = coln ? f.gamma[n] : 0.f, bet = coln ? f.beta[n] : 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        float t = 0.f;
        for (int w = 0; w < nw; ++w) t += part[1][w][4 * kq + r];
        const float rstd = rsqrtf(t / p.N + f.eps);
        const int m = m0 + r;
        if (m < p.M) {
            if (coln) f.y[(long)m * f.ldy + n] = (z[r] - mean[r]) * rstd * gam + bet;
            if (wave == 0 && q == 0) { f.mean[m] = mean[r]; f.rstd[m] = rstd; }
        }
    }
}

// LayerNorm ABSORBED by the Linear that consumes it (round 4): out = act(LayerNorm(z) W^T + bias) and y = LayerNorm(z) (+ mean, rstd) in ONE
// launch -- the encoder's norm1 -> linear1, norm2 -> the next layer's in_proj and the last norm2 -> fingerprint_fc (R:75-82): twelve
// LayerNorm launches leave the encoder's forward chain, where every launch costs 8-15 us whatever its size.  Algebra, per output element:
//     LayerNorm(z)_k = (z_k - mu) rstd gamma_k + beta_k
//     out[m][n] = rstd_m ( sum_k (z[m][k] - x0_m) gamma_k W[n][k]  -  (mu_m - x0_m) c_n ) + d_n + bias_n,
//     c_n = sum_k gamma_k W[n][k],   d_n = sum_k beta_k W[n][k],   x0_m = z[m][0] (a pivot: sums of z - x0 do not cancel like sums of z).
// A wave of the small-product kernel walks ALL of K for its 16 rows and 16 columns (K <= 192: one K slice), so everything above is in its
// own operand stream: the row sums (sum and sum of squares of z - x0: mean and variance), c_n and d_n from the weight rows it loads anyway,
// gamma / beta from 1.5 KB of LDS; the MFMAs run on (z - x0) gamma, and the row statistics reach the accumulator layout through two
// ds_bpermutes per accumulator register.  No prepared weights, no extra pass, no barrier in the K loop.
// y, mean and rstd (the residual of the next block, linear1's weight gradient and the LayerNorm backward need them) are written by ONE
// extra column of work-groups of the same launch that run the plain row kernel (rowops.hip: layernorm_fwd_kernel without dropout and
// residual -- those moved into the epilogue of the GEMM that PRODUCES z): same statistics as the stand-alone launch, bit for bit.
struct LnaParams {
    const float* Z; int ldz;                 // [M][K] pre-norm rows
    const float* gamma; const float* beta; float eps;
    const float* W; int ldw; const float* bias;      // [N][K], [N]
    float* C; int ldc;
    int M, N, K, act;
    float drop_p, drop_inv_keep; unsigned long long drop_seed; const unsigned long long* seed_base;
    float* Y; int ldy; float* mean; float* rstd;
    int gx;                                  // column blocks of the product; blockIdx.x == gx: the LayerNorm writers
};
constexpr int LNA_MAX_K = 192;

__global__ __launch_bounds__(256) void gemm_direct_lna_kernel(LnaParams p) {
    BBBP_HIGH_PRIO();
    __shared__ __attribute__((aligned(16))) float gb[2][LNA_MAX_K];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if ((int)blockIdx.x == p.gx) {
        // ---- LayerNorm writers: 32 rows per work-group, one wave per row at a time, the row in registers (three elements per lane) ----
#pragma unroll 1
        for (int i = 0; i < 8; ++i) {
            const int row = blockIdx.y * 32 + wave * 8 + i;
            if (row >= p.M) break;
            const float* zr = p.Z + (long)row * p.ldz;
            float v[3]; float s = 0.f;
#pragma unroll
            for (int e = 0; e < 3; ++e) { const int c = lane + 64 * e; v[e] = c < p.K ? zr[c] : 0.f; s += v[e]; }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
            const float mean = s / p.K;
            float q = 0.f;
#pragma unroll
            for (int e = 0; e < 3; ++e) { const int c = lane + 64 * e; const float d = c < p.K ? v[e] - mean : 0.f; q += d * d; }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o);
            const float rstd = rsqrtf(q / p.K + p.eps);
            float* yr = p.Y + (long)row * p.ldy;
#pragma unroll
            for (int e = 0; e < 3; ++e) { const int c = lane + 64 * e; if (c < p.K) yr[c] = (v[e] - mean) * rstd * p.gamma[c] + p.beta[c]; }
            if (lane == 0) { p.mean[row] = mean; p.rstd[row] = rstd; }
        }
        return;
    }
    constexpr int D = 3;                                          // chunks in flight (4 spill at the 64-register cap that keeps 8 waves per SIMD)
    const int q = lane & 15, kq = lane >> 4;
    const int wm = wave >> 1, wn = wave & 1;
    DirectOperand<1, false> opa, opb;
    opa.init(p.Z, p.ldz, p.M, (blockIdx.y * 2 + wm) * 16, lane);
    opb.init(p.W, p.ldw, p.N, (blockIdx.x * 2 + wn) * 16, lane);
    const int nch = (p.K + 15) / 16, nfast = nch - 1;
    // the first chunks and the row pivot are in flight before the LDS fill's barrier
    float ra[D][4][1], rb[D][4][1];
    if (nfast > 0) {
#pragma unroll
        for (int d = 0; d < D; ++d) { const int c = min(d, nfast - 1); opa.fetch(c, ra[d]); opb.fetch(c, rb[d]); }
    }
    const float x0 = *(opa.ptr[0] - 4 * kq);
    for (int i = threadIdx.x; i < LNA_MAX_K; i += 256) { gb[0][i] = i < p.K ? p.gamma[i] : 0.f; gb[1][i] = i < p.K ? p.beta[i] : 0.f; }
    __syncthreads();
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    float s1 = 0.f, s2 = 0.f, cn = 0.f, dn = 0.f;
    auto chunk = [&](const float (&a)[4][1], const float (&b)[4][1], int c, bool tail) __attribute__((always_inline)) {
        const f32x4 g = *reinterpret_cast<const f32x4*>(&gb[0][16 * c + 4 * kq]);
        const f32x4 bt = *reinterpret_cast<const f32x4*>(&gb[1][16 * c + 4 * kq]);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float as = a[j][0] - x0;
            if (tail) as = (16 * c + 4 * kq + j < p.K) ? as : 0.f;          // the zero-filled tail of K is not part of the row
            s1 += as; s2 = fmaf(as, as, s2);
            cn = fmaf(b[j][0], g[j], cn); dn = fmaf(b[j][0], bt[j], dn);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(as * g[j], b[j][0], acc, 0, 0, 0);
        }
    };
    if (nfast > 0) {
        for (int i0 = 0; i0 < nfast; i0 += D) {
#pragma unroll
            for (int d = 0; d < D; ++d) {
                if (i0 + d < nfast) chunk(ra[d], rb[d], i0 + d, false);
                const int c = min(i0 + d + D, nfast - 1);
                opa.fetch(c, ra[d]); opb.fetch(c, rb[d]);
            }
        }
    }
    {
        float ta[4][1], tb[4][1];
        opa.fetch_tail(nch - 1, p.K, ta); opb.fetch_tail(nch - 1, p.K, tb);
        chunk(ta, tb, nch - 1, true);
    }
    // lanes (q, 0..3) hold the four K quarters of row q (A side) and of column q (B side)
    s1 += __shfl_xor(s1, 16); s1 += __shfl_xor(s1, 32);
    s2 += __shfl_xor(s2, 16); s2 += __shfl_xor(s2, 32);
    cn += __shfl_xor(cn, 16); cn += __shfl_xor(cn, 32);
    dn += __shfl_xor(dn, 16); dn += __shfl_xor(dn, 32);
    const float mus = s1 / p.K;                                   // mean - pivot
    const float rstd = rsqrtf(fmaxf(s2 / p.K - mus * mus, 0.f) + p.eps);
    const int n = opb.index(0, q);
    const float bv = (p.bias && n < p.N) ? p.bias[n] : 0.f;
    const uint64_t seed = effective_seed(p.drop_seed, p.seed_base);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int m = opa.index(0, 4 * kq + r);
        // accumulator register r of lane (q, kq) is row 4 kq + r, column q: that row's statistics sit in lane 4 kq + r
        const float mu_r = __shfl(mus, 4 * kq + r), rs_r = __shfl(rstd, 4 * kq + r);
        if (m < p.M && n < p.N) {
            float v = apply_act(rs_r * (acc[r] - mu_r * cn) + dn + bv, p.act);
            if (p.drop_p > 0.f) v *= dropout_scale(seed, (uint64_t)m * p.N + n, p.drop_p, p.drop_inv_keep);
            p.C[(long)m * p.ldc + n] = v;
        }
    }
}

// Two independent products in one launch (dV | dP and dQ | dK of the attention backward, which become ready together):
// the chain of small launches is bound by per-launch latency, not by work, so halving the launches halves the time.
// blockIdx.z < p0.batch -> problem 0, else problem 1; work-groups outside a problem's own grid exit at once.
template <int L0, int T0, int L1, int T1>
__global__ __launch_bounds__(256) void gemm_direct_pair_kernel(DirectParams p0, DirectParams p1) {
    BBBP_HIGH_PRIO();
    extern __shared__ float red[];
    if ((int)blockIdx.z < p0.batch) {
        if ((int)blockIdx.x < p0.gx && (int)blockIdx.y < p0.gy) gemm_direct_body<L0, T0, T0>(p0, blockIdx.z, red);
    } else {
        if ((int)blockIdx.x < p1.gx && (int)blockIdx.y < p1.gy) gemm_direct_body<L1, T1, T1>(p1, blockIdx.z - p0.batch, red);
    }
}

struct DirectPlan { bool use; int t, wsm, wsn, ks; };

// The direct path serves launches that cannot fill the chip with 64x64 tiles anyway; big GEMMs keep the LDS tiling.
DirectPlan direct_plan(int M, int N, int K, int batch) {
    static const int enabled = [] { const char* e = getenv("BBBP_GEMM_DIRECT"); return e ? atoi(e) : 1; }();
    DirectPlan d{false, 1, 1, 1, 1};
    const double flops = 2.0 * M * N * (double)K * batch;
    if (!enabled || flops > 1.2e9 || K > 8192) return d;
    d.use = true;
    const int nch = cdiv(K, 16);
    // work-groups stay at 4 waves (one per SIMD, <= 64 VGPRs for the 16x16 variant): a 16-wave group needs 256 free
    // VGPRs on every SIMD of one CU and cannot start while the persistent conv work-groups of the other stream hold
    // theirs (measured: the conv beside it slowed from 0.72 to 1.25 ms and the encoder gained nothing).  Whole training
    // step, B = 512 (tools/exp_step.py): K slices capped at 1 / 2 / 4 -> 3.95 / 3.75 / 3.78 ms; LDS-tiled path 4.25 ms.
    static const int max_ks = [] { const char* e = getenv("BBBP_GEMM_DIRECT_KS"); return e ? atoi(e) : 2; }();
    d.ks = nch <= 12 ? 1 : (cdiv(nch, 8) < max_ks ? cdiv(nch, 8) : max_ks);
    if (d.ks == 3) d.ks = 4;
    const long wt = (long)cdiv(M, 16) * cdiv(N, 16) * batch;
    // 32x32 wave tiles (t = 2) need ~100 VGPRs: beside conv_wgrad32 (2 x 224 VGPRs per SIMD) such a wave cannot be
    // placed until the conv kernel ends -- rocprofv3 showed the dhff GEMM of every layer waiting up to 0.65 ms.  The
    // 16x16 variant fits the 64 VGPRs that are left, so it serves everything up to 4096 wave tiles (all of B = 512,
    // F = 167: whole step 3.81 -> 3.75 ms); larger outputs (B = 4096 screening) take 32x32 tiles for the operand reuse.
    static const int max_t = [] { const char* e = getenv("BBBP_GEMM_DIRECT_T"); return e ? atoi(e) : 2; }();
    d.t = (wt * d.ks <= 4096 || max_t < 2) ? 1 : 2;
    // many heads with a tiny head dimension (F = 2048: 256 heads of 8) are hundreds of thousands of nearly empty wave
    // tiles: those stay on the LDS-tiled kernel
    const long waves = (long)cdiv(M, 16 * d.t) * cdiv(N, 16 * d.t) * batch * d.ks;
    const int mind = M < N ? (M < K ? M : K) : (N < K ? N : K);
    if (waves > 16384 || (batch >= 8 && mind < 16)) d.use = false;
    if (d.ks >= 4) { d.wsm = 1; d.wsn = 1; }
    else if (d.ks >= 2) { d.wsm = 2; d.wsn = 1; }
    else { d.wsm = 2; d.wsn = 2; }
    if ((enabled & 2) && d.ks > 1) d.use = false;      // A/B knobs: BBBP_GEMM_DIRECT=0 off, 3 no K slices, 5 no 32x32 wave tiles
    if ((enabled & 4) && d.t > 1) d.use = false;
    return d;
}

inline size_t direct_lds(const DirectParams& p, int t) {
    const int waves = p.wsm * p.wsn * p.ks;
    size_t lds = p.ks > 1 ? (size_t)waves * t * t * 4 * 64 * sizeof(float) : 0;
    return lds < g_bbbp_small_lds_pad ? g_bbbp_small_lds_pad : lds;
}

template <int LAYOUT, int T>
void launch_direct_one(const DirectParams& p, hipStream_t st) {
    const int waves = p.wsm * p.wsn * p.ks;
    hipLaunchKernelGGL((gemm_direct_kernel<LAYOUT, T, T>), dim3(p.gx, p.gy, p.batch), dim3(64 * waves), direct_lds(p, T), st, p);
}

void launch_direct(const DirectParams& p, int layout, int t, hipStream_t st) {
    if (t == 1) {
        if (layout == 0) launch_direct_one<0, 1>(p, st);
        else if (layout == 1) launch_direct_one<1, 1>(p, st);
        else launch_direct_one<2, 1>(p, st);
    } else {
        if (layout == 0) launch_direct_one<0, 2>(p, st);
        else if (layout == 1) launch_direct_one<1, 2>(p, st);
        else launch_direct_one<2, 2>(p, st);
    }
}

template <int L0, int T0, int L1, int T1>
void launch_pair_one(const DirectParams& a, const DirectParams& b, hipStream_t st) {
    const size_t la = direct_lds(a, T0), lb = direct_lds(b, T1);
    dim3 grid(a.gx > b.gx ? a.gx : b.gx, a.gy > b.gy ? a.gy : b.gy, a.batch + b.batch);
    hipLaunchKernelGGL((gemm_direct_pair_kernel<L0, T0, L1, T1>), grid, dim3(256), la > lb ? la : lb, st, a, b);
}

// the pairs the engine issues; anything else runs as two launches
bool launch_pair(const DirectParams& a, int la, int ta, const DirectParams& b, int lb, int tb, hipStream_t st) {
    if (a.wsm * a.wsn * a.ks != 4 || b.wsm * b.wsn * b.ks != 4) return false;
    if ((long)a.batch + b.batch > 65535) return false;
    if (ta == 1 && tb == 1) {
        if (la == 2 && lb == 0) { launch_pair_one<2, 1, 0, 1>(a, b, st); return true; }      // dV = Pd^T dO | dPd = dO V^T
        if (la == 1 && lb == 2) { launch_pair_one<1, 1, 2, 1>(a, b, st); return true; }      // dQ = dS K   | dK = dS^T Q
        if (la == 2 && lb == 2) { launch_pair_one<2, 1, 2, 1>(a, b, st); return true; }      // [dWq; dWk] = dQK^T x | dW' = dVW^T x (fold.hip)
    }
    return false;
}

template <int BM, int BN, int LAYOUT, bool VEC, int BKT = 0>
void launch_one(const GemmParams& p, dim3 grid, hipStream_t st) {
    constexpr int BK = BKT ? BKT : bk_of(BM);
    constexpr size_t lds = (size_t)2 * BK * ((BM + (LAYOUT == 2 ? 4 : 1)) + (BN + (LAYOUT != 0 ? 4 : 1))) * sizeof(float);
    // a refused opt-in makes the launch itself fail, which BBBP_CHECK_LAUNCH reports
    if (lds > 64 * 1024) (void)bbbp_ensure_dyn_lds(reinterpret_cast<const void*>(gemm_f32_kernel<BM, BN, LAYOUT, VEC, BKT>), lds);
    hipLaunchKernelGGL((gemm_f32_kernel<BM, BN, LAYOUT, VEC, BKT>), grid, dim3(256), lds > g_bbbp_small_lds_pad ? lds : g_bbbp_small_lds_pad, st, p);
}

template <int BM, int BN>
void launch_tile(const GemmParams& p, int layout, dim3 grid, hipStream_t st) {
    const bool vec = p.vecA && p.vecB;
    if (BM == 128 && p.short_k && vec) {           // short K over a large output: four work-groups per CU (see the kernel)
        if (layout == 0) launch_one<128, 128, 0, true, 16>(p, grid, st);
        else if (layout == 1) launch_one<128, 128, 1, true, 16>(p, grid, st);
        else launch_one<128, 128, 2, true, 16>(p, grid, st);
        return;
    }
    if (layout == 0) { if (vec) launch_one<BM, BN, 0, true>(p, grid, st); else launch_one<BM, BN, 0, false>(p, grid, st); }
    else if (layout == 1) { if (vec) launch_one<BM, BN, 1, true>(p, grid, st); else launch_one<BM, BN, 1, false>(p, grid, st); }
    else { if (vec) launch_one<BM, BN, 2, true>(p, grid, st); else launch_one<BM, BN, 2, false>(p, grid, st); }
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

int g_gemm_b3 = -1;
int gemm_b3_on() {
    if (g_gemm_b3 < 0) { const char* e = getenv("BBBP_GEMM_SPLIT_BF16"); g_gemm_b3 = e ? (atoi(e) != 0) : 1; }
    return g_gemm_b3;
}
// Arrival counters of the split-bf16 kernel's in-kernel split-K reduction: one zeroed region per (device, stream) -- launches on one
// stream run one after another and every launch leaves its counters at zero, launches on different streams never share a region.
constexpr int ARRIVAL_REGION = 8192, ARRIVAL_REGIONS = 32;
int g_gemm_fold_reduce = -1;
int gemm_fold_reduce_on() {
    if (g_gemm_fold_reduce < 0) { const char* e = getenv("BBBP_GEMM_FOLD_REDUCE"); g_gemm_fold_reduce = e ? (atoi(e) != 0) : 0; }      // default off: measured slower, see DESIGN.md section 3
    return g_gemm_fold_reduce;
}
unsigned* arrival_counters(hipStream_t st, long tiles) {
    if (!gemm_fold_reduce_on() || tiles > ARRIVAL_REGION) return nullptr;
    static std::mutex mu;
    static unsigned* base[64] = {};
    static std::unordered_map<hipStream_t, int> region[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
    std::lock_guard<std::mutex> lock(mu);
    if (!base[dev]) {
        hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
        (void)hipStreamIsCapturing(st, &cap);
        if (cap != hipStreamCaptureStatusNone) return nullptr;          // no allocation inside a capture: this launch keeps the reduce kernel
        unsigned* ptr = nullptr;
        const size_t bytes = (size_t)ARRIVAL_REGION * ARRIVAL_REGIONS * sizeof(unsigned);
        if (hipMalloc(&ptr, bytes) != hipSuccess) return nullptr;
        if (hipMemset(ptr, 0, bytes) != hipSuccess || hipDeviceSynchronize() != hipSuccess) { (void)hipFree(ptr); return nullptr; }
        base[dev] = ptr;
    }
    auto it = region[dev].find(st);
    if (it == region[dev].end()) {
        if ((int)region[dev].size() >= ARRIVAL_REGIONS) return nullptr;
        it = region[dev].emplace(st, (int)region[dev].size()).first;
    }
    return base[dev] + (size_t)it->second * ARRIVAL_REGION;
}

// the split-bf16 form serves 128 x 128 plans whose operands can be read in aligned-extent quads
bool b3_eligible(const GemmParams& p, int layout) {
    if (!gemm_b3_on() || p.K < 32) return false;          // any K >= 32: a partial last stage is handled by B3Loader::load_tail
    const bool a_ok = layout != 2 || (p.M % 4 == 0 && p.M >= 4);
    const bool b_ok = layout == 0 || (p.N % 4 == 0 && p.N >= 4);
    return a_ok && b_ok;
}
// the 64 x 64 split-bf16 tile: the 128-tile grid would leave CUs idle (or split K) while 64-tiles fill the chip, and K is deep enough to
// amortise the tile's prologue.  BBBP_GEMM_B3_SMALL=0 keeps the 128-tile plans.
bool b3_small_tile(int M, int N, int K, int batch) {
    static const int on = [] { const char* e = getenv("BBBP_GEMM_B3_SMALL"); return e ? atoi(e) : 1; }();
    if (!on || K < 512) return false;
    const long t128 = (long)cdiv(M, 128) * cdiv(N, 128) * batch, t64 = (long)cdiv(M, 64) * cdiv(N, 64) * batch;
    const int ncu = bbbp_num_cus();
    return t128 < ncu && t64 * 2 >= (long)ncu * 5;
}
template <int LAYOUT>
int launch_b3_one(const GemmParams& p, dim3 grid, hipStream_t st) {
    { int rc_ = bbbp_ensure_dyn_lds(reinterpret_cast<const void*>(gemm_b3_kernel<LAYOUT>), (size_t)B3_LDS); if (rc_) return rc_; }
    static const bool probe = [] { const char* e = getenv("BBBP_GEMM_B3_PROBE"); return e && atoi(e) != 0; }();
    if (probe) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_b3_probe_kernel<LAYOUT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)B3_LDS);
        hipLaunchKernelGGL((gemm_b3_probe_kernel<LAYOUT>), grid, dim3(256), B3_LDS, st, p);
    } else {
        hipLaunchKernelGGL((gemm_b3_kernel<LAYOUT>), grid, dim3(256), B3_LDS, st, p);
    }
    return BBBP_OK;
}

}  // namespace

// Split-K plan shared by the launcher and bbbp_gemm_workspace_bytes.
static void gemm_plan(int M, int N, int K, int batch, int* tile, int* splits, int* kchunk) {
    long t128 = (long)cdiv(M, 128) * cdiv(N, 128) * batch;
    long t64 = (long)cdiv(M, 64) * cdiv(N, 64) * batch;
    int ncu = bbbp_num_cus();
    *tile = (t128 >= (long)ncu * 3 / 4) ? 128 : 64;
    // Very deep K over a small output (the 65536-wide image FC forward): the 128x128 tile does 64 MFMAs per wave per
    // stage, enough to cover the global-load latency of the next stage, and split-K supplies the parallelism; the
    // 64x64 tile (8 MFMAs per stage) is latency-bound there (measured 36 vs ~90 TFLOP/s).
    if (K >= 8192 && M >= 128 && N >= 128) *tile = 128;
    // Deep K with a large, 128-divisible output (the F = 2048 encoder's FFN / projection GEMMs): same argument.
    {
        const long covered = (long)cdiv(M, 128) * 128 * (long)cdiv(N, 128) * 128;
        if (K >= 512 && M >= 256 && N >= 256 && covered * 100 <= (long)M * N * 115) *tile = 128;
    }
    // A deep K over many rows and one or two column tiles (linear2 of a 4096-row screening batch: N = 167, K = 2048): the 64 x 64 f32
    // plan runs it at 49 TFLOP/s (66 us); on the bf16 pipe the padded columns (167 -> 256) cost less than the pipe gains.
    // BBBP_GEMM_TALL_B3=0 keeps the old plan.
    {
        static const int tall = [] { const char* e = getenv("BBBP_GEMM_TALL_B3"); return e ? atoi(e) : 1; }();
        const long covered = (long)cdiv(M, 128) * 128 * (long)cdiv(N, 128) * 128;
        if (tall && K >= 1024 && M >= 1024 && N >= 128 && covered * 10 <= (long)M * N * 16) *tile = 128;
    }
    long tiles = (*tile == 128) ? t128 : t64;
    // Few output tiles and a deep K (weight gradients over the batch, the 65536-wide image FC, FFN2): these
    // launches are latency-bound at one work-group per tile, so spread K over ~2 work-groups per CU.
    int s = 1;
    // BBBP_GEMM_SPLIT_X10: target work-groups per CU x 10 when K is split (default 20 = two per CU)
    static const int split_x10 = [] { const char* e = getenv("BBBP_GEMM_SPLIT_X10"); return e ? atoi(e) : 20; }();
    // BBBP_GEMM_SPLIT_FULL: also split when the tiles already cover the chip once but not twice (one work-group per CU leaves
    // every barrier stall of its single wave per SIMD exposed)
    static const int split_full = [] { const char* e = getenv("BBBP_GEMM_SPLIT_FULL"); return e ? atoi(e) : 0; }();
    if ((tiles < ncu || (split_full && tiles < 2 * ncu && K >= 512)) && K >= 256) {
        s = (int)(((long)split_x10 * ncu / 10 + tiles - 1) / tiles);
        int maxs = K / 64;
        if (s > maxs) s = maxs;
        if (s < 1) s = 1;
    }
    const int BK = bk_of(*tile);
    int kc = cdiv(cdiv(K, s), BK) * BK;
    if (kc < BK) kc = BK;
    s = cdiv(K, kc);
    if (s < 1) s = 1;
    *splits = s;
    *kchunk = kc;
}

extern "C" int bbbp_gemm_folds_asum(int M, int N, int K, int batch) {
    if (M <= 0 || N <= 0 || K <= 0 || batch < 1) return 0;
    const DirectPlan dp = direct_plan(M, N, K, batch);
    return (dp.use && batch <= 65535 && cdiv(M, 16 * dp.t * dp.wsm) <= 65535) ? 1 : 0;
}

extern "C" size_t bbbp_gemm_workspace_bytes(int M, int N, int K, int batch) {
    if (M <= 0 || N <= 0 || direct_plan(M, N, K, batch).use) return 0;
    int tile, splits, kchunk;
    gemm_plan(M, N, K, batch, &tile, &splits, &kchunk);
    return splits > 1 ? (size_t)batch * splits * M * N * sizeof(float) : 0;
}

namespace {

int gemm_validate(const bbbp_gemm_desc& g) {
    BBBP_CHECK_ARG(g.M >= 0 && g.N >= 0 && g.K >= 0 && g.batch >= 1, "gemm: bad sizes M=%d N=%d K=%d batch=%d", g.M, g.N, g.K, g.batch);
    BBBP_CHECK_ARG(g.act >= 0 && g.act <= 2, "gemm: bad act %d", g.act);
    BBBP_CHECK_ARG(!(g.transA && g.transB), "gemm: layout TT (A^T B^T) is not on the hot path");
    if (g.M == 0 || g.N == 0) return BBBP_OK;
    BBBP_CHECK_ARG(g.A && g.B && g.C, "gemm: null operand");
    BBBP_CHECK_ARG(g.lda >= (g.transA ? g.M : g.K), "gemm: lda %d too small", g.lda);
    BBBP_CHECK_ARG(g.ldb >= (g.transB ? g.K : g.N), "gemm: ldb %d too small", g.ldb);
    BBBP_CHECK_ARG(g.ldc >= g.N, "gemm: ldc %d too small", g.ldc);
    BBBP_CHECK_ARG(!g.gate || g.ldg >= g.N, "gemm: ldg %d too small", g.ldg);
    BBBP_CHECK_ARG(g.drop_p >= 0.f && g.drop_p < 1.f, "gemm: drop_p %g outside [0, 1)", (double)g.drop_p);
    return BBBP_OK;
}

// layout: NT = (transA 0, transB 1); NN = (0, 0); TN = (1, 0)
inline int layout_of(const bbbp_gemm_desc& g) { return g.transA ? 2 : (g.transB ? 0 : 1); }

bool direct_params(const bbbp_gemm_desc& g, DirectParams* d, int* t) {
    const DirectPlan dp = direct_plan(g.M, g.N, g.K, g.batch);
    if (!dp.use || g.batch > 65535 || cdiv(g.M, 16 * dp.t * dp.wsm) > 65535) return false;
    d->A = g.A; d->B = g.B; d->C = g.C; d->bias = g.bias; d->R = g.residual;
    d->M = g.M; d->N = g.N; d->K = g.K; d->lda = g.lda; d->ldb = g.ldb; d->ldc = g.ldc; d->ldr = g.ldr;
    d->sA = g.strideA; d->sB = g.strideB; d->sC = g.strideC; d->sR = g.strideR;
    d->alpha = g.alpha; d->act = g.act;
    d->gate = g.gate; d->ldg = g.ldg; d->sG = g.strideG; d->gate_scale = g.gate_scale; d->gate_after = g.gate_after_residual;
    d->wsm = dp.wsm; d->wsn = dp.wsn; d->ks = dp.ks;
    d->asum = (g.transA && !g.transB) ? g.asum : nullptr;
    d->drop_p = g.drop_p; d->drop_inv_keep = g.drop_p > 0.f ? 1.f / (1.f - g.drop_p) : 1.f; d->drop_seed = g.drop_seed; d->seed_base = g_bbbp_seed_base;
    d->batch = g.batch; d->gx = cdiv(g.N + (d->asum ? 1 : 0), 16 * dp.t * dp.wsn); d->gy = cdiv(g.M, 16 * dp.t * dp.wsm);
    *t = dp.t;
    return true;
}

int gemm_run(hipStream_t st, const bbbp_gemm_desc& g, void* workspace, size_t workspace_bytes) {
    TRY_RC(gemm_validate(g));
    if (g.M == 0 || g.N == 0) return BBBP_OK;
    const int layout = layout_of(g);
    DirectParams d; int t;
    if (direct_params(g, &d, &t)) {
        launch_direct(d, layout, t, st);
        BBBP_CHECK_LAUNCH();
        return BBBP_OK;
    }
    BBBP_CHECK_ARG(!g.asum, "gemm: asum is only produced by the small-product path (bbbp_gemm_folds_asum(%d, %d, %d, %d) == 0)", g.M, g.N, g.K, g.batch);
    BBBP_CHECK_ARG(!(g.drop_p > 0.f), "gemm: output dropout is only applied by the small-product path (bbbp_gemm_folds_asum(%d, %d, %d, %d) == 0)", g.M, g.N, g.K, g.batch);
    const int M = g.M, N = g.N, K = g.K, batch = g.batch;
    GemmParams p;
    p.A = g.A; p.B = g.B; p.C = g.C; p.bias = g.bias; p.R = g.residual;
    p.M = M; p.N = N; p.K = K; p.lda = g.lda; p.ldb = g.ldb; p.ldc = g.ldc; p.ldr = g.ldr;
    p.sA = g.strideA; p.sB = g.strideB; p.sC = g.strideC; p.sR = g.strideR;
    p.alpha = g.alpha; p.act = g.act;
    p.gate = g.gate; p.ldg = g.ldg; p.sG = g.strideG; p.gate_scale = g.gate_scale; p.gate_after = g.gate_after_residual;
    // 4-wide loads need a contiguous extent that is a multiple of 4 (so that a quad is either fully inside or fully
    // outside the matrix): K for an [M][K] / [N][K] operand, M or N for a [K][.] one.  Bases and strides may be odd
    // (the image-FC weight sits at an odd offset of the flat parameter buffer when F = 167): the loads are unaligned dwordx4.
    p.vecA = ((g.transA ? M : K) % 4 == 0);
    p.vecB = ((g.transB ? K : N) % 4 == 0);
    int tile;
    gemm_plan(M, N, K, batch, &tile, &p.splits, &p.kchunk);
    p.slab = nullptr;
    if (p.splits > 1) {
        size_t need = (size_t)batch * p.splits * M * N * sizeof(float);
        if (!workspace || workspace_bytes < need) {   // no room: fall back to a single pass
            p.splits = 1;
            p.kchunk = cdiv(K, bk_of(tile)) * bk_of(tile);
        } else {
            p.slab = static_cast<float*>(workspace);
        }
    }
    if (K == 0) { p.splits = 1; p.kchunk = bk_of(tile); }
    static const int short_k_max = [] { const char* e = getenv("BBBP_GEMM_SHORT_K"); return e ? atoi(e) : 256; }();
    static const int short_k_tiles = [] { const char* e = getenv("BBBP_GEMM_SHORT_TILES"); return e ? atoi(e) : 4; }();
    p.short_k = (tile == 128 && p.splits == 1 && K <= short_k_max &&
                 (long)cdiv(M, 128) * cdiv(N, 128) * batch >= (long)short_k_tiles * bbbp_num_cus()) ? 1 : 0;
    dim3 grid(cdiv(N, tile), cdiv(M, tile), batch * p.splits);
    BBBP_CHECK_ARG(grid.y <= 65535 && grid.z <= 65535, "gemm: grid too large");
    p.arrivals = nullptr;
    // few 128 x 128 tiles, many 64 x 64 ones, deep K: the small split-bf16 tile walks the whole K in one launch (gemm_b3s_kernel)
    if (tile == 128 && layout != 2 && b3_eligible(p, layout) && b3_small_tile(M, N, K, batch)) {
        p.splits = 1; p.kchunk = cdiv(K, B3_BK) * B3_BK; p.slab = nullptr;
        dim3 gs(cdiv(N, B3S_T), cdiv(M, B3S_T), batch);
        BBBP_CHECK_ARG(gs.y <= 65535 && gs.z <= 65535, "gemm: grid too large");
        if (layout == 0) {
            TRY_RC(bbbp_ensure_dyn_lds(reinterpret_cast<const void*>(gemm_b3s_kernel<0, B3S_DEPTH_NT>), B3S_LDS));
            hipLaunchKernelGGL((gemm_b3s_kernel<0, B3S_DEPTH_NT>), gs, dim3(256), B3S_LDS, st, p);
        } else {
            TRY_RC(bbbp_ensure_dyn_lds(reinterpret_cast<const void*>(gemm_b3s_kernel<1, B3S_DEPTH_NN>), B3S_LDS));
            hipLaunchKernelGGL((gemm_b3s_kernel<1, B3S_DEPTH_NN>), gs, dim3(256), B3S_LDS, st, p);
        }
        BBBP_CHECK_LAUNCH();
        return BBBP_OK;
    }
    if (tile == 128 && b3_eligible(p, layout)) {
        if (p.splits > 1) p.arrivals = arrival_counters(st, (long)grid.x * grid.y * batch);
        TRY_RC(layout == 0 ? launch_b3_one<0>(p, grid, st) : layout == 1 ? launch_b3_one<1>(p, grid, st) : launch_b3_one<2>(p, grid, st));
    } else if (tile == 128) launch_tile<128, 128>(p, layout, grid, st);
    else launch_tile<64, 64>(p, layout, grid, st);
    BBBP_CHECK_LAUNCH();
    if (p.splits > 1 && !p.arrivals) {
        long mn = (long)M * N;
        int gx = (int)((mn + 255) / 256);
        if (gx > 4096) gx = 4096;
        auto al16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
        static const int vec4_on = [] { const char* e = getenv("BBBP_GEMM_REDUCE_VEC4"); return e ? atoi(e) : 1; }();
        const bool vec4 = vec4_on && N % 4 == 0 && p.ldc % 4 == 0 && p.sC % 4 == 0 && al16(p.C) && al16(p.slab) && (!p.bias || al16(p.bias)) &&
                          (!p.R || (p.ldr % 4 == 0 && p.sR % 4 == 0 && al16(p.R))) && (!p.gate || (p.ldg % 4 == 0 && p.sG % 4 == 0 && al16(p.gate)));
        if (vec4) {
            int g4 = (int)((mn / 4 + 255) / 256);
            if (g4 > 4096) g4 = 4096;
            hipLaunchKernelGGL(gemm_splitk_reduce4_kernel, dim3(g4, batch), dim3(256), g_bbbp_small_lds_pad, st, p);
        } else
        hipLaunchKernelGGL(gemm_splitk_reduce_kernel, dim3(gx, batch), dim3(256), g_bbbp_small_lds_pad, st, p);
        BBBP_CHECK_LAUNCH();
    }
    return BBBP_OK;
}

}  // namespace

extern "C" int bbbp_gemm_f32(void* stream, int transA, int transB, int M, int N, int K, float alpha,
                             const float* A, int lda, const float* B, int ldb, float* C, int ldc,
                             const float* bias, const float* residual, int ldr, int act,
                             int batch, long strideA, long strideB, long strideC, long strideR,
                             void* workspace, size_t workspace_bytes) {
    bbbp_gemm_desc g;
    g.transA = transA; g.transB = transB; g.M = M; g.N = N; g.K = K; g.alpha = alpha;
    g.A = A; g.lda = lda; g.B = B; g.ldb = ldb; g.C = C; g.ldc = ldc;
    g.bias = bias; g.residual = residual; g.ldr = ldr; g.act = act;
    g.gate = nullptr; g.ldg = 0; g.gate_scale = 1.f; g.gate_after_residual = 0; g.asum = nullptr; g.drop_p = 0.f; g.drop_seed = 0;
    g.batch = batch; g.strideA = strideA; g.strideB = strideB; g.strideC = strideC; g.strideR = strideR; g.strideG = 0;
    return gemm_run(static_cast<hipStream_t>(stream), g, workspace, workspace_bytes);
}

extern "C" int bbbp_gemm_f32_grouped(void* stream, const bbbp_gemm_desc* problems, int count, void* workspace,
                                     size_t workspace_bytes) {
    BBBP_CHECK_ARG(count >= 0 && (count == 0 || problems), "gemm_grouped: bad arguments");
    hipStream_t st = static_cast<hipStream_t>(stream);
    int i = 0;
    while (i < count) {
        if (i + 1 < count) {                       // try to issue problems i and i + 1 as one launch
            const bbbp_gemm_desc &a = problems[i], &b = problems[i + 1];
            TRY_RC(gemm_validate(a));
            TRY_RC(gemm_validate(b));
            DirectParams da, db; int ta, tb;
            if (a.M > 0 && a.N > 0 && b.M > 0 && b.N > 0 && direct_params(a, &da, &ta) && direct_params(b, &db, &tb) &&
                launch_pair(da, layout_of(a), ta, db, layout_of(b), tb, st)) {
                BBBP_CHECK_LAUNCH();
                i += 2;
                continue;
            }
        }
        // the split-K scratch is shared: sequential launches on one stream may reuse it
        TRY_RC(gemm_run(st, problems[i], workspace, workspace_bytes));
        ++i;
    }
    return BBBP_OK;
}

extern "C" int bbbp_set_gemm_fold_reduce(int on) {
    const int prev = gemm_fold_reduce_on();
    g_gemm_fold_reduce = on ? 1 : 0;
    return prev;
}

extern "C" int bbbp_set_gemm_split_bf16(int on) {
    const int prev = gemm_b3_on();
    g_gemm_b3 = on ? 1 : 0;
    return prev;
}

extern "C" int bbbp_gemm_split_bf16_phases(unsigned long long* phases7) {
    BBBP_CHECK_ARG(phases7 != nullptr, "gemm_split_bf16_phases: null output");
    BBBP_CHECK_HIP(hipDeviceSynchronize());
    BBBP_CHECK_HIP(hipMemcpyFromSymbol(phases7, HIP_SYMBOL(g_gemm_b3_phase), 7 * sizeof(unsigned long long)));
    return BBBP_OK;
}

// Linear + dropout + residual + LayerNorm in one launch (gemm_direct_ln_kernel): narrow outputs only
extern "C" int bbbp_linear_layernorm_supported(int M, int N, int K) {
    return (M >= 1 && N >= 1 && N <= 256 && K >= 1 && K <= 8192 && (M + 15) / 16 <= 65535) ? 1 : 0;
}

extern "C" int bbbp_linear_layernorm_fwd(void* stream, const float* x, int ldx, const float* W, const float* bias, const float* residual, int ldr,
                                         float* z, int ldz, float* y, int ldy, const float* gamma, const float* beta, float* mean, float* rstd,
                                         int M, int N, int K, float eps, float dropout_p, uint64_t seed) {
    BBBP_CHECK_ARG(bbbp_linear_layernorm_supported(M, N, K), "linear_layernorm: M=%d N=%d K=%d not supported (N <= 256, K <= 8192)", M, N, K);
    BBBP_CHECK_ARG(x && W && z && y && gamma && beta && mean && rstd, "linear_layernorm: null pointer");
    BBBP_CHECK_ARG(ldx >= K && ldz >= N && ldy >= N && (!residual || ldr >= N), "linear_layernorm: leading dimension too small");
    BBBP_CHECK_ARG(dropout_p >= 0.f && dropout_p < 1.f, "linear_layernorm: bad dropout %f", (double)dropout_p);
    DirectParams d;
    memset(&d, 0, sizeof(d));
    d.A = x; d.B = W; d.C = z; d.bias = bias; d.R = residual;
    d.M = M; d.N = N; d.K = K; d.lda = ldx; d.ldb = K; d.ldc = ldz; d.ldr = ldr;
    d.alpha = 1.f; d.wsm = 1; d.wsn = cdiv(N, 16); d.ks = 1; d.batch = 1; d.gx = 1; d.gy = cdiv(M, 16);
    d.drop_p = dropout_p; d.drop_inv_keep = dropout_p > 0.f ? 1.f / (1.f - dropout_p) : 1.f; d.drop_seed = seed; d.seed_base = g_bbbp_seed_base;
    LnFuse f{gamma, beta, y, ldy, mean, rstd, eps};
    hipLaunchKernelGGL(gemm_direct_ln_kernel, dim3(1, d.gy), dim3(64 * d.wsn), g_bbbp_small_lds_pad, static_cast<hipStream_t>(stream), d, f);
    BBBP_CHECK_LAUNCH();
    return BBBP_OK;
}

// LayerNorm absorbed by the Linear that consumes it (gemm_direct_lna_kernel)
extern "C" int bbbp_layernorm_linear_supported(int M, int N, int K) {
    return (M >= 1 && N >= 1 && K >= 1 && K <= LNA_MAX_K && (M + 31) / 32 <= 65535 && (N + 31) / 32 < 65535) ? 1 : 0;
}

// engine: absorb the LayerNorm only where the plain product would run on the same 16 x 16 wave tiles anyway (larger outputs take 32 x 32
// wave tiles or the split-bf16 kernels, which the absorbing kernel would be slower than)
bool bbbp_layernorm_linear_preferred(int M, int N, int K) {
    if (!bbbp_layernorm_linear_supported(M, N, K)) return false;
    const DirectPlan dp = direct_plan(M, N, K, 1);
    return dp.use && dp.t == 1 && dp.ks == 1;
}

extern "C" int bbbp_layernorm_linear_fwd(void* stream, const float* z, int ldz, const float* gamma, const float* beta, float eps,
                                         const float* W, const float* bias, float* out, int ldo, int act, float dropout_p, uint64_t seed,
                                         float* y, int ldy, float* mean, float* rstd, int M, int N, int K) {
    BBBP_CHECK_ARG(bbbp_layernorm_linear_supported(M, N, K), "layernorm_linear: M=%d N=%d K=%d not supported (K <= %d)", M, N, K, LNA_MAX_K);
    BBBP_CHECK_ARG(z && gamma && beta && W && out && y && mean && rstd, "layernorm_linear: null pointer");
    BBBP_CHECK_ARG(ldz >= K && ldo >= N && ldy >= K, "layernorm_linear: leading dimension too small");
    BBBP_CHECK_ARG(act >= 0 && act <= 2, "layernorm_linear: bad act %d", act);
    BBBP_CHECK_ARG(dropout_p >= 0.f && dropout_p < 1.f, "layernorm_linear: bad dropout %f", (double)dropout_p);
    LnaParams p;
    p.Z = z; p.ldz = ldz; p.gamma = gamma; p.beta = beta; p.eps = eps; p.W = W; p.ldw = K; p.bias = bias; p.C = out; p.ldc = ldo;
    p.M = M; p.N = N; p.K = K; p.act = act;
    p.drop_p = dropout_p; p.drop_inv_keep = dropout_p > 0.f ? 1.f / (1.f - dropout_p) : 1.f; p.drop_seed = seed; p.seed_base = g_bbbp_seed_base;
    p.Y = y; p.ldy = ldy; p.mean = mean; p.rstd = rstd;
    p.gx = cdiv(N, 32);
    hipLaunchKernelGGL(gemm_direct_lna_kernel, dim3(p.gx + 1, cdiv(M, 32)), dim3(256), g_bbbp_small_lds_pad, static_cast<hipStream_t>(stream), p);
    BBBP_CHECK_LAUNCH();
    return BBBP_OK;
}
