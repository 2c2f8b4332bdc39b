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
#include "common.h"
#include "bbbp_hip.h"

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
    int vecA, vecB;          // 16-byte global loads allowed
};

// Four consecutive elements starting at p, `valid` (0..4) of them inside the matrix; the rest read as 0.
// The loads are UNCONDITIONAL: out-of-range lanes read a clamped in-range address (or `safe`, the matrix base) and
// are zeroed by selects.  A load under a per-lane `if` makes hipcc wait vmcnt(0) at the join, which serialised
// every K stage behind a full memory round trip.  VEC: one 16-byte load (the host guarantees valid is 0 or 4).
template <bool VEC>
__device__ __forceinline__ float4 ld4(const float* p, const float* safe, int valid) {
    const float* q = valid > 0 ? p : safe;
    if (VEC) return *reinterpret_cast<const float4*>(q);
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
template <int BM, int BN, int LAYOUT, bool VEC>
__global__ __launch_bounds__(256) void gemm_f32_kernel(GemmParams p) {
    BBBP_HIGH_PRIO();
    constexpr int BK = bk_of(BM);
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
#pragma unroll
        for (int kk = 0; kk < BK; kk += 2) {
            float a[TM], b[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) a[i] = as[kk * LDAS + i * 32];
#pragma unroll
            for (int j = 0; j < TN; ++j) b[j] = bs[kk * LDBS + j * 32];
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = mfma32(a[i], b[j], acc[i][j]);
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
                    if (R) v += R[(long)m * p.ldr + n];
                    C[(long)m * p.ldc + n] = v;
                }
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
        float s = 0.f;
        for (int k = 0; k < p.splits; ++k) s += S[(long)k * mn + idx];
        int m = (int)(idx / p.N), n = (int)(idx % p.N);
        float v = apply_act(p.alpha * s + (p.bias ? p.bias[n] : 0.f), p.act);
        if (R) v += R[(long)m * p.ldr + n];
        C[(long)m * p.ldc + n] = v;
    }
}

template <int BM, int BN, int LAYOUT, bool VEC>
void launch_one(const GemmParams& p, dim3 grid, hipStream_t st) {
    constexpr int BK = bk_of(BM);
    constexpr size_t lds = (size_t)2 * BK * ((BM + (LAYOUT == 2 ? 4 : 1)) + (BN + (LAYOUT != 0 ? 4 : 1))) * sizeof(float);
    static bool attr_set = false;        // > 64 KB of dynamic LDS needs the opt-in once per kernel
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_f32_kernel<BM, BN, LAYOUT, VEC>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set = true;
    }
    hipLaunchKernelGGL((gemm_f32_kernel<BM, BN, LAYOUT, VEC>), grid, dim3(256), lds > g_bbbp_small_lds_pad ? lds : g_bbbp_small_lds_pad, st, p);
}

template <int BM, int BN>
void launch_tile(const GemmParams& p, int layout, dim3 grid, hipStream_t st) {
    const bool vec = p.vecA && p.vecB;
    if (layout == 0) { if (vec) launch_one<BM, BN, 0, true>(p, grid, st); else launch_one<BM, BN, 0, false>(p, grid, st); }
    else if (layout == 1) { if (vec) launch_one<BM, BN, 1, true>(p, grid, st); else launch_one<BM, BN, 1, false>(p, grid, st); }
    else { if (vec) launch_one<BM, BN, 2, true>(p, grid, st); else launch_one<BM, BN, 2, false>(p, grid, st); }
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

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
    long tiles = (*tile == 128) ? t128 : t64;
    // Few output tiles and a deep K (weight gradients over the batch, the 65536-wide image FC, FFN2): these
    // launches are latency-bound at one work-group per tile, so spread K over ~2 work-groups per CU.
    int s = 1;
    if (tiles < ncu && K >= 256) {
        s = (int)((2L * ncu + tiles - 1) / tiles);
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

extern "C" size_t bbbp_gemm_workspace_bytes(int M, int N, int K, int batch) {
    int tile, splits, kchunk;
    gemm_plan(M, N, K, batch, &tile, &splits, &kchunk);
    return splits > 1 ? (size_t)batch * splits * M * N * sizeof(float) : 0;
}

extern "C" int bbbp_gemm_f32(void* stream, int transA, int transB, int M, int N, int K, float alpha,
                             const float* A, int lda, const float* B, int ldb, float* C, int ldc,
                             const float* bias, const float* residual, int ldr, int act,
                             int batch, long strideA, long strideB, long strideC, long strideR,
                             void* workspace, size_t workspace_bytes) {
    BBBP_CHECK_ARG(M >= 0 && N >= 0 && K >= 0 && batch >= 1, "gemm: bad sizes M=%d N=%d K=%d batch=%d", M, N, K, batch);
    BBBP_CHECK_ARG(act >= 0 && act <= 2, "gemm: bad act %d", act);
    BBBP_CHECK_ARG(!(transA && transB), "gemm: layout TT (A^T B^T) is not on the hot path");
    if (M == 0 || N == 0) return BBBP_OK;
    BBBP_CHECK_ARG(A && B && C, "gemm: null operand");
    // layout: NT = (transA 0, transB 1); NN = (0, 0); TN = (1, 0)
    int layout = transA ? 2 : (transB ? 0 : 1);
    BBBP_CHECK_ARG(lda >= (transA ? M : K), "gemm: lda %d too small", lda);
    BBBP_CHECK_ARG(ldb >= (transB ? K : N), "gemm: ldb %d too small", ldb);
    BBBP_CHECK_ARG(ldc >= N, "gemm: ldc %d too small", ldc);
    GemmParams p;
    p.A = A; p.B = B; p.C = C; p.bias = bias; p.R = residual;
    p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldb = ldb; p.ldc = ldc; p.ldr = ldr;
    p.sA = strideA; p.sB = strideB; p.sC = strideC; p.sR = strideR;
    p.alpha = alpha; p.act = act;
    // 16-byte loads need aligned bases/strides AND a contiguous extent that is a multiple of 4 (so that a quad is
    // either fully inside or fully outside the matrix): K for an [M][K] / [N][K] operand, M or N for a [K][.] one
    p.vecA = aligned16(A) && (lda % 4 == 0) && (strideA % 4 == 0) && ((transA ? M : K) % 4 == 0);
    p.vecB = aligned16(B) && (ldb % 4 == 0) && (strideB % 4 == 0) && ((transB ? K : N) % 4 == 0);
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
    hipStream_t st = static_cast<hipStream_t>(stream);
    dim3 grid(cdiv(N, tile), cdiv(M, tile), batch * p.splits);
    BBBP_CHECK_ARG(grid.y <= 65535 && grid.z <= 65535, "gemm: grid too large");
    if (tile == 128) launch_tile<128, 128>(p, layout, grid, st);
    else launch_tile<64, 64>(p, layout, grid, st);
    BBBP_CHECK_LAUNCH();
    if (p.splits > 1) {
        long mn = (long)M * N;
        int gx = (int)((mn + 255) / 256);
        if (gx > 4096) gx = 4096;
        hipLaunchKernelGGL(gemm_splitk_reduce_kernel, dim3(gx, batch), dim3(256), g_bbbp_small_lds_pad, st, p);
        BBBP_CHECK_LAUNCH();
    }
    return BBBP_OK;
}
