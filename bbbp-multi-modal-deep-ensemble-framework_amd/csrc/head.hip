// Fused forward of the fusion block and the regression head (R:60-65 MultiHeadAttentionFusion.forward, R:98-107 fc):
//   combined[B,256] -> 4 x (Linear(256,128) -> Tanh -> Linear(128,1)) -> softmax over heads -> sum_h w_h * combined
//   -> Linear(256,256)+ReLU -> BatchNorm1d(256) -> Linear(256,128)+ReLU -> Linear(128,64)+ReLU -> Linear(64,1)
// Between the join of the two branches and the loss nothing else runs on the GPU, and as ten separate launches (eight
// GEMMs of a few MFLOP, the head-softmax kernel, BatchNorm) this stretch cost 0.10 ms of pure launch latency per step.
// Here it is TWO launches, split where BatchNorm needs statistics over the whole batch:
//   head_a: one work-group per 16 rows keeps its rows in LDS and runs fusion hidden layers, head softmax, weighted sum
//           and fc.0; it also leaves per-block (mean, M2) of every fc.0 column for the BatchNorm;
//   head_b: every work-group merges those partials in block order (Chan's parallel-variance update: no E[x^2]-E[x]^2
//           cancellation), normalises its 16 rows and runs fc.3, fc.5, fc.7.
// GEMMs use v_mfma_f32_16x16x4_f32 as in gemm.hip's direct kernel: a wave owns 16-column output tiles, the A operand
// (the block's rows) comes from LDS as one ds_read_b128 per 16-deep chunk, the B operand (Linear weight [out][in]) as
// one 16-byte global load per lane and chunk, both in the k-permuted order (lane (i, kq) holds k = 16c + 4kq .. +3).
// Every intermediate the backward pass reads (hid, attn, fused, h, hb, BN mean / rstd, h2, h3) is still written.
#include "common.h"
#include "bbbp_hip.h"
#include "head_sync.h"

namespace {

constexpr int ROWS = 16;         // rows per work-group = MFMA M
constexpr int NTH = 512;         // 8 waves: the tiles of a layer are spread over them, 2 waves per SIMD hide each other's latencies
constexpr int NW = NTH / 64;
constexpr int COMB = 256, FHID = 128, NHEADS = 4, H1 = 256, H2 = 128, H3 = 64;
constexpr int LD = COMB + 4;     // LDS row stride (16-byte aligned rows)
typedef float f32x4g __attribute__((ext_vector_type(4), aligned(4)));      // parameters are only 4-byte aligned
static_assert(NTH >= H1, "one thread per BatchNorm column");

struct HeadParams {
    const float* comb;                                   // [B][256]
    const float* fw1[NHEADS]; const float* fb1[NHEADS]; const float* fw2[NHEADS]; const float* fb2[NHEADS];
    const float* w0; const float* b0;                    // fc.0 [256][256]
    const float* gamma; const float* beta; float* running_mean; float* running_var;
    const float* w3; const float* b3;                    // fc.3 [128][256]
    const float* w5; const float* b5;                    // fc.5 [64][128]
    const float* w7; const float* b7;                    // fc.7 [1][64]
    float* hid; float* attn; float* fused; float* h; float* hb; float* bn_mean; float* bn_rstd; float* h2; float* h3; float* out;
    float* partial;                                      // [blocks][2][256]: per-block mean and M2 of h
    int B, training;
    float eps, momentum;
    int concat;                                          // 1: fused = combined (torch.cat fusion), the fusion block is skipped
    int world, rank;                                     // batch sharded over `world` ranks (B = this rank's rows): partial is [world][blocks][2][256]
};

// B fragments of one 16-column tile: W rows n = 16 * tile + (lane & 15), k = 16c + 4 (lane >> 4) .. +3
template <int NCH>
__device__ __forceinline__ void load_b(f32x4 (&bf)[NCH], const float* Wrow /* W + n * K + 4 * kq */) {
#pragma unroll
    for (int c = 0; c < NCH; ++c) { f32x4g v = *reinterpret_cast<const f32x4g*>(Wrow + 16 * c); bf[c] = f32x4{v[0], v[1], v[2], v[3]}; }
}
// 16 x 16 tile of rows * W^T over K = 16 * NCH; two accumulator chains (the MFMA's dependent latency exceeds its issue time)
template <int NCH>
__device__ __forceinline__ f32x4 mma_tile(const f32x4 (&af)[NCH], const f32x4 (&bf)[NCH]) {
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < NCH; ++c)
#pragma unroll
        for (int j = 0; j < 4; j += 2) {
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(af[c][j], bf[c][j], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(af[c][j + 1], bf[c][j + 1], acc1, 0, 0, 0);
        }
    return acc0 + acc1;
}

template <int NCH>
__device__ __forceinline__ void load_a(f32x4 (&af)[NCH], const float* s, int ld, int lane) {
    const int i = lane & 15, kq = lane >> 4;
#pragma unroll
    for (int c = 0; c < NCH; ++c) af[c] = *reinterpret_cast<const f32x4*>(s + i * ld + 16 * c + 4 * kq);
}

// sum over the 16 lanes that share (lane >> 4)
__device__ __forceinline__ float sum16(float v) {
    v += __shfl_xor(v, 1); v += __shfl_xor(v, 2); v += __shfl_xor(v, 4); v += __shfl_xor(v, 8);
    return v;
}

__global__ __launch_bounds__(NTH) void head_a_kernel(HeadParams p) {
    __shared__ __attribute__((aligned(16))) float sC[ROWS * LD];          // combined, later fused
    __shared__ float sLogit[NW][NHEADS][ROWS];
    __shared__ float sP[ROWS][NHEADS];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int q = lane & 15, kq = lane >> 4;
    const int r0 = blockIdx.x * ROWS, nrows = min(ROWS, p.B - r0);
    // ---- rows of `combined` -> LDS (rows past B: zeros) ----
    for (int idx = t; idx < ROWS * (COMB / 4); idx += NTH) {
        const int r = idx / (COMB / 4), c4 = idx % (COMB / 4);
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (r < nrows) v = *reinterpret_cast<const f32x4*>(p.comb + (long)(r0 + r) * COMB + c4 * 4);
        *reinterpret_cast<f32x4*>(sC + r * LD + c4 * 4) = v;
    }
    __syncthreads();
    f32x4 af[COMB / 16];
    load_a<COMB / 16>(af, sC, LD, lane);
    // ---- fusion hidden layers: 4 heads x 8 column tiles; wave w owns tile w of every head.  The B fragments of the
    //      next tile are in flight while the current tile's MFMAs run. ----
    float lg[NHEADS][4];
    f32x4 bf[2][COMB / 16];
    const int n = wave * 16 + q;
    if (p.concat) {
        // torch.cat fusion (Descriptors/..._round_2_transformer_cnn.py:99): `fused` IS `combined`; straight to fc.0
        load_b<COMB / 16>(bf[0], p.w0 + (long)n * COMB + 4 * kq);
    } else {
    load_b<COMB / 16>(bf[0], p.fw1[0] + (long)n * COMB + 4 * kq);
#pragma unroll
    for (int h = 0; h < NHEADS; ++h) {
        if (h + 1 < NHEADS) load_b<COMB / 16>(bf[(h + 1) & 1], p.fw1[h + 1] + (long)n * COMB + 4 * kq);
        else load_b<COMB / 16>(bf[(h + 1) & 1], p.w0 + (long)n * COMB + 4 * kq);          // first fc.0 tile of this wave
        f32x4 acc = mma_tile<COMB / 16>(af, bf[h & 1]);
        const float b = p.fb1[h][n], w2 = p.fw2[h][n];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = 4 * kq + r;
            const float v = tanhf(acc[r] + b);
            if (row < nrows) p.hid[((long)h * p.B + r0 + row) * FHID + n] = v;
            lg[h][r] = sum16(v * w2);
        }
    }
    if (q == 0) {
#pragma unroll
        for (int h = 0; h < NHEADS; ++h)
#pragma unroll
            for (int r = 0; r < 4; ++r) sLogit[wave][h][4 * kq + r] = lg[h][r];
    }
    __syncthreads();
    // ---- softmax over the heads, one thread per row ----
    if (t < ROWS) {
        float a[NHEADS], m = -INFINITY;
#pragma unroll
        for (int h = 0; h < NHEADS; ++h) {
            float z = 0.f;
#pragma unroll
            for (int w = 0; w < NW; ++w) z += sLogit[w][h][t];          // fixed order
            a[h] = z + p.fb2[h][0];
            m = fmaxf(m, a[h]);
        }
        float den = 0.f;
#pragma unroll
        for (int h = 0; h < NHEADS; ++h) { a[h] = __expf(a[h] - m); den += a[h]; }
#pragma unroll
        for (int h = 0; h < NHEADS; ++h) {
            a[h] /= den;
            sP[t][h] = a[h];
            if (t < nrows) p.attn[(long)(r0 + t) * NHEADS + h] = a[h];
        }
    }
    __syncthreads();
    // ---- fused = sum_h a_h * combined (in place in LDS) ----
    for (int idx = t; idx < ROWS * COMB; idx += NTH) {
        const int r = idx / COMB, c = idx % COMB;
        const float x = sC[r * LD + c];
        float s = 0.f;
#pragma unroll
        for (int h = 0; h < NHEADS; ++h) s += sP[r][h] * x;
        sC[r * LD + c] = s;
        if (r < nrows) p.fused[(long)(r0 + r) * COMB + c] = s;
    }
    __syncthreads();
    load_a<COMB / 16>(af, sC, LD, lane);
    }
    // ---- fc.0 + ReLU, and the block's (mean, M2) per column for the BatchNorm: 16 tiles, 2 per wave ----
    // (bf[0] already holds this wave's first fc.0 tile: it was requested under the last fusion tile)
#pragma unroll
    for (int tt = 0; tt < 2; ++tt) {
        const int n0 = (wave + NW * tt) * 16 + q;
        if (tt == 0) load_b<COMB / 16>(bf[1], p.w0 + (long)((wave + NW) * 16 + q) * COMB + 4 * kq);
        f32x4 acc = mma_tile<COMB / 16>(af, bf[tt & 1]);
        const float b = p.b0[n0];
        float v[4], s = 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = 4 * kq + r;
            v[r] = fmaxf(acc[r] + b, 0.f);
            if (row < nrows) { p.h[(long)(r0 + row) * H1 + n0] = v[r]; s += v[r]; }
        }
        s += __shfl_xor(s, 16); s += __shfl_xor(s, 32);
        const float mean = s / nrows;
        float m2 = 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) if (4 * kq + r < nrows) { const float d = v[r] - mean; m2 += d * d; }
        m2 += __shfl_xor(m2, 16); m2 += __shfl_xor(m2, 32);
        if (kq == 0) {
            const long slot = (long)p.rank * gridDim.x + blockIdx.x;
            p.partial[(slot * 2 + 0) * H1 + n0] = mean;
            p.partial[(slot * 2 + 1) * H1 + n0] = m2;
        }
    }
}

__global__ __launch_bounds__(NTH) void head_b_kernel(HeadParams p) {
    __shared__ __attribute__((aligned(16))) float sA[ROWS * LD];          // hb
    __shared__ __attribute__((aligned(16))) float sB[ROWS * (H2 + 4)];    // h2
    __shared__ __attribute__((aligned(16))) float sD[ROWS * (H3 + 4)];    // h3
    __shared__ float sMean[H1], sRstd[H1];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int q = lane & 15, kq = lane >> 4;
    const int r0 = blockIdx.x * ROWS, nrows = min(ROWS, p.B - r0);
    const int nblocks = (p.B + ROWS - 1) / ROWS;
    // the weights of the first GEMM do not depend on anything computed here: request them before the statistics
    f32x4 bf3[H1 / 16];
    load_b<H1 / 16>(bf3, p.w3 + (long)(wave * 16 + q) * H1 + 4 * kq);
    // ---- batch statistics of column t: merge the blocks' (count, mean, M2) in block order ----
    if (t < H1) {
        const int c = t;
        float mean, rstd;
        if (p.training) {
            float na = 0.f, ma = 0.f, m2a = 0.f;
            for (int b = 0; b < nblocks * p.world; ++b) {          // rank by rank, block by block: the same order on every rank
                const float nb = (float)min(ROWS, p.B - (b % nblocks) * ROWS);
                const float mb = p.partial[((long)b * 2 + 0) * H1 + c], m2b = p.partial[((long)b * 2 + 1) * H1 + c];
                const float d = mb - ma, n = na + nb;
                ma += d * (nb / n);
                m2a += m2b + d * d * (na * nb / n);
                na = n;
            }
            const float nall = (float)p.B * (float)p.world;
            const float var = m2a / nall;
            mean = ma; rstd = rsqrtf(var + p.eps);
            if (blockIdx.x == 0) {
                p.running_mean[c] = (1.f - p.momentum) * p.running_mean[c] + p.momentum * mean;
                p.running_var[c] = (1.f - p.momentum) * p.running_var[c] + p.momentum * var * (nall / (nall - 1.f));
            }
        } else {
            mean = p.running_mean[c]; rstd = rsqrtf(p.running_var[c] + p.eps);
        }
        sMean[c] = mean; sRstd[c] = rstd;
        if (blockIdx.x == 0) { p.bn_mean[c] = mean; p.bn_rstd[c] = rstd; }
    }
    __syncthreads();
    // ---- BatchNorm of this block's rows ----
    for (int idx = t; idx < ROWS * H1; idx += NTH) {
        const int r = idx / H1, c = idx % H1;
        float v = 0.f;
        if (r < nrows) {
            v = (p.h[(long)(r0 + r) * H1 + c] - sMean[c]) * sRstd[c] * p.gamma[c] + p.beta[c];
            p.hb[(long)(r0 + r) * H1 + c] = v;
        }
        sA[r * LD + c] = v;
    }
    __syncthreads();
    // ---- fc.3 + ReLU: 8 column tiles, 1 per wave ----
    {
        f32x4 af[H1 / 16];
        load_a<H1 / 16>(af, sA, LD, lane);
        const int n = wave * 16 + q;
        f32x4 acc = mma_tile<H1 / 16>(af, bf3);
        const float b = p.b3[n];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = 4 * kq + r;
            const float v = fmaxf(acc[r] + b, 0.f);
            sB[row * (H2 + 4) + n] = v;
            if (row < nrows) p.h2[(long)(r0 + row) * H2 + n] = v;
        }
    }
    __syncthreads();
    // ---- fc.5 + ReLU: 4 column tiles, waves 0..3 ----
    if (wave < H3 / 16) {
        f32x4 af[H2 / 16], bf[H2 / 16];
        const int n = wave * 16 + q;
        load_b<H2 / 16>(bf, p.w5 + (long)n * H2 + 4 * kq);
        load_a<H2 / 16>(af, sB, H2 + 4, lane);
        f32x4 acc = mma_tile<H2 / 16>(af, bf);
        const float b = p.b5[n];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = 4 * kq + r;
            const float v = fmaxf(acc[r] + b, 0.f);
            sD[row * (H3 + 4) + n] = v;
            if (row < nrows) p.h3[(long)(r0 + row) * H3 + n] = v;
        }
    }
    __syncthreads();
    // ---- fc.7: one wave, 4 lanes per row ----
    if (wave == 0) {
        const int row = lane >> 2, part = lane & 3;
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < H3 / 4; ++k) s += sD[row * (H3 + 4) + part * (H3 / 4) + k] * p.w7[part * (H3 / 4) + k];
        s += __shfl_xor(s, 1); s += __shfl_xor(s, 2);
        if (part == 0 && row < nrows) p.out[r0 + row] = s + p.b7[0];
    }
}

// ------------------------------------------------------------------------------------------------------------------
// Backward of the same stretch (autograd of R:60-65, 98-107 under loss.backward(), R:190), input gradients only -- the
// weight and bias gradients stay GEMM / column-sum leaves on the leaf stream and read the intermediates written here.
// Two launches, split again where BatchNorm needs sums over the whole batch:
//   head_bwd_a: dh3 = dout W7 (.) [h3 > 0];  dh2 = dh3 W5 (.) [h2 > 0];  dhb = dh2 W3;  per-block column sums of dhb and
//               dhb * xhat for the BatchNorm backward
//   head_bwd_c: merges those sums in block order (= dbeta, dgamma), dh = BatchNorm backward (.) [h > 0], dfused = dh W0,
//               the fusion block's backward (dlogit, dpre through the Tanh), dcomb = dfused sum_h a_h + sum_h dpre_h W1_h,
//               masked by the two branches' output ReLUs.
// Input-gradient products contract over the weight's ROW index (Linear weight [out][in]), so the weight is the k-major
// operand: a lane reads T consecutive columns of row k with one load and feeds T MFMAs (a wave owns T * 16 consecutive
// output columns, column = base + T * (lane & 15) + u), the A operand (the block's rows) comes from LDS as in the forward.
struct HeadBwdParams {
    const float* dout;                                   // [B] (the loss gradient of out[B,1])
    const float* comb; const float* hid; const float* attn; const float* h; const float* h2; const float* h3;
    const float* bn_mean; const float* bn_rstd; const float* gamma;
    const float* fw1[NHEADS]; const float* fw2[NHEADS];
    const float* w0; const float* w3; const float* w5; const float* w7;
    float* dh3; float* dh2; float* dhb; float* dh; float* dlogit; float* dpre; float* dcomb;
    float* dgamma; float* dbeta;
    float* partial;                                      // [world][blocks][2][256]: per-block sums of dhb * xhat and dhb
    int B, training;
    int world, rank;
};

typedef float f32x2g __attribute__((ext_vector_type(2), aligned(4)));

// acc[u] (16 rows x 16 columns base + T q + u) += A[16 x K] (LDS, row stride lda) * W[K][ldw] columns
template <int K, int T>
__device__ __forceinline__ void mma_kmajor(f32x4 (&acc)[T], const float* sA, int lda, const float* W, int ldw, int colbase, int lane) {
    const int q = lane & 15, kq = lane >> 4;
    const float* wcol = W + colbase + T * q;
#pragma unroll 4
    for (int c = 0; c < K / 16; ++c) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(sA + q * lda + 16 * c + 4 * kq);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float* wr = wcol + (long)(16 * c + 4 * kq + j) * ldw;
            if constexpr (T == 2) {
                const f32x2g b = *reinterpret_cast<const f32x2g*>(wr);
                acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j], b[0], acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j], b[1], acc[1], 0, 0, 0);
            } else {
                acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j], wr[0], acc[0], 0, 0, 0);
            }
        }
    }
}

constexpr int LD2 = H2 + 4, LD3 = H3 + 4, LDP = FHID + 4;

__global__ __launch_bounds__(NTH) void head_bwd_a_kernel(HeadBwdParams p) {
    __shared__ __attribute__((aligned(16))) float sD3[ROWS * LD3];        // dh3
    __shared__ __attribute__((aligned(16))) float sD2[ROWS * LD2];        // dh2
    __shared__ float sPa[4][H1], sPb[4][H1];                              // per row-quad column sums of dhb * xhat, dhb
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int q = lane & 15, kq = lane >> 4;
    const int r0 = blockIdx.x * ROWS, nrows = min(ROWS, p.B - r0);
    // ---- dh3 = dout W7, masked by fc.5's ReLU output ----
    for (int idx = t; idx < ROWS * H3; idx += NTH) {
        const int r = idx / H3, c = idx % H3;
        float v = 0.f;
        if (r < nrows) {
            v = p.h3[(long)(r0 + r) * H3 + c] > 0.f ? p.dout[r0 + r] * p.w7[c] : 0.f;
            p.dh3[(long)(r0 + r) * H3 + c] = v;
        }
        sD3[r * LD3 + c] = v;
    }
    __syncthreads();
    // ---- dh2 = dh3 W5 (.) [h2 > 0]: 8 column tiles, one per wave ----
    {
        f32x4 acc[1] = {{0.f, 0.f, 0.f, 0.f}};
        mma_kmajor<H3, 1>(acc, sD3, LD3, p.w5, H2, wave * 16, lane);
        const int n = wave * 16 + q;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = 4 * kq + r;
            float v = 0.f;
            if (row < nrows) {
                v = p.h2[(long)(r0 + row) * H2 + n] > 0.f ? acc[0][r] : 0.f;
                p.dh2[(long)(r0 + row) * H2 + n] = v;
            }
            sD2[row * LD2 + n] = v;
        }
    }
    __syncthreads();
    // ---- dhb = dh2 W3: 32 columns per wave; the block's column sums for the BatchNorm backward ----
    {
        f32x4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
        mma_kmajor<H2, 2>(acc, sD2, LD2, p.w3, H1, wave * 32, lane);
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int n = wave * 32 + 2 * q + u;
            const float mu = p.bn_mean[n], rs = p.bn_rstd[n];
            float pa = 0.f, pb = 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = 4 * kq + r;
                if (row < nrows) {
                    const float g = acc[u][r];
                    p.dhb[(long)(r0 + row) * H1 + n] = g;
                    pa += g * (p.h[(long)(r0 + row) * H1 + n] - mu) * rs;
                    pb += g;
                }
            }
            sPa[kq][n] = pa; sPb[kq][n] = pb;
        }
    }
    __syncthreads();
    if (t < H1) {
        const long slot = (long)p.rank * gridDim.x + blockIdx.x;
        p.partial[(slot * 2 + 0) * H1 + t] = ((sPa[0][t] + sPa[1][t]) + sPa[2][t]) + sPa[3][t];       // fixed order
        p.partial[(slot * 2 + 1) * H1 + t] = ((sPb[0][t] + sPb[1][t]) + sPb[2][t]) + sPb[3][t];
    }
}

__global__ __launch_bounds__(NTH) void head_bwd_c_kernel(HeadBwdParams p) {
    __shared__ __attribute__((aligned(16))) float sX[ROWS * LD];                      // dh, later dfused, finally the unmasked dcomb
    __shared__ __attribute__((aligned(16))) float sP[NHEADS * ROWS * LDP];            // dpre of the four heads
    __shared__ float sSa[H1], sSb[H1];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int q = lane & 15, kq = lane >> 4;
    const int r0 = blockIdx.x * ROWS, nrows = min(ROWS, p.B - r0);
    const int nblocks = (p.B + ROWS - 1) / ROWS;
    // ---- BatchNorm: sums over the batch, blocks merged in block order ----
    if (t < H1) {
        // the sums over the WHOLE (global) batch feed the input gradient; dgamma / dbeta are this rank's rows only (the ranks' parameter
        // gradients are summed / averaged by the caller like every other one)
        float sa = 0.f, sb = 0.f, la = 0.f, lb = 0.f;
        for (int b = 0; b < nblocks * p.world; ++b) {
            const float va = p.partial[((long)b * 2 + 0) * H1 + t], vb = p.partial[((long)b * 2 + 1) * H1 + t];
            sa += va; sb += vb;
            if (b / nblocks == p.rank) { la += va; lb += vb; }
        }
        sSa[t] = sa; sSb[t] = sb;
        if (blockIdx.x == 0) { p.dgamma[t] = la; p.dbeta[t] = lb; }
    }
    __syncthreads();
    // ---- dh = BatchNorm backward of dhb, masked by fc.0's ReLU output h ----
    for (int idx = t; idx < ROWS * H1; idx += NTH) {
        const int r = idx / H1, c = idx % H1;
        float v = 0.f;
        if (r < nrows) {
            const float d = p.dhb[(long)(r0 + r) * H1 + c], xv = p.h[(long)(r0 + r) * H1 + c];
            const float g = p.gamma[c], mu = p.bn_mean[c], rs = p.bn_rstd[c];
            const float nall = (float)p.B * (float)p.world;
            v = p.training ? g * rs * (d - sSb[c] / nall - (xv - mu) * rs * (sSa[c] / nall)) : d * g * rs;
            v = xv > 0.f ? v : 0.f;
            p.dh[(long)(r0 + r) * H1 + c] = v;
        }
        sX[r * LD + c] = v;
    }
    __syncthreads();
    // ---- dfused = dh W0: 32 columns per wave (kept in registers until every wave has read dh) ----
    f32x4 df[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    mma_kmajor<H1, 2>(df, sX, LD, p.w0, COMB, wave * 32, lane);
    __syncthreads();
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int r = 0; r < 4; ++r) sX[(4 * kq + r) * LD + wave * 32 + 2 * q + u] = df[u][r];
    __syncthreads();
    // ---- fusion block backward, two rows per wave: dcomb0 = dfused sum_h a_h; dlogit_h = a_h (t - t sum_k a_k), t = dfused . comb;
    //      dpre_h = dlogit_h w2_h (1 - hid_h^2) ----
#pragma unroll
    for (int rr = 0; rr < 2; ++rr) {
        const int r = wave * 2 + rr;
        const bool live = r < nrows;
        float a[NHEADS], asum = 0.f;
#pragma unroll
        for (int hh = 0; hh < NHEADS; ++hh) { a[hh] = live ? p.attn[(long)(r0 + r) * NHEADS + hh] : 0.f; asum += a[hh]; }
        float tt = 0.f;
#pragma unroll
        for (int c = lane; c < COMB; c += 64) {
            const float g = sX[r * LD + c];
            tt += live ? g * p.comb[(long)(r0 + r) * COMB + c] : 0.f;
            sX[r * LD + c] = g * asum;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) tt += __shfl_xor(tt, o);
#pragma unroll
        for (int hh = 0; hh < NHEADS; ++hh) {
            const float dl = a[hh] * (tt - tt * asum);
            if (lane == 0 && live) p.dlogit[(long)hh * p.B + r0 + r] = dl;
#pragma unroll
            for (int c = lane; c < FHID; c += 64) {
                float v = 0.f;
                if (live) {
                    const float y = p.hid[((long)hh * p.B + r0 + r) * FHID + c];
                    v = dl * p.fw2[hh][c] * (1.f - y * y);
                    p.dpre[((long)hh * p.B + r0 + r) * FHID + c] = v;
                }
                sP[(hh * ROWS + r) * LDP + c] = v;
            }
        }
    }
    __syncthreads();
    // ---- dcomb = dcomb0 + sum_h dpre_h W1_h, masked by the ReLUs that produced combined = [fp_out | img_out] ----
    f32x4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
    for (int hh = 0; hh < NHEADS; ++hh) mma_kmajor<FHID, 2>(acc, sP + hh * ROWS * LDP, LDP, p.fw1[hh], COMB, wave * 32, lane);
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int n = wave * 32 + 2 * q + u;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = 4 * kq + r;
            if (row < nrows) {
                const float v = acc[u][r] + sX[row * LD + n];
                p.dcomb[(long)(r0 + row) * COMB + n] = p.comb[(long)(r0 + row) * COMB + n] > 0.f ? v : 0.f;
            }
        }
    }
}

}  // namespace

// Internal entry point (engine.hip): enqueue the fused fusion-block + head forward.  `partial` needs
// ceil(B / 16) * 2 * 256 floats.
int bbbp_head_forward_fused(hipStream_t st, const float* comb, const float* const* fw1, const float* const* fb1,
                            const float* const* fw2, const float* const* fb2, const float* w0, const float* b0, const float* gamma,
                            const float* beta, float* running_mean, float* running_var, const float* w3, const float* b3,
                            const float* w5, const float* b5, const float* w7, const float* b7, float* hid, float* attn, float* fused,
                            float* h, float* hb, float* bn_mean, float* bn_rstd, float* h2, float* h3, float* out, float* partial,
                            int B, int training, int concat) {
    return bbbp_head_forward_fused_sync(st, comb, fw1, fb1, fw2, fb2, w0, b0, gamma, beta, running_mean, running_var, w3, b3, w5, b5, w7, b7,
                                        hid, attn, fused, h, hb, bn_mean, bn_rstd, h2, h3, out, partial, B, training, concat, nullptr);
}

int bbbp_head_forward_fused_sync(hipStream_t st, const float* comb, const float* const* fw1, const float* const* fb1,
                                 const float* const* fw2, const float* const* fb2, const float* w0, const float* b0, const float* gamma,
                                 const float* beta, float* running_mean, float* running_var, const float* w3, const float* b3,
                                 const float* w5, const float* b5, const float* w7, const float* b7, float* hid, float* attn, float* fused,
                                 float* h, float* hb, float* bn_mean, float* bn_rstd, float* h2, float* h3, float* out, float* partial,
                                 int B, int training, int concat, const bbbp_head_sync* sync) {
    const int world = sync ? sync->world : 1, rank = sync ? sync->rank : 0;
    BBBP_CHECK_ARG(B >= 1, "head: empty batch");
    BBBP_CHECK_ARG(world >= 1 && rank >= 0 && rank < world, "head: rank %d of %d", rank, world);
    BBBP_CHECK_ARG(!(training && (long)B * world <= 1), "Expected more than 1 value per channel when training, got input size [%d, %d]", B, H1);
    HeadParams p;
    p.comb = comb;
    for (int i = 0; i < NHEADS; ++i) { p.fw1[i] = fw1[i]; p.fb1[i] = fb1[i]; p.fw2[i] = fw2[i]; p.fb2[i] = fb2[i]; }
    p.w0 = w0; p.b0 = b0; p.gamma = gamma; p.beta = beta; p.running_mean = running_mean; p.running_var = running_var;
    p.w3 = w3; p.b3 = b3; p.w5 = w5; p.b5 = b5; p.w7 = w7; p.b7 = b7;
    p.hid = hid; p.attn = attn; p.fused = fused; p.h = h; p.hb = hb; p.bn_mean = bn_mean; p.bn_rstd = bn_rstd; p.h2 = h2; p.h3 = h3;
    p.out = out; p.partial = partial; p.B = B; p.training = training; p.eps = 1e-5f; p.momentum = 0.1f; p.concat = concat;
    p.world = world; p.rank = rank;
    const int blocks = cdiv(B, ROWS);
    hipLaunchKernelGGL(head_a_kernel, dim3(blocks), dim3(NTH), 0, st, p);
    BBBP_CHECK_LAUNCH();
    if (sync && sync->between && training) { const int rc = sync->between(); if (rc) return rc; }
    hipLaunchKernelGGL(head_b_kernel, dim3(blocks), dim3(NTH), 0, st, p);
    BBBP_CHECK_LAUNCH();
    return BBBP_OK;
}

// Internal entry point (engine.hip): the input-gradient chain of the head and fusion block in two launches.  Writes dh3, dh2,
// dhb, dh, dlogit [4][B], dpre [4][B][128], dcomb [B][256] (masked by combined > 0) and the BatchNorm's dgamma / dbeta;
// `partial` needs ceil(B / 16) * 2 * 256 floats.
int bbbp_head_backward_fused(hipStream_t st, const float* dout, const float* comb, const float* hid, const float* attn, const float* h,
                             const float* h2, const float* h3, const float* bn_mean, const float* bn_rstd, const float* gamma,
                             const float* const* fw1, const float* const* fw2, const float* w0, const float* w3, const float* w5,
                             const float* w7, float* dh3, float* dh2, float* dhb, float* dh, float* dlogit, float* dpre, float* dcomb,
                             float* dgamma, float* dbeta, float* partial, int B, int training) {
    return bbbp_head_backward_fused_sync(st, dout, comb, hid, attn, h, h2, h3, bn_mean, bn_rstd, gamma, fw1, fw2, w0, w3, w5, w7, dh3, dh2, dhb, dh,
                                         dlogit, dpre, dcomb, dgamma, dbeta, partial, B, training, nullptr);
}

int bbbp_head_backward_fused_sync(hipStream_t st, const float* dout, const float* comb, const float* hid, const float* attn, const float* h,
                                  const float* h2, const float* h3, const float* bn_mean, const float* bn_rstd, const float* gamma,
                                  const float* const* fw1, const float* const* fw2, const float* w0, const float* w3, const float* w5,
                                  const float* w7, float* dh3, float* dh2, float* dhb, float* dh, float* dlogit, float* dpre, float* dcomb,
                                  float* dgamma, float* dbeta, float* partial, int B, int training, const bbbp_head_sync* sync) {
    const int world = sync ? sync->world : 1, rank = sync ? sync->rank : 0;
    BBBP_CHECK_ARG(world >= 1 && rank >= 0 && rank < world, "head backward: rank %d of %d", rank, world);
    BBBP_CHECK_ARG(B >= 1, "head backward: empty batch");
    HeadBwdParams p;
    p.dout = dout; p.comb = comb; p.hid = hid; p.attn = attn; p.h = h; p.h2 = h2; p.h3 = h3;
    p.bn_mean = bn_mean; p.bn_rstd = bn_rstd; p.gamma = gamma;
    for (int i = 0; i < NHEADS; ++i) { p.fw1[i] = fw1[i]; p.fw2[i] = fw2[i]; }
    p.w0 = w0; p.w3 = w3; p.w5 = w5; p.w7 = w7;
    p.dh3 = dh3; p.dh2 = dh2; p.dhb = dhb; p.dh = dh; p.dlogit = dlogit; p.dpre = dpre; p.dcomb = dcomb;
    p.dgamma = dgamma; p.dbeta = dbeta; p.partial = partial; p.B = B; p.training = training;
    p.world = world; p.rank = rank;
    const int blocks = cdiv(B, ROWS);
    hipLaunchKernelGGL(head_bwd_a_kernel, dim3(blocks), dim3(NTH), 0, st, p);
    BBBP_CHECK_LAUNCH();
    if (sync && sync->between && training) { const int rc = sync->between(); if (rc) return rc; }
    hipLaunchKernelGGL(head_bwd_c_kernel, dim3(blocks), dim3(NTH), 0, st, p);
    BBBP_CHECK_LAUNCH();
    return BBBP_OK;
}
