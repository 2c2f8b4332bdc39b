// Fused ROW-LOCAL stretches of the post-norm encoder layer (nn.TransformerEncoderLayer as built at
// Models/multi_input_data_regression_opt_transformer_cnn_20250113.py:75-78 and called at :110-111 with [B,1,F]: sequence = batch).
// Only the attention couples the rows of a mini-batch; everything between two attentions is a function of ONE row:
//   forward :  ctx -> out_proj (+bias, dropout, +residual) -> LayerNorm1 -> linear1 (+bias, ReLU, dropout) -> linear2 (+bias,
//              dropout, +residual) -> LayerNorm2 -> the NEXT layer's in_proj (or fingerprint_fc + ReLU after the last layer)
//   backward:  (the layer above's in_proj input gradient + residual) -> LayerNorm2 backward -> linear2 input gradient (.) ReLU /
//              dropout gate -> linear1 input gradient + residual -> LayerNorm1 backward -> out_proj input gradient
// As separate launches that is 6 + 6 kernels per layer of 4-30 us each on a dependency chain (the engine's
// side stream was ~200 launches per step).  Here each stretch is ONE launch: a work-group owns 16 rows (= the M of
// v_mfma_f32_16x16x4_f32), keeps them in LDS from stage to stage and streams the weights from L2 straight into the MFMA register
// layout, as gemm.hip's direct kernel does (lane (i, kq) takes k = 16c + 4kq .. +3 with one 16-byte load; the same permutation
// of K on both operands is invisible to a dot product).  Every intermediate the backward pass or a weight-gradient GEMM reads is
// still written to the workspace, in the layout of the launch-per-op schedule, so both schedules share one plan and one set of
// leaf kernels.  Dropout masks come from the same Philox streams (site, element index) as rowops.hip's kernels: the two
// schedules draw identical masks.
// Sized to run BESIDE the image branch's persistent conv work-groups: 4 waves (one per SIMD), <= 64 VGPRs, ~23 KB of LDS.
#include "common.h"
#include "bbbp_hip.h"
#include <cstddef>
#include "encoder_sliced.h"

namespace {

constexpr int ROWS = 16;
constexpr int NTH = 512, NW = NTH / 64;      // two waves per SIMD: the second one's loads and MFMAs fill the first one's load latency
constexpr int PD = 4;                 // chunks of weight loads in flight per wave
typedef float f32x4g __attribute__((ext_vector_type(4), aligned(4)));
typedef float f32x2g __attribute__((ext_vector_type(2), aligned(4)));

__device__ __forceinline__ float wsum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// keep-scales of 4 consecutive elements idx0 .. idx0 + 3 of dropout stream `seed` (same stream as common.h: dropout_scale)
__device__ __forceinline__ void dropout_scale4(uint64_t seed, uint64_t idx0, float p, float inv_keep, float (&s)[4]) {
    const uint4 a = philox4(seed, idx0 >> 2);
    const uint32_t va[4] = {a.x, a.y, a.z, a.w};
    const int off = (int)(idx0 & 3);
    uint32_t v[4];
    if (off == 0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = va[i];
    } else {
        const uint4 b = philox4(seed, (idx0 >> 2) + 1);
        const uint32_t vb[4] = {b.x, b.y, b.z, b.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) { const int j = off + i; v[i] = j < 4 ? va[j & 3] : vb[j & 3]; }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) s[i] = ((float)(v[i] >> 8) * (1.0f / 16777216.0f)) >= p ? inv_keep : 0.f;
}

// ---- A operand (the block's 16 rows): from LDS (zero-padded to a multiple of 16 columns) or from global memory ----------------
struct ALds {
    const float* s; int ld;
    __device__ __forceinline__ f32x4 frag(int c, int q, int kq) const { return *reinterpret_cast<const f32x4*>(s + q * ld + 16 * c + 4 * kq); }
    __device__ __forceinline__ f32x4 frag_tail(int c, int q, int kq, int) const { return frag(c, q, kq); }
};
struct AGlb {
    const float* g;            // row 0 of the block; rows past `nrows` are clamped (their results are never stored)
    int ld, nrows;
    __device__ __forceinline__ f32x4 frag(int c, int q, int kq) const {
        const f32x4g v = *reinterpret_cast<const f32x4g*>(g + (long)min(q, nrows - 1) * ld + 16 * c + 4 * kq);
        return f32x4{v[0], v[1], v[2], v[3]};
    }
    __device__ __forceinline__ f32x4 frag_tail(int c, int q, int kq, int K) const {
        const float* r = g + (long)min(q, nrows - 1) * ld;
        f32x4 v;
#pragma unroll
        for (int j = 0; j < 4; ++j) { const int k = 16 * c + 4 * kq + j; const float x = r[k < K ? k : 0]; v[j] = k < K ? x : 0.f; }
        return v;
    }
};

// ---- out^T tiles: out[16 x N] = A[16 x K] * W[N][K]^T (W = nn.Linear weight, k-contiguous).  The MFMA runs as
// D[n][row] = W_tile * A^T, so lane (q, kq) ends up with row q and the 4 CONSECUTIVE columns n0 = 16 t + 4 kq .. +3:
// one Philox block per lane, 16-byte stores.  epi(n0, row, acc4) is called for every (tile, lane); columns >= N must be skipped
// by the callee.  A must be zero beyond K (LDS rows are padded; AGlb zero-fills its tail), W's tail k is clamped.
// Column tiles [tbeg, tend) only (tend < 0: all): the sliced persistent kernel gives every work-group of a row block its own tiles.
template <int TPW, class ASrc, class Epi>
__device__ __forceinline__ void rb_gemm_nt(const ASrc& A, const float* __restrict__ W, int ldw, int N, int K, int wave, int lane, Epi&& epi,
                                           int tbeg = 0, int tend = -1) {
    const int q = lane & 15, kq = lane >> 4;
    const int nfull = K >> 4, tail = K & 15;
    const int ntile = tend < 0 ? (N + 15) >> 4 : min(tend, (N + 15) >> 4);
    for (int t0 = tbeg + wave * TPW; t0 < ntile; t0 += NW * TPW) {
        const float* wrow[TPW];
#pragma unroll
        for (int u = 0; u < TPW; ++u) wrow[u] = W + (long)min((t0 + u) * 16 + q, N - 1) * ldw + 4 * kq;
        f32x4 acc[TPW];
#pragma unroll
        for (int u = 0; u < TPW; ++u) acc[u] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (nfull > 0) {
            f32x4 wf[PD][TPW], af[PD];
#pragma unroll
            for (int d = 0; d < PD; ++d) {
                const int c = min(d, nfull - 1);
                af[d] = A.frag(c, q, kq);
#pragma unroll
                for (int u = 0; u < TPW; ++u) { const f32x4g v = *reinterpret_cast<const f32x4g*>(wrow[u] + 16 * c); wf[d][u] = f32x4{v[0], v[1], v[2], v[3]}; }
            }
            for (int c0 = 0; c0 < nfull; c0 += PD) {
#pragma unroll
                for (int d = 0; d < PD; ++d) {
                    if (c0 + d < nfull) {
#pragma unroll
                        for (int j = 0; j < 4; ++j)
#pragma unroll
                            for (int u = 0; u < TPW; ++u) acc[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[d][u][j], af[d][j], acc[u], 0, 0, 0);
                    }
                    const int c = min(c0 + d + PD, nfull - 1);
                    af[d] = A.frag(c, q, kq);
#pragma unroll
                    for (int u = 0; u < TPW; ++u) { const f32x4g v = *reinterpret_cast<const f32x4g*>(wrow[u] + 16 * c); wf[d][u] = f32x4{v[0], v[1], v[2], v[3]}; }
                }
            }
        }
        if (tail) {
            const f32x4 a = A.frag_tail(nfull, q, kq, K);
#pragma unroll
            for (int u = 0; u < TPW; ++u) {
                f32x4 w;
#pragma unroll
                for (int j = 0; j < 4; ++j) w[j] = (wrow[u] - 4 * kq)[min(16 * nfull + 4 * kq + j, K - 1)];
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[j], a[j], acc[u], 0, 0, 0);
            }
        }
#pragma unroll
        for (int u = 0; u < TPW; ++u)
            if (t0 + u < ntile) epi((t0 + u) * 16 + 4 * kq, q, acc[u]);
    }
}

// ---- out[16 x N] = A[16 x K] * W[K][N] (contraction over the weight's ROW index: the input gradient of a Linear).
// A wave pass covers 16 * T consecutive columns; lane (q, kq) owns columns cb + T q .. + T - 1 (one T-wide load per k) of rows
// 4 kq .. + 3.  epi(col0, row0, acc[T]) : acc[u][r] = out[row0 + r][col0 + u]; columns >= N must be skipped by the callee.
// Column passes [pbeg, pend) only (pend < 0: all).
template <int T, class ASrc, class Epi>
__device__ __forceinline__ void rb_gemm_kmajor(const ASrc& A, const float* __restrict__ W, int ldw, int N, int K, int wave, int lane, Epi&& epi,
                                               int pbeg = 0, int pend = -1) {
    const int q = lane & 15, kq = lane >> 4;
    const int nfull = K >> 4, tail = K & 15;
    const int npass = pend < 0 ? (N + 16 * T - 1) / (16 * T) : min(pend, (N + 16 * T - 1) / (16 * T));
    for (int ps = pbeg + wave; ps < npass; ps += NW) {
        const int col = ps * 16 * T + T * q;
        const bool edge = (ps + 1) * 16 * T > N;            // wave-uniform: the last pass may hang over the matrix
        f32x4 acc[T];
#pragma unroll
        for (int u = 0; u < T; ++u) acc[u] = f32x4{0.f, 0.f, 0.f, 0.f};
        auto loadw = [&](int k, float (&w)[T]) __attribute__((always_inline)) {
            const float* r = W + (long)k * ldw;
            if (!edge) {
                if constexpr (T == 4) { const f32x4g v = *reinterpret_cast<const f32x4g*>(r + col); w[0] = v[0]; w[1] = v[1]; w[2] = v[2]; w[3] = v[3]; }
                else if constexpr (T == 2) { const f32x2g v = *reinterpret_cast<const f32x2g*>(r + col); w[0] = v[0]; w[1] = v[1]; }
                else w[0] = r[col];
            } else {
#pragma unroll
                for (int u = 0; u < T; ++u) w[u] = r[min(col + u, N - 1)];
            }
        };
        const int nch = nfull + (tail ? 1 : 0);
        // rows k >= K are clamped to K - 1: A is zero there
        float wf[PD][4][T]; f32x4 af[PD];
#pragma unroll
        for (int d = 0; d < PD; ++d) {
            const int c = min(d, nch - 1);
            af[d] = c < nfull ? A.frag(c, q, kq) : A.frag_tail(c, q, kq, K);
#pragma unroll
            for (int j = 0; j < 4; ++j) loadw(min(16 * c + 4 * kq + j, K - 1), wf[d][j]);
        }
        for (int c0 = 0; c0 < nch; c0 += PD) {
#pragma unroll
            for (int d = 0; d < PD; ++d) {
                if (c0 + d < nch) {
#pragma unroll
                    for (int j = 0; j < 4; ++j)
#pragma unroll
                        for (int u = 0; u < T; ++u) acc[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[d][j], wf[d][j][u], acc[u], 0, 0, 0);
                }
                const int c = min(c0 + d + PD, nch - 1);
                af[d] = c < nfull ? A.frag(c, q, kq) : A.frag_tail(c, q, kq, K);
#pragma unroll
                for (int j = 0; j < 4; ++j) loadw(min(16 * c + 4 * kq + j, K - 1), wf[d][j]);
            }
        }
        epi(col, 4 * kq, acc);
    }
}

// rows of a [B][F] matrix -> LDS rows (zero beyond F up to `ldz` columns and beyond the block's last valid row)
__device__ __forceinline__ void rows_to_lds(float* s, int ld, int ldz, const float* g, int F, int nrows, int t) {
    for (int idx = t; idx < ROWS * ldz; idx += NTH) {
        const int r = idx / ldz, c = idx % ldz;
        s[r * ld + c] = (r < nrows && c < F) ? g[(long)r * F + c] : 0.f;
    }
}

// LayerNorm of the block's rows: z (LDS, columns < F) -> y = (z - mean) rstd gamma + beta into `ys` (LDS, zero-padded) and `yg`
// (global); the pre-norm rows go to `zg`, the statistics to mean / rstd.  4 rows per wave, eps 1e-5 (nn.LayerNorm default).
__device__ __forceinline__ void ln_fwd_rows(const float* zs, float* ys, int ld, int ldz, int F, int nrows, long row0, const float* gamma,
                                            const float* beta, float* zg, float* yg, float* mean_out, float* rstd_out, int wave, int lane,
                                            bool store = true) {
    for (int r = wave; r < ROWS; r += NW) {
        const float* z = zs + r * ld;
        float s = 0.f;
        for (int c = lane; c < F; c += 64) s += z[c];
        const float mean = wsum(s) / F;
        float qv = 0.f;
        for (int c = lane; c < F; c += 64) { const float d = z[c] - mean; qv += d * d; }
        const float rstd = rsqrtf(wsum(qv) / F + 1e-5f);
        const bool live = r < nrows;
        for (int c = lane; c < ldz; c += 64) {
            float y = 0.f;
            if (c < F) {
                y = (z[c] - mean) * rstd * gamma[c] + beta[c];
                if (live && store) { zg[(row0 + r) * F + c] = z[c]; yg[(row0 + r) * F + c] = y; }
            }
            ys[r * ld + c] = live ? y : 0.f;
        }
        if (live && store && lane == 0) { mean_out[row0 + r] = mean; rstd_out[row0 + r] = rstd; }
    }
}

// LayerNorm backward of the block's rows: dy (LDS) -> dz = rstd (g - mean(g) - xhat mean(g xhat)), g = dy gamma.
// dz -> `dzg` (global); its dropped copy (the gradient of the sublayer output) -> `ds` (LDS, zero-padded) and `dxg` (global).
__device__ __forceinline__ void ln_bwd_rows(const float* dys, float* ds, int ld, int ldz, int F, int nrows, long row0, const float* zg,
                                            const float* gamma, const float* mean, const float* rstd, float* dzg, float* dxg, float p,
                                            uint64_t seed, int wave, int lane, bool store = true) {
    const float inv_keep = p > 0.f ? 1.f / (1.f - p) : 1.f;
    for (int r = wave; r < ROWS; r += NW) {
        const bool live = r < nrows;
        const long row = row0 + (live ? r : 0);
        const float mu = mean[row], rs = rstd[row];
        const float* dy = dys + r * ld;
        const float* z = zg + row * F;
        float s1 = 0.f, s2 = 0.f;
        for (int c = lane; c < F; c += 64) { const float g = dy[c] * gamma[c]; s1 += g; s2 += g * (z[c] - mu) * rs; }
        s1 = wsum(s1) / F; s2 = wsum(s2) / F;
        for (int c = lane; c < ldz; c += 64) {
            float v = 0.f, vd = 0.f;
            if (c < F && live) {
                const float g = dy[c] * gamma[c];
                v = rs * (g - s1 - (z[c] - mu) * rs * s2);
                vd = p > 0.f ? v * dropout_scale(seed, (uint64_t)row * F + c, p, inv_keep) : v;
                if (store) {
                    dzg[row * F + c] = v;
                    if (dxg != dzg) dxg[row * F + c] = vd;
                }
            }
            ds[r * ld + c] = vd;
        }
    }
}

struct RowFwdParams {
    const float* ctx; const float* xin;                          // [B][F]: attention output, layer input (residual)
    const float *wo, *bo, *g1, *be1, *w1, *b1, *w2, *b2, *g2, *be2;
    const float *wn, *bnx; float* outn; int nn, ldn, actn;       // the next projection: in_proj of layer l + 1 ([3F], ld 3F) or fingerprint_fc ([128] + ReLU into combined, ld 256)
    float *z1, *y1, *hff, *z2, *y2, *mean1, *rstd1, *mean2, *rstd2;
    int B, F, DFF;
    float p; uint64_t seed1, seed2, seed3; const unsigned long long* seed_base;
};

constexpr int MAXF = 192;             // d_model up to 192 (MACCS: 167); LDS rows are padded to a multiple of 16 columns
constexpr int LD = MAXF + 4;

__global__ __launch_bounds__(NTH) void enc_row_fwd_kernel(RowFwdParams P) {
    BBBP_HIGH_PRIO();
    __shared__ __attribute__((aligned(16))) float sA[ROWS * LD];          // ctx, later y1, later y2
    __shared__ __attribute__((aligned(16))) float sB[ROWS * LD];          // z1, later z2
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const long row0 = (long)blockIdx.x * ROWS;
    const int nrows = min(ROWS, P.B - (int)row0);
    const int F = P.F, ldz = (F + 15) & ~15;
    const bool drop = P.p > 0.f;
    const float inv_keep = drop ? 1.f / (1.f - P.p) : 1.f;
    const uint64_t s1 = effective_seed(P.seed1, P.seed_base), s2 = effective_seed(P.seed2, P.seed_base), s3 = effective_seed(P.seed3, P.seed_base);

    rows_to_lds(sA, LD, ldz, P.ctx + row0 * F, F, nrows, t);
    __syncthreads();
    // ---- z1 = dropout(ctx Wo^T + bo) + x ----
    rb_gemm_nt<2>(ALds{sA, LD}, P.wo, F, F, F, wave, lane, [&](int n0, int r, const f32x4& acc) {
        const long row = row0 + min(r, nrows - 1);
        float ks[4] = {1.f, 1.f, 1.f, 1.f};
        if (drop) dropout_scale4(s1, (uint64_t)row * F + n0, P.p, inv_keep, ks);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int n = n0 + i;
            if (n < F) sB[r * LD + n] = (acc[i] + P.bo[n]) * ks[i] + P.xin[row * F + n];
        }
    });
    __syncthreads();
    ln_fwd_rows(sB, sA, LD, ldz, F, nrows, row0, P.g1, P.be1, P.z1, P.y1, P.mean1, P.rstd1, wave, lane);
    __syncthreads();
    // ---- hff = dropout(relu(y1 W1^T + b1)) -> global (the backward pass and linear2 read it) ----
    rb_gemm_nt<2>(ALds{sA, LD}, P.w1, F, P.DFF, F, wave, lane, [&](int n0, int r, const f32x4& acc) {
        if (r >= nrows) return;
        const long row = row0 + r;
        float ks[4] = {1.f, 1.f, 1.f, 1.f};
        if (drop) dropout_scale4(s2, (uint64_t)row * P.DFF + n0, P.p, inv_keep, ks);
        if (n0 + 3 < P.DFF) {
            const f32x4g b = *reinterpret_cast<const f32x4g*>(P.b1 + n0);
            f32x4g v;
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] = fmaxf(acc[i] + b[i], 0.f) * ks[i];
            *reinterpret_cast<f32x4g*>(P.hff + row * P.DFF + n0) = v;
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) if (n0 + i < P.DFF) P.hff[row * P.DFF + n0 + i] = fmaxf(acc[i] + P.b1[n0 + i], 0.f) * ks[i];
        }
    });
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    // ---- z2 = dropout(hff W2^T + b2) + y1 ----
    rb_gemm_nt<2>(AGlb{P.hff + row0 * P.DFF, P.DFF, nrows}, P.w2, P.DFF, F, P.DFF, wave, lane, [&](int n0, int r, const f32x4& acc) {
        const long row = row0 + min(r, nrows - 1);
        float ks[4] = {1.f, 1.f, 1.f, 1.f};
        if (drop) dropout_scale4(s3, (uint64_t)row * F + n0, P.p, inv_keep, ks);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int n = n0 + i;
            if (n < F) sB[r * LD + n] = (acc[i] + P.b2[n]) * ks[i] + sA[r * LD + n];
        }
    });
    __syncthreads();
    ln_fwd_rows(sB, sA, LD, ldz, F, nrows, row0, P.g2, P.be2, P.z2, P.y2, P.mean2, P.rstd2, wave, lane);
    __syncthreads();
    // ---- the next projection of these rows: in_proj of the next layer, or fingerprint_fc + ReLU ----
    if (P.wn) {
        rb_gemm_nt<2>(ALds{sA, LD}, P.wn, F, P.nn, F, wave, lane, [&](int n0, int r, const f32x4& acc) {
            if (r >= nrows) return;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int n = n0 + i;
                if (n < P.nn) { const float v = acc[i] + P.bnx[n]; P.outn[(row0 + r) * P.ldn + n] = P.actn ? fmaxf(v, 0.f) : v; }
            }
        });
    }
}

struct RowBwdParams {
    // stage 0 (optional): dyout = dqkv_up Win_up + dz1_up -- the in_proj input gradient of the layer ABOVE, whose rows are these rows
    const float* dqkv_up; const float* win_up; const float* dz1_up; int k_up;      // k_up = 3F; win_up [3F][F]; null -> dyout is read
    float* dyout;                                                // [B][F] gradient of this layer's output (read, or written by stage 0)
    const float *z2, *mean2, *rstd2, *g2, *w2, *hff, *w1, *z1, *mean1, *rstd1, *g1, *wo;
    float *dz2, *dff, *dhff, *dy1, *dz1, *dsa, *dctx;
    int B, F, DFF;
    float p, inv_keep; uint64_t seed1, seed3; const unsigned long long* seed_base;
};

__global__ __launch_bounds__(NTH) void enc_row_bwd_kernel(RowBwdParams P) {
    BBBP_HIGH_PRIO();
    __shared__ __attribute__((aligned(16))) float sA[ROWS * LD];          // dyout -> dff -> dsa
    __shared__ __attribute__((aligned(16))) float sB[ROWS * LD];          // dy1
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const long row0 = (long)blockIdx.x * ROWS;
    const int nrows = min(ROWS, P.B - (int)row0);
    const int F = P.F, ldz = (F + 15) & ~15;
    const uint64_t s1 = effective_seed(P.seed1, P.seed_base), s3 = effective_seed(P.seed3, P.seed_base);

    if (P.dqkv_up) {
        // pad columns of sA are zeroed once; the epilogue fills columns < F
        for (int idx = t; idx < ROWS * LD; idx += NTH) sA[idx] = 0.f;
        __syncthreads();
        rb_gemm_kmajor<2>(AGlb{P.dqkv_up + row0 * P.k_up, P.k_up, nrows}, P.win_up, F, F, P.k_up, wave, lane,
                          [&](int col0, int r0, const f32x4 (&acc)[2]) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int rr = r0 + r;
                if (rr >= nrows) continue;
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int n = col0 + u;
                    if (n < F) { const float v = acc[u][r] + P.dz1_up[(row0 + rr) * F + n]; sA[rr * LD + n] = v; P.dyout[(row0 + rr) * F + n] = v; }
                }
            }
        });
    } else {
        rows_to_lds(sA, LD, ldz, P.dyout + row0 * F, F, nrows, t);
    }
    __syncthreads();
    // ---- LayerNorm2 backward: dz2 (residual gradient, global) and its dropped copy dff (gradient of linear2's output) ----
    ln_bwd_rows(sA, sA, LD, ldz, F, nrows, row0, P.z2, P.g2, P.mean2, P.rstd2, P.dz2, P.dff, P.p, s3, wave, lane);
    __syncthreads();
    // ---- dhff = (dff W2) (.) [hff > 0] / keep: hff is the POST-dropout activation, > 0 <=> active and kept ----
    rb_gemm_kmajor<4>(ALds{sA, LD}, P.w2, P.DFF, P.DFF, F, wave, lane, [&](int col0, int r0, const f32x4 (&acc)[4]) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int rr = r0 + r;
            if (rr >= nrows) continue;
            const long o = (row0 + rr) * P.DFF + col0;
            if (col0 + 3 < P.DFF) {
                const f32x4g h = *reinterpret_cast<const f32x4g*>(P.hff + o);
                f32x4g v;
#pragma unroll
                for (int u = 0; u < 4; ++u) v[u] = h[u] > 0.f ? acc[u][r] * P.inv_keep : 0.f;
                *reinterpret_cast<f32x4g*>(P.dhff + o) = v;
            } else {
#pragma unroll
                for (int u = 0; u < 4; ++u) if (col0 + u < P.DFF) P.dhff[o + u] = P.hff[o + u] > 0.f ? acc[u][r] * P.inv_keep : 0.f;
            }
        }
    });
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    // ---- dy1 = dhff W1 + dz2 ----
    rb_gemm_kmajor<2>(AGlb{P.dhff + row0 * P.DFF, P.DFF, nrows}, P.w1, F, F, P.DFF, wave, lane, [&](int col0, int r0, const f32x4 (&acc)[2]) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int rr = r0 + r;
            const long row = row0 + min(rr, nrows - 1);
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int n = col0 + u;
                if (n < F) {
                    const float v = rr < nrows ? acc[u][r] + P.dz2[row * F + n] : 0.f;
                    sB[rr * LD + n] = v;
                    if (rr < nrows) P.dy1[row * F + n] = v;
                }
            }
        }
    });
    __syncthreads();
    // ---- LayerNorm1 backward: dz1 (global: the residual of the next stage 0) and dsa = dropped copy (gradient of out_proj's output) ----
    ln_bwd_rows(sB, sA, LD, ldz, F, nrows, row0, P.z1, P.g1, P.mean1, P.rstd1, P.dz1, P.dsa, P.p, s1, wave, lane);
    __syncthreads();
    // ---- dctx = dsa Wo ----
    rb_gemm_kmajor<2>(ALds{sA, LD}, P.wo, F, F, F, wave, lane, [&](int col0, int r0, const f32x4 (&acc)[2]) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int rr = r0 + r;
            if (rr >= nrows) continue;
#pragma unroll
            for (int u = 0; u < 2; ++u) if (col0 + u < F) P.dctx[(row0 + rr) * F + col0 + u] = acc[u][r];
        }
    });
}

// LayerNorm weight / bias gradients of up to 12 norms in ONE launch (they are leaves: nothing waits for them before the optimizer):
// dgamma = sum_rows dy * xhat, dbeta = sum_rows dy.  grid (ceil(F / 16), n_norms); fixed-order LDS tree (bit-reproducible).
constexpr int LN_MAX = 12, LCW = 16, LRL = 32;
struct LnGradParams {
    const float* dy[LN_MAX]; const float* z[LN_MAX]; const float* mean[LN_MAX]; const float* rstd[LN_MAX];
    float* dgamma[LN_MAX]; float* dbeta[LN_MAX];
    int rows, cols;
};
__global__ __launch_bounds__(LCW * LRL) void ln_param_grad_multi_kernel(LnGradParams P) {
    BBBP_HIGH_PRIO();
    __shared__ float s1[LRL][LCW], s2[LRL][LCW];
    const int i = blockIdx.y;
    const float* dy = P.dy[i]; const float* z = P.z[i]; const float* mean = P.mean[i]; const float* rstd = P.rstd[i];
    const int cl = threadIdx.x % LCW, rl = threadIdx.x / LCW;
    const int c = blockIdx.x * LCW + cl;
    float a = 0.f, b = 0.f;
    if (c < P.cols)
        for (int r = rl; r < P.rows; r += LRL) {
            const float g = dy[(long)r * P.cols + c];
            a += g * (z[(long)r * P.cols + c] - mean[r]) * rstd[r];
            b += g;
        }
    s1[rl][cl] = a; s2[rl][cl] = b;
    __syncthreads();
    if (rl == 0 && c < P.cols) {
        float ta = 0.f, tb = 0.f;
#pragma unroll 8
        for (int k = 0; k < LRL; ++k) { ta += s1[k][cl]; tb += s2[k][cl]; }
        P.dgamma[i][c] = ta; P.dbeta[i][c] = tb;
    }
}


// ==================================================================================================================================
// Sliced persistent forward (round 3): the whole encoder chain of a SMALL batch (B <= 128) as one launch.
//
// The launch-per-op schedule needs ~78 dependent launches for the forward chain whatever the batch: at the reference's own batch size
// (DataLoader(batch_size=32), ...20250113.py:167-168) each is at its 5-10 us latency floor and the step is bound by their count.
// The row-fused kernels above cut the count but leave ONE work-group per 16 rows to stream a whole layer's weights (3.2 MB) by itself:
// 0.18 ms per layer.  Here a 16-row block is shared by S work-groups ("slices") that split the COLUMN tiles of every product, so
// S x ceil(B / 16) = 64 work-groups stream the weights; what the row kernel kept in LDS between stages goes through the workspace
// buffers the backward pass reads anyway.  Stages of a layer, per (row block, slice):
//   attention : every slice computes the block's 16 query rows against ALL keys redundantly (1.4 MFLOP at B = 128) -- scores in
//               registers (one 16-key tile per wave), softmax across the 8 waves through LDS, dropout from the softmax kernel's Philox
//               stream, P.V through LDS -- so the context rows are in every slice's LDS without a barrier; slice 0 stores prob / pd / ctx;
//   out_proj  : its column tiles -> z1 (global)                                        | row-block barrier
//   LN1 + linear1: LayerNorm of the 16 rows (every slice, from z1), its tiles of hff   | row-block barrier
//   linear2   : its K range of all column tiles -> partial sums                        | row-block barrier
//   reduce + LN2 (every slice, partials added in slice order), next in_proj tiles      | GRID barrier (the next attention reads every row)
// Barriers are monotonic counters in device memory (zeroed before the launch): arrive = every wave drains its stores, work-group
// barrier, lane 0: agent-scope release fence + relaxed add; wait = relaxed polling with s_sleep, then an agent-scope acquire fence
// (invalidates this CU's L1; MI355X_MICROARCH.md "barrier-counter").  All work-groups must be resident at once: the grid is
// <= 128 work-groups of 512 threads / 26 KB LDS; a wait that exceeds SL_SPIN_LIMIT polls sets the abort word and falls through, so a
// launch can never hang the GPU (the host reads the word: bbbp_enc_sliced_aborted).
// Same saved activations, same dropout streams as the launch-per-op schedule: the backward pass does not know which forward ran.
constexpr unsigned SL_SPIN_LIMIT = 1u << 22;
constexpr int SL_LDP = 176;            // row stride of the linear2 partials
constexpr int SL_KSLICES = 8;          // K ranges of linear2 per row block

struct SlFwdParams {
    bbbp_enc_sliced_fwd_args a;
    int S, NRB;
    unsigned* bar;                     // [0] grid counter, [1 + rb] row-block counters, [SL_ABORT] abort word
    const unsigned long long* seed_base;
};
constexpr int SL_ABORT = 63;

__device__ __forceinline__ void sl_barrier(unsigned* ctr, unsigned target, unsigned* abort_word) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");              // this wave's stores have left
    __syncthreads();
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // the fence's write-back has completed before the arrival is visible
        __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned spins = 0;
        while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(2);
            if (++spins > SL_SPIN_LIMIT) { __hip_atomic_store(abort_word, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
}

// rows of a matrix with leading dimension ldg -> LDS rows (zero beyond F up to ldz columns and beyond the block's last valid row)
__device__ __forceinline__ void rows_to_lds_ld(float* s, int ld, int ldz, const float* g, int ldg, int F, int nrows, int t) {
    for (int idx = t; idx < ROWS * ldz; idx += NTH) {
        const int r = idx / ldz, c = idx % ldz;
        s[r * ld + c] = (r < nrows && c < F) ? g[(long)r * ldg + c] : 0.f;
    }
}

__global__ __launch_bounds__(NTH) void enc_sliced_fwd_kernel(SlFwdParams P) {
    BBBP_HIGH_PRIO();
    __shared__ __attribute__((aligned(16))) float sA[ROWS * LD];          // x -> ctx -> y1 -> y2
    __shared__ __attribute__((aligned(16))) float sB[ROWS * LD];          // Q rows -> dropped probabilities [key][16] -> z1 -> z2
    __shared__ float sred[2][NW][16];
    // every kernel argument used below is copied into a local first (scalars) or per layer (`Y`): a reference into the by-value
    // argument block would make hipcc spill the whole 2 KB block to scratch
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, q = lane & 15, kq = lane >> 4;
    const int S = P.S, rb = blockIdx.x / S, sl = blockIdx.x % S, G = gridDim.x;
    const int B = P.a.B, F = P.a.F, DFF = P.a.DFF, L = P.a.L, F3 = 3 * F, ldz = (F + 15) & ~15, Bp = (B + 15) & ~15;
    const float pdrop = P.a.p, scale = P.a.scale;
    const float* const x0 = P.a.x0;
    float* const part = P.a.part;
    unsigned* const bar = P.bar;
    const unsigned long long* const seed_base = P.seed_base;
    const long row0 = (long)rb * ROWS;
    const int nrows = min(ROWS, B - (int)row0);
    const bool drop = pdrop > 0.f;
    const float inv_keep = drop ? 1.f / (1.f - pdrop) : 1.f;
    unsigned gphase = 0, rphase = 0;
    auto grid_bar = [&]() { sl_barrier(bar, (++gphase) * (unsigned)G, bar + SL_ABORT); };
    auto group_bar = [&]() { sl_barrier(bar + 1 + rb, (++rphase) * (unsigned)S, bar + SL_ABORT); };
    auto my_tiles = [&](int n, int& tb, int& te) { const int nt = (n + 15) >> 4, tps = (nt + S - 1) / S; tb = sl * tps; te = min(nt, tb + tps); };
    // the projection out[row][n] = act(rows(sA) W^T + b) of this slice's column tiles
    auto project = [&](const float* W, const float* bias, int N, float* out, int ldo, bool relu) {
        int tb, te; my_tiles(N, tb, te);
        rb_gemm_nt<2>(ALds{sA, LD}, W, F, N, F, wave, lane, [&](int n0, int r, const f32x4& acc) {
            if (r >= nrows) return;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int n = n0 + i;
                if (n < N) { const float v = acc[i] + bias[n]; out[(row0 + r) * ldo + n] = relu ? fmaxf(v, 0.f) : v; }
            }
        }, tb, te);
    };

    // ---- in_proj of the first layer ----
    rows_to_lds(sA, LD, ldz, x0 + row0 * F, F, nrows, t);
    __syncthreads();
    {
        const float* w = P.a.lay[0].win; const float* bv = P.a.lay[0].bin; float* o = P.a.lay[0].qkv;
        project(w, bv, F3, o, F3, false);
    }
    grid_bar();

    const float* xin = x0;
    for (int l = 0; l < L; ++l) {
        const bbbp_enc_sliced_layer Y = P.a.lay[l];                // by value: scalar loads from the argument block
        const uint64_t s0 = effective_seed(Y.seed0, seed_base), s1 = effective_seed(Y.seed1, seed_base),
                       s2 = effective_seed(Y.seed2, seed_base), s3 = effective_seed(Y.seed3, seed_base);
        // ================= attention of the block's 16 queries over all keys =================
        rows_to_lds_ld(sB, LD, ldz, Y.qkv + row0 * F3, F3, F, nrows, t);              // Q rows (zero-padded columns)
        __syncthreads();
        // S^T tile of this wave: keys 16 wave .. + 15 (rows), the 16 queries (columns); lane (q, kq) register r <-> key 16 wave + 4 kq + r
        const int kt = wave;
        const bool tile_on = kt * 16 < Bp;
        f32x4 st = {0.f, 0.f, 0.f, 0.f};
        if (tile_on) {
            const float* krow = Y.qkv + (long)min(kt * 16 + q, B - 1) * F3 + F + 4 * kq;      // K row of key (kt, lane & 15)
            const int nch = ldz >> 4;                     // the K row is read up to 16 * nch <= F + 15 columns: still inside the [3F] row
#pragma unroll 2
            for (int c = 0; c < nch; ++c) {
                const f32x4g kv = *reinterpret_cast<const f32x4g*>(krow + 16 * c);
                const f32x4 qv = *reinterpret_cast<const f32x4*>(sB + q * LD + 16 * c + 4 * kq);       // zero beyond F
#pragma unroll
                for (int j = 0; j < 4; ++j) st = __builtin_amdgcn_mfma_f32_16x16x4f32(kv[j], qv[j], st, 0, 0, 0);
            }
        }
        float mloc = -INFINITY;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            st[r] = (tile_on && kt * 16 + 4 * kq + r < B) ? st[r] * scale : -INFINITY;
            mloc = fmaxf(mloc, st[r]);
        }
        mloc = fmaxf(mloc, __shfl_xor(mloc, 16)); mloc = fmaxf(mloc, __shfl_xor(mloc, 32));
        if (kq == 0) sred[0][wave][q] = mloc;
        __syncthreads();                                  // every wave is done with the Q rows in sB, too
        float M = sred[0][0][q];
#pragma unroll
        for (int w = 1; w < NW; ++w) M = fmaxf(M, sred[0][w][q]);
        float e[4], sloc = 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) { e[r] = __expf(st[r] - M); sloc += e[r]; }        // exp(-inf) = 0 for masked keys
        sloc += __shfl_xor(sloc, 16); sloc += __shfl_xor(sloc, 32);
        if (kq == 0) sred[1][wave][q] = sloc;
        __syncthreads();
        float tot = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) tot += sred[1][w][q];
        const float inv = 1.f / tot;
        if (tile_on) {
            const int query = (int)row0 + q, key0 = kt * 16 + 4 * kq;
            float ks[4] = {1.f, 1.f, 1.f, 1.f};
            if (drop) dropout_scale4(s0, ((uint64_t)min(query, B - 1)) * B + key0, pdrop, inv_keep, ks);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float pr = e[r] * inv, pdv = pr * ks[r];
                sB[(key0 + r) * 16 + q] = pdv;                                         // [key][query] for the P.V product
                if (sl == 0 && q < nrows && key0 + r < B) {
                    Y.prob[(long)query * B + key0 + r] = pr;
                    if (Y.pd != Y.prob) Y.pd[(long)query * B + key0 + r] = pdv;
                }
            }
        }
        __syncthreads();
        // ctx^T[d][query] = sum_key V^T[d][key] Pd^T[key][query]: wave w takes the d tiles w, w + NW; a step contracts 4 keys
        for (int dt = wave; dt * 16 < ldz; dt += NW) {
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            const float* vcol = Y.qkv + 2 * F + 16 * dt + q;                           // V[.][d = 16 dt + (lane & 15)]: inside the row (d < F + 15)
#pragma unroll 4
            for (int k4 = 0; k4 < Bp; k4 += 4) {
                const float a = vcol[(long)min(k4 + kq, B - 1) * F3];
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, sB[(k4 + kq) * 16 + q], acc, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int d = 16 * dt + 4 * kq + r;                                    // lane (query q, kq), register r
                const float v = d < F ? acc[r] : 0.f;
                sA[q * LD + d] = v;
                if (sl == 0 && q < nrows && d < F) Y.ctx[(row0 + q) * F + d] = v;
            }
        }
        __syncthreads();
        // ================= z1 = dropout(ctx Wo^T + bo) + x : this slice's column tiles =================
        {
            int tb, te; my_tiles(F, tb, te);
            rb_gemm_nt<1>(ALds{sA, LD}, Y.wo, F, F, F, wave, lane, [&](int n0, int r, const f32x4& acc) {
                if (r >= nrows) return;
                const long row = row0 + r;
                float ks[4] = {1.f, 1.f, 1.f, 1.f};
                if (drop) dropout_scale4(s1, (uint64_t)row * F + n0, pdrop, inv_keep, ks);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int n = n0 + i;
                    if (n < F) Y.z1[row * F + n] = (acc[i] + Y.bo[n]) * ks[i] + xin[row * F + n];
                }
            }, tb, te);
        }
        group_bar();
        // ================= LayerNorm1 of the block (every slice), then this slice's tiles of hff =================
        rows_to_lds(sB, LD, ldz, Y.z1 + row0 * F, F, nrows, t);
        __syncthreads();
        ln_fwd_rows(sB, sA, LD, ldz, F, nrows, row0, Y.g1, Y.be1, Y.z1, Y.y1, Y.mean1, Y.rstd1, wave, lane, sl == 0);
        __syncthreads();
        {
            int tb, te; my_tiles(DFF, tb, te);
            rb_gemm_nt<2>(ALds{sA, LD}, Y.w1, F, DFF, F, wave, lane, [&](int n0, int r, const f32x4& acc) {
                if (r >= nrows) return;
                const long row = row0 + r;
                float ks[4] = {1.f, 1.f, 1.f, 1.f};
                if (drop) dropout_scale4(s2, (uint64_t)row * DFF + n0, pdrop, inv_keep, ks);
#pragma unroll
                for (int i = 0; i < 4; ++i) if (n0 + i < DFF) Y.hff[row * DFF + n0 + i] = fmaxf(acc[i] + Y.b1[n0 + i], 0.f) * ks[i];
            }, tb, te);
        }
        group_bar();
        // ================= linear2: this slice's K range of every column tile -> partial sums =================
        // at most SL_KSLICES slices take a K range (the others idle through this stage): the partials are re-read by EVERY slice of the
        // row block, and 32 of them per element made that reduction the longest stage of the layer
        const int nchunk = DFF >> 4, ksl = min(S, SL_KSLICES), cps = (nchunk + ksl - 1) / ksl, nact = (nchunk + cps - 1) / cps;
        float* mypart = part + ((long)(rb * S + sl) * ROWS) * SL_LDP;
        if (sl < nact) {
            const int k0 = sl * cps * 16, klen = min(DFF - k0, cps * 16);
            rb_gemm_nt<2>(AGlb{Y.hff + row0 * DFF + k0, DFF, nrows}, Y.w2 + k0, DFF, F, klen, wave, lane, [&](int n0, int r, const f32x4& acc) {
#pragma unroll
                for (int i = 0; i < 4; ++i) if (n0 + i < SL_LDP) mypart[r * SL_LDP + n0 + i] = acc[i];
            });
        }
        group_bar();
        // ================= z2 = dropout(sum of the partials + b2) + y1, LayerNorm2 (every slice) =================
        for (int idx = t; idx < ROWS * ldz; idx += NTH) {
            const int r = idx / ldz, n = idx % ldz;
            float v = 0.f;
            if (n < F) {
                const float* pp = part + ((long)(rb * S) * ROWS + r) * SL_LDP + n;
                float pv[SL_KSLICES];
#pragma unroll
                for (int s = 0; s < SL_KSLICES; ++s) pv[s] = pp[(long)min(s, nact - 1) * ROWS * SL_LDP];      // independent loads in flight
#pragma unroll
                for (int s = 0; s < SL_KSLICES; ++s) v += s < nact ? pv[s] : 0.f;                                 // added in slice order
                const long row = row0 + min(r, nrows - 1);
                v += Y.b2[n];
                if (drop) v *= dropout_scale(s3, (uint64_t)row * F + n, pdrop, inv_keep);
                v += sA[r * LD + n];
            }
            sB[r * LD + n] = v;
        }
        __syncthreads();
        ln_fwd_rows(sB, sA, LD, ldz, F, nrows, row0, Y.g2, Y.be2, Y.z2, Y.y2, Y.mean2, Y.rstd2, wave, lane, sl == 0);
        __syncthreads();
        // ================= the next projection of these rows =================
        xin = Y.y2;
        if (l + 1 < L) {
            const float* w = P.a.lay[l + 1].win; const float* bv = P.a.lay[l + 1].bin; float* o = P.a.lay[l + 1].qkv;
            project(w, bv, F3, o, F3, false);
            grid_bar();
        } else if (P.a.wfc) {
            const float* w = P.a.wfc; const float* bv = P.a.bfc; float* o = P.a.comb;
            project(w, bv, P.a.nfc, o, P.a.ldcomb, true);
        }
    }
}


// ==================================================================================================================================
// Sliced persistent BACKWARD chain (round 3): the input-gradient chain of the encoder for a small batch as one launch, the mirror
// image of enc_sliced_fwd_kernel.  Per layer, top to bottom, for (row block, slice):
//   LayerNorm2 backward of the block's 16 rows (every slice; slice 0 stores dz2 / dff)
//   dhff = (dff W2) (.) gate      : its column passes                                  | row-block barrier
//   dy1  = dhff W1 + dz2          : its K range -> partials                            | row-block barrier, then reduce (every slice)
//   LayerNorm1 backward (every slice; slice 0 stores dy1 / dz1 / dsa), dctx = dsa Wo (every slice, 16 x 167 x 167: redundant is cheaper
//   than a barrier)
//   attention backward of the block's 16 queries: dP = dctx V^T, dS = P (.) (dP (.) keep - rowsum) in registers (one 16-key tile per
//   wave), dQ = scale dS K (row-local, slice 0 stores it); the block's SHARE of dK = scale dS^T Q and dV = Pd^T dctx for all keys: its
//   (key tile, d tile) pairs -> kvpart[row block]                                      | GRID barrier
//   dK | dV rows of the block = sum over row blocks, in block order: its element range | row-block barrier
//   dyout of the layer below = dqkv Win + dz1 (every slice: 16 x 501 x 167)
// Weight / bias / LayerNorm-parameter gradients are NOT computed here: they stay leaf launches over the buffers this kernel stores.
struct SlBwdParams {
    bbbp_enc_sliced_bwd_args a;
    int S, NRB;
    unsigned* bar;
    const unsigned long long* seed_base;
};

__global__ __launch_bounds__(NTH) void enc_sliced_bwd_kernel(SlBwdParams P) {
    BBBP_HIGH_PRIO();
    __shared__ __attribute__((aligned(16))) float sA[ROWS * LD];          // dyout -> dff -> dsa -> (dropped probabilities^T) -> dyout of the layer below
    __shared__ __attribute__((aligned(16))) float sB[ROWS * LD];          // dy1 -> dctx
    __shared__ __attribute__((aligned(16))) float sS[16 * NW * 16];       // dS^T [key][16 queries]
    __shared__ float sred[NW][16];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, q = lane & 15, kq = lane >> 4;
    const int S = P.S, NRB = P.NRB, rb = blockIdx.x / S, sl = blockIdx.x % S, G = gridDim.x;
    const int B = P.a.B, F = P.a.F, DFF = P.a.DFF, L = P.a.L, F3 = 3 * F, ldz = (F + 15) & ~15, Bp = (B + 15) & ~15;
    const float pdrop = P.a.p, scale = P.a.scale;
    float* const part = P.a.part;
    float* const kvpart = P.a.kvpart;
    unsigned* const bar = P.bar;
    const unsigned long long* const seed_base = P.seed_base;
    const long row0 = (long)rb * ROWS;
    const int nrows = min(ROWS, B - (int)row0);
    const bool drop = pdrop > 0.f;
    const float inv_keep = drop ? 1.f / (1.f - pdrop) : 1.f;
    // the per-layer table stays in the kernel-argument segment and is read field by field with scalar loads where it is used: a by-value
    // copy of one layer is 54 SGPRs held across every stage
    typedef const __attribute__((address_space(4))) bbbp_enc_sliced_bwd_layer* LayerTable;
    const LayerTable lay = (LayerTable)((const __attribute__((address_space(4))) char*)
        __builtin_amdgcn_kernarg_segment_ptr() + offsetof(SlBwdParams, a) + offsetof(bbbp_enc_sliced_bwd_args, lay));
    unsigned gphase = 0, rphase = 0;
    auto grid_bar = [&]() { sl_barrier(bar, (++gphase) * (unsigned)G, bar + SL_ABORT); };
    auto group_bar = [&]() { sl_barrier(bar + 1 + rb, (++rphase) * (unsigned)S, bar + SL_ABORT); };
    float* sP = sA;                                                        // Pd^T [key][16 queries] while sA is free (attention stage)

    // ---- dyout of the top layer = dcomb[:, :nfc] Wfc (every slice) ----
    for (int idx = t; idx < ROWS * LD; idx += NTH) sA[idx] = 0.f;
    __syncthreads();
    {
        const float* dcomb = P.a.dcomb; const float* wfc = P.a.wfc; const int ldc = P.a.ldcomb, nfc = P.a.nfc;
        float* dy = lay[L - 1].dyout;
        rb_gemm_kmajor<2>(AGlb{dcomb + row0 * ldc, ldc, nrows}, wfc, F, F, nfc, wave, lane, [&](int col0, int r0, const f32x4 (&acc)[2]) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int rr = r0 + r;
                if (rr >= nrows) continue;
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int n = col0 + u;
                    if (n < F) { sA[rr * LD + n] = acc[u][r]; if (sl == 0) dy[(row0 + rr) * F + n] = acc[u][r]; }
                }
            }
        });
    }
    __syncthreads();

    for (int l = L - 1; l >= 0; --l) {
#define Y lay[l]
        const uint64_t s0 = effective_seed(Y.seed0, seed_base), s1 = effective_seed(Y.seed1, seed_base), s3 = effective_seed(Y.seed3, seed_base);
        // ================= LayerNorm2 backward: dz2 (global) and its dropped copy dff (LDS + global) =================
        ln_bwd_rows(sA, sA, LD, ldz, F, nrows, row0, Y.z2, Y.g2, Y.mean2, Y.rstd2, Y.dz2, Y.dff, pdrop, s3, wave, lane, sl == 0);
        __syncthreads();
        // ================= dhff = (dff W2) (.) [hff > 0] / keep : this slice's column passes =================
        {
            const int np = (DFF + 63) / 64, pps = (np + S - 1) / S, pb = sl * pps, pe = min(np, pb + pps);
            rb_gemm_kmajor<4>(ALds{sA, LD}, Y.w2, DFF, DFF, F, wave, lane, [&](int col0, int r0, const f32x4 (&acc)[4]) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int rr = r0 + r;
                    if (rr >= nrows) continue;
                    const long o = (row0 + rr) * DFF + col0;
#pragma unroll
                    for (int u = 0; u < 4; ++u) if (col0 + u < DFF) Y.dhff[o + u] = Y.hff[o + u] > 0.f ? acc[u][r] * inv_keep : 0.f;
                }
            }, pb, pe);
        }
        group_bar();
        // ================= dy1 = dhff W1 + dz2 : this slice's K range -> partials; reduce after the barrier =================
        const int nchunk = DFF >> 4, ksl = min(S, SL_KSLICES), cps = (nchunk + ksl - 1) / ksl, nact = (nchunk + cps - 1) / cps;
        if (sl < nact) {
            float* mypart = part + ((long)(rb * S + sl) * ROWS) * SL_LDP;
            const int k0 = sl * cps * 16, klen = min(DFF - k0, cps * 16);
            rb_gemm_kmajor<2>(AGlb{Y.dhff + row0 * DFF + k0, DFF, nrows}, Y.w1 + (long)k0 * F, F, F, klen, wave, lane,
                              [&](int col0, int r0, const f32x4 (&acc)[2]) {
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int u = 0; u < 2; ++u) if (col0 + u < SL_LDP) mypart[(r0 + r) * SL_LDP + col0 + u] = acc[u][r];
            });
        }
        group_bar();
        for (int idx = t; idx < ROWS * ldz; idx += NTH) {
            const int r = idx / ldz, n = idx % ldz;
            float v = 0.f;
            if (n < F && r < nrows) {
                const float* pp = part + ((long)(rb * S) * ROWS + r) * SL_LDP + n;
                float pv[SL_KSLICES];
#pragma unroll
                for (int s = 0; s < SL_KSLICES; ++s) pv[s] = pp[(long)min(s, nact - 1) * ROWS * SL_LDP];
#pragma unroll
                for (int s = 0; s < SL_KSLICES; ++s) v += s < nact ? pv[s] : 0.f;
                v += Y.dz2[(row0 + r) * F + n];
                if (sl == 0) Y.dy1[(row0 + r) * F + n] = v;
            }
            sB[r * LD + n] = v;
        }
        __syncthreads();
        // ================= LayerNorm1 backward: dz1 (global), dsa = dropped copy (LDS + global) =================
        ln_bwd_rows(sB, sA, LD, ldz, F, nrows, row0, Y.z1, Y.g1, Y.mean1, Y.rstd1, Y.dz1, Y.dsa, pdrop, s1, wave, lane, sl == 0);
        __syncthreads();
        // ================= dctx = dsa Wo (every slice) -> sB =================
        rb_gemm_kmajor<2>(ALds{sA, LD}, Y.wo, F, F, F, wave, lane, [&](int col0, int r0, const f32x4 (&acc)[2]) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int u = 0; u < 2; ++u) if (col0 + u < ldz) sB[(r0 + r) * LD + col0 + u] = (col0 + u < F && r0 + r < nrows) ? acc[u][r] : 0.f;
        });
        __syncthreads();
        // ================= attention backward of the block's 16 queries =================
        // dPd^T tile of this wave: keys 16 wave .. + 15 (rows) x 16 queries: sum_d V[key][d] dctx[query][d]
        const int kt = wave;
        const bool tile_on = kt * 16 < Bp;
        f32x4 dpt = {0.f, 0.f, 0.f, 0.f};
        if (tile_on) {
            const float* vrow = Y.qkv + (long)min(kt * 16 + q, B - 1) * F3 + 2 * F + 4 * kq;
            const int nch = ldz >> 4;
#pragma unroll 2
            for (int c = 0; c < nch; ++c) {
                const int dcol = 16 * c + 4 * kq;
                f32x4 vv;
                if (dcol + 3 < F) { const f32x4g x = *reinterpret_cast<const f32x4g*>(vrow + 16 * c); vv = f32x4{x[0], x[1], x[2], x[3]}; }
                else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) vv[j] = dcol + j < F ? vrow[16 * c + j] : 0.f;        // the V row ends the [3F] row: no overrun
                }
                const f32x4 dv = *reinterpret_cast<const f32x4*>(sB + q * LD + 16 * c + 4 * kq);        // dctx, zero beyond F
#pragma unroll
                for (int j = 0; j < 4; ++j) dpt = __builtin_amdgcn_mfma_f32_16x16x4f32(vv[j], dv[j], dpt, 0, 0, 0);
            }
        }
        // lane (q, kq) register r <-> key 16 wave + 4 kq + r, query row0 + q
        const int query = (int)row0 + q, key0 = kt * 16 + 4 * kq;
        float pr[4] = {0.f, 0.f, 0.f, 0.f}, ks[4] = {1.f, 1.f, 1.f, 1.f}, dp[4];
        if (tile_on && q < nrows) {
#pragma unroll
            for (int r = 0; r < 4; ++r) pr[r] = key0 + r < B ? Y.prob[(long)query * B + key0 + r] : 0.f;
        }
        if (drop && tile_on) dropout_scale4(s0, ((uint64_t)min(query, B - 1)) * B + key0, pdrop, inv_keep, ks);
        float dot = 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) { dp[r] = dpt[r] * ks[r]; dot += dp[r] * pr[r]; }
        dot += __shfl_xor(dot, 16); dot += __shfl_xor(dot, 32);
        if (kq == 0) sred[wave][q] = dot;
        __syncthreads();                                   // also: every wave is done with dsa in sA (its Wo product) -> sP may be written
        float rowdot = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) rowdot += sred[w][q];
        if (tile_on) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float dsv = pr[r] * (dp[r] - rowdot);                   // dS[query][key]
                sS[(key0 + r) * 16 + q] = dsv;
                sP[(key0 + r) * 16 + q] = pr[r] * ks[r];                       // Pd[query][key] (what the forward stored as pd)
            }
        }
        __syncthreads();
        // ---- dQ^T[d][query] = scale sum_key K^T[d][key] dS^T[key][query]: wave w takes d tiles w, w + NW; slice 0 stores ----
        if (sl == 0) {
            for (int dt = wave; dt * 16 < ldz; dt += NW) {
                f32x4 acc = {0.f, 0.f, 0.f, 0.f};
                const float* kcol = Y.qkv + F + min(16 * dt + q, F - 1);
#pragma unroll 4
                for (int k4 = 0; k4 < Bp; k4 += 4) {
                    const float a = kcol[(long)min(k4 + kq, B - 1) * F3];
                    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, sS[(k4 + kq) * 16 + q], acc, 0, 0, 0);
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int d = 16 * dt + 4 * kq + r;
                    if (q < nrows && d < F) Y.dqkv[(row0 + q) * F3 + d] = acc[r] * scale;
                }
            }
        }
        // ---- this block's share of dK[key][d] = scale sum_q dS[q][key] Q[q][d] and dV[key][d] = sum_q Pd[q][key] dctx[q][d]:
        //      (key tile, d tile, which) triples dealt over the slices and waves; a product contracts the 16 queries in 4 MFMAs ----
        {
            const int nkt = Bp >> 4, ndt = ldz >> 4, ntrip = nkt * ndt * 2;
            float* kvp = kvpart + ((long)(l & 1) * NRB + rb) * B * 2 * F;     // [B][2F]: dK | dV share of row block rb; two copies by
                                                                               // layer parity (a block may run one layer ahead of its readers)
            for (int tr = sl * NW + wave; tr < ntrip; tr += S * NW) {
                const int which = tr % 2, dt2 = (tr / 2) % ndt, kt2 = tr / (2 * ndt);
                const float* lhs = which ? sP : sS;                            // [key][16 q]
                f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int s4 = 0; s4 < 4; ++s4) {
                    const int qq = 4 * s4 + kq;                                // contraction index: query
                    const float a = lhs[(kt2 * 16 + q) * 16 + qq];            // A[i = key (lane & 15)][k = query]
                    float bv;
                    if (which) bv = sB[qq * LD + 16 * dt2 + q];                // dctx[query][d = 16 dt + (lane & 15)]
                    else { const int d = 16 * dt2 + q; bv = (qq < nrows && d < F) ? Y.qkv[(row0 + qq) * F3 + d] : 0.f; }        // Q[query][d]
                    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bv, acc, 0, 0, 0);
                }
                // lane (col = d, kq), register r <-> key 16 kt + 4 kq + r
                const int d = 16 * dt2 + q;
                if (d < F) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int key = kt2 * 16 + 4 * kq + r;
                        if (key < B) kvp[(long)key * 2 * F + which * F + d] = which ? acc[r] : acc[r] * scale;
                    }
                }
            }
        }
        grid_bar();
        // ================= dK | dV rows of this block: sum of every row block's share, in block order =================
        {
            const int nel = nrows * 2 * F, eps = (nel + S - 1) / S, eb = sl * eps, ee = min(nel, eb + eps);
            for (int e = eb + t; e < ee; e += NTH) {
                const int r = e / (2 * F), c = e % (2 * F);
                float v = 0.f;
                for (int b2 = 0; b2 < NRB; ++b2) v += kvpart[(((long)(l & 1) * NRB + b2) * B + row0 + r) * 2 * F + c];
                Y.dqkv[(row0 + r) * F3 + F + c] = v;
            }
        }
        group_bar();
        // ================= dyout of the layer below = dqkv Win + dz1 (every slice) -> sA =================
        if (l > 0) {
            float* dylow = lay[l - 1].dyout;
            for (int idx = t; idx < ROWS * LD; idx += NTH) sA[idx] = 0.f;
            __syncthreads();
            rb_gemm_kmajor<2>(AGlb{Y.dqkv + row0 * F3, F3, nrows}, Y.win, F, F, F3, wave, lane, [&](int col0, int r0, const f32x4 (&acc)[2]) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int rr = r0 + r;
                    if (rr >= nrows) continue;
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        const int n = col0 + u;
                        if (n < F) {
                            const float v = acc[u][r] + Y.dz1[(row0 + rr) * F + n];
                            sA[rr * LD + n] = v;
                            if (sl == 0) dylow[(row0 + rr) * F + n] = v;
                        }
                    }
                }
            });
            __syncthreads();
        }
    }
#undef Y
}

}  // namespace

// ---- internal entry points (engine.hip) -----------------------------------------------------------------------------------------
bool bbbp_enc_rows_supported(int F, int nhead, int dff) { return nhead >= 1 && F >= 4 && F <= MAXF && dff >= 16; }

int bbbp_enc_row_fwd(hipStream_t st, const bbbp_enc_row_fwd_args* a) {
    RowFwdParams P;
    P.ctx = a->ctx; P.xin = a->xin; P.wo = a->wo; P.bo = a->bo; P.g1 = a->g1; P.be1 = a->be1; P.w1 = a->w1; P.b1 = a->b1;
    P.w2 = a->w2; P.b2 = a->b2; P.g2 = a->g2; P.be2 = a->be2; P.wn = a->wn; P.bnx = a->bn; P.outn = a->outn; P.nn = a->nn; P.ldn = a->ldn;
    P.actn = a->actn; P.z1 = a->z1; P.y1 = a->y1; P.hff = a->hff; P.z2 = a->z2; P.y2 = a->y2; P.mean1 = a->mean1; P.rstd1 = a->rstd1;
    P.mean2 = a->mean2; P.rstd2 = a->rstd2; P.B = a->B; P.F = a->F; P.DFF = a->DFF; P.p = a->p; P.seed1 = a->seed1; P.seed2 = a->seed2;
    P.seed3 = a->seed3; P.seed_base = g_bbbp_seed_base;
    hipLaunchKernelGGL(enc_row_fwd_kernel, dim3(cdiv(a->B, ROWS)), dim3(NTH), 0, st, P);
    BBBP_CHECK_LAUNCH();
    return BBBP_OK;
}

int bbbp_enc_row_bwd(hipStream_t st, const bbbp_enc_row_bwd_args* a) {
    RowBwdParams P;
    P.dqkv_up = a->dqkv_up; P.win_up = a->win_up; P.dz1_up = a->dz1_up; P.k_up = 3 * a->F; P.dyout = a->dyout;
    P.z2 = a->z2; P.mean2 = a->mean2; P.rstd2 = a->rstd2; P.g2 = a->g2; P.w2 = a->w2; P.hff = a->hff; P.w1 = a->w1; P.z1 = a->z1;
    P.mean1 = a->mean1; P.rstd1 = a->rstd1; P.g1 = a->g1; P.wo = a->wo; P.dz2 = a->dz2; P.dff = a->dff; P.dhff = a->dhff; P.dy1 = a->dy1;
    P.dz1 = a->dz1; P.dsa = a->dsa; P.dctx = a->dctx; P.B = a->B; P.F = a->F; P.DFF = a->DFF; P.p = a->p;
    P.inv_keep = a->p > 0.f ? 1.f / (1.f - a->p) : 1.f; P.seed1 = a->seed1; P.seed3 = a->seed3; P.seed_base = g_bbbp_seed_base;
    hipLaunchKernelGGL(enc_row_bwd_kernel, dim3(cdiv(a->B, ROWS)), dim3(NTH), 0, st, P);
    BBBP_CHECK_LAUNCH();
    return BBBP_OK;
}

int bbbp_ln_param_grad_multi(hipStream_t st, int n, const float* const* dy, const float* const* z, const float* const* mean,
                             const float* const* rstd, float* const* dgamma, float* const* dbeta, int rows, int cols) {
    for (int i0 = 0; i0 < n; i0 += LN_MAX) {
        LnGradParams P;
        const int m = n - i0 < LN_MAX ? n - i0 : LN_MAX;
        for (int i = 0; i < LN_MAX; ++i) {
            const int j = i0 + (i < m ? i : 0);
            P.dy[i] = dy[j]; P.z[i] = z[j]; P.mean[i] = mean[j]; P.rstd[i] = rstd[j]; P.dgamma[i] = dgamma[j]; P.dbeta[i] = dbeta[j];
        }
        P.rows = rows; P.cols = cols;
        hipLaunchKernelGGL(ln_param_grad_multi_kernel, dim3(cdiv(cols, LCW), m), dim3(LCW * LRL), 0, st, P);
        BBBP_CHECK_LAUNCH();
    }
    return BBBP_OK;
}

// ---- sliced persistent forward ---------------------------------------------------------------------------------------------------
bool bbbp_enc_sliced_supported(int B, int F, int nhead, int dff, int layers) {
    return B >= 1 && B <= 16 * NW && nhead == 1 && F >= 16 && F <= MAXF - 16 && dff >= 16 && dff % 16 == 0 && layers >= 1 &&
           layers <= BBBP_SLICED_MAX_LAYERS;
}
static int sliced_slices(int B) {
    const int nrb = (B + ROWS - 1) / ROWS;
    int s = 64 / nrb;
    return s < 1 ? 1 : (s > 32 ? 32 : s);
}
size_t bbbp_enc_sliced_sync_bytes() { return 64 * sizeof(unsigned); }
size_t bbbp_enc_sliced_part_bytes(int B, int F) {
    (void)F;
    const int nrb = (B + ROWS - 1) / ROWS;
    return (size_t)nrb * sliced_slices(B) * ROWS * SL_LDP * sizeof(float);
}
int bbbp_enc_sliced_fwd(hipStream_t st, const bbbp_enc_sliced_fwd_args* a) {
    BBBP_CHECK_ARG(a && a->sync && a->part && a->x0, "enc_sliced_fwd: null pointer");
    BBBP_CHECK_ARG(bbbp_enc_sliced_supported(a->B, a->F, 1, a->DFF, a->L), "enc_sliced_fwd: B=%d F=%d dff=%d layers=%d not supported", a->B, a->F, a->DFF, a->L);
    SlFwdParams P;
    P.a = *a;
    P.NRB = (a->B + ROWS - 1) / ROWS;
    P.S = sliced_slices(a->B);
    P.bar = static_cast<unsigned*>(a->sync);
    P.seed_base = g_bbbp_seed_base;
    BBBP_CHECK_ARG(P.NRB * P.S <= bbbp_num_cus(), "enc_sliced_fwd: the grid must be resident at once");
    BBBP_CHECK_HIP(hipMemsetAsync(a->sync, 0, bbbp_enc_sliced_sync_bytes(), st));
    hipLaunchKernelGGL(enc_sliced_fwd_kernel, dim3(P.NRB * P.S), dim3(NTH), 0, st, P);
    BBBP_CHECK_LAUNCH();
    return BBBP_OK;
}
int bbbp_enc_sliced_aborted(hipStream_t st, const void* sync, int* aborted) {
    BBBP_CHECK_ARG(sync && aborted, "enc_sliced_aborted: null pointer");
    unsigned w = 0;
    BBBP_CHECK_HIP(hipMemcpyAsync(&w, static_cast<const unsigned*>(sync) + SL_ABORT, sizeof(w), hipMemcpyDeviceToHost, st));
    BBBP_CHECK_HIP(hipStreamSynchronize(st));
    *aborted = w != 0;
    return BBBP_OK;
}

size_t bbbp_enc_sliced_kvpart_bytes(int B, int F) {
    const int nrb = (B + ROWS - 1) / ROWS;
    return (size_t)2 * nrb * B * 2 * F * sizeof(float);
}
int bbbp_enc_sliced_bwd(hipStream_t st, const bbbp_enc_sliced_bwd_args* a) {
    BBBP_CHECK_ARG(a && a->sync && a->part && a->kvpart && a->dcomb && a->wfc, "enc_sliced_bwd: null pointer");
    BBBP_CHECK_ARG(bbbp_enc_sliced_supported(a->B, a->F, 1, a->DFF, a->L), "enc_sliced_bwd: B=%d F=%d dff=%d layers=%d not supported", a->B, a->F, a->DFF, a->L);
    SlBwdParams P;
    P.a = *a;
    P.NRB = (a->B + ROWS - 1) / ROWS;
    P.S = sliced_slices(a->B);
    P.bar = static_cast<unsigned*>(a->sync);
    P.seed_base = g_bbbp_seed_base;
    BBBP_CHECK_ARG(P.NRB * P.S <= bbbp_num_cus(), "enc_sliced_bwd: the grid must be resident at once");
    BBBP_CHECK_HIP(hipMemsetAsync(a->sync, 0, bbbp_enc_sliced_sync_bytes(), st));
    hipLaunchKernelGGL(enc_sliced_bwd_kernel, dim3(P.NRB * P.S), dim3(NTH), 0, st, P);
    BBBP_CHECK_LAUNCH();
    return BBBP_OK;
}
