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
template <int TPW, class ASrc, class Epi>
__device__ __forceinline__ void rb_gemm_nt(const ASrc& A, const float* __restrict__ W, int ldw, int N, int K, int wave, int lane, Epi&& epi) {
    const int q = lane & 15, kq = lane >> 4;
    const int ntile = (N + 15) >> 4, nfull = K >> 4, tail = K & 15;
    for (int t0 = wave * TPW; t0 < ntile; t0 += NW * TPW) {
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
template <int T, class ASrc, class Epi>
__device__ __forceinline__ void rb_gemm_kmajor(const ASrc& A, const float* __restrict__ W, int ldw, int N, int K, int wave, int lane, Epi&& epi) {
    const int q = lane & 15, kq = lane >> 4;
    const int npass = (N + 16 * T - 1) / (16 * T), nfull = K >> 4, tail = K & 15;
    for (int ps = wave; ps < npass; ps += NW) {
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
                                            const float* beta, float* zg, float* yg, float* mean_out, float* rstd_out, int wave, int lane) {
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
                if (live) { zg[(row0 + r) * F + c] = z[c]; yg[(row0 + r) * F + c] = y; }
            }
            ys[r * ld + c] = live ? y : 0.f;
        }
        if (live && lane == 0) { mean_out[row0 + r] = mean; rstd_out[row0 + r] = rstd; }
    }
}

// LayerNorm backward of the block's rows: dy (LDS) -> dz = rstd (g - mean(g) - xhat mean(g xhat)), g = dy gamma.
// dz -> `dzg` (global); its dropped copy (the gradient of the sublayer output) -> `ds` (LDS, zero-padded) and `dxg` (global).
__device__ __forceinline__ void ln_bwd_rows(const float* dys, float* ds, int ld, int ldz, int F, int nrows, long row0, const float* zg,
                                            const float* gamma, const float* mean, const float* rstd, float* dzg, float* dxg, float p,
                                            uint64_t seed, int wave, int lane) {
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
                dzg[row * F + c] = v;
                if (dxg != dzg) dxg[row * F + c] = vd;
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
