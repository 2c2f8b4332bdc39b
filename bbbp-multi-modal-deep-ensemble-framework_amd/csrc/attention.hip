// Self-attention of nn.MultiheadAttention (inside nn.TransformerEncoderLayer, Models/multi_input_data_regression_opt_transformer_cnn_20250113.py:75-78)
// for MANY heads of a SMALL head dimension: the reference's head rule (:71-73) gives the Morgan / RDKit width F = 2048 nhead = 256 and
// head_dim = 8 (F = 64 -> 8 x 8, F = 128 -> 16 x 8), and the encoder is called with [B,1,F], so every head attends over the B
// molecules of the mini-batch.  As separate GEMMs that is 256 products of K = 8 on 128 x 128 tiles (>90 % of the MFMA work is padding)
// around a [256, B, B] probability tensor that is written and re-read five times per layer (268 MB at B = 512: softmax forward alone
// was 0.39 ms per layer, the whole attention ~1 ms of a layer's 1.2 ms forward).
// Here one work-group owns one head: K_h, V_h (and in backward Q_h, dO_h) sit in LDS, scores never leave registers.
//   forward : per 16-query block, S^T = K Q^T on v_mfma_f32_16x16x4_f32 in chunks of 128 keys, online softmax (running max / sum per
//             query), dropout from the same Philox stream as rowops.hip's softmax kernel (element (h B + q) B + k), O^T += V^T Pd^T with
//             the score registers fed STRAIGHT back as the MFMA's B operand (lane (q, kq) of S^T tile register r holds key 4 kq + r of
//             query q -- exactly B[k = kq][j = q] of step r).  Saves logsumexp per (head, query) for the backward pass.
//   backward: recomputes P from Q, K and the saved logsumexp (flash-style).  Sweep A -- a wave owns key tiles and walks the query
//             blocks: dV += Pd^T dO, dK += dS^T Q (the transposed tiles pass through 1 KB of wave-private LDS to reach the A-operand
//             layout).  Sweep B -- a wave owns query blocks and walks the key tiles: dQ^T += K^T dS^T, again with the registers as B
//             operand.  No atomics: every output element has one owner, results are bit-reproducible.
#include "common.h"
#include "bbbp_hip.h"

namespace {

constexpr int NTH = 512, NW = NTH / 64;
constexpr int KT = 8;                  // key tiles (of 16) per online-softmax chunk in the forward pass

struct AttnParams {
    const float* qkv;                  // [B][3F]: Q | K | V, head h at columns h D .. h D + D - 1 of each third
    float* ctx;                        // [B][F] forward output (concatenated heads); backward: input O
    float* lse;                        // [NH][B] logsumexp of the scaled scores
    const float* dctx;                 // backward: dO [B][F]
    float* dqkv;                       // backward: [B][3F]
    int B, F, NH, D;
    float scale, p, inv_keep;
    uint64_t seed; const unsigned long long* seed_base;
};

__device__ __forceinline__ void keep4(uint64_t seed, uint64_t idx0, float p, float inv_keep, float (&s)[4]) {
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

// rows [0, B) of one head's slice of a [B][ld] matrix -> LDS [Bp][DP] (rows >= B and columns >= D are zero)
__device__ __forceinline__ void head_to_lds(float* s, int DP, const float* g, int ld, int B, int Bp, int D, int t) {
    for (int idx = t; idx < Bp * DP; idx += NTH) {
        const int r = idx / DP, c = idx % DP;
        s[idx] = (r < B && c < D) ? g[(long)r * ld + c] : 0.f;
    }
}

// S^T tile [16 keys x 16 queries] = X[key rows kbase ..][d] * Y[query rows qbase ..][d]^T over d < 4 * NS (zero-padded rows)
template <int NS>
__device__ __forceinline__ f32x4 tile_xyT(const float* X, int kbase, const float* Y, int qbase, int DP, int q, int kq) {
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < NS; ++s)
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(X[(kbase + q) * DP + 4 * s + kq], Y[(qbase + q) * DP + 4 * s + kq], acc, 0, 0, 0);
    return acc;
}

template <int NS>       // head_dim = 4 * NS  (8 or 16)
__global__ __launch_bounds__(NTH) void attn_small_fwd_kernel(AttnParams P) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int D = 4 * NS, DP = D + 1;
    const int B = P.B, Bp = (B + 15) & ~15, F = P.F, h = blockIdx.x;
    float* sK = smem; float* sV = sK + Bp * DP; float* sQ = sV + Bp * DP;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, q = lane & 15, kq = lane >> 4;
    head_to_lds(sK, DP, P.qkv + F + h * D, 3 * F, B, Bp, D, t);
    head_to_lds(sV, DP, P.qkv + 2 * F + h * D, 3 * F, B, Bp, D, t);
    head_to_lds(sQ, DP, P.qkv + h * D, 3 * F, B, Bp, D, t);
    __syncthreads();
    const uint64_t seed = effective_seed(P.seed, P.seed_base);
    const bool drop = P.p > 0.f;
    const int ntile = Bp >> 4;
    for (int qb = wave + NW * blockIdx.y; qb < ntile; qb += NW * gridDim.y) {
        const int query = qb * 16 + q;                           // this lane's query (column of every S^T tile)
        float m = -INFINITY, l = 0.f;
        f32x4 o = {0.f, 0.f, 0.f, 0.f};                          // O^T[d = 4 kq + r][query]
        for (int t0 = 0; t0 < ntile; t0 += KT) {
            f32x4 s[KT];
            float cm = -INFINITY;
#pragma unroll
            for (int i = 0; i < KT; ++i) {
                if (t0 + i < ntile) {
                    s[i] = tile_xyT<NS>(sK, (t0 + i) * 16, sQ, qb * 16, DP, q, kq);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int key = (t0 + i) * 16 + 4 * kq + r;
                        s[i][r] = key < B ? s[i][r] * P.scale : -INFINITY;
                        cm = fmaxf(cm, s[i][r]);
                    }
                }
            }
            cm = fmaxf(cm, __shfl_xor(cm, 16)); cm = fmaxf(cm, __shfl_xor(cm, 32));
            const float mn = fmaxf(m, cm);
            const float corr = __expf(m - mn);                   // 0 on the first chunk (m = -inf)
            float ps = 0.f;
            o = o * corr;
#pragma unroll
            for (int i = 0; i < KT; ++i) {
                if (t0 + i < ntile) {
                    float ks[4] = {1.f, 1.f, 1.f, 1.f};
                    if (drop) keep4(seed, ((uint64_t)h * B + min(query, B - 1)) * B + (t0 + i) * 16 + 4 * kq, P.p, P.inv_keep, ks);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float pv = __expf(s[i][r] - mn);
                        ps += pv;
                        // O^T += V^T[d][key] * Pd^T[key][query]: step r contracts keys 4 kq + r
                        const float va = q < D ? sV[((t0 + i) * 16 + 4 * kq + r) * DP + q] : 0.f;       // V^T[d = q][key]
                        o = __builtin_amdgcn_mfma_f32_16x16x4f32(va, pv * ks[r], o, 0, 0, 0);
                    }
                }
            }
            ps += __shfl_xor(ps, 16); ps += __shfl_xor(ps, 32);
            l = l * corr + ps;
            m = mn;
        }
        const float inv = 1.f / l;
        if (query < B) {
            if (4 * kq < D) {
                float* dst = P.ctx + (long)query * F + h * D + 4 * kq;
#pragma unroll
                for (int r = 0; r < 4; ++r) dst[r] = o[r] * inv;
            }
            if (kq == 0) P.lse[(long)h * B + query] = m + __logf(l);
        }
    }
}

template <int NS>
__global__ __launch_bounds__(NTH) void attn_small_bwd_kernel(AttnParams P) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int D = 4 * NS, DP = D + 1;
    const int B = P.B, Bp = (B + 15) & ~15, F = P.F, h = blockIdx.x;
    float* sK = smem; float* sV = sK + Bp * DP; float* sQ = sV + Bp * DP; float* sdO = sQ + Bp * DP;
    float* sL = sdO + Bp * DP; float* sDelta = sL + Bp; float* sT = sDelta + Bp;          // sT: [NW][2][16][17] wave-private transposes
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, q = lane & 15, kq = lane >> 4;
    head_to_lds(sK, DP, P.qkv + F + h * D, 3 * F, B, Bp, D, t);
    head_to_lds(sV, DP, P.qkv + 2 * F + h * D, 3 * F, B, Bp, D, t);
    head_to_lds(sQ, DP, P.qkv + h * D, 3 * F, B, Bp, D, t);
    head_to_lds(sdO, DP, P.dctx + h * D, F, B, Bp, D, t);
    for (int r = t; r < Bp; r += NTH) {
        float dl = 0.f;
        if (r < B) {
#pragma unroll
            for (int c = 0; c < D; ++c) dl += P.dctx[(long)r * F + h * D + c] * P.ctx[(long)r * F + h * D + c];
        }
        sDelta[r] = dl;
        sL[r] = r < B ? P.lse[(long)h * B + r] : INFINITY;       // exp(s - inf) = 0 for padded queries
    }
    __syncthreads();
    const uint64_t seed = effective_seed(P.seed, P.seed_base);
    const bool drop = P.p > 0.f;
    const int ntile = Bp >> 4;
    float* tp = sT + wave * (2 * 16 * 17);                       // Pd^T tile, then dS^T tile: [query][key] with stride 17
    float* td = tp + 16 * 17;

    // P^T and dS^T of tile (key tile kt, query block qb) in registers: lane (q, kq), register r <-> key 4 kq + r, query q
    auto tiles = [&](int kt, int qb, f32x4& pd, f32x4& ds) __attribute__((always_inline)) {
        const f32x4 s = tile_xyT<NS>(sK, kt * 16, sQ, qb * 16, DP, q, kq);
        const f32x4 dp = tile_xyT<NS>(sV, kt * 16, sdO, qb * 16, DP, q, kq);
        const int query = qb * 16 + q;
        const float L = sL[query], dl = sDelta[query];
        float ks[4] = {1.f, 1.f, 1.f, 1.f};
        if (drop) keep4(seed, ((uint64_t)h * B + min(query, B - 1)) * B + kt * 16 + 4 * kq, P.p, P.inv_keep, ks);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int key = kt * 16 + 4 * kq + r;
            const float pv = key < B ? __expf(s[r] * P.scale - L) : 0.f;
            pd[r] = pv * ks[r];
            ds[r] = pv * (dp[r] * ks[r] - dl) * P.scale;
        }
    };

    // ---- sweep A: dK, dV of the key tiles this wave owns ----
    for (int kt = wave + NW * blockIdx.y; kt < ntile; kt += NW * gridDim.y) {
        f32x4 dv = {0.f, 0.f, 0.f, 0.f}, dk = {0.f, 0.f, 0.f, 0.f};          // [key = 4 kq + r][d = q]
        for (int qb = 0; qb < ntile; ++qb) {
            f32x4 pd, ds;
            tiles(kt, qb, pd, ds);
            // transpose through LDS: row = query q, columns = keys 4 kq .. + 3
#pragma unroll
            for (int r = 0; r < 4; ++r) { tp[q * 17 + 4 * kq + r] = pd[r]; td[q * 17 + 4 * kq + r] = ds[r]; }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            // dV[key][d] += sum_query Pd^T[key][query] dO[query][d];  dK[key][d] += sum_query dS^T[key][query] Q[query][d]
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const int ql = 4 * s + kq;                       // A: [i = key q][k = query ql];  B: [k = query ql][j = d q]
                const float bo = q < D ? sdO[(qb * 16 + ql) * DP + q] : 0.f;
                const float bq = q < D ? sQ[(qb * 16 + ql) * DP + q] : 0.f;
                dv = __builtin_amdgcn_mfma_f32_16x16x4f32(tp[ql * 17 + q], bo, dv, 0, 0, 0);
                dk = __builtin_amdgcn_mfma_f32_16x16x4f32(td[ql * 17 + q], bq, dk, 0, 0, 0);
            }
            __builtin_amdgcn_wave_barrier();
        }
        if (q < D) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int key = kt * 16 + 4 * kq + r;
                if (key < B) {
                    P.dqkv[(long)key * 3 * F + 2 * F + h * D + q] = dv[r];
                    P.dqkv[(long)key * 3 * F + F + h * D + q] = dk[r];
                }
            }
        }
    }
    // ---- sweep B: dQ of the query blocks this wave owns ----
    for (int qb = wave + NW * blockIdx.y; qb < ntile; qb += NW * gridDim.y) {
        f32x4 dq = {0.f, 0.f, 0.f, 0.f};                          // dQ^T[d = 4 kq + r][query = q]
        for (int kt = 0; kt < ntile; ++kt) {
            f32x4 pd, ds;
            tiles(kt, qb, pd, ds);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float a = q < D ? sK[(kt * 16 + 4 * kq + r) * DP + q] : 0.f;       // K^T[d = q][key 4 kq + r]
                dq = __builtin_amdgcn_mfma_f32_16x16x4f32(a, ds[r], dq, 0, 0, 0);
            }
        }
        const int query = qb * 16 + q;
        if (query < B && 4 * kq < D) {
            float* dst = P.dqkv + (long)query * 3 * F + h * D + 4 * kq;
#pragma unroll
            for (int r = 0; r < 4; ++r) dst[r] = dq[r];
        }
    }
}

size_t fwd_lds(int B, int D) { const size_t Bp = (B + 15) & ~15; return 3 * Bp * (D + 1) * sizeof(float); }
size_t bwd_lds(int B, int D) { const size_t Bp = (B + 15) & ~15; return (4 * Bp * (D + 1) + 2 * Bp + NW * 2 * 16 * 17) * sizeof(float); }
constexpr size_t LDS_MAX = 160 * 1024;
// few heads (F = 64: 8) cannot fill the chip by themselves: split a head's query blocks / key tiles over several work-groups
int head_parts(int nhead, int B) {
    int parts = 256 / (nhead > 0 ? nhead : 1);
    const int blocks = (B + 15) / 16;
    if (parts > (blocks + NW - 1) / NW) parts = (blocks + NW - 1) / NW;
    return parts < 1 ? 1 : (parts > 16 ? 16 : parts);
}

template <class K>
int set_dyn_lds(K kernel, size_t bytes) {
    if (bytes <= 64 * 1024) return BBBP_OK;
    BBBP_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    return BBBP_OK;
}

}  // namespace

// The fused path serves head_dim 8 and 16 whenever one head's K, V, Q, dO fit in LDS (B <= 1024 at head_dim 8).
bool bbbp_attn_small_supported(int B, int nhead, int head_dim) {
    return nhead > 1 && (head_dim == 8 || head_dim == 16) && B >= 1 && bwd_lds(B, head_dim) <= LDS_MAX;
}

int bbbp_attn_small_fwd(hipStream_t st, const float* qkv, float* ctx, float* lse, int B, int F, int nhead, float scale, float p, uint64_t seed) {
    const int D = F / nhead;
    BBBP_CHECK_ARG(bbbp_attn_small_supported(B, nhead, D), "attn_small: B=%d nhead=%d head_dim=%d not supported", B, nhead, D);
    AttnParams P{qkv, ctx, lse, nullptr, nullptr, B, F, nhead, D, scale, p, p > 0.f ? 1.f / (1.f - p) : 1.f, seed, g_bbbp_seed_base};
    const size_t lds = fwd_lds(B, D);
    if (D == 8) { int rc = set_dyn_lds(attn_small_fwd_kernel<2>, lds); if (rc) return rc; hipLaunchKernelGGL(attn_small_fwd_kernel<2>, dim3(nhead, head_parts(nhead, B)), dim3(NTH), lds, st, P); }
    else { int rc = set_dyn_lds(attn_small_fwd_kernel<4>, lds); if (rc) return rc; hipLaunchKernelGGL(attn_small_fwd_kernel<4>, dim3(nhead, head_parts(nhead, B)), dim3(NTH), lds, st, P); }
    BBBP_CHECK_LAUNCH();
    return BBBP_OK;
}

int bbbp_attn_small_bwd(hipStream_t st, const float* qkv, const float* ctx, const float* lse, const float* dctx, float* dqkv, int B, int F,
                        int nhead, float scale, float p, uint64_t seed) {
    const int D = F / nhead;
    BBBP_CHECK_ARG(bbbp_attn_small_supported(B, nhead, D), "attn_small: B=%d nhead=%d head_dim=%d not supported", B, nhead, D);
    AttnParams P{qkv, const_cast<float*>(ctx), const_cast<float*>(lse), dctx, dqkv, B, F, nhead, D, scale, p, p > 0.f ? 1.f / (1.f - p) : 1.f, seed,
                 g_bbbp_seed_base};
    const size_t lds = bwd_lds(B, D);
    if (D == 8) { int rc = set_dyn_lds(attn_small_bwd_kernel<2>, lds); if (rc) return rc; hipLaunchKernelGGL(attn_small_bwd_kernel<2>, dim3(nhead, head_parts(nhead, B)), dim3(NTH), lds, st, P); }
    else { int rc = set_dyn_lds(attn_small_bwd_kernel<4>, lds); if (rc) return rc; hipLaunchKernelGGL(attn_small_bwd_kernel<4>, dim3(nhead, head_parts(nhead, B)), dim3(NTH), lds, st, P); }
    BBBP_CHECK_LAUNCH();
    return BBBP_OK;
}
