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
#include <stdlib.h>

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
    uint8_t* keep;                     // optional [NH][ceil(B/16)][ceil(B/16)][64]: dropout keep bits per (head, query block, key tile, lane),
                                       // bit r of lane (q, kq) <-> key 4 kq + r of query q; written by the forward pass, read by backward
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
                    if (drop) {
                        keep4(seed, ((uint64_t)h * B + min(query, B - 1)) * B + (t0 + i) * 16 + 4 * kq, P.p, P.inv_keep, ks);
                        // the backward pass reads the decisions back instead of drawing the Philox blocks again (its dominant cost)
                        if (P.keep)
                            P.keep[(((long)h * ntile + qb) * ntile + (t0 + i)) * 64 + lane] =
                                (uint8_t)((ks[0] != 0.f) | ((ks[1] != 0.f) << 1) | ((ks[2] != 0.f) << 2) | ((ks[3] != 0.f) << 3));
                    }
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
        if (drop) {
            if (P.keep) {
                const unsigned bits = P.keep[(((long)h * ntile + qb) * ntile + kt) * 64 + lane];
#pragma unroll
                for (int r = 0; r < 4; ++r) ks[r] = (bits >> r) & 1u ? P.inv_keep : 0.f;
            } else {
                keep4(seed, ((uint64_t)h * B + min(query, B - 1)) * B + kt * 16 + 4 * kq, P.p, P.inv_keep, ks);
            }
        }
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

// Backward in ONE sweep (round 3; one work-group per head, B <= 16 * NW * NKT).  The two-sweep kernel above evaluates every (key tile,
// query block) pair twice -- S and dP on the matrix pipe, four exp and, with dropout, a Philox block per lane each time -- and moves the
// transposed tiles through wave-private LDS behind two wave barriers per pair: at head_dim 8 a pair is a ~1600-cycle dependent chain
// with two waves per SIMD to cover it (0.27 ms per layer at F = 2048, B = 512).  Here
//   * wave w owns key tiles w, w + 8, ... (dV^T, dK^T accumulators in registers for the whole sweep); all eight waves walk the query
//     blocks in step, and a block's dQ^T partials (one per wave: its key tiles' share) are added through LDS in wave order by 256
//     threads -- one barrier per query block, a fixed summation order, no atomics;
//   * a pair's probabilities are produced in BOTH register layouts by the matrix pipe (S^T = K Q^T: rows = keys, the B operand of
//     dQ^T += K^T dS^T; S = Q K^T: rows = queries, the B operand of dV^T += dO^T Pd and dK^T += Q^T dS), so nothing is transposed
//     through LDS and there is no barrier inside a pair: 20 independent-enough MFMAs and 8 exp per pair, four pairs in flight;
//   * the dropout decisions come back as the bytes the forward pass saved (AttnParams::keep, S^T layout) instead of being drawn
//     again; the S layout reads them out of four wave-wide ballots.
template <int NS, int NKT>
__global__ __launch_bounds__(NTH) void attn_small_bwd1_kernel(AttnParams P) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int D = 4 * NS, DP = D + 1;
    const int B = P.B, Bp = (B + 15) & ~15, F = P.F, h = blockIdx.x;
    float* sK = smem; float* sV = sK + Bp * DP; float* sQ = sV + Bp * DP; float* sdO = sQ + Bp * DP;
    float* sL = sdO + Bp * DP; float* sDelta = sL + Bp;
    float* sR = sDelta + Bp;                                                               // [2][NW][256] dQ^T partials, double-buffered
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
    const bool drop = P.p > 0.f;                                  // the launcher guarantees P.keep when drop
    const int ntile = Bp >> 4;
    const uint8_t* keep = drop ? P.keep + (long)h * ntile * ntile * 64 + lane : nullptr;

    // Vector work per pair is what is left to save (the f32 MFMA and the vector ALU share a SIMD's FMA lanes): the softmax scale and
    // log2(e) are folded into the saved logsumexp and one multiply per score (exp2 of a difference), the scale of dS into the K^T / Q^T
    // operands of dQ^T / dK^T, and ragged tiles need no selects -- a padded key's K row is zero, so its dS reaches dQ^T through a zero
    // operand and its dV^T / dK^T columns are never stored; a padded query has logsumexp = +inf, i.e. probability 0.
    const float LOG2E = 1.4426950408889634f;
    const float sc2 = P.scale * LOG2E;
    f32x4 dv[NKT], dk[NKT];                                      // dV^T, dK^T [d = 4 kq + r][key = q] of the wave's key tiles
    float ak[NKT][4];                                            // scale * K^T[d = q][key 4 kq + r] of the wave's key tiles: A operand of dQ^T
#pragma unroll
    for (int i = 0; i < NKT; ++i) {
        const int kt = wave + NW * i;
        dv[i] = f32x4{0.f, 0.f, 0.f, 0.f}; dk[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int r = 0; r < 4; ++r) ak[i][r] = (kt < ntile && q < D) ? P.scale * sK[(kt * 16 + 4 * kq + r) * DP + q] : 0.f;
    }
    unsigned kb_next[NKT];
#pragma unroll
    for (int i = 0; i < NKT; ++i) kb_next[i] = (keep && wave + NW * i < ntile) ? keep[(long)(wave + NW * i) * 64] : 0xfu;
    // S layout: lane (q, kq), register r <-> query 4 kq + r, key q.  Its keep bit sits in the S^T layout's lane (query, key >> 2), bit
    // key & 3, i.e. bit ((q >> 2) * 16 + 4 kq + r) of the ballot of bit (q & 3)
    const int bal_sel = q & 3, bal_shift = (q >> 2) * 16 + 4 * kq;

    for (int qb = 0; qb < ntile; ++qb) {
        unsigned kb[NKT];
#pragma unroll
        for (int i = 0; i < NKT; ++i) {
            kb[i] = kb_next[i];
            // the next query block's bytes travel while this block is computed
            if (keep && qb + 1 < ntile && wave + NW * i < ntile) kb_next[i] = keep[((long)(qb + 1) * ntile + wave + NW * i) * 64];
        }
        const float Lt = sL[qb * 16 + q] * LOG2E, dlt = sDelta[qb * 16 + q];
        float Ln[4], dln[4], adO[4], aQ[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = qb * 16 + 4 * kq + r;
            Ln[r] = sL[row] * LOG2E; dln[r] = sDelta[row];
            adO[r] = q < D ? sdO[row * DP + q] : 0.f;              // dO^T[d = q][query 4 kq + r]
            aQ[r] = q < D ? P.scale * sQ[row * DP + q] : 0.f;      // scale * Q^T[d = q][query 4 kq + r]
        }
        f32x4 dq = {0.f, 0.f, 0.f, 0.f};                          // dQ^T[d = 4 kq + r][query = q], this wave's key tiles only
#pragma unroll
        for (int i = 0; i < NKT; ++i) {
            const int kt = wave + NW * i;
            if (kt < ntile) {                                     // wave-uniform
                const f32x4 st = tile_xyT<NS>(sK, kt * 16, sQ, qb * 16, DP, q, kq);       // S^T: rows keys
                const f32x4 dpt = tile_xyT<NS>(sV, kt * 16, sdO, qb * 16, DP, q, kq);
                const f32x4 sn = tile_xyT<NS>(sQ, qb * 16, sK, kt * 16, DP, q, kq);       // S: rows queries
                const f32x4 dpn = tile_xyT<NS>(sdO, qb * 16, sV, kt * 16, DP, q, kq);
                if (drop) {
                    const unsigned long long b0 = __ballot(kb[i] & 1u), b1 = __ballot(kb[i] & 2u), b2 = __ballot(kb[i] & 4u), b3 = __ballot(kb[i] & 8u);
                    const unsigned mine = (unsigned)((bal_sel == 0 ? b0 : bal_sel == 1 ? b1 : bal_sel == 2 ? b2 : b3) >> bal_shift);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float kst = (kb[i] >> r) & 1u ? P.inv_keep : 0.f, ksn = (mine >> r) & 1u ? P.inv_keep : 0.f;
                        const float pt = __builtin_amdgcn_exp2f(st[r] * sc2 - Lt);
                        // dQ^T[d][query] += (scale K^T)[d][key] (dS^T / scale)[key][query]: the score registers are the B operand as they stand
                        dq = __builtin_amdgcn_mfma_f32_16x16x4f32(ak[i][r], pt * (dpt[r] * kst - dlt), dq, 0, 0, 0);
                        const float pn = __builtin_amdgcn_exp2f(sn[r] * sc2 - Ln[r]);
                        // dV^T[d][key] += dO^T[d][query] Pd[query][key];  dK^T[d][key] += (scale Q^T)[d][query] (dS / scale)[query][key]
                        dv[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(adO[r], pn * ksn, dv[i], 0, 0, 0);
                        dk[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(aQ[r], pn * (dpn[r] * ksn - dln[r]), dk[i], 0, 0, 0);
                    }
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float pt = __builtin_amdgcn_exp2f(st[r] * sc2 - Lt);
                        dq = __builtin_amdgcn_mfma_f32_16x16x4f32(ak[i][r], pt * (dpt[r] - dlt), dq, 0, 0, 0);
                        const float pn = __builtin_amdgcn_exp2f(sn[r] * sc2 - Ln[r]);
                        dv[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(adO[r], pn, dv[i], 0, 0, 0);
                        dk[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(aQ[r], pn * (dpn[r] - dln[r]), dk[i], 0, 0, 0);
                    }
                }
            }
        }
        // the eight partial dQ^T tiles of this query block -> LDS; 256 threads add them in wave order
        float* rb = sR + (qb & 1) * (NW * 256);
        *reinterpret_cast<f32x4*>(rb + wave * 256 + lane * 4) = dq;
        __syncthreads();
        if (t < 256) {
            float sum = 0.f;
#pragma unroll
            for (int w = 0; w < NW; ++w) sum += rb[w * 256 + t];
            const int ql = (t >> 2) & 15, d = 4 * (t >> 6) + (t & 3), qrow = qb * 16 + ql;
            if (d < D && qrow < B) P.dqkv[(long)qrow * 3 * F + h * D + d] = sum;
        }
        // the buffer written two blocks later is this one: every thread passes the next block's barrier only after these reads
    }
#pragma unroll
    for (int i = 0; i < NKT; ++i) {
        const int key = (wave + NW * i) * 16 + q;
        if (wave + NW * i < ntile && key < B && 4 * kq < D) {
            float* dvp = P.dqkv + (long)key * 3 * F + 2 * F + h * D + 4 * kq;
            float* dkp = P.dqkv + (long)key * 3 * F + F + h * D + 4 * kq;
#pragma unroll
            for (int r = 0; r < 4; ++r) { dvp[r] = dv[i][r]; dkp[r] = dk[i][r]; }
        }
    }
}

size_t fwd_lds(int B, int D) { const size_t Bp = (B + 15) & ~15; return 3 * Bp * (D + 1) * sizeof(float); }
size_t bwd_lds(int B, int D) { const size_t Bp = (B + 15) & ~15; return (4 * Bp * (D + 1) + 2 * Bp + NW * 2 * 16 * 17) * sizeof(float); }
size_t bwd1_lds(int B, int D) { const size_t Bp = (B + 15) & ~15; return (4 * Bp * (D + 1) + 2 * Bp + 2 * NW * 256) * sizeof(float); }
constexpr int BWD1_NKT = 4;            // key tiles per wave in the single-sweep backward: B <= 16 * NW * BWD1_NKT = 512
constexpr size_t LDS_MAX = 160 * 1024;
// few heads (F = 64: 8) cannot fill the chip by themselves: split a head's query blocks / key tiles over several work-groups
int head_parts(int nhead, int B) {
    int parts = 256 / (nhead > 0 ? nhead : 1);
    const int blocks = (B + 15) / 16;
    if (parts > (blocks + NW - 1) / NW) parts = (blocks + NW - 1) / NW;
    return parts < 1 ? 1 : (parts > 16 ? 16 : parts);
}

// forward only (the single-sweep backward wants a whole head per work-group): BBBP_ATTN_FWD_WGS = target number of work-groups.  Default 512:
// at F = 2048 (256 heads) a head's query blocks are split over two groups, and two groups per CU cover each other's exp / Philox / LDS
// latencies: 135 -> 124 us per layer at B = 512 (256: one group per head, 1024: 138 us)
int fwd_parts(int nhead, int B) {
    static const int target = [] { const char* e = getenv("BBBP_ATTN_FWD_WGS"); return e ? atoi(e) : 512; }();
    int parts = (target > 0 ? target : 512) / (nhead > 0 ? nhead : 1);
    const int blocks = (B + 15) / 16;
    if (parts > (blocks + NW - 1) / NW) parts = (blocks + NW - 1) / NW;
    return parts < 1 ? 1 : (parts > 16 ? 16 : parts);
}

// ------------------------------------------------------------------------------------------------------------------------------
// ONE (or a few) WIDE heads: the MACCS width F = 167 is prime, the reference's head rule gives nhead = 1, head_dim = 167 (R:71-73).
// As separate launches the attention of a layer was QK^T -> softmax(+dropout) -> PV forward and (dV | dPd) -> softmax backward ->
// (dQ | dK) backward: seven small launches around [B, B] tensors on a latency-bound chain.  Here it is one launch each way.
//   * operands come STRAIGHT from global memory (L2: Q, K, V, dO of B = 512 are 342 KB each) into the MFMA register layout, as in
//     gemm.hip's latency path: a row is read as 16-byte quads (lane (i, kq) holds X[row i][16c + 4kq + j]; MFMA j of chunk c then
//     contracts k = 16c + 4kq + j -- a permutation of the head dimension applied to both operands alike), the transposed operands
//     (V^T, K^T, Q^T, dO^T) as four consecutive columns of a row (column groups of 64: tile (g, u) row i <-> d = 64g + 4i + u);
//   * forward: a work-group owns 16 queries, its 8 waves split the key tiles; every wave runs its own online softmax and the
//     (max, sum, O^T) partials are merged through LDS.  Scores go back into the MFMA as B operand from the registers they were
//     produced in (rows of the accumulator layout are the contraction index), as in the small-head kernel above;
//   * backward: blockIdx.z = 0: key-owner sweep (dV^T, dK^T of 16 keys; tiles computed as S = Q K^T so that the contraction
//     index -- the query -- is the accumulator row); blockIdx.z = 1: query-owner sweep (dQ^T of 16 queries; tiles as S^T).  P is
//     recomputed from the saved logsumexp; delta = rowsum(dO * O) is recomputed from the operands' own quads.  4 waves per group
//     split the other index; partials are summed through LDS in wave order.  No atomics, bit-reproducible.
constexpr int WNC = 11;                 // 16-deep chunks of the head dimension: 160 < head_dim <= 176
constexpr int WDG = 3;                  // 64-wide column groups: head_dim <= 192
constexpr int WFW = 8, WBW = 4;         // waves per work-group: forward, backward
typedef float f32x4w __attribute__((ext_vector_type(4), aligned(4)));

__device__ __forceinline__ void wide_quads(const float* row, int D, int kq, f32x4 (&v)[WNC]) {
#pragma unroll
    for (int c = 0; c < WNC; ++c) {
        const int k = 16 * c + 4 * kq;
        if (c < WNC - 1) {
            const f32x4w x = *reinterpret_cast<const f32x4w*>(row + k);
            v[c] = f32x4{x[0], x[1], x[2], x[3]};
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) { const float x = row[min(k + j, D - 1)]; v[c][j] = k + j < D ? x : 0.f; }
        }
    }
}
// columns min(64g + 4i, D - 4) .. + 3 of a row
__device__ __forceinline__ void wide_cols(const float* row, int D, int i, f32x4 (&v)[WDG]) {
#pragma unroll
    for (int g = 0; g < WDG; ++g) {
        const f32x4w x = *reinterpret_cast<const f32x4w*>(row + min(64 * g + 4 * i, D - 4));
        v[g] = f32x4{x[0], x[1], x[2], x[3]};
    }
}
// [16 rows of a] x [16 rows of b]^T over the head dimension: lane (col = b row, kq), register r <-> a row 4 kq + r
__device__ __forceinline__ f32x4 wide_dot(const f32x4 (&a)[WNC], const f32x4 (&b)[WNC]) {
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < WNC; ++c)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[c][j], b[c][j], acc, 0, 0, 0);
    return acc;
}
__device__ __forceinline__ float wide_sum_kq(float v) { v += __shfl_xor(v, 16); v += __shfl_xor(v, 32); return v; }
__device__ __forceinline__ float wide_dot_rows(const f32x4 (&a)[WNC], const f32x4 (&b)[WNC]) {
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < WNC; ++c)
#pragma unroll
        for (int j = 0; j < 4; ++j) s += a[c][j] * b[c][j];
    return wide_sum_kq(s);
}
// element (group g, sub-tile u, register r) of a transposed accumulator held by lane (col, kq): head-dimension index, or -1 when the
// slot is a duplicate of a clamped quad / beyond the head dimension
__device__ __forceinline__ int wide_d_of(int g, int u, int r, int kq, int D) {
    const int start = 64 * g + 4 * (4 * kq + r);
    const int d = min(start, D - 4) + u;
    return (d >= start && d < D) ? d : -1;
}

__global__ __launch_bounds__(64 * WFW) void attn_wide_fwd_kernel(AttnParams P) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* so = smem;                                   // [WFW][48][64]
    float* sm = so + WFW * 48 * 64;                     // [WFW][16]
    float* sl = sm + WFW * 16;                          // [WFW][16]
    float* sw = sl + WFW * 16;                          // [WFW][16] merge weights; [WFW * 16 ..]: logsumexp of the 16 queries
    const int B = P.B, F = P.F, D = P.D, ld = 3 * F, h = blockIdx.x, qb = blockIdx.y;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, q = lane & 15, kq = lane >> 4;
    const float* Qb = P.qkv + h * D; const float* Kb = Qb + F; const float* Vb = Qb + 2 * F;
    const int query = qb * 16 + q;
    const uint64_t seed = effective_seed(P.seed, P.seed_base);
    const bool drop = P.p > 0.f;
    const int ntile = (B + 15) >> 4;

    f32x4 qv[WNC];
    wide_quads(Qb + (long)min(query, B - 1) * ld, D, kq, qv);
    float m = -INFINITY, l = 0.f;
    f32x4 o[WDG][4];
#pragma unroll
    for (int g = 0; g < WDG; ++g)
#pragma unroll
        for (int u = 0; u < 4; ++u) o[g][u] = f32x4{0.f, 0.f, 0.f, 0.f};

    for (int kt = wave; kt < ntile; kt += WFW) {
        f32x4 kv[WNC];
        wide_quads(Kb + (long)min(kt * 16 + q, B - 1) * ld, D, kq, kv);
        f32x4 vc[4][WDG];
#pragma unroll
        for (int r = 0; r < 4; ++r) wide_cols(Vb + (long)min(kt * 16 + 4 * kq + r, B - 1) * ld, D, q, vc[r]);
        f32x4 s = wide_dot(kv, qv);                     // S^T[key 4 kq + r][query q]
        float cm = -INFINITY;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int key = kt * 16 + 4 * kq + r;
            s[r] = key < B ? s[r] * P.scale : -INFINITY;
            cm = fmaxf(cm, s[r]);
        }
        cm = fmaxf(cm, __shfl_xor(cm, 16)); cm = fmaxf(cm, __shfl_xor(cm, 32));
        const float mn = fmaxf(m, cm);
        const float corr = __expf(m - mn);
        float ks[4] = {1.f, 1.f, 1.f, 1.f};
        if (drop) keep4(seed, ((uint64_t)h * B + min(query, B - 1)) * B + kt * 16 + 4 * kq, P.p, P.inv_keep, ks);
        float ps = 0.f, pd[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) { const float pv = __expf(s[r] - mn); ps += pv; pd[r] = pv * ks[r]; }
#pragma unroll
        for (int g = 0; g < WDG; ++g)
#pragma unroll
            for (int u = 0; u < 4; ++u) o[g][u] = o[g][u] * corr;
        // O^T[d][query] += V^T[d][key] Pd^T[key][query]: step r contracts keys 4 kq + r
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int g = 0; g < WDG; ++g)
#pragma unroll
                for (int u = 0; u < 4; ++u) o[g][u] = __builtin_amdgcn_mfma_f32_16x16x4f32(vc[r][g][u], pd[r], o[g][u], 0, 0, 0);
        l = l * corr + wide_sum_kq(ps);
        m = mn;
    }
    // ---- merge the waves' partials ----
#pragma unroll
    for (int g = 0; g < WDG; ++g)
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int r = 0; r < 4; ++r) so[(wave * 48 + (g * 4 + u) * 4 + r) * 64 + lane] = o[g][u][r];
    if (kq == 0) { sm[wave * 16 + q] = m; sl[wave * 16 + q] = l; }
    __syncthreads();
    if (t < 16) {
        float M = -INFINITY;
        for (int w = 0; w < WFW; ++w) M = fmaxf(M, sm[w * 16 + t]);
        float L = 0.f;
        for (int w = 0; w < WFW; ++w) L += sl[w * 16 + t] * __expf(sm[w * 16 + t] - M);
        const float inv = 1.f / L;
        for (int w = 0; w < WFW; ++w) sw[w * 16 + t] = __expf(sm[w * 16 + t] - M) * inv;
        if (qb * 16 + t < B) P.lse[(long)h * B + qb * 16 + t] = M + __logf(L);
    }
    __syncthreads();
    for (int idx = t; idx < 48 * 64; idx += 64 * WFW) {
        const int a = idx >> 6, ln = idx & 63, qq = ln & 15, kk = ln >> 4;
        const int d = wide_d_of(a >> 4, (a >> 2) & 3, a & 3, kk, D);
        const int qy = qb * 16 + qq;
        if (d < 0 || qy >= B) continue;
        float v = 0.f;
#pragma unroll
        for (int w = 0; w < WFW; ++w) v += so[(w * 48 + a) * 64 + ln] * sw[w * 16 + qq];
        P.ctx[(long)qy * F + h * D + d] = v;
    }
}

__device__ __forceinline__ float keep1(uint64_t seed, uint64_t idx, float p, float inv_keep) {
    const uint4 a = philox4(seed, idx >> 2);
    const int off = (int)(idx & 3);
    const uint32_t v = off == 0 ? a.x : off == 1 ? a.y : off == 2 ? a.z : a.w;
    return ((float)(v >> 8) * (1.0f / 16777216.0f)) >= p ? inv_keep : 0.f;
}

__global__ __launch_bounds__(64 * WBW) void attn_wide_bwd_kernel(AttnParams P) {
    extern __shared__ __attribute__((aligned(16))) float smem[];       // [WBW][96][64]
    const int B = P.B, F = P.F, D = P.D, ld = 3 * F, h = blockIdx.x, own = blockIdx.y;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, q = lane & 15, kq = lane >> 4;
    const float* Qb = P.qkv + h * D; const float* Kb = Qb + F; const float* Vb = Qb + 2 * F;
    const float* dOb = P.dctx + h * D; const float* Ob = P.ctx + h * D;
    const uint64_t seed = effective_seed(P.seed, P.seed_base);
    const bool drop = P.p > 0.f;
    const int ntile = (B + 15) >> 4;
    f32x4 acc0[WDG][4], acc1[WDG][4];
#pragma unroll
    for (int g = 0; g < WDG; ++g)
#pragma unroll
        for (int u = 0; u < 4; ++u) { acc0[g][u] = f32x4{0.f, 0.f, 0.f, 0.f}; acc1[g][u] = f32x4{0.f, 0.f, 0.f, 0.f}; }

    if (blockIdx.z == 0) {
        // ---- key owner: dV^T[d][key], dK^T[d][key] of keys own * 16 .. + 15; tiles S[query 4 kq + r][key q] ----
        const int key = own * 16 + q;
        f32x4 kv[WNC], vv[WNC];
        wide_quads(Kb + (long)min(key, B - 1) * ld, D, kq, kv);
        wide_quads(Vb + (long)min(key, B - 1) * ld, D, kq, vv);
        for (int qt = wave; qt < ntile; qt += WBW) {
            const int qrow = min(qt * 16 + q, B - 1);
            f32x4 qa[WNC], doa[WNC];
            wide_quads(Qb + (long)qrow * ld, D, kq, qa);
            wide_quads(dOb + (long)qrow * F, D, kq, doa);
            float delta_i, lse_i;
            {
                f32x4 ca[WNC];
                wide_quads(Ob + (long)qrow * F, D, kq, ca);
                delta_i = wide_dot_rows(doa, ca);
                lse_i = qt * 16 + q < B ? P.lse[(long)h * B + qt * 16 + q] : INFINITY;
            }
            const f32x4 s = wide_dot(qa, kv);
            const f32x4 dp = wide_dot(doa, vv);
            float pd[4], ds[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int qy = qt * 16 + 4 * kq + r;
                const float dl = __shfl(delta_i, 4 * kq + r), L = __shfl(lse_i, 4 * kq + r);
                const float keep = drop ? keep1(seed, ((uint64_t)h * B + min(qy, B - 1)) * B + min(key, B - 1), P.p, P.inv_keep) : 1.f;
                const float pv = key < B ? __expf(s[r] * P.scale - L) : 0.f;
                pd[r] = pv * keep;
                ds[r] = pv * (dp[r] * keep - dl) * P.scale;
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int qy = min(qt * 16 + 4 * kq + r, B - 1);
                f32x4 doc[WDG], qc[WDG];
                wide_cols(dOb + (long)qy * F, D, q, doc);
                wide_cols(Qb + (long)qy * ld, D, q, qc);
#pragma unroll
                for (int g = 0; g < WDG; ++g)
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        acc0[g][u] = __builtin_amdgcn_mfma_f32_16x16x4f32(doc[g][u], pd[r], acc0[g][u], 0, 0, 0);
                        acc1[g][u] = __builtin_amdgcn_mfma_f32_16x16x4f32(qc[g][u], ds[r], acc1[g][u], 0, 0, 0);
                    }
            }
        }
    } else {
        // ---- query owner: dQ^T[d][query] of queries own * 16 .. + 15; tiles S^T[key 4 kq + r][query q] ----
        const int query = own * 16 + q;
        const int qrow = min(query, B - 1);
        f32x4 qv[WNC], dov[WNC];
        wide_quads(Qb + (long)qrow * ld, D, kq, qv);
        wide_quads(dOb + (long)qrow * F, D, kq, dov);
        float delta;
        {
            f32x4 cv[WNC];
            wide_quads(Ob + (long)qrow * F, D, kq, cv);
            delta = wide_dot_rows(dov, cv);
        }
        const float L = query < B ? P.lse[(long)h * B + query] : INFINITY;
        for (int kt = wave; kt < ntile; kt += WBW) {
            const int krow = min(kt * 16 + q, B - 1);
            f32x4 ka[WNC], va[WNC];
            wide_quads(Kb + (long)krow * ld, D, kq, ka);
            wide_quads(Vb + (long)krow * ld, D, kq, va);
            const f32x4 s = wide_dot(ka, qv);
            const f32x4 dp = wide_dot(va, dov);
            float ks[4] = {1.f, 1.f, 1.f, 1.f};
            if (drop) keep4(seed, ((uint64_t)h * B + qrow) * B + kt * 16 + 4 * kq, P.p, P.inv_keep, ks);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int key = kt * 16 + 4 * kq + r;
                const float pv = key < B ? __expf(s[r] * P.scale - L) : 0.f;
                const float ds = pv * (dp[r] * ks[r] - delta) * P.scale;
                f32x4 kc[WDG];
                wide_cols(Kb + (long)min(key, B - 1) * ld, D, q, kc);
#pragma unroll
                for (int g = 0; g < WDG; ++g)
#pragma unroll
                    for (int u = 0; u < 4; ++u) acc0[g][u] = __builtin_amdgcn_mfma_f32_16x16x4f32(kc[g][u], ds, acc0[g][u], 0, 0, 0);
            }
        }
    }
    // ---- sum the waves' partials in wave order ----
    const int nacc = blockIdx.z == 0 ? 96 : 48;
#pragma unroll
    for (int g = 0; g < WDG; ++g)
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                smem[(wave * 96 + (g * 4 + u) * 4 + r) * 64 + lane] = acc0[g][u][r];
                if (blockIdx.z == 0) smem[(wave * 96 + 48 + (g * 4 + u) * 4 + r) * 64 + lane] = acc1[g][u][r];
            }
    __syncthreads();
    for (int idx = t; idx < nacc * 64; idx += 64 * WBW) {
        const int a = idx >> 6, ln = idx & 63, cc = ln & 15, kk = ln >> 4, a48 = a % 48;
        const int d = wide_d_of(a48 >> 4, (a48 >> 2) & 3, a48 & 3, kk, D);
        const int row = own * 16 + cc;                  // key (sweep A) or query (sweep B)
        if (d < 0 || row >= B) continue;
        float v = 0.f;
#pragma unroll
        for (int w = 0; w < WBW; ++w) v += smem[(w * 96 + a) * 64 + ln];
        // sweep A: a < 48 -> dV (third block of dqkv), else dK (second); sweep B: dQ (first)
        const int blk = blockIdx.z == 0 ? (a < 48 ? 2 : 1) : 0;
        P.dqkv[(long)row * ld + blk * F + h * D + d] = v;
    }
}

constexpr size_t WIDE_FWD_LDS = (size_t)(WFW * 48 * 64 + 4 * WFW * 16) * sizeof(float);
constexpr size_t WIDE_BWD_LDS = (size_t)(WBW * 96 * 64) * sizeof(float);
bool wide_supported(int B, int nhead, int head_dim) {
    return nhead >= 1 && nhead <= 8 && head_dim > 16 * (WNC - 1) && head_dim <= 16 * WNC && B >= 1 && (B + 15) / 16 <= 65535;
}

template <class K>
int set_dyn_lds(K kernel, size_t bytes) {
    if (bytes <= 64 * 1024) return BBBP_OK;
    return bbbp_ensure_dyn_lds(reinterpret_cast<const void*>(kernel), bytes);
}

}  // namespace

// The fused path serves head_dim 8 and 16 whenever one head's K, V, Q, dO fit in LDS (B <= 1024 at head_dim 8).
static bool small_supported(int B, int nhead, int head_dim) {
    return nhead > 1 && (head_dim == 8 || head_dim == 16) && B >= 1 && bwd_lds(B, head_dim) <= LDS_MAX;
}
bool bbbp_attn_small_supported(int B, int nhead, int head_dim) { return small_supported(B, nhead, head_dim); }
// ... and one (or a few) wide heads of 161 .. 176 columns (the MACCS width 167 with the reference's nhead = 1) on attn_wide_*; opt-in
// (bbbp_set_flash_attention bit 1): measured at B = 512 the two launches take 28 / 62 us alone against ~18 / ~30 us for the seven
// latency-path launches they replace, because 32 work-groups cannot use more than 32 CUs' matrix pipes.
bool bbbp_attn_wide_supported(int B, int nhead, int head_dim) { return wide_supported(B, nhead, head_dim); }

// bytes of the dropout keep mask the small-head forward can leave for the backward pass (0: the shape runs on the wide-head kernels)
size_t bbbp_attn_small_keep_bytes(int B, int nhead, int head_dim) {
    if (!small_supported(B, nhead, head_dim)) return 0;
    const size_t nt = (B + 15) / 16;
    return (size_t)nhead * nt * nt * 64;
}

int bbbp_attn_small_fwd(hipStream_t st, const float* qkv, float* ctx, float* lse, int B, int F, int nhead, float scale, float p, uint64_t seed,
                        uint8_t* keep) {
    const int D = F / nhead;
    BBBP_CHECK_ARG(small_supported(B, nhead, D) || wide_supported(B, nhead, D), "attn_small: B=%d nhead=%d head_dim=%d not supported", B, nhead, D);
    AttnParams P{qkv, ctx, lse, nullptr, nullptr, B, F, nhead, D, scale, p, p > 0.f ? 1.f / (1.f - p) : 1.f, seed, g_bbbp_seed_base, keep};
    if (!small_supported(B, nhead, D)) {
        int rc = set_dyn_lds(attn_wide_fwd_kernel, WIDE_FWD_LDS); if (rc) return rc;
        hipLaunchKernelGGL(attn_wide_fwd_kernel, dim3(nhead, (B + 15) / 16), dim3(64 * WFW), WIDE_FWD_LDS, st, P);
        BBBP_CHECK_LAUNCH();
        return BBBP_OK;
    }
    const size_t lds = fwd_lds(B, D);
    if (D == 8) { int rc = set_dyn_lds(attn_small_fwd_kernel<2>, lds); if (rc) return rc; hipLaunchKernelGGL(attn_small_fwd_kernel<2>, dim3(nhead, fwd_parts(nhead, B)), dim3(NTH), lds, st, P); }
    else { int rc = set_dyn_lds(attn_small_fwd_kernel<4>, lds); if (rc) return rc; hipLaunchKernelGGL(attn_small_fwd_kernel<4>, dim3(nhead, fwd_parts(nhead, B)), dim3(NTH), lds, st, P); }
    BBBP_CHECK_LAUNCH();
    return BBBP_OK;
}

int bbbp_attn_small_bwd(hipStream_t st, const float* qkv, const float* ctx, const float* lse, const float* dctx, float* dqkv, int B, int F,
                        int nhead, float scale, float p, uint64_t seed, const uint8_t* keep) {
    const int D = F / nhead;
    BBBP_CHECK_ARG(small_supported(B, nhead, D) || wide_supported(B, nhead, D), "attn_small: B=%d nhead=%d head_dim=%d not supported", B, nhead, D);
    AttnParams P{qkv, const_cast<float*>(ctx), const_cast<float*>(lse), dctx, dqkv, B, F, nhead, D, scale, p, p > 0.f ? 1.f / (1.f - p) : 1.f, seed,
                 g_bbbp_seed_base, const_cast<uint8_t*>(keep)};
    if (!small_supported(B, nhead, D)) {
        int rc = set_dyn_lds(attn_wide_bwd_kernel, WIDE_BWD_LDS); if (rc) return rc;
        hipLaunchKernelGGL(attn_wide_bwd_kernel, dim3(nhead, (B + 15) / 16, 2), dim3(64 * WBW), WIDE_BWD_LDS, st, P);
        BBBP_CHECK_LAUNCH();
        return BBBP_OK;
    }
    // one work-group per head and at most NW * BWD1_NKT key tiles: the single-sweep kernel (BBBP_ATTN_BWD1=0: the two-sweep kernel)
    static const int bwd1 = [] { const char* e = getenv("BBBP_ATTN_BWD1"); return e ? atoi(e) : 1; }();
    if (bwd1 && (p == 0.f || keep != nullptr) && head_parts(nhead, B) == 1 && (B + 15) / 16 <= NW * BWD1_NKT && bwd1_lds(B, D) <= LDS_MAX) {
        const size_t lds1 = bwd1_lds(B, D);
        if (D == 8) { int rc = set_dyn_lds(attn_small_bwd1_kernel<2, BWD1_NKT>, lds1); if (rc) return rc; hipLaunchKernelGGL((attn_small_bwd1_kernel<2, BWD1_NKT>), dim3(nhead), dim3(NTH), lds1, st, P); }
        else { int rc = set_dyn_lds(attn_small_bwd1_kernel<4, BWD1_NKT>, lds1); if (rc) return rc; hipLaunchKernelGGL((attn_small_bwd1_kernel<4, BWD1_NKT>), dim3(nhead), dim3(NTH), lds1, st, P); }
        BBBP_CHECK_LAUNCH();
        return BBBP_OK;
    }
    const size_t lds = bwd_lds(B, D);
    if (D == 8) { int rc = set_dyn_lds(attn_small_bwd_kernel<2>, lds); if (rc) return rc; hipLaunchKernelGGL(attn_small_bwd_kernel<2>, dim3(nhead, head_parts(nhead, B)), dim3(NTH), lds, st, P); }
    else { int rc = set_dyn_lds(attn_small_bwd_kernel<4>, lds); if (rc) return rc; hipLaunchKernelGGL(attn_small_bwd_kernel<4>, dim3(nhead, head_parts(nhead, B)), dim3(NTH), lds, st, P); }
    BBBP_CHECK_LAUNCH();
    return BBBP_OK;
}
