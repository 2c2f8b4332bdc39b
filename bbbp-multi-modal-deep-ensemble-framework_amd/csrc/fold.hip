// out_proj folded into the value projection of a ONE-head encoder layer (nn.TransformerEncoderLayer as the reference builds it at
// Models/multi_input_data_regression_opt_transformer_cnn_20250113.py:75-78 for a prime fingerprint width: 167 -> nhead 1).
//
// With one head, ctx = Pd V and out_proj(ctx) = Pd (V Wo^T) + bo with V = x Wv^T + bv, so
//     V Wo^T = x (Wo Wv)^T + 1 (Wo bv)^T =: x W'^T + 1 b'^T                     (W' = Wo Wv, b' = Wo bv: weights only)
// and a layer's attention block is  [Q | K | VW] = x [Wq; Wk; W']^T + [bq; bk; b'],  z = Pd VW + bo:  the out_proj GEMM leaves the
// forward chain, its input-gradient GEMM leaves the backward chain (dVW = Pd^T dz, dPd = dz VW^T take dz directly), and the
// K = B weight-gradient GEMM of out_proj becomes two K = F products of the folded gradient:
//     dW'|db' = dVW^T [x | 1]          (the V block of the in_proj weight-gradient GEMM, as before)
//     dWo = dW' Wv^T + db' bv^T,   d[Wv | bv] = Wo^T [dW' | db'],   dbo = column sums of dz.
// Every launch on the encoder's chain costs 11-20 us at B = 512 whatever its size, so two fewer per layer and
// direction is what this buys; the products here are 167^3 and run as plain float32 FMA loops in a fixed order (bit-reproducible).
#include "common.h"

namespace {

constexpr int FOLD_MAX_F = 256;          // a reduction or a row of columns covers at most F + 1 <= 256 elements
constexpr int NW = 8, NT = 64 * NW;      // waves / threads of a work-group: K is split eight ways
constexpr int UR = 4;                    // output rows per work-group
constexpr int KB = 7;                    // loads in flight per lane and round (21 k per wave at F = 167: three rounds); 64 VGPRs, no scratch (8+: spills at the cap)
constexpr int FILL_IT = (UR * (FOLD_MAX_F + KB) + NT - 1) / NT;      // LDS-fill loads per thread, all issued before the first wait

struct FoldLds {
    float u[8][FOLD_MAX_F + KB];         // rows 0 .. UR-1: the work-group's rows of the row operand, all k (8 rows: the transposing copy)
    float red[NW][UR][64];               // the waves' partial sums
};

// C[r0 + rr][c] = sum_k U[r0 + rr][k] V[k][c] for UR rows and the 64 columns c = c0 + lane: the rows of U sit in LDS (every lane reads the
// same word: a broadcast), V is read coalesced, K is split over the eight waves and the partial sums are added in wave order --
// bit-reproducible.  The kernels are bound by load LATENCY (167^3 products, everything L2-resident), so every phase issues all of its
// loads before it waits: the LDS fill (the first version waited per element: 6 round trips), then KB operand loads per round.
// ROW_FAST: consecutive lanes of the LDS fill take consecutive ROWS of U (for a U stored k-major, i.e. read as U^T)
template <bool ROW_FAST = false, class LoadU, class LoadV, class Store>
__device__ __forceinline__ void small_product(FoldLds& L, int K, int rows, int cols, int r0, int c0, LoadU lu, LoadV lv, Store st) {
    const int t = threadIdx.x, w = t >> 6, lane = t & 63, c = c0 + lane;
    const int Kp = K + KB;                  // the KB words past K are read (times a zero operand) by the last round: they must be finite
    float fill[FILL_IT];
#pragma unroll
    for (int i = 0; i < FILL_IT; ++i) {
        const int idx = t + i * NT;
        const int rr = ROW_FAST ? idx % UR : idx / Kp, k = ROW_FAST ? idx / UR : idx - rr * Kp;
        fill[i] = (idx < UR * Kp && r0 + rr < rows && k < K) ? lu(r0 + rr, k) : 0.f;
    }
#pragma unroll
    for (int i = 0; i < FILL_IT; ++i) {
        const int idx = t + i * NT;
        const int rr = ROW_FAST ? idx % UR : idx / Kp, k = ROW_FAST ? idx / UR : idx - rr * Kp;
        if (idx < UR * Kp) L.u[rr][k] = fill[i];
    }
    __syncthreads();
    const int kq = (K + NW - 1) / NW, k0 = w * kq, k1 = min(K, k0 + kq);
    const bool c_ok = c < cols;
    float acc[UR];
#pragma unroll
    for (int rr = 0; rr < UR; ++rr) acc[rr] = 0.f;
    for (int k = k0; k < k1; k += KB) {
        float v[KB];
#pragma unroll
        for (int j = 0; j < KB; ++j) v[j] = (c_ok && k + j < k1) ? lv(k + j, c) : 0.f;
#pragma unroll
        for (int j = 0; j < KB; ++j)
#pragma unroll
            for (int rr = 0; rr < UR; ++rr) acc[rr] = fmaf(L.u[rr][k + j], v[j], acc[rr]);          // k + j < K + KB: zero words times zero
    }
#pragma unroll
    for (int rr = 0; rr < UR; ++rr) L.red[w][rr][lane] = acc[rr];
    __syncthreads();
    if (w == 0 && c_ok) {
#pragma unroll
        for (int rr = 0; rr < UR; ++rr)
            if (r0 + rr < rows) {
                float sum = L.red[0][rr][lane];
#pragma unroll
                for (int g = 1; g < NW; ++g) sum += L.red[g][rr][lane];
                st(r0 + rr, c, sum);
            }
    }
}

struct FoldArgs {
    const float* win[32]; const float* bin[32]; const float* wo[32]; const float* bo[32];      // bo: added to b' when not null
    float* wf[32]; float* bf[32]; float* wvt[32];
    int F;
};

// grid (nrt * nct + copy blocks, L), nrt = ceil(F / UR) row tiles, nct = ceil((F + 1) / 64) column tiles of W' | b' = Wo [Wv | bv];
// the copy blocks write [Wq; Wk | bq; bk] (rows 0 .. 2F-1 of wf | bf) and [Wv | bv]^T (F + 1 rows of F), 8 rows each
__global__ __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu(8, 8))) void outproj_fold_kernel(FoldArgs a) {
    BBBP_HIGH_PRIO();
    __shared__ FoldLds L;
    const int l = blockIdx.y, F = a.F, t = threadIdx.x;
    const float* __restrict__ win = a.win[l];
    const float* __restrict__ bin = a.bin[l];
    const float* __restrict__ wo = a.wo[l];
    const float* __restrict__ wv = win + (long)2 * F * F;
    const float* __restrict__ bv = bin + 2 * F;
    float* __restrict__ wf = a.wf[l];
    float* __restrict__ bf = a.bf[l];
    const float* __restrict__ bo = a.bo[l];
    const int nrt = (F + UR - 1) / UR, nct = (F + 1 + 63) / 64;
    int b = blockIdx.x;
    if (b < nrt * nct) {
        const int r0 = (b / nct) * UR, c0 = (b % nct) * 64;
        small_product(L, F, F, F + 1, r0, c0,
                      [&](int r, int k) { return wo[(long)r * F + k]; },
                      [&](int k, int c) { return c < F ? wv[(long)k * F + c] : bv[k]; },
                      [&](int r, int c, float v) { if (c < F) wf[(long)(2 * F + r) * F + c] = v; else bf[2 * F + r] = bo ? v + bo[r] : v; });
        return;
    }
    b -= nrt * nct;
    const int ncopy = (2 * F + 7) / 8;
    if (b < ncopy) {
        for (int idx = t; idx < 8 * (F + 1); idx += NT) {
            const int r = b * 8 + idx / (F + 1), c = idx % (F + 1);
            if (r < 2 * F) { if (c < F) wf[(long)r * F + c] = win[(long)r * F + c]; else bf[r] = bin[r]; }
        }
        return;
    }
    b -= ncopy;
    // [Wv | bv]^T: rows c = 8 b .. + 7 of F columns k; read through LDS so that both sides are coalesced
    for (int idx = t; idx < 8 * F; idx += NT) {
        const int k = idx / 8, cc = idx % 8, c = b * 8 + cc;
        L.u[cc][k] = c < F ? wv[(long)k * F + c] : (c == F ? bv[k] : 0.f);
    }
    __syncthreads();
    for (int idx = t; idx < 8 * F; idx += NT) {
        const int cc = idx / F, k = idx % F, c = b * 8 + cc;
        if (c <= F) a.wvt[l][(long)c * F + k] = L.u[cc][k];
    }
}

struct UnfoldArgs {
    const float* tdw; const float* tdb;          // dW' [F][F], db' [F]
    const float* wo; const float* wvt;           // Wo [F][F]; [Wv | bv]^T [F + 1][F]
    const float* dz; int lddz; int B;            // gradient of the out_proj output (pre-dropout), [B][F]
    float* g_outw; float* g_outb;                // out_proj.weight / bias gradients
    float* g_inw_v; float* g_inb_v;              // rows 2F .. 3F-1 of the in_proj weight / bias gradients
    int F;
};

// work-groups: nrt x ceil(F / 64) tiles of dWo = [dW' | db'] [Wv | bv]^T; nrt x ceil((F + 1) / 64) tiles of d[Wv | bv] = Wo^T [dW' | db'];
// then 32 columns of dbo = column sums of dz each
__global__ __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu(8, 8))) void outproj_unfold_kernel(UnfoldArgs a) {
    BBBP_HIGH_PRIO();
    __shared__ FoldLds L;
    const int F = a.F, t = threadIdx.x;
    const float* __restrict__ tdw = a.tdw;
    const float* __restrict__ tdb = a.tdb;
    const int nrt = (F + UR - 1) / UR, nca = (F + 63) / 64, ncb = (F + 1 + 63) / 64;
    int b = blockIdx.x;
    if (b < nrt * nca) {
        const float* __restrict__ wvt = a.wvt;
        float* __restrict__ out = a.g_outw;
        small_product(L, F + 1, F, F, (b / nca) * UR, (b % nca) * 64,
                      [&](int r, int k) { return k < F ? tdw[(long)r * F + k] : tdb[r]; },
                      [&](int k, int c) { return wvt[(long)k * F + c]; },
                      [&](int r, int c, float v) { out[(long)r * F + c] = v; });
        return;
    }
    b -= nrt * nca;
    if (b < nrt * ncb) {
        const float* __restrict__ wo = a.wo;
        float* __restrict__ gw = a.g_inw_v;
        float* __restrict__ gb = a.g_inb_v;
        small_product<true>(L, F, F, F + 1, (b / ncb) * UR, (b % ncb) * 64,
                      [&](int r, int k) { return wo[(long)k * F + r]; },
                      [&](int k, int c) { return c < F ? tdw[(long)k * F + c] : tdb[k]; },
                      [&](int r, int c, float v) { if (c < F) gw[(long)r * F + c] = v; else gb[r] = v; });
        return;
    }
    b -= nrt * ncb;
    // column sums of dz: 32 columns per work-group, 16 row groups, added in group order
    float* part = &L.red[0][0][0];               // [16][32]
    const int cc = t & 31, rg = t >> 5, c = b * 32 + cc;
    float s = 0.f;
    if (c < F)
        for (int m = rg; m < a.B; m += NT / 32) s += a.dz[(long)m * a.lddz + c];
    part[rg * 32 + cc] = s;
    __syncthreads();
    if (rg == 0 && c < F) {
        float tot = 0.f;
#pragma unroll
        for (int g = 0; g < NT / 32; ++g) tot += part[g * 32 + cc];
        a.g_outb[c] = tot;
    }
}

}  // namespace

bool bbbp_outproj_fold_supported(int F, int nhead, int layers) { return nhead == 1 && F >= 1 && F <= FOLD_MAX_F - 1 && layers >= 1 && layers <= 32; }

// per layer: wf [3F][F], bf [3F], wvt [F + 1][F], tdw [F][F], tdb [F]
size_t bbbp_outproj_fold_floats(int F, int which) {
    const size_t f = (size_t)F;
    return which == 0 ? 3 * f * f : which == 1 ? 3 * f : which == 2 ? (f + 1) * f : which == 3 ? f * f : f;
}

int bbbp_outproj_fold(hipStream_t st, int layers, int F, const float* const* win, const float* const* bin, const float* const* wo,
                      const float* const* bo, float* const* wf, float* const* bf, float* const* wvt) {
    BBBP_CHECK_ARG(bbbp_outproj_fold_supported(F, 1, layers), "outproj_fold: F %d, %d layers", F, layers);
    FoldArgs a;
    for (int l = 0; l < layers; ++l) {
        BBBP_CHECK_ARG(win[l] && bin[l] && wo[l] && wf[l] && bf[l] && wvt[l], "outproj_fold: null pointer (layer %d)", l);
        a.win[l] = win[l]; a.bin[l] = bin[l]; a.wo[l] = wo[l]; a.bo[l] = bo ? bo[l] : nullptr; a.wf[l] = wf[l]; a.bf[l] = bf[l]; a.wvt[l] = wvt[l];
    }
    for (int l = layers; l < 32; ++l) { a.win[l] = a.bin[l] = a.wo[l] = a.bo[l] = nullptr; a.wf[l] = a.bf[l] = a.wvt[l] = nullptr; }
    a.F = F;
    const int nrt = (F + UR - 1) / UR, nct = (F + 1 + 63) / 64;
    hipLaunchKernelGGL(outproj_fold_kernel, dim3(nrt * nct + (2 * F + 7) / 8 + (F + 1 + 7) / 8, layers), dim3(NT), g_bbbp_small_lds_pad, st, a);
    BBBP_CHECK_LAUNCH();
    return BBBP_OK;
}

int bbbp_outproj_unfold(hipStream_t st, int F, int B, const float* tdw, const float* tdb, const float* wo, const float* wvt, const float* dz,
                        int lddz, float* g_outw, float* g_outb, float* g_inw_v, float* g_inb_v) {
    BBBP_CHECK_ARG(bbbp_outproj_fold_supported(F, 1, 1) && B >= 1 && lddz >= F, "outproj_unfold: F %d, B %d, lddz %d", F, B, lddz);
    BBBP_CHECK_ARG(tdw && tdb && wo && wvt && dz && g_outw && g_outb && g_inw_v && g_inb_v, "outproj_unfold: null pointer");
    UnfoldArgs a{tdw, tdb, wo, wvt, dz, lddz, B, g_outw, g_outb, g_inw_v, g_inb_v, F};
    const int nrt = (F + UR - 1) / UR;
    hipLaunchKernelGGL(outproj_unfold_kernel, dim3(nrt * ((F + 63) / 64) + nrt * ((F + 1 + 63) / 64) + (F + 31) / 32), dim3(NT), g_bbbp_small_lds_pad, st, a);
    BBBP_CHECK_LAUNCH();
    return BBBP_OK;
}
