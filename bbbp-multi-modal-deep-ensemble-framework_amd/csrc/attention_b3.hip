// Self-attention forward for ONE (or a few) WIDE heads at screening batch sizes, on the BF16 matrix pipe with float32 operands split
// into three bf16 pieces (common.h: split2; six bf16 MFMAs per float32 product block, f32 accumulate -- float32 accuracy).
//
// The MACCS encoder (F = 167, prime => nhead = 1, head_dim = 167: ...20250113.py:71-78) attends across the B molecules of the batch; at
// the screening batch of BASELINE config 5 (B = 4096, eval mode: no dropout, nothing kept for a backward pass) that is 11.2 GFLOP per
// layer.  attention.hip's attn_wide_fwd_kernel runs it on v_mfma_f32_16x16x4_f32 with operands straight from L2 at 61 TFLOP/s
// (0.18 ms per layer, 1.1 of config 5's 7.2 ms).  Here:
//   * a work-group of 8 waves owns 128 queries (16 per wave: Q^T pre-scaled by scale * log2(e), pre-split, in 72 registers per lane)
//     and a RANGE of keys (blockIdx.z: the key axis is split so that 4096 queries still fill the chip; partial (max, sum, O) triples are
//     merged by a second, tiny kernel);
//   * a tile of 32 keys is staged ONCE per work-group in LDS, already split: K row-major ([plane][key][192 + 8] bf16) and V TRANSPOSED
//     ([plane][d][32 + 8]) with the keys of a row in the order the score accumulators hold them -- so that
//   * S^T = K Q^T (v_mfma_f32_16x16x32_bf16, two 16-key sub-tiles) leaves lane (query, g) with eight probabilities that ARE its
//     B-operand fragment of O^T += V^T P^T: no transpose, no LDS round trip for P; the online softmax (running max / sum per query,
//     exp2) touches each score once;
//   * the head dimension is padded to 192 in LDS and registers only (zeros), never in HBM.
// 144 MFMAs of 16 cycles per wave and tile (2304 matrix cycles for 32 keys x 16 queries x 192 x 2 products).
#include "common.h"
#include "attention_b3.h"
#include <stdlib.h>

namespace {

constexpr int DPAD = 192;                  // padded head dimension: six k-steps of 32
constexpr int KSTEPS = DPAD / 32;
constexpr int NWV = 8;                     // waves per work-group, 16 queries each
constexpr int QBLK = 16 * NWV;             // queries per work-group
constexpr int KT = 32;                     // keys per tile
// row strides chosen by exhaustive search over the lane groups ds_read_b128 serves together ({0-3,12-15,20-27}, {4-11,16-19,28-31}, ...):
// 104 / 24 dwords are conflict-free for the fragment pattern row = lane & 15, column quad = lane >> 4 (100 / 20 dwords were 2-way)
constexpr int K_LD = DPAD + 16;            // bf16 per key row (416 B)
constexpr int V_LD = KT + 16;              // bf16 per head-dimension row (96 B)
constexpr int K_PLANE = KT * K_LD;
constexpr int V_PLANE = DPAD * V_LD;
constexpr size_t LDS_BYTES = (size_t)(3 * K_PLANE + 3 * V_PLANE) * 2;      // 95 KB: one work-group per CU (its 8 waves x 232 registers fill the SIMDs anyway)
constexpr int NTH = 64 * NWV;
constexpr int QUADS = KT * (DPAD / 4);     // float4 items of a tile per operand
constexpr int NIT = QUADS / NTH;           // 3 per thread

struct B3AttnParams {
    const float* qkv;                  // [B][3F]: Q | K | V, head h at columns h D .. of each third
    float* ctx;                        // [B][F]
    float* part;                       // [nsplit][NH][B][D + 2]: unnormalised O, running max (log2 domain), running sum
    int B, F, NH, D;
    float qscale;                      // softmax scale * log2(e), folded into Q
    int nsplit, keys_per_split;
    int exp;                           // experiment bits (BBBP_ATTN_B3_EXP): 1 no S MFMAs, 2 no PV MFMAs, 4 no tile stores, 8 no tile loads
};

typedef float f32x4w __attribute__((ext_vector_type(4), aligned(4)));

// eight floats -> three bf16x8 planes
__device__ __forceinline__ void split8(const float (&v)[8], bf16x8 (&out)[3]) {
    uint32_t hi[4], mid[4], lo[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) split2(v[2 * j], v[2 * j + 1], hi[j], mid[j], lo[j]);
    out[0] = __builtin_bit_cast(bf16x8, u32x4{hi[0], hi[1], hi[2], hi[3]});
    out[1] = __builtin_bit_cast(bf16x8, u32x4{mid[0], mid[1], mid[2], mid[3]});
    out[2] = __builtin_bit_cast(bf16x8, u32x4{lo[0], lo[1], lo[2], lo[3]});
}
// the six piece products of a split-bf16 block, smallest terms first
__device__ __forceinline__ f32x4 mfma6(const bf16x8 (&a)[3], const bf16x8 (&b)[3], f32x4 c) {
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1], b[1], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[2], b[0], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[2], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1], b[0], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[1], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[0], c, 0, 0, 0);
    return c;
}
// position of key k (0..31) of a tile inside a V^T row: lane (query, g) of the score accumulators holds keys 4 g + r of sub-tile 0 and
// 16 + 4 g + r of sub-tile 1; as the B operand of a 32-deep MFMA step it supplies k-slots 8 g .. 8 g + 7 in that order
__device__ __forceinline__ int v_slot(int k) { return k < 16 ? 8 * (k >> 2) + (k & 3) : 8 * ((k - 16) >> 2) + 4 + (k & 3); }

__global__ __launch_bounds__(NTH) void attn_b3_fwd_kernel(B3AttnParams p) {
    extern __shared__ __attribute__((aligned(16))) uint16_t smem[];
    uint16_t* Ks = smem;                    // [plane][key][K_LD]
    uint16_t* Vt = smem + 3 * K_PLANE;      // [plane][d][V_LD]
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, q = lane & 15, g = lane >> 4;
    const int B = p.B, D = p.D, ld = 3 * p.F, h = blockIdx.y, split = blockIdx.z;
    const float* Qb = p.qkv + h * D; const float* Kb = Qb + p.F; const float* Vb = Qb + 2 * p.F;
    const int query = blockIdx.x * QBLK + wave * 16 + q;
    const int kbeg = split * p.keys_per_split, kend = min(B, kbeg + p.keys_per_split);
    const int ntiles = (kend - kbeg + KT - 1) / KT;

    // ---- Q^T fragments: lane (query q, g) holds d = 32 ks + 8 g .. + 7 of its query, scaled, split ----
    bf16x8 qf[KSTEPS][3];
    {
        const float* row = Qb + (long)min(query, B - 1) * ld;
        const bool qok = query < B;
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ++ks) {
            float v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int d = 32 * ks + 8 * g + e;
                const float x = row[min(d, D - 1)];
                v[e] = (qok && d < D) ? x * p.qscale : 0.f;
            }
            split8(v, qf[ks]);
        }
    }

    // ---- tile loader.  K: item = (key, quad of the head dimension): one 16-byte load, one 8-byte LDS write per plane (lanes along d:
    //      coalesced, conflict-free).  V goes in TRANSPOSED: item = (quad of keys, d): four dword loads (each coalesced along d across the
    //      wave), and the four keys of a quad are four consecutive positions of a V^T row -- one 8-byte LDS write per plane (the first
    //      version wrote 2-byte pieces at a 320-byte lane stride: 32-way bank conflicts, the kernel's bottleneck) ----
    int ikey[NIT], id0[NIT], vkq[NIT], vd[NIT];
#pragma unroll
    for (int i = 0; i < NIT; ++i) {
        const int idx = t + i * NTH;
        ikey[i] = idx / (DPAD / 4); id0[i] = 4 * (idx % (DPAD / 4));
        vkq[i] = idx / DPAD; vd[i] = idx % DPAD;                    // 8 key quads x 192 d
    }
    // loads are UNCONDITIONAL (clamped addresses; the zeroing selects are applied when the registers are split and written to LDS): a load
    // under a per-item `if` makes hipcc wait for it at the join, i.e. before this tile's MFMAs instead of after them
    f32x4 kr[NIT], vr[NIT];
    int kd0c[NIT];
#pragma unroll
    for (int i = 0; i < NIT; ++i) kd0c[i] = min(id0[i], D - 4);       // a quad that would overrun the row's head slice is read shifted
    int tile_k0 = 0;
    auto load_tile = [&](int k0) __attribute__((always_inline)) {
        tile_k0 = k0;
#pragma unroll
        for (int i = 0; i < NIT; ++i) {
            const long roff = (long)min(k0 + ikey[i], B - 1) * ld;
            const f32x4w x = *reinterpret_cast<const f32x4w*>(Kb + roff + kd0c[i]);
            kr[i] = f32x4{x[0], x[1], x[2], x[3]};
            const int d = min(vd[i], D - 1);
#pragma unroll
            for (int j = 0; j < 4; ++j) vr[i][j] = Vb[(long)min(k0 + 4 * vkq[i] + j, B - 1) * ld + d];
        }
    };
    auto store_tile = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < NIT; ++i) {
            const bool kok = tile_k0 + ikey[i] < kend;
            const int sh = id0[i] - kd0c[i];                         // 0 for whole quads; 1..3 for the quad that straddles D; >= 4 beyond D
            float kv[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int src = e + sh;
                const float x = src == 0 ? kr[i][0] : src == 1 ? kr[i][1] : src == 2 ? kr[i][2] : kr[i][3];
                kv[e] = (kok && id0[i] + e < D) ? x : 0.f;
            }
            uint32_t h0, m0, l0, h1, m1, l1;
            split2(kv[0], kv[1], h0, m0, l0);
            split2(kv[2], kv[3], h1, m1, l1);
            uint16_t* kd = Ks + ikey[i] * K_LD + id0[i];
            *reinterpret_cast<u32x2*>(kd) = u32x2{h0, h1};
            *reinterpret_cast<u32x2*>(kd + K_PLANE) = u32x2{m0, m1};
            *reinterpret_cast<u32x2*>(kd + 2 * K_PLANE) = u32x2{l0, l1};
            float vv[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) vv[j] = (vd[i] < D && tile_k0 + 4 * vkq[i] + j < kend) ? vr[i][j] : 0.f;
            split2(vv[0], vv[1], h0, m0, l0);
            split2(vv[2], vv[3], h1, m1, l1);
            uint16_t* vdst = Vt + vd[i] * V_LD + v_slot(4 * vkq[i]);   // keys 4 kq .. + 3 sit at four consecutive positions
            *reinterpret_cast<u32x2*>(vdst) = u32x2{h0, h1};
            *reinterpret_cast<u32x2*>(vdst + V_PLANE) = u32x2{m0, m1};
            *reinterpret_cast<u32x2*>(vdst + 2 * V_PLANE) = u32x2{l0, l1};
        }
    };

    float m = -INFINITY, lsum = 0.f;                 // running max (log2 domain) of this lane's query; this lane's share of the sum
    f32x4 o[DPAD / 16];                              // O^T[d = 16 dt + 4 g + r][query q]
#pragma unroll
    for (int dt = 0; dt < DPAD / 16; ++dt) o[dt] = f32x4{0.f, 0.f, 0.f, 0.f};

    if (ntiles > 0) { load_tile(kbeg); store_tile(); }
    __syncthreads();
    const uint16_t* kfrag = Ks + q * K_LD + 8 * g;              // + 16 t rows, + 32 ks columns, + plane
    const uint16_t* vfrag = Vt + q * V_LD + 8 * g;              // + 16 dt rows, + plane
    for (int tile = 0; tile < ntiles; ++tile) {
        const int k0 = kbeg + tile * KT;
        if (tile + 1 < ntiles && !(p.exp & 8)) load_tile(k0 + KT);      // global loads in flight under this tile's MFMAs
        // ---- S^T = K Q^T for the two 16-key sub-tiles: twelve (sub-tile, k-step) groups of six MFMAs; the three fragments of group
        //      i + 1 are fetched while group i runs (hipcc otherwise reads, waits and issues group by group) ----
        f32x4 s[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
        bf16x8 fa[2][3];
        auto fetch_k = [&](int gi, int slot) __attribute__((always_inline)) {
#pragma unroll
            for (int pl = 0; pl < 3; ++pl)
                fa[slot][pl] = *reinterpret_cast<const bf16x8*>(kfrag + pl * K_PLANE + (gi / KSTEPS) * 16 * K_LD + 32 * (gi % KSTEPS));
        };
        auto fetch_v = [&](int dt, int slot) __attribute__((always_inline)) {
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) fa[slot][pl] = *reinterpret_cast<const bf16x8*>(vfrag + pl * V_PLANE + dt * 16 * V_LD);
        };
        fetch_k(0, 0);
#pragma unroll
        for (int gi = 0; gi < 2 * KSTEPS; ++gi) {
            if (gi + 1 < 2 * KSTEPS) fetch_k(gi + 1, (gi + 1) & 1);
            else fetch_v(0, (gi + 1) & 1);                       // the first V^T fragments travel under the softmax arithmetic
            if (!(p.exp & 1)) s[gi / KSTEPS] = mfma6(fa[gi & 1], qf[gi % KSTEPS], s[gi / KSTEPS]);
        }
        // ---- online softmax: lane (q, g) holds keys k0 + 16 st + 4 g + r of query q ----
        float cm = -INFINITY;
#pragma unroll
        for (int st = 0; st < 2; ++st)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (k0 + 16 * st + 4 * g + r >= kend) s[st][r] = -INFINITY;
                cm = fmaxf(cm, s[st][r]);
            }
        cm = fmaxf(cm, __shfl_xor(cm, 16)); cm = fmaxf(cm, __shfl_xor(cm, 32));
        const float mn = fmaxf(m, cm);                           // finite: every tile holds at least one valid key
        const float corr = __builtin_amdgcn_exp2f(m - mn);       // 0 on the first tile (m = -inf)
        float pv[8];
        float ps = 0.f;
#pragma unroll
        for (int st = 0; st < 2; ++st)
#pragma unroll
            for (int r = 0; r < 4; ++r) { pv[4 * st + r] = __builtin_amdgcn_exp2f(s[st][r] - mn); ps += pv[4 * st + r]; }
        lsum = lsum * corr + ps;
        m = mn;
        bf16x8 pb[3];
        split8(pv, pb);
        // ---- O^T = O^T * corr + V^T P^T ----
#pragma unroll
        for (int dt = 0; dt < DPAD / 16; ++dt) {
            if (dt + 1 < DPAD / 16) fetch_v(dt + 1, (dt + 1) & 1);
            if (!(p.exp & 2)) o[dt] = mfma6(fa[dt & 1], pb, o[dt] * corr);
        }
        __syncthreads();                                         // every wave is done reading this tile
        if (tile + 1 < ntiles && !(p.exp & 4)) store_tile();
        __syncthreads();
    }
    lsum += __shfl_xor(lsum, 16); lsum += __shfl_xor(lsum, 32);
    if (query >= B) return;
    if (p.nsplit == 1) {
        const float inv = 1.f / lsum;
        float* dst = p.ctx + (long)query * p.F + h * D;
#pragma unroll
        for (int dt = 0; dt < DPAD / 16; ++dt)
#pragma unroll
            for (int r = 0; r < 4; ++r) { const int d = 16 * dt + 4 * g + r; if (d < D) dst[d] = o[dt][r] * inv; }
    } else {
        float* dst = p.part + (((long)split * p.NH + h) * B + query) * (D + 2);
#pragma unroll
        for (int dt = 0; dt < DPAD / 16; ++dt)
#pragma unroll
            for (int r = 0; r < 4; ++r) { const int d = 16 * dt + 4 * g + r; if (d < D) dst[d] = o[dt][r]; }
        if (g == 0) { dst[D] = m; dst[D + 1] = lsum; }
    }
}

// partial (max, sum, O) triples of the key ranges -> ctx: O = sum_i 2^(m_i - M) O_i / sum_i 2^(m_i - M) l_i, summed in range order
__global__ __launch_bounds__(256) void attn_b3_merge_kernel(B3AttnParams p) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);          // (head, query)
    if (row >= (long)p.NH * p.B) return;
    const int h = (int)(row / p.B), query = (int)(row % p.B), D = p.D;
    const long stride = (long)p.NH * p.B * (D + 2);
    const float* base = p.part + ((long)h * p.B + query) * (D + 2);
    float M = -INFINITY;
    for (int i = 0; i < p.nsplit; ++i) M = fmaxf(M, base[i * stride + D]);
    float L = 0.f;
    for (int i = 0; i < p.nsplit; ++i) L += __builtin_amdgcn_exp2f(base[i * stride + D] - M) * base[i * stride + D + 1];
    const float inv = 1.f / L;
    for (int d = lane; d < D; d += 64) {
        float acc = 0.f;
        for (int i = 0; i < p.nsplit; ++i) acc += __builtin_amdgcn_exp2f(base[i * stride + D] - M) * base[i * stride + d];
        p.ctx[(long)query * p.F + h * D + d] = acc * inv;
    }
}

int plan_splits(int B, int NH, int* keys_per_split) {
    const int nqb = (B + QBLK - 1) / QBLK;
    static const int wg_per_cu_x2 = [] { const char* e = getenv("BBBP_ATTN_B3_WG_X2"); return e ? atoi(e) : 2; }();      // tuning: work-groups per CU x 2 (one per CU: two cannot co-reside)
    int ns = (wg_per_cu_x2 * bbbp_num_cus() / 2 + nqb * NH - 1) / (nqb * NH);
    const int max_ns = (B + 255) / 256;                                 // at least 256 keys (8 tiles) per range
    if (ns > max_ns) ns = max_ns;
    if (ns < 1) ns = 1;
    int kps = ((B + ns - 1) / ns + KT - 1) / KT * KT;
    ns = (B + kps - 1) / kps;
    *keys_per_split = kps;
    return ns;
}

}  // namespace

bool bbbp_attn_b3_supported(int B, int nhead, int head_dim) {
    return nhead >= 1 && nhead <= 16 && head_dim > 96 && head_dim <= DPAD && B >= 1 && (B + QBLK - 1) / QBLK <= 65535;
}

size_t bbbp_attn_b3_workspace_bytes(int B, int nhead, int head_dim) {
    if (!bbbp_attn_b3_supported(B, nhead, head_dim)) return 0;
    int kps;
    const int ns = plan_splits(B, nhead, &kps);
    return ns > 1 ? (size_t)ns * nhead * B * (head_dim + 2) * sizeof(float) : 0;
}

int bbbp_attn_b3_fwd(hipStream_t st, const float* qkv, float* ctx, int B, int F, int nhead, float scale, float* workspace, size_t workspace_bytes) {
    const int D = F / nhead;
    BBBP_CHECK_ARG(bbbp_attn_b3_supported(B, nhead, D) && F == nhead * D, "attn_b3: B=%d nhead=%d head_dim=%d not supported", B, nhead, D);
    BBBP_CHECK_ARG(qkv && ctx, "attn_b3: null pointer");
    static const int exp_bits = [] { const char* e = getenv("BBBP_ATTN_B3_EXP"); return e ? atoi(e) : 0; }();
    B3AttnParams p{qkv, ctx, workspace, B, F, nhead, D, scale * 1.4426950408889634f, 1, 0, exp_bits};
    p.nsplit = plan_splits(B, nhead, &p.keys_per_split);
    if (p.nsplit > 1) BBBP_CHECK_ARG(workspace && workspace_bytes >= bbbp_attn_b3_workspace_bytes(B, nhead, D), "attn_b3: workspace too small");
    { int rc = bbbp_ensure_dyn_lds(reinterpret_cast<const void*>(attn_b3_fwd_kernel), LDS_BYTES); if (rc) return rc; }
    hipLaunchKernelGGL(attn_b3_fwd_kernel, dim3((B + QBLK - 1) / QBLK, nhead, p.nsplit), dim3(NTH), LDS_BYTES, st, p);
    BBBP_CHECK_LAUNCH();
    if (p.nsplit > 1) {
        const long rows = (long)nhead * B;
        hipLaunchKernelGGL(attn_b3_merge_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, st, p);
        BBBP_CHECK_LAUNCH();
    }
    return BBBP_OK;
}
