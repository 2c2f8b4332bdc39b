// Whole-model orchestration of MixedInputModel forward / backward on one MI355X: the host-side
// schedule that strings the gfx950 kernels (gemm.hip, conv.hip, rowops.hip) together on one HIP stream
// with a single caller-provided workspace (no allocation, no synchronisation: graph-capturable).
//
// Reference: MixedInputModel.forward, Models/multi_input_data_regression_opt_transformer_cnn_20250113.py:109-119
// (R below) and its autograd under loss.backward() (R:190).  The encoder is called with a [B,1,F]
// tensor and batch_first=False (R:110-111), so self-attention runs ACROSS the mini-batch: S = B, N = 1.
#include "common.h"
#include "attention_b3.h"
#include "encoder_sliced.h"
#include "head_sync.h"
#include <optional>
#include "bbbp_hip.h"
#include <mutex>
#include <stdlib.h>

namespace {

constexpr int IMG = 128, C1 = 32, C2 = 64, FC = 128, NHEADS_FUSION = 4, FUS_HID = 128;
constexpr int H1 = 256, H2 = 128, H3 = 64;
constexpr int COMB = 2 * FC;          // 256
constexpr int IMG_FLAT = C2 * (IMG / 4) * (IMG / 4);   // 65536

// parameter indices in named_parameters() order
enum { L_INW = 0, L_INB, L_OUTW, L_OUTB, L_W1, L_B1, L_W2, L_B2, L_N1W, L_N1B, L_N2W, L_N2B, L_COUNT };
struct PIdx {
    int L, NF;             // NF: parameter tensors of the fusion block (16 for MultiHeadAttentionFusion, 0 for torch.cat)
    PIdx(int layers, int fusion) : L(layers), NF(fusion == 0 ? 4 * NHEADS_FUSION : 0) {}
    explicit PIdx(const bbbp_mixed_desc* d) : PIdx(d->num_layers, d->fusion) {}
    int layer(int l, int k) const { return l * L_COUNT + k; }
    int base() const { return L * L_COUNT; }
    int fpfc_w() const { return base() + 0; }
    int fpfc_b() const { return base() + 1; }
    int c1_w() const { return base() + 2; }
    int c1_b() const { return base() + 3; }
    int c2_w() const { return base() + 4; }
    int c2_b() const { return base() + 5; }
    int ifc_w() const { return base() + 6; }
    int ifc_b() const { return base() + 7; }
    int fus(int h, int k) const { return base() + 8 + h * 4 + k; }     // k: 0 W1, 1 b1, 2 w2, 3 b2
    int fc0_w() const { return base() + 8 + NF; }
    int fc0_b() const { return base() + 9 + NF; }
    int bn_w() const { return base() + 10 + NF; }
    int bn_b() const { return base() + 11 + NF; }
    int fc3_w() const { return base() + 12 + NF; }
    int fc3_b() const { return base() + 13 + NF; }
    int fc5_w() const { return base() + 14 + NF; }
    int fc5_b() const { return base() + 15 + NF; }
    int fc7_w() const { return base() + 16 + NF; }
    int fc7_b() const { return base() + 17 + NF; }
    int count() const { return base() + 18 + NF; }
};

struct Bump {
    size_t off = 0;
    size_t take(size_t bytes) { size_t o = off; off = align_up(off + bytes, 256); return o; }
    size_t f(size_t n) { return take(n * sizeof(float)); }
};

struct LayerOff { size_t qkv, prob, pd, ctx, z1, y1, hff, z2, y2, mean1, rstd1, mean2, rstd2, lse, keep, kvg; };      // keep: 0 = none; kvg: exact-global-batch mode only
// per-layer gradient buffers: weight-gradient kernels read them on a third stream while the dependency
// chain moves on to the next layer, so they must not be recycled within one backward pass
struct LayerGrad { size_t dyout, dz2, dz2d, dhff, dy1, dz1, dz1d, dqkv; };

struct Plan {
    int B, F, NH, D, L, DFF;
    bool drop, concat, inference;
    bool flash;            // many heads of head_dim 8 / 16: fused attention (attention.hip), no [NH,B,B] probability tensors
    size_t sl_sync, sl_part;   // sliced persistent forward: barrier counters, linear2 partials (allocated whenever the shape is supported)
    size_t sl_kvpart;          // sliced persistent backward: every row block's share of dK | dV (training plans only)
    // exact-global-batch mode (bbbp_mixed_desc.collective): B is this rank's shard, keys / values / BatchNorm statistics span Bg = world * B rows
    bool exact; int world, rank; size_t Bg;
    size_t dkvg, dkvl;         // this rank's queries' share of dK | dV for all Bg keys; the reduce-scattered rows of this rank
    bool attn_b3;          // forward-only plan, wide head, >= 1024 rows: split-bf16 attention (attention_b3.hip)
    // one-head layers on the launch-per-op schedule: out_proj folded into the value projection (fold.hip); per layer the folded in_proj
    // weight / bias, [Wv | bv]^T, and (training plans) the folded gradient dW' | db'
    bool fold; size_t fwf[32], fbf[32], fwvt[32], ftdw[32], ftdb[32];
    size_t attn_part, attn_part_bytes;
    LayerOff layer[32];
    LayerGrad lgrad[32];
    size_t scratch3, scratch3_bytes;
    size_t seed_slot;      // the call's dropout seed, in device memory (common.h: effective_seed)
    size_t pool1, mask1, pool2, mask2;
    size_t combined, hid, attn, fused, h, hb, bn_mean, bn_rstd, h2, h3, head_partial;
    // temporaries
    size_t scratch, scratch_bytes, scratch2, scratch2_bytes;
    size_t dA, dB, dqkv, dprob, dctx, dhff, dpool2, dpool1, dcomb, dfused, dlogit, dpre, dh, dhb, dh2, dh3;
    size_t total;
};

// Fused attention for many small heads (attention.hip); 0 selects the GEMM + softmax schedule with materialised probabilities.
int g_flash_attention = -1;

// out_proj folded into the value projection for one-head layers (fold.hip): two launches fewer per layer on the encoder's forward and
// backward chains.  A bit mask (default 3): bit 0 the launch-per-op schedule with materialised probabilities; bit 1 also the
// forward-only split-bf16 attention kernel of 2048+ row plans -- there the encoder alone gets faster (1.88 -> 1.76 ms at B = 4096); beside the
// conv kernels on a second stream the step measured slower with it (7.54 -> 7.79 ms; profiles/r03_fold_outproj.txt), which is one of the
// reasons those plans run on one stream (forward_enqueue).  BBBP_FOLD_OUTPROJ=0 / bbbp_set_fold_outproj(0) keep the reference's operation order.
int g_fold_outproj = -1;
int g_fused_encoder = -1;              // bit 0: row-fused kernels (opt-in); bits 1 / 2: sliced persistent forward / backward for small batches
int fold_outproj_on() {
    if (g_fold_outproj < 0) { const char* e = getenv("BBBP_FOLD_OUTPROJ"); g_fold_outproj = e ? (atoi(e) & 3) : 3; }
    return g_fold_outproj;
}
// LayerNorm absorbed by the consuming Linear (gemm.hip: gemm_direct_lna_kernel).  Default OFF: measured SLOWER inside the B = 512 step
// (profiles/r04_ln_absorb.txt: the encoder's forward chain 0.826 -> 0.857 ms on the device timeline, step 2.487 -> 2.504 ms -- the absorbing
// GEMMs pay more for their row statistics / gamma / beta work beside the conv kernels than the twelve LayerNorm launches cost).
int g_ln_absorb = -1;
int ln_absorb_on() {
    if (g_ln_absorb < 0) { const char* e = getenv("BBBP_LN_ABSORB"); g_ln_absorb = e ? (atoi(e) != 0) : 0; }
    return g_ln_absorb;
}
int fused_encoder_mode() {
    if (g_fused_encoder < 0) { const char* e = getenv("BBBP_FUSED_ENCODER"); g_fused_encoder = e ? atoi(e) & 7 : 0; }
    return g_fused_encoder;
}

int make_plan(const bbbp_mixed_desc* d, Plan* p) {
    BBBP_CHECK_ARG(d, "null desc");
    BBBP_CHECK_ARG(d->batch >= 1, "batch %d < 1", d->batch);
    BBBP_CHECK_ARG(d->fingerprint_size >= 1 && d->nhead >= 1 && d->fingerprint_size % d->nhead == 0,
                   "fingerprint_size %d not divisible by nhead %d", d->fingerprint_size, d->nhead);
    BBBP_CHECK_ARG(d->num_layers >= 0 && d->num_layers <= 32, "num_layers %d not in [0, 32]", d->num_layers);
    BBBP_CHECK_ARG(d->dim_feedforward >= 1, "dim_feedforward");
    BBBP_CHECK_ARG(d->dropout_p >= 0.f && d->dropout_p < 1.f, "dropout_p %f", d->dropout_p);
    BBBP_CHECK_ARG(d->fusion == 0 || d->fusion == 1, "fusion %d (0 attention fusion, 1 concat)", d->fusion);
    BBBP_CHECK_ARG(!(d->inference && d->training), "an inference workspace cannot serve a training-mode call");
    p->concat = d->fusion == 1; p->inference = d->inference != 0;
    p->B = d->batch; p->F = d->fingerprint_size; p->NH = d->nhead; p->D = p->F / p->NH; p->L = d->num_layers;
    p->DFF = d->dim_feedforward;
    p->drop = d->training && d->dropout_p > 0.f;
    p->exact = d->collective != nullptr;
    BBBP_CHECK_ARG(p->exact || d->world <= 1, "world = %d needs a collective callback", d->world);
    p->world = p->exact && d->world > 1 ? d->world : 1; p->rank = p->exact ? d->rank : 0;
    BBBP_CHECK_ARG(p->rank >= 0 && p->rank < p->world, "rank %d of %d", d->rank, p->world);
    p->Bg = (size_t)p->B * p->world;
    if (g_flash_attention < 0) { const char* e = getenv("BBBP_FLASH_ATTENTION"); g_flash_attention = e ? atoi(e) & 31 : 13; }
    // bit 3 (round 3): forward-only plans of 2048 rows and more with a wide head run the split-bf16 attention kernel (attention_b3.hip);
    // bit 4: that kernel from 256 rows on (tests; below ~2048 rows its 128-query work-groups leave most CUs idle)
    p->attn_b3 = (g_flash_attention & 8) && p->inference && p->L > 0 && p->B >= ((g_flash_attention & 16) ? 256 : 2048) &&
                 bbbp_attn_b3_supported(p->B, p->NH, p->D) && !p->exact;
    // bit 2: the wide-head kernel where it wins -- forward-only plans of 2048 rows and more (256+ work-groups fill the chip and the
    // [B, B] probability tensor, 67 MB per layer at B = 4096, is never written): config 5 8.28 -> 7.84 ms per 4096 molecules
    const bool wide = (g_flash_attention & 2) || ((g_flash_attention & 4) && p->inference && p->B >= 2048);
    p->flash = p->L > 0 && !p->exact && (p->attn_b3 || ((g_flash_attention & 1) && bbbp_attn_small_supported(p->B, p->NH, p->D)) ||
                            (wide && bbbp_attn_wide_supported(p->B, p->NH, p->D)));
    // the fused / sliced encoder schedules (opt-in) and the fused attention kernels keep the reference's operation order
    // (the forward-only split-bf16 attention kernel reads VW where it read V: its rows of P sum to one, so bo rides in b')
    p->fold = (((fold_outproj_on() & 1) && !p->flash) || ((fold_outproj_on() & 2) && p->attn_b3)) && p->L > 0 && !p->exact && fused_encoder_mode() == 0 &&
              bbbp_outproj_fold_supported(p->F, p->NH, p->L);
    const size_t B = p->B, F = p->F, NH = p->NH, DFF = p->DFF, Bg = p->Bg;
    Bump b;
    p->seed_slot = b.take(256);
    for (int l = 0; l < p->L; ++l) {
        LayerOff& o = p->layer[l];
        // forward only: nothing is kept for a backward pass, every layer runs in layer 0's buffers (layer l reads its input
        // y2 before it writes y2 again: qkv <- y2, y1 <- LN(z1 + y2), y2 <- LN(z2 + y1))
        // (not for batches the sliced persistent forward can take: its row blocks run layers out of step, so a layer's buffers must not be
        // the previous layer's; at B <= 128 a layer's activations are < 2 MB)
        if (p->inference && l > 0 && !bbbp_enc_sliced_supported(p->B, p->F, p->NH, p->DFF, p->L)) { o = p->layer[0]; continue; }
        o.qkv = b.f(B * 3 * F); o.prob = p->flash ? 0 : b.f(NH * B * Bg); o.ctx = b.f(B * F);
        o.kvg = p->exact ? b.f(Bg * 2 * F) : 0;          // K | V of every rank's rows, kept for the backward pass
        // dropped attention weights are KEPT per layer (1 MB at B = 512, one head) rather than recomputed in backward:
        // every launch on the encoder's chain costs more than the bytes
        o.pd = (p->drop && !p->flash) ? b.f(NH * B * Bg) : o.prob;
        o.lse = b.f(NH * B);
        // fused small-head attention with dropout: the forward pass leaves its keep decisions (one byte per lane and tile pair, 17 MB
        // per layer at F = 2048, B = 512) for the backward pass, whose dominant cost was drawing them again
        {
            const size_t kbytes = (p->flash && p->drop && !p->inference) ? bbbp_attn_small_keep_bytes(p->B, p->NH, p->D) : 0;
            o.keep = kbytes ? b.take(kbytes) : 0;          // offset 0 is the seed slot: 0 means "no mask kept"
        }
        o.z1 = b.f(B * F); o.y1 = b.f(B * F); o.hff = b.f(B * DFF); o.z2 = b.f(B * F); o.y2 = b.f(B * F);
        o.mean1 = b.f(B); o.rstd1 = b.f(B); o.mean2 = b.f(B); o.rstd2 = b.f(B);
    }
    if (p->L > 0 && bbbp_enc_sliced_supported(p->B, p->F, p->NH, p->DFF, p->L)) {
        p->sl_sync = b.take(bbbp_enc_sliced_sync_bytes());
        p->sl_part = b.take(bbbp_enc_sliced_part_bytes(p->B, p->F));
        p->sl_kvpart = p->inference ? 0 : b.take(bbbp_enc_sliced_kvpart_bytes(p->B, p->F));
    } else {
        p->sl_sync = p->sl_part = p->sl_kvpart = 0;
    }
    for (int l = 0; l < p->L; ++l) {
        p->fwf[l] = p->fold ? b.f(bbbp_outproj_fold_floats(p->F, 0)) : 0; p->fbf[l] = p->fold ? b.f(bbbp_outproj_fold_floats(p->F, 1)) : 0;
        p->fwvt[l] = p->fold ? b.f(bbbp_outproj_fold_floats(p->F, 2)) : 0;
        p->ftdw[l] = (p->fold && !p->inference) ? b.f(bbbp_outproj_fold_floats(p->F, 3)) : 0;
        p->ftdb[l] = (p->fold && !p->inference) ? b.f(bbbp_outproj_fold_floats(p->F, 4)) : 0;
    }
    p->attn_part_bytes = p->attn_b3 ? bbbp_attn_b3_workspace_bytes(p->B, p->NH, p->D) : 0;
    p->attn_part = p->attn_part_bytes ? b.take(p->attn_part_bytes) : 0;
    // forward-only plans keep NO pooling decisions (nothing reads them: 0.8 GB per step at B = 4096): offset 0 = "no mask"
    p->pool1 = b.f(B * C1 * (IMG / 2) * (IMG / 2)); p->mask1 = p->inference ? 0 : b.take(B * C1 * (IMG / 2) * (IMG / 2));
    p->pool2 = b.f(B * IMG_FLAT); p->mask2 = p->inference ? 0 : b.take(B * IMG_FLAT);
    p->combined = b.f(B * COMB); p->hid = b.f(NHEADS_FUSION * B * FUS_HID); p->attn = b.f(B * NHEADS_FUSION);
    p->fused = p->concat ? p->combined : b.f(B * COMB); p->h = b.f(B * H1); p->hb = b.f(B * H1); p->bn_mean = b.f(H1); p->bn_rstd = b.f(H1);
    p->h2 = b.f(B * H2); p->h3 = b.f(B * H3);
    p->head_partial = b.f((size_t)p->world * ((B + 15) / 16) * 2 * H1);
    size_t cw = bbbp_conv3x3_workspace_bytes(p->B, 32, 64, 64, 64);
    size_t cw1 = bbbp_conv3x3_workspace_bytes(p->B, 3, 32, 128, 128);
    size_t sb = cw > cw1 ? cw : cw1;
    size_t gw = bbbp_gemm_workspace_bytes(p->B, FC, IMG_FLAT, 1);
    if (gw > sb) sb = gw;
    if (sb < ((size_t)32 << 20)) sb = (size_t)32 << 20;
    p->scratch_bytes = sb;
    p->scratch = b.take(sb);
    // the fingerprint branch runs on its own stream beside the image branch: it needs its own split-K scratch
    p->scratch2_bytes = (size_t)32 << 20;
    p->scratch2 = b.take(p->scratch2_bytes);
    if (p->inference) { p->total = b.off; return BBBP_OK; }
    // backward temporaries
    for (int l = 0; l < p->L; ++l) {
        LayerGrad& g = p->lgrad[l];
        g.dyout = b.f(B * F); g.dz2 = b.f(B * F); g.dz2d = p->drop ? b.f(B * F) : g.dz2; g.dhff = b.f(B * DFF);
        g.dy1 = b.f(B * F); g.dz1 = b.f(B * F); g.dz1d = p->drop ? b.f(B * F) : g.dz1; g.dqkv = b.f(B * 3 * F);
    }
    p->scratch3_bytes = (size_t)32 << 20;
    p->scratch3 = b.take(p->scratch3_bytes);
    p->dA = b.f(B * F); p->dB = b.f(B * F); p->dqkv = b.f(B * 3 * F); p->dprob = p->flash ? 0 : b.f(NH * B * Bg); p->dctx = b.f(B * F);
    p->dkvg = p->exact ? b.f(Bg * 2 * F) : 0; p->dkvl = p->exact ? b.f(B * 2 * F) : 0;
    p->dhff = b.f(B * DFF); p->dpool2 = b.f(B * IMG_FLAT); p->dpool1 = b.f(B * C1 * (IMG / 2) * (IMG / 2));
    p->dcomb = b.f(B * COMB); p->dfused = b.f(B * COMB); p->dlogit = b.f(NHEADS_FUSION * B);
    p->dpre = b.f(NHEADS_FUSION * B * FUS_HID); p->dh = b.f(B * H1); p->dhb = b.f(B * H1); p->dh2 = b.f(B * H2);
    p->dh3 = b.f(B * H3);
    p->total = b.off;
    return BBBP_OK;
}

// exact-global-batch mode: K | V of this rank's rows into its slot of the gathered buffer, and the reduce-scattered dK | dV back into dqkv
__global__ __launch_bounds__(256) void kv_pack_kernel(const float* __restrict__ qkv, float* __restrict__ kv, long n, int F) {
    BBBP_HIGH_PRIO();
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) { const long r = i / (2 * F); const int c = (int)(i % (2 * F)); kv[i] = qkv[r * 3 * F + F + c]; }
}
__global__ __launch_bounds__(256) void kv_unpack_kernel(const float* __restrict__ dkv, float* __restrict__ dqkv, long n, int F) {
    BBBP_HIGH_PRIO();
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) { const long r = i / (2 * F); const int c = (int)(i % (2 * F)); dqkv[r * 3 * F + F + c] = dkv[i]; }
}
int run_collective(const bbbp_mixed_desc* d, int op, int what, int layer, size_t send_off, size_t recv_off, size_t count, hipStream_t st) {
    BBBP_CHECK_ARG(d->collective, "no collective callback");
    const int rc = d->collective(d->collective_ctx, op, what, layer, send_off, recv_off, count, st);
    if (rc) { bbbp_set_error("collective callback (op %d, what %d, layer %d) returned %d", op, what, layer, rc); return BBBP_ERR_ARG; }
    return BBBP_OK;
}

struct Ctx {
    hipStream_t st;
    char* ws;
    const Plan* p;
    int side = 0;          // 0 main stream, 1 fingerprint-branch chain, 2 weight-gradient leaves: own scratch each
    float* f(size_t off) const { return reinterpret_cast<float*>(ws + off); }
    uint8_t* u8(size_t off) const { return reinterpret_cast<uint8_t*>(ws + off); }
    void* scratch() const { return ws + (side == 0 ? p->scratch : side == 1 ? p->scratch2 : p->scratch3); }
    size_t scratch_bytes() const { return side == 0 ? p->scratch_bytes : side == 1 ? p->scratch2_bytes : p->scratch3_bytes; }
};

// ---- two-branch overlap -------------------------------------------------------------------------
// The fingerprint branch (encoder, F = 167: ~270 launches of a few microseconds each, latency-bound) and the image
// branch (five persistent MFMA-bound conv kernels) are independent between the input and the fusion block.  They
// are enqueued on two HIP streams (fork/join with events, graph-capturable) so the small kernels run in the shadow
// of the conv kernels: their work-groups are sized to co-reside on a CU with a conv work-group.
constexpr int NEV = 32;
struct SideStream { hipStream_t s = nullptr; hipStream_t leaf = nullptr; hipEvent_t fork = nullptr, join = nullptr, join2 = nullptr;
                    hipEvent_t ev[NEV]; int next = 0; };
SideStream g_side[64];
// recorded on the caller's stream when the image-FC weight gradient (33.5 MB of the 53.9 MB of gradients at F = 167) is
// final, i.e. after the first GEMM of the image branch's backward: a data-parallel caller can start reducing that bucket
// while the remaining ~2 ms of the backward pass run (bbbp_mixed_backward_wait_bucket)
hipEvent_t g_bucket_event[64];
hipEvent_t g_bucket0_released[64];      // recorded after the last READ of the image-FC weight in a backward pass ...
bool g_release_events = false;          // ... only on request (bbbp_set_release_events): one more record on the image branch's stream
bool g_bucket0_released_recorded[64];
bool g_bucket_recorded[64];
// bucket 1: everything except the image-FC weight and the four conv tensors -- final when the fingerprint branch's chain
// and all weight-gradient leaves are done (~0.25 ms before the image branch's last kernel); recorded on the leaf stream
hipEvent_t g_bucket1_event[64];
bool g_bucket1_recorded[64];
// buckets 2 + l: the twelve tensors of encoder layer l (one contiguous slice in named_parameters order) -- final when that
// layer's weight-gradient leaves are, layer L-1 first.  At F = 2048 these are 100 MB each and 94 % of all gradient bytes.
hipEvent_t g_layer_event[64][32];
bool g_layer_recorded[64][32];
int record_layer_bucket(hipStream_t leaf, int layer) {
    int dev = 0;
    BBBP_CHECK_HIP(hipGetDevice(&dev));
    if (dev < 0 || dev >= 64 || layer < 0 || layer >= 32) return BBBP_OK;
    if (!g_layer_event[dev][layer]) BBBP_CHECK_HIP(hipEventCreateWithFlags(&g_layer_event[dev][layer], hipEventDisableTiming));
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    (void)hipStreamIsCapturing(leaf, &cap);
    g_layer_recorded[dev][layer] = cap == hipStreamCaptureStatusNone;       // an event recorded inside a graph capture means nothing outside it
    if (g_layer_recorded[dev][layer]) BBBP_CHECK_HIP(hipEventRecord(g_layer_event[dev][layer], leaf));
    return BBBP_OK;
}
int g_overlap = -1;
int g_fused_head_bwd = -1;
// Row-local stretches of an encoder layer as single launches (encoder.hip) instead of 6 + 6 per layer: BBBP_FUSED_ENCODER=1 /
// bbbp_set_fused_encoder(1).  Correct (tests compare the two schedules) but OFF by default: a 16-row work-group streams a whole
// layer's weights by itself (3.2 MB forward) through ONE wave per SIMD, which is bound by load latency -- 176 / 289 us per
// forward / backward launch against ~50 us for the launch-per-op chain whose GEMMs spread over all CUs.
bool fused_encoder(const Plan& p) {
    return (fused_encoder_mode() & 1) && p.L > 0 && !p.flash && !p.exact && bbbp_enc_rows_supported(p.F, p.NH, p.DFF);
}
// the whole forward chain of a small batch as one persistent launch (encoder.hip: enc_sliced_fwd_kernel)
bool sliced_encoder(const Plan& p, int bit = 2) {
    return (fused_encoder_mode() & bit) && p.L > 0 && !p.flash && !p.exact && bbbp_enc_sliced_supported(p.B, p.F, p.NH, p.DFF, p.L);
}
bool overlap_enabled() {
    if (g_overlap < 0) { const char* e = getenv("BBBP_SINGLE_STREAM"); g_overlap = (e && e[0] == '1') ? 0 : 1; }
    return g_overlap == 1;
}
int reserved_cus() {
    static int v = -1;
    // default 0: measured (tools/exp_overlap.py, bench sweeps) the reservation costs the conv kernels what it gives
    if (v < 0) { const char* e = getenv("BBBP_RESERVED_CUS"); v = e ? atoi(e) : 0; if (v < 0 || v > 128) v = 0; }
    return v;
}
// scoped CU partition (common.h): conv grids leave `reserved_cus()` CUs to the side stream's small kernels
// The Winograd conv2 kernels take a whole CU each (496 registers per lane, 110-154 KB LDS): nothing of the side stream
// can start beside them, so while the branches overlap their grids leave `wino_side_cus()` CUs entirely to the side
// stream.  Measured (tools/exp_wino.sh, ms/step): 256 CUs 3.65, 224 3.51, 192 3.42, 160 3.51, 128 3.76.
int wino_side_cus() {
    static int v = -1;
    if (v < 0) { const char* e = getenv("BBBP_WINO_SIDE_CUS"); v = e ? atoi(e) : 64; if (v < 0 || v > 192) v = 64; }
    return v;
}
struct Partition {
    bool on, side;
    explicit Partition(bool enable) : on(enable && reserved_cus() > 0), side(enable) {
        if (on) { g_bbbp_reserved_cus = reserved_cus(); g_bbbp_small_lds_pad = 48 * 1024; }
        if (side) g_bbbp_wino_side_cus = wino_side_cus();
    }
    ~Partition() {
        if (on) { g_bbbp_reserved_cus = 0; g_bbbp_small_lds_pad = 0; }
        if (side) g_bbbp_wino_side_cus = 0;
    }
};
int get_side(SideStream** out) {
    int dev = 0;
    BBBP_CHECK_HIP(hipGetDevice(&dev));
    BBBP_CHECK_ARG(dev >= 0 && dev < 64, "device index %d", dev);
    SideStream& ss = g_side[dev];
    if (!ss.s) {
        // (queue priorities -- hipStreamCreateWithPriority, chain stream highest / leaf stream lowest -- were measured on the headline step:
        // 2.497 ms default, 2.501 / 2.508 with them, profiles/r03_rebalance.txt; the wave priority set inside the kernels is what matters)
        BBBP_CHECK_HIP(hipStreamCreateWithFlags(&ss.s, hipStreamNonBlocking));
        BBBP_CHECK_HIP(hipEventCreateWithFlags(&ss.fork, hipEventDisableTiming));
        BBBP_CHECK_HIP(hipEventCreateWithFlags(&ss.join, hipEventDisableTiming));
        BBBP_CHECK_HIP(hipStreamCreateWithFlags(&ss.leaf, hipStreamNonBlocking));
        BBBP_CHECK_HIP(hipEventCreateWithFlags(&ss.join2, hipEventDisableTiming));
        for (int i = 0; i < NEV; ++i) BBBP_CHECK_HIP(hipEventCreateWithFlags(&ss.ev[i], hipEventDisableTiming));
    }
    *out = &ss;
    return BBBP_OK;
}
int fork_side(hipStream_t main, SideStream* ss) {
    BBBP_CHECK_HIP(hipEventRecord(ss->fork, main));
    BBBP_CHECK_HIP(hipStreamWaitEvent(ss->s, ss->fork, 0));
    return BBBP_OK;
}
// make stream `waiter` wait for everything enqueued so far on stream `producer`
int after(SideStream* ss, hipStream_t producer, hipStream_t waiter) {
    if (producer == waiter) return BBBP_OK;
    hipEvent_t e = ss->ev[ss->next];
    ss->next = (ss->next + 1) % NEV;
    BBBP_CHECK_HIP(hipEventRecord(e, producer));
    BBBP_CHECK_HIP(hipStreamWaitEvent(waiter, e, 0));
    return BBBP_OK;
}
int join_side(hipStream_t main, SideStream* ss) {
    BBBP_CHECK_HIP(hipEventRecord(ss->join, ss->s));
    BBBP_CHECK_HIP(hipStreamWaitEvent(main, ss->join, 0));
    return BBBP_OK;
}

#define TRY(expr) do { int _rc = (expr); if (_rc) return _rc; } while (0)

// ---- optional per-section timing with HIP events on the launch stream (bench.py roofline leg) ----
enum { SEC_CONV1_FWD = 0, SEC_CONV2_FWD, SEC_IMGFC_FWD, SEC_ENCODER_FWD, SEC_HEAD_FWD, SEC_HEAD_BWD, SEC_IMGFC_BWD,
       SEC_CONV2_WGRAD, SEC_CONV2_DGRAD, SEC_CONV1_WGRAD, SEC_ENCODER_BWD, SEC_FFN1_FWD, SEC_FFN2_FWD,
       // the other per-layer kernels of the launch-per-op encoder schedule (one instance per layer and step): chain ...
       SEC_QKV_FWD, SEC_ATTN_FWD, SEC_OUTPROJ_FWD, SEC_LN_FWD, SEC_LN_BWD, SEC_FFN2_DGRAD, SEC_FFN1_DGRAD, SEC_OUTPROJ_DGRAD, SEC_ATTN_BWD,
       SEC_QKV_DGRAD,
       // ... and weight-gradient leaves (leaf stream)
       SEC_FFN2_WGRAD, SEC_FFN1_WGRAD, SEC_OUTPROJ_WGRAD, SEC_QKV_WGRAD, SEC_COUNT };
static_assert(SEC_COUNT <= 32, "bbbp_profile_select takes a 32-bit mask");
const char* const kSectionNames[SEC_COUNT] = {"conv1_fwd", "conv2_fwd", "imgfc_fwd", "encoder_fwd", "head_fwd", "head_bwd",
                                              "imgfc_bwd", "conv2_wgrad", "conv2_dgrad", "conv1_wgrad", "encoder_bwd",
                                              "ffn1_fwd", "ffn2_fwd",          // the encoder's two FFN GEMMs, one instance per layer
                                              "qkv_fwd", "attn_fwd", "outproj_fwd", "ln_fwd", "ln_bwd", "ffn2_dgrad", "ffn1_dgrad",
                                              "outproj_dgrad", "attn_bwd", "qkv_dgrad", "ffn2_wgrad", "ffn1_wgrad", "outproj_wgrad",
                                              "qkv_wgrad"};
constexpr int PROF_MAX = 16384;
struct ProfState {
    bool on = false;
    unsigned mask = ~0u;          // sections that record events (bbbp_profile_select)
    int n = 0;
    int sec[PROF_MAX];
    hipEvent_t a[PROF_MAX], b[PROF_MAX];
    int created = 0;
} g_prof;

struct Section {
    hipStream_t st; int idx;
    Section(hipStream_t s, int sec) : st(s), idx(-1) {
        if (!g_prof.on || !((g_prof.mask >> sec) & 1u) || g_prof.n >= PROF_MAX) return;
        idx = g_prof.n++;
        if (idx >= g_prof.created) {
            if (hipEventCreate(&g_prof.a[idx]) != hipSuccess || hipEventCreate(&g_prof.b[idx]) != hipSuccess) { idx = -1; --g_prof.n; return; }
            g_prof.created = idx + 1;
        }
        g_prof.sec[idx] = sec;
        (void)hipEventRecord(g_prof.a[idx], st);
    }
    ~Section() { if (idx >= 0) (void)hipEventRecord(g_prof.b[idx], st); }
};

// y[M,N] = act(x[M,K] W[N,K]^T + b) + res
int linear_fwd(const Ctx& c, const float* x, int ldx, const float* W, const float* b, float* y, int ldy, int M, int N, int K,
               int act, const float* res = nullptr, int ldr = 0) {
    return bbbp_gemm_f32(c.st, 0, 1, M, N, K, 1.f, x, ldx, W, K, y, ldy, b, res, ldr, act, 1, 0, 0, 0, 0, c.scratch(),
                         c.scratch_bytes());
}
// dx[M,K] = dy[M,N] W[N,K] + res
int linear_bwd_input(const Ctx& c, const float* dy, int lddy, const float* W, float* dx, int lddx, int M, int N, int K,
                     const float* res = nullptr, int ldr = 0) {
    return bbbp_gemm_f32(c.st, 0, 0, M, K, N, 1.f, dy, lddy, W, K, dx, lddx, nullptr, res, ldr, 0, 1, 0, 0, 0, 0, c.scratch(),
                         c.scratch_bytes());
}
// dW[N,K] = dy[M,N]^T x[M,K]
int linear_bwd_weight(const Ctx& c, const float* dy, int lddy, const float* x, int ldx, float* dW, int M, int N, int K) {
    return bbbp_gemm_f32(c.st, 1, 0, N, K, M, 1.f, dy, lddy, x, ldx, dW, K, nullptr, nullptr, 0, 0, 1, 0, 0, 0, 0, c.scratch(),
                         c.scratch_bytes());
}

bbbp_gemm_desc gemm_desc(int transA, int transB, int M, int N, int K, float alpha, const float* A, int lda, const float* B, int ldb,
                         float* C, int ldc, int batch, long sA, long sB, long sC);
// dW[N,K] = dy[M,N]^T x[M,K] and db[N] = column sums of dy: ONE launch when the product takes the small-GEMM path (the bias
// gradient rides through the same MFMAs as a virtual all-ones column of x), else the GEMM plus a column-sum kernel
int linear_bwd_weight_bias(const Ctx& c, const float* dy, int lddy, const float* x, int ldx, float* dW, float* db, int M, int N, int K) {
    static const int fold = [] { const char* e = getenv("BBBP_FOLD_BIAS_GRAD"); return e ? atoi(e) : 1; }();
    if (fold && bbbp_gemm_folds_asum(N, K, M, 1)) {
        bbbp_gemm_desc g = gemm_desc(1, 0, N, K, M, 1.f, dy, lddy, x, ldx, dW, K, 1, 0, 0, 0);
        g.asum = db;
        return bbbp_gemm_f32_grouped(c.st, &g, 1, c.scratch(), c.scratch_bytes());
    }
    TRY(linear_bwd_weight(c, dy, lddy, x, ldx, dW, M, N, K));
    return bbbp_bias_act_bwd(c.st, const_cast<float*>(dy), lddy, nullptr, 0, db, M, N, 0, 1.f);
}

bbbp_gemm_desc gemm_desc(int transA, int transB, int M, int N, int K, float alpha, const float* A, int lda, const float* B, int ldb,
                         float* C, int ldc, int batch, long sA, long sB, long sC) {
    bbbp_gemm_desc g;
    g.transA = transA; g.transB = transB; g.M = M; g.N = N; g.K = K; g.alpha = alpha;
    g.A = A; g.lda = lda; g.B = B; g.ldb = ldb; g.C = C; g.ldc = ldc;
    g.bias = nullptr; g.residual = nullptr; g.ldr = 0; g.act = 0; g.gate = nullptr; g.ldg = 0; g.gate_scale = 1.f;
    g.gate_after_residual = 0; g.asum = nullptr; g.drop_p = 0.f; g.drop_seed = 0;
    g.batch = batch; g.strideA = sA; g.strideB = sB; g.strideC = sC; g.strideR = 0; g.strideG = 0;
    return g;
}

bbbp_gemm_desc gemm_desc(int transA, int transB, int M, int N, int K, float alpha, const float* A, int lda, const float* B, int ldb,
                         float* C, int ldc) {
    return gemm_desc(transA, transB, M, N, K, alpha, A, lda, B, ldb, C, ldc, 1, 0, 0, 0);
}

// per-site salt; the kernels mix it with the call's seed read from device memory (effective_seed)
uint64_t site_seed(uint64_t /*seed: lives in the workspace*/, int layer, int site) { return (uint64_t)(layer * 8 + site + 1); }

__global__ void set_seed_kernel(unsigned long long* slot, unsigned long long seed) { *slot = seed; }

struct SeedScope {
    const unsigned long long* prev;
    explicit SeedScope(const unsigned long long* slot) : prev(g_bbbp_seed_base) { g_bbbp_seed_base = slot; }
    ~SeedScope() { g_bbbp_seed_base = prev; }        // (the op-level base a caller set with bbbp_set_seed_base survives a whole-model call)
};

// ---- HIP-graph replay of the two whole-model calls ---------------------------------------------------------------
// A call enqueues 85 (forward) or 175 (backward) launches plus ~100 event operations: ~0.5 / 0.9 ms of host time.  For
// fixed arguments the enqueued work is identical from step to step (the dropout seed is read from device memory), so
// the second time a call arrives with the same arguments its enqueue is captured (all three streams: the fork/join
// events pull the side streams into the capture) and from then on replayed with one hipGraphLaunch.  Anything that
// changes an argument (another batch tensor, re-flattened parameters, a different batch size) is a different key;
// the cache holds a few entries, least recently used first out.  Opt-in (see graphs_mode); section profiling bypasses it.
struct GraphKey {
    int kind;                      // 0 forward, 1 backward
    bbbp_mixed_desc d;             // seed zeroed
    const void* ptr[6];
    uint64_t phash;                // hash of the parameter / gradient / BN pointer arrays
    int overlap;
    hipStream_t st;
    bool operator==(const GraphKey& o) const {
        return kind == o.kind && d.batch == o.d.batch && d.fingerprint_size == o.d.fingerprint_size && d.nhead == o.d.nhead &&
               d.num_layers == o.d.num_layers && d.dim_feedforward == o.d.dim_feedforward && d.training == o.d.training &&
               d.dropout_p == o.d.dropout_p && d.need_input_grad == o.d.need_input_grad && phash == o.phash &&
               overlap == o.overlap && st == o.st && ptr[0] == o.ptr[0] && ptr[1] == o.ptr[1] && ptr[2] == o.ptr[2] &&
               ptr[3] == o.ptr[3] && ptr[4] == o.ptr[4] && ptr[5] == o.ptr[5];
    }
};
struct GraphEntry { GraphKey key; hipGraphExec_t exec = nullptr; unsigned long long stamp = 0; bool seen_only = true; };
constexpr int GRAPH_SLOTS = 12;
GraphEntry g_graphs[GRAPH_SLOTS];
unsigned long long g_graph_clock = 0;
long g_graph_replays = 0, g_graph_captures = 0;

// Default OFF: on ROCm 7.2 replaying the captured three-stream graph is SLOWER on the GPU than the hand-scheduled
// streams (whole step, B = 512: 3.67 ms eager, 4.50 ms replayed; the host loop is not shorter either, hipGraphLaunch of
// a 260-node graph costs about what the eager enqueue does).  Kept as an opt-in (BBBP_GRAPHS=1 / bbbp_set_graphs) with a
// bit-exactness test, for runtimes where graph launch is cheaper.
int g_graphs_mode = -1;
int graphs_mode() {
    if (g_graphs_mode < 0) { const char* e = getenv("BBBP_GRAPHS"); g_graphs_mode = e ? atoi(e) : 0; }
    return g_graphs_mode;
}
uint64_t hash_ptrs(const void* const* a, int n, uint64_t h = 1469598103934665603ull) {
    for (int i = 0; i < n; ++i) { h ^= (uint64_t)reinterpret_cast<uintptr_t>(a[i]); h *= 1099511628211ull; }
    return h;
}
// 0: run eagerly; 1: capture now into *slot; 2: replay *slot
int graph_lookup(const GraphKey& k, GraphEntry** slot) {
    if (!graphs_mode() || g_prof.on) return 0;
    ++g_graph_clock;
    GraphEntry* lru = &g_graphs[0];
    for (auto& e : g_graphs) {
        if (e.stamp && e.key == k) {
            e.stamp = g_graph_clock; *slot = &e;
            return e.seen_only ? 1 : 2;
        }
        if (e.stamp < lru->stamp) lru = &e;
    }
    if (lru->exec) { (void)hipGraphExecDestroy(lru->exec); lru->exec = nullptr; }
    lru->key = k; lru->stamp = g_graph_clock; lru->seen_only = true;       // first sighting: eager
    return 0;
}

}  // namespace

// Branch overlap on/off at run time (default on; env BBBP_SINGLE_STREAM=1 starts with it off).  Returns the old value.
extern "C" int bbbp_set_fused_head_bwd(int on) {
    const int prev = g_fused_head_bwd > 0 ? 1 : 0;
    g_fused_head_bwd = on ? 1 : 0;
    return prev;
}

extern "C" int bbbp_set_fused_encoder(int mode) {
    const int prev = fused_encoder_mode();
    g_fused_encoder = mode & 7;
    return prev;
}

extern "C" int bbbp_set_fold_outproj(int on) {
    const int prev = fold_outproj_on();
    g_fold_outproj = on & 3;
    return prev;
}

extern "C" int bbbp_set_flash_attention(int on) {
    if (g_flash_attention < 0) { const char* e = getenv("BBBP_FLASH_ATTENTION"); g_flash_attention = e ? atoi(e) & 31 : 13; }
    const int prev = g_flash_attention;
    g_flash_attention = on & 31;
    return prev;
}

extern "C" int bbbp_set_ln_absorb(int on) { const int prev = ln_absorb_on(); g_ln_absorb = on ? 1 : 0; return prev; }

extern "C" int bbbp_set_overlap(int on) { int old = overlap_enabled() ? 1 : 0; g_overlap = on ? 1 : 0; return old; }

// Profiling: enable, run steps, synchronise the stream, then collect {sum of ms, launches} per section.
extern "C" int bbbp_profile_enable(int on) { g_prof.on = on != 0; g_prof.n = 0; return BBBP_OK; }
extern "C" int bbbp_profile_select(unsigned section_mask) { g_prof.mask = section_mask ? section_mask : ~0u; return BBBP_OK; }
extern "C" int bbbp_profile_num_sections(void) { return SEC_COUNT; }
extern "C" const char* bbbp_profile_section_name(int i) { return (i >= 0 && i < SEC_COUNT) ? kSectionNames[i] : ""; }
extern "C" int bbbp_profile_collect(float* ms_sum, int* count) {
    for (int i = 0; i < SEC_COUNT; ++i) { ms_sum[i] = 0.f; count[i] = 0; }
    for (int i = 0; i < g_prof.n; ++i) {
        float ms = 0.f;
        BBBP_CHECK_HIP(hipEventElapsedTime(&ms, g_prof.a[i], g_prof.b[i]));
        ms_sum[g_prof.sec[i]] += ms;
        count[g_prof.sec[i]] += 1;
    }
    g_prof.n = 0;
    return BBBP_OK;
}

// Absolute placement of the recorded section instances (does not clear them): start / end in ms after the first instance's
// start.  Events of different streams share the device timeline, so the gaps between sections are visible.
extern "C" int bbbp_profile_timeline(int* section, float* start_ms, float* end_ms, int max_entries) {
    BBBP_CHECK_ARG(section && start_ms && end_ms && max_entries >= 0, "profile_timeline: bad arguments");
    const int n = g_prof.n < max_entries ? g_prof.n : max_entries;
    for (int i = 0; i < n; ++i) {
        section[i] = g_prof.sec[i];
        BBBP_CHECK_HIP(hipEventElapsedTime(&start_ms[i], g_prof.a[0], g_prof.a[i]));
        BBBP_CHECK_HIP(hipEventElapsedTime(&end_ms[i], g_prof.a[0], g_prof.b[i]));
    }
    return n;
}

extern "C" int bbbp_mixed_num_params(const bbbp_mixed_desc* d) {
    if (!d) return -1;
    if (d->num_layers < 0 || d->num_layers > 32 || (d->fusion != 0 && d->fusion != 1)) return -1;
    return PIdx(d).count();
}

extern "C" size_t bbbp_mixed_workspace_bytes(const bbbp_mixed_desc* d) {
    Plan p;
    if (make_plan(d, &p)) return 0;
    return p.total;
}

static int forward_enqueue(void* stream, const bbbp_mixed_desc* d, const float* const* P, float* const* bn_running,
                           const float* fingerprint, const float* image, float* out, void* workspace,
                           size_t workspace_bytes) {
    Plan plan;
    TRY(make_plan(d, &plan));
    SeedScope seed_scope(reinterpret_cast<const unsigned long long*>(static_cast<char*>(workspace) + plan.seed_slot));
    BBBP_CHECK_ARG(P && fingerprint && image && out && workspace && bn_running, "mixed_forward: null pointer");
    if (workspace_bytes < plan.total) {
        bbbp_set_error("mixed_forward: workspace %zu < %zu bytes", workspace_bytes, plan.total);
        return BBBP_ERR_WORKSPACE;
    }
    Ctx c{static_cast<hipStream_t>(stream), static_cast<char*>(workspace), &plan};
    const PIdx ix(d);
    // a deferred optimizer slice that is NOT this model's image-FC weight (another model's step, another tensor): wait before anything runs
    if (bbbp_param_pending_elsewhere(P[ix.ifc_w()])) (void)bbbp_param_wait(c.st, nullptr);
    const int B = plan.B, F = plan.F, NH = plan.NH, D = plan.D, DFF = plan.DFF;
    const int Bk = (int)plan.Bg;            // attention keys: this rank's rows, or every rank's in exact-global-batch mode
    const float p_drop = plan.drop ? d->dropout_p : 0.f;
    const float scale = 1.0f / sqrtf((float)D);

    // ---- fingerprint branch: encoder (R:75-78, 110-111), on the side stream -------------------
    Ctx ce = c;
    SideStream* ss = nullptr;
    // Screening plans (forward-only, 2048+ rows on the bf16 attention kernel) run on ONE stream: there both branches are bound by the matrix
    // pipe, so the second stream only time-shares it -- measured equal on one box (6.66 / 6.83 ms overlapped, 6.59 / 6.85 one stream) and
    // WORSE than the sum of the branches on others (7.6 ms against 5.29 + 1.85 + 0.06; profiles/r03_config5_streams.txt); alone, the
    // encoder also takes the out_proj fold's gain (fold mask bit 1).  BBBP_SCREEN_OVERLAP=1 restores the two-stream schedule.
    static const bool screen_overlap = [] { const char* e = getenv("BBBP_SCREEN_OVERLAP"); return e && atoi(e) != 0; }();
    if (overlap_enabled() && (screen_overlap || !plan.attn_b3)) {
        TRY(get_side(&ss));
        TRY(fork_side(c.st, ss));
        ce.st = ss->s;
        ce.side = 1;
    }
    std::optional<Partition> part;           // scoped sections / partitions end early with reset(), or at any return
    part.emplace(ss != nullptr);
    float* comb = c.f(plan.combined);

    // ---- image branch (R:84-94, 114-115): enqueued first so the GPU is busy while the host feeds the encoder's launches -------------------------------------------------------
    float* pool1 = c.f(plan.pool1); float* pool2 = c.f(plan.pool2);
    {
        Section s1(c.st, SEC_CONV1_FWD);
        // beside a training step's encoder chain the f32 form stays (common.h: g_bbbp_conv1_fwd_f32); screening batches, eval loops and
        // the encoder-less two-branch model take the split-bf16 form when the conv mask selects it (bit 6, default)
        // (the rule looks at the plan only, not at the stream mode: one stream or three give bit-identical steps)
        // round 4: training steps run the software-pipelined split-bf16 kernel (conv_b3c1.hip) too, ONE work-group per CU beside the chain
        // (BBBP_C1_TRAIN=0: the f32 kernel of rounds 1-3 there; 2: two work-groups per CU, measured slower for the step)
        static const int c1_train = [] { const char* e = getenv("BBBP_C1_TRAIN"); return e ? atoi(e) : 1; }();      // default 1: step 2.517 -> 2.487 ms (profiles/r04_c1_pipe.txt)
        const bool beside_chain = !plan.inference && plan.L > 0;
        g_bbbp_conv1_fwd_f32 = (beside_chain && !c1_train) ? 1 : 0;
        g_bbbp_conv1_fwd_per_cu = (beside_chain && c1_train) ? (c1_train >= 2 ? 2 : 1) : 0;      // BBBP_C1_TRAIN=2: two work-groups per CU there too
        const int rc1 = bbbp_conv3x3_relu_pool_fwd(c.st, image, P[ix.c1_w()], P[ix.c1_b()], pool1, plan.inference ? nullptr : c.u8(plan.mask1), B, 3, C1, IMG, IMG,
                                                   c.scratch(), c.scratch_bytes());
        g_bbbp_conv1_fwd_f32 = 0; g_bbbp_conv1_fwd_per_cu = 0;
        TRY(rc1);
    }
    {
        Section s2(c.st, SEC_CONV2_FWD);
        // round 4: beside a training step's encoder chain the software-pipelined one-work-group-per-CU kernel (conv_b3.hip): 0.40 instead of
        // 0.387 ms alone, but the chain -- the forward half's critical path -- keeps three quarters of every SIMD: step 2.41 -> 2.34 ms
        // (profiles/r04_conv2_pipe.txt).  BBBP_C2_TRAIN=0: the two-work-group kernel there too.  Bit-identical outputs either way.
        static const int c2_train = [] { const char* e = getenv("BBBP_C2_TRAIN"); return e ? atoi(e) : 1; }();
        g_bbbp_conv2_fwd_pipe = (!plan.inference && plan.L > 0 && c2_train) ? 1 : 0;
        const int rc2 = bbbp_conv3x3_relu_pool_fwd(c.st, pool1, P[ix.c2_w()], P[ix.c2_b()], pool2, plan.inference ? nullptr : c.u8(plan.mask2), B, C1, C2, IMG / 2,
                                                   IMG / 2, c.scratch(), c.scratch_bytes());
        g_bbbp_conv2_fwd_pipe = 0;
        TRY(rc2);
    }
    {
        Section s3(c.st, SEC_IMGFC_FWD);
        (void)bbbp_param_wait(c.st, P[ix.ifc_w()]);        // the optimizer's deferred image-FC slice (bbbp_adamw_step_deferred): first read here
        TRY(linear_fwd(c, pool2, IMG_FLAT, P[ix.ifc_w()], P[ix.ifc_b()], comb + FC, COMB, B, FC, IMG_FLAT, BBBP_ACT_RELU));
    }


    // ---- fingerprint branch body (side stream)
    const float* x = fingerprint;
    std::optional<Section> sec_enc;
    sec_enc.emplace(ce.st, SEC_ENCODER_FWD);
    const bool sliced = sliced_encoder(plan) && plan.sl_sync;
    if (sliced) {
        bbbp_enc_sliced_fwd_args a;
        memset(&a, 0, sizeof(a));
        a.x0 = fingerprint; a.L = plan.L; a.B = B; a.F = F; a.DFF = DFF; a.p = p_drop; a.scale = scale;
        a.wfc = P[ix.fpfc_w()]; a.bfc = P[ix.fpfc_b()]; a.comb = comb; a.nfc = FC; a.ldcomb = COMB;
        a.sync = c.u8(plan.sl_sync); a.part = c.f(plan.sl_part);
        for (int l = 0; l < plan.L; ++l) {
            const LayerOff& o = plan.layer[l];
            bbbp_enc_sliced_layer& y = a.lay[l];
            y.win = P[ix.layer(l, L_INW)]; y.bin = P[ix.layer(l, L_INB)]; y.wo = P[ix.layer(l, L_OUTW)]; y.bo = P[ix.layer(l, L_OUTB)];
            y.g1 = P[ix.layer(l, L_N1W)]; y.be1 = P[ix.layer(l, L_N1B)]; y.w1 = P[ix.layer(l, L_W1)]; y.b1 = P[ix.layer(l, L_B1)];
            y.w2 = P[ix.layer(l, L_W2)]; y.b2 = P[ix.layer(l, L_B2)]; y.g2 = P[ix.layer(l, L_N2W)]; y.be2 = P[ix.layer(l, L_N2B)];
            y.qkv = c.f(o.qkv); y.prob = c.f(o.prob); y.pd = c.f(o.pd); y.ctx = c.f(o.ctx); y.z1 = c.f(o.z1); y.y1 = c.f(o.y1);
            y.hff = c.f(o.hff); y.z2 = c.f(o.z2); y.y2 = c.f(o.y2); y.mean1 = c.f(o.mean1); y.rstd1 = c.f(o.rstd1);
            y.mean2 = c.f(o.mean2); y.rstd2 = c.f(o.rstd2);
            y.seed0 = site_seed(d->seed, l, 0); y.seed1 = site_seed(d->seed, l, 1); y.seed2 = site_seed(d->seed, l, 2); y.seed3 = site_seed(d->seed, l, 3);
        }
        TRY(bbbp_enc_sliced_fwd(ce.st, &a));
    }
    if (plan.fold) {
        // W' = Wo Wv, b' = Wo bv of every layer in one launch at the head of the chain (the parameters change every step)
        const float* win[32]; const float* bin[32]; const float* wo[32]; const float* bo[32]; float* wf[32]; float* bf[32]; float* wvt[32];
        for (int l = 0; l < plan.L; ++l) {
            win[l] = P[ix.layer(l, L_INW)]; bin[l] = P[ix.layer(l, L_INB)]; wo[l] = P[ix.layer(l, L_OUTW)];
            // fused attention (forward-only, no dropout): softmax rows sum to one, so out_proj's bias is carried by b' = Wo bv + bo
            bo[l] = plan.attn_b3 ? P[ix.layer(l, L_OUTB)] : nullptr;
            wf[l] = c.f(plan.fwf[l]); bf[l] = c.f(plan.fbf[l]); wvt[l] = c.f(plan.fwvt[l]);
        }
        TRY(bbbp_outproj_fold(ce.st, plan.L, F, win, bin, wo, bo, wf, bf, wvt));
    }
    const bool fused_rows = !sliced && fused_encoder(plan);
    if (fused_rows) {
        // in_proj of layer 0; every later in_proj (and fingerprint_fc) is the tail of the previous layer's row kernel
        TRY(linear_fwd(ce, x, F, P[ix.layer(0, L_INW)], P[ix.layer(0, L_INB)], c.f(plan.layer[0].qkv), 3 * F, B, 3 * F, F, 0));
        for (int l = 0; l < plan.L; ++l) {
            const LayerOff& o = plan.layer[l];
            float* qkv = c.f(o.qkv); float* prob = c.f(o.prob); float* ctx = c.f(o.ctx); float* pd = c.f(o.pd);
            TRY(bbbp_gemm_f32(ce.st, 0, 1, B, B, D, scale, qkv, 3 * F, qkv + F, 3 * F, prob, B, nullptr, nullptr, 0, 0, NH, D, D,
                              (long)B * B, 0, ce.scratch(), ce.scratch_bytes()));
            TRY(bbbp_softmax_fwd(ce.st, prob, pd, (long)NH * B, B, p_drop, site_seed(d->seed, l, 0)));
            TRY(bbbp_gemm_f32(ce.st, 0, 0, B, D, B, 1.f, pd, B, qkv + 2 * F, 3 * F, ctx, F, nullptr, nullptr, 0, 0, NH, (long)B * B,
                              D, D, 0, ce.scratch(), ce.scratch_bytes()));
            bbbp_enc_row_fwd_args a;
            a.ctx = ctx; a.xin = x;
            a.wo = P[ix.layer(l, L_OUTW)]; a.bo = P[ix.layer(l, L_OUTB)]; a.g1 = P[ix.layer(l, L_N1W)]; a.be1 = P[ix.layer(l, L_N1B)];
            a.w1 = P[ix.layer(l, L_W1)]; a.b1 = P[ix.layer(l, L_B1)]; a.w2 = P[ix.layer(l, L_W2)]; a.b2 = P[ix.layer(l, L_B2)];
            a.g2 = P[ix.layer(l, L_N2W)]; a.be2 = P[ix.layer(l, L_N2B)];
            const bool last = l + 1 == plan.L;
            a.wn = last ? P[ix.fpfc_w()] : P[ix.layer(l + 1, L_INW)]; a.bn = last ? P[ix.fpfc_b()] : P[ix.layer(l + 1, L_INB)];
            a.outn = last ? comb : c.f(plan.layer[l + 1].qkv); a.nn = last ? FC : 3 * F; a.ldn = last ? COMB : 3 * F; a.actn = last ? 1 : 0;
            a.z1 = c.f(o.z1); a.y1 = c.f(o.y1); a.hff = c.f(o.hff); a.z2 = c.f(o.z2); a.y2 = c.f(o.y2);
            a.mean1 = c.f(o.mean1); a.rstd1 = c.f(o.rstd1); a.mean2 = c.f(o.mean2); a.rstd2 = c.f(o.rstd2);
            a.B = B; a.F = F; a.DFF = DFF; a.p = p_drop;
            a.seed1 = site_seed(d->seed, l, 1); a.seed2 = site_seed(d->seed, l, 2); a.seed3 = site_seed(d->seed, l, 3);
            TRY(bbbp_enc_row_fwd(ce.st, &a));
            x = c.f(o.y2);
        }
    }
    // LayerNorm absorbed by the Linear that consumes it (gemm.hip: gemm_direct_lna_kernel; BBBP_LN_ABSORB=0 keeps the stand-alone launches):
    // norm1 -> linear1, norm2 -> the next in_proj / fingerprint_fc.  The dropout + residual that the LayerNorm launch applied to its input
    // move into the epilogue of the GEMM that produces it (same Philox elements), so z1 / z2, y1 / y2 and the row statistics the backward
    // pass reads are the same tensors as before.
    struct { const float* z; const float* gamma; const float* beta; float* y; float* mean; float* rstd; } pend = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    const bool lna_ok = ln_absorb_on() && !plan.exact && plan.L > 0 && bbbp_layernorm_linear_preferred(B, DFF, F) && bbbp_layernorm_linear_preferred(B, 3 * F, F) &&
                        bbbp_layernorm_linear_preferred(B, FC, F) &&
                        // training: the producers' dropout rides in the small-product GEMM's epilogue only
                        (!plan.drop || (bbbp_gemm_folds_asum(B, F, DFF, 1) && bbbp_gemm_folds_asum(B, F, plan.fold ? Bk : F, 1)));
    for (int l = 0; l < ((fused_rows || sliced) ? 0 : plan.L); ++l) {
        const LayerOff& o = plan.layer[l];
        float* qkv = c.f(o.qkv); float* prob = c.f(o.prob); float* ctx = c.f(o.ctx);
        const bool absorb1 = lna_ok && !(plan.attn_b3 && plan.fold);      // (that attention kernel writes z1 itself: no epilogue for the residual)
        const bool absorb2 = lna_ok;
        {
            Section sq(ce.st, SEC_QKV_FWD);
            // folded plan: [Q | K | VW] = x [Wq; Wk; Wo Wv]^T + [bq; bk; Wo bv]
            const float* w_in = plan.fold ? c.f(plan.fwf[l]) : P[ix.layer(l, L_INW)];
            const float* b_in = plan.fold ? c.f(plan.fbf[l]) : P[ix.layer(l, L_INB)];
            if (pend.z) {
                // the previous layer's norm2 is absorbed here: x = LayerNorm(z2) is written by the same launch (round 4)
                TRY(bbbp_layernorm_linear_fwd(ce.st, pend.z, F, pend.gamma, pend.beta, 1e-5f, w_in, b_in, qkv, 3 * F, 0, 0.f, 0, pend.y, F, pend.mean,
                                              pend.rstd, B, 3 * F, F));
                pend.z = nullptr;
            } else {
                TRY(linear_fwd(ce, x, F, w_in, b_in, qkv, 3 * F, B, 3 * F, F, 0));
            }
        }
        std::optional<Section> sec_attn;
        sec_attn.emplace(ce.st, SEC_ATTN_FWD);
        if (plan.attn_b3) {
            TRY(bbbp_attn_b3_fwd(ce.st, qkv, plan.fold ? c.f(o.z1) : ctx, B, F, NH, scale, plan.attn_part_bytes ? c.f(plan.attn_part) : nullptr, plan.attn_part_bytes));
        } else if (plan.flash) {
            TRY(bbbp_attn_small_fwd(ce.st, qkv, ctx, c.f(o.lse), B, F, NH, scale, p_drop, site_seed(d->seed, l, 0), o.keep ? c.u8(o.keep) : nullptr));
        } else {
        // keys and values: this rank's rows of qkv, or (exact-global-batch mode) the rows of every rank, gathered once per layer
        const float* kmat = qkv + F; const float* vmat = qkv + 2 * F; int ldkv = 3 * F;
        if (plan.exact) {
            float* kvg = c.f(o.kvg);
            const long n = (long)B * 2 * F;
            hipLaunchKernelGGL(kv_pack_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), g_bbbp_small_lds_pad, ce.st, qkv,
                               kvg + (size_t)plan.rank * n, n, F);
            BBBP_CHECK_LAUNCH();
            TRY(run_collective(d, BBBP_COLL_ALLGATHER, BBBP_COLL_KV, l, o.kvg + (size_t)plan.rank * n * sizeof(float), o.kvg, (size_t)n, ce.st));
            kmat = kvg; vmat = kvg + F; ldkv = 2 * F;
        }
        // scores_h = scale * Q_h K_h^T
        TRY(bbbp_gemm_f32(ce.st, 0, 1, B, Bk, D, scale, qkv, 3 * F, kmat, ldkv, prob, Bk, nullptr, nullptr, 0, 0, NH, D, D,
                          (long)B * Bk, 0, ce.scratch(), ce.scratch_bytes()));
        float* pd = c.f(o.pd);
        TRY(bbbp_softmax_fwd(ce.st, prob, pd, (long)NH * B, Bk, p_drop, site_seed(d->seed, l, 0)));
        // ctx_h = Pd_h V_h; folded plan (one head): z1 = Pd VW + bo, the out_proj output itself
        if (plan.fold && absorb1) {
            // ... and with norm1 absorbed by linear1, z1 = dropout(Pd VW + bo) + x: what bbbp_layernorm_fwd would have left in z1
            bbbp_gemm_desc g = gemm_desc(0, 0, B, D, Bk, 1.f, pd, Bk, vmat, ldkv, c.f(o.z1), F, 1, 0, 0, 0);
            g.bias = P[ix.layer(l, L_OUTB)]; g.residual = x; g.ldr = F; g.drop_p = p_drop; g.drop_seed = site_seed(d->seed, l, 1);
            TRY(bbbp_gemm_f32_grouped(ce.st, &g, 1, ce.scratch(), ce.scratch_bytes()));
        } else {
        TRY(bbbp_gemm_f32(ce.st, 0, 0, B, D, Bk, 1.f, pd, Bk, vmat, ldkv, plan.fold ? c.f(o.z1) : ctx, F, plan.fold ? P[ix.layer(l, L_OUTB)] : nullptr,
                          nullptr, 0, 0, NH, (long)B * Bk, D, D, 0, ce.scratch(), ce.scratch_bytes()));
        }
        }
        sec_attn.reset();
        float* z1 = c.f(o.z1); float* y1 = c.f(o.y1);
        // out_proj -> (dropout) + residual -> norm1: one launch when the output is narrow (gemm.hip: gemm_direct_ln_kernel)
        // OPT-IN (BBBP_FUSED_LINEAR_LN=1), measured slower at B = 512: the 11-wave work-groups need three wave slots on three SIMDs of a CU
        // beside the resident conv work-groups (encoder forward 0.98 -> 1.15 ms in the step, 0.58 -> 0.60 alone); linear2 -> norm2 is never
        // fused (one wave per tile walking K = 2048 alone: 90 us against 22 + 4)
        static const bool ln_opt_in = [] { const char* e = getenv("BBBP_FUSED_LINEAR_LN"); return e && atoi(e) != 0; }();
        const bool ln_fused = !plan.fold && ln_opt_in && bbbp_linear_layernorm_supported(B, F, F) && F <= 512;
        if (ln_fused) {
            TRY(bbbp_linear_layernorm_fwd(ce.st, ctx, F, P[ix.layer(l, L_OUTW)], P[ix.layer(l, L_OUTB)], x, F, z1, F, y1, F, P[ix.layer(l, L_N1W)],
                                          P[ix.layer(l, L_N1B)], c.f(o.mean1), c.f(o.rstd1), B, F, F, 1e-5f, p_drop, site_seed(d->seed, l, 1)));
        } else {
        if (!plan.fold) {
            Section so(ce.st, SEC_OUTPROJ_FWD);
            if (absorb1) {
                bbbp_gemm_desc g = gemm_desc(0, 1, B, F, F, 1.f, ctx, F, P[ix.layer(l, L_OUTW)], F, z1, F, 1, 0, 0, 0);
                g.bias = P[ix.layer(l, L_OUTB)]; g.residual = x; g.ldr = F; g.drop_p = p_drop; g.drop_seed = site_seed(d->seed, l, 1);
                TRY(bbbp_gemm_f32_grouped(ce.st, &g, 1, ce.scratch(), ce.scratch_bytes()));
            } else {
                TRY(linear_fwd(ce, ctx, F, P[ix.layer(l, L_OUTW)], P[ix.layer(l, L_OUTB)], z1, F, B, F, F, 0));
            }
        }
        if (!absorb1) {
        Section sl(ce.st, SEC_LN_FWD);
        TRY(bbbp_layernorm_fwd(ce.st, z1, x, y1, P[ix.layer(l, L_N1W)], P[ix.layer(l, L_N1B)], c.f(o.mean1), c.f(o.rstd1), B, F,
                               1e-5f, p_drop, site_seed(d->seed, l, 1)));
        }
        }
        float* hff = c.f(o.hff); float* z2 = c.f(o.z2); float* y2 = c.f(o.y2);
        // linear1 + ReLU (+ the FFN dropout in the same epilogue when the product takes the small-GEMM path: same Philox elements as bbbp_dropout)
        const bool drop_in_gemm = plan.drop && bbbp_gemm_folds_asum(B, DFF, F, 1);
        {
            Section sf(ce.st, SEC_FFN1_FWD);
            if (absorb1) {
                // norm1 absorbed: hff = dropout(ReLU(LayerNorm(z1) W1^T + b1)); y1, mean1, rstd1 written by the same launch
                TRY(bbbp_layernorm_linear_fwd(ce.st, z1, F, P[ix.layer(l, L_N1W)], P[ix.layer(l, L_N1B)], 1e-5f, P[ix.layer(l, L_W1)], P[ix.layer(l, L_B1)],
                                              hff, DFF, BBBP_ACT_RELU, p_drop, site_seed(d->seed, l, 2), y1, F, c.f(o.mean1), c.f(o.rstd1), B, DFF, F));
            } else if (drop_in_gemm) {
                bbbp_gemm_desc g = gemm_desc(0, 1, B, DFF, F, 1.f, y1, F, P[ix.layer(l, L_W1)], F, hff, DFF, 1, 0, 0, 0);
                g.bias = P[ix.layer(l, L_B1)]; g.act = BBBP_ACT_RELU; g.drop_p = p_drop; g.drop_seed = site_seed(d->seed, l, 2);
                TRY(bbbp_gemm_f32_grouped(ce.st, &g, 1, ce.scratch(), ce.scratch_bytes()));
            } else {
                TRY(linear_fwd(ce, y1, F, P[ix.layer(l, L_W1)], P[ix.layer(l, L_B1)], hff, DFF, B, DFF, F, BBBP_ACT_RELU));
            }
        }
        if (plan.drop && !drop_in_gemm && !absorb1) TRY(bbbp_dropout(ce.st, hff, hff, (long)B * DFF, p_drop, site_seed(d->seed, l, 2)));
        {
            Section sf(ce.st, SEC_FFN2_FWD);
            if (absorb2) {
                // norm2 is absorbed by the next in_proj (or fingerprint_fc): z2 = dropout(h W2^T + b2) + y1 here, normalised there
                bbbp_gemm_desc g = gemm_desc(0, 1, B, F, DFF, 1.f, hff, DFF, P[ix.layer(l, L_W2)], DFF, z2, F, 1, 0, 0, 0);
                g.bias = P[ix.layer(l, L_B2)]; g.residual = y1; g.ldr = F; g.drop_p = p_drop; g.drop_seed = site_seed(d->seed, l, 3);
                TRY(bbbp_gemm_f32_grouped(ce.st, &g, 1, ce.scratch(), ce.scratch_bytes()));
                pend.z = z2; pend.gamma = P[ix.layer(l, L_N2W)]; pend.beta = P[ix.layer(l, L_N2B)]; pend.y = y2; pend.mean = c.f(o.mean2); pend.rstd = c.f(o.rstd2);
            } else {
                TRY(linear_fwd(ce, hff, DFF, P[ix.layer(l, L_W2)], P[ix.layer(l, L_B2)], z2, F, B, F, DFF, 0));
            }
        }
        if (!absorb2) {
            Section sl(ce.st, SEC_LN_FWD);
            TRY(bbbp_layernorm_fwd(ce.st, z2, y1, y2, P[ix.layer(l, L_N2W)], P[ix.layer(l, L_N2B)], c.f(o.mean2), c.f(o.rstd2), B, F,
                                   1e-5f, p_drop, site_seed(d->seed, l, 3)));
        }
        x = y2;
    }
    // fingerprint_fc (R:79-82, 112) -> combined[:, 0:128]
    if (!fused_rows && !sliced) {
        if (pend.z) {
            TRY(bbbp_layernorm_linear_fwd(ce.st, pend.z, F, pend.gamma, pend.beta, 1e-5f, P[ix.fpfc_w()], P[ix.fpfc_b()], comb, COMB, BBBP_ACT_RELU, 0.f, 0,
                                          pend.y, F, pend.mean, pend.rstd, B, FC, F));
            pend.z = nullptr;
        } else {
            TRY(linear_fwd(ce, x, F, P[ix.fpfc_w()], P[ix.fpfc_b()], comb, COMB, B, FC, F, BBBP_ACT_RELU));
        }
    }
    sec_enc.reset();

    if (ss) TRY(join_side(c.st, ss));        // fusion needs both halves of `combined`
    part.reset();
    Section sec_head(c.st, SEC_HEAD_FWD);

    static const int fused_head = [] { const char* e = getenv("BBBP_FUSED_HEAD"); return e ? atoi(e) : 1; }();
    BBBP_CHECK_ARG(!plan.exact || (fused_head && NHEADS_FUSION == 4), "the exact-global-batch mode needs the fused head (BBBP_FUSED_HEAD=1)");
    if (fused_head && NHEADS_FUSION == 4) {
        // fusion block + head in two launches (head.hip); with torch.cat fusion the first launch starts at fc.0
        const float *fw1[4] = {}, *fb1[4] = {}, *fw2[4] = {}, *fb2[4] = {};
        if (!plan.concat)
            for (int h = 0; h < 4; ++h) { fw1[h] = P[ix.fus(h, 0)]; fb1[h] = P[ix.fus(h, 1)]; fw2[h] = P[ix.fus(h, 2)]; fb2[h] = P[ix.fus(h, 3)]; }
        // exact-global-batch mode: the (mean, M2) blocks of every rank are gathered between the two launches
        bbbp_head_sync sync;
        sync.world = plan.world; sync.rank = plan.rank;
        const size_t pcount = (size_t)((B + 15) / 16) * 2 * H1;
        if (plan.exact)
            sync.between = [&]() -> int {
                return run_collective(d, BBBP_COLL_ALLGATHER, BBBP_COLL_BN_FWD, -1, plan.head_partial + (size_t)plan.rank * pcount * sizeof(float),
                                      plan.head_partial, pcount, c.st);
            };
        return bbbp_head_forward_fused_sync(c.st, comb, fw1, fb1, fw2, fb2, P[ix.fc0_w()], P[ix.fc0_b()], P[ix.bn_w()], P[ix.bn_b()],
                                            bn_running[0], bn_running[1], P[ix.fc3_w()], P[ix.fc3_b()], P[ix.fc5_w()], P[ix.fc5_b()],
                                            P[ix.fc7_w()], P[ix.fc7_b()], c.f(plan.hid), c.f(plan.attn), c.f(plan.fused), c.f(plan.h),
                                            c.f(plan.hb), c.f(plan.bn_mean), c.f(plan.bn_rstd), c.f(plan.h2), c.f(plan.h3), out,
                                            c.f(plan.head_partial), B, d->training, plan.concat ? 1 : 0, plan.exact ? &sync : nullptr);
    }
    // ---- attention fusion (R:60-65, 117) -------------------------------------------------------
    float* hid = c.f(plan.hid);
    float* fused = c.f(plan.fused);              // == combined under torch.cat fusion
    if (!plan.concat) {
        const float* w2[NHEADS_FUSION]; const float* b2[NHEADS_FUSION];
        for (int h = 0; h < NHEADS_FUSION; ++h) {
            TRY(linear_fwd(c, comb, COMB, P[ix.fus(h, 0)], P[ix.fus(h, 1)], hid + (size_t)h * B * FUS_HID, FUS_HID, B, FUS_HID, COMB,
                           BBBP_ACT_TANH));
            w2[h] = P[ix.fus(h, 2)]; b2[h] = P[ix.fus(h, 3)];
        }
        TRY(bbbp_fusion_combine_fwd(c.st, comb, hid, w2, b2, fused, c.f(plan.attn), B, COMB, FUS_HID, NHEADS_FUSION));
    }

    // ---- regression head (R:98-107, 118) -------------------------------------------------------
    float* h = c.f(plan.h); float* hb = c.f(plan.hb); float* h2 = c.f(plan.h2); float* h3 = c.f(plan.h3);
    TRY(linear_fwd(c, fused, COMB, P[ix.fc0_w()], P[ix.fc0_b()], h, H1, B, H1, COMB, BBBP_ACT_RELU));
    TRY(bbbp_batchnorm1d_fwd(c.st, h, hb, P[ix.bn_w()], P[ix.bn_b()], bn_running[0], bn_running[1], c.f(plan.bn_mean),
                             c.f(plan.bn_rstd), B, H1, 1e-5f, 0.1f, d->training));
    TRY(linear_fwd(c, hb, H1, P[ix.fc3_w()], P[ix.fc3_b()], h2, H2, B, H2, H1, BBBP_ACT_RELU));
    TRY(linear_fwd(c, h2, H2, P[ix.fc5_w()], P[ix.fc5_b()], h3, H3, B, H3, H2, BBBP_ACT_RELU));
    TRY(linear_fwd(c, h3, H3, P[ix.fc7_w()], P[ix.fc7_b()], out, 1, B, 1, H3, 0));
    return BBBP_OK;
}

static int backward_enqueue(void* stream, const bbbp_mixed_desc* d, const float* const* P, float* const* G,
                            const float* fingerprint, const float* image, const float* dout, void* workspace,
                            size_t workspace_bytes) {
    Plan plan;
    TRY(make_plan(d, &plan));
    SeedScope seed_scope(reinterpret_cast<const unsigned long long*>(static_cast<char*>(workspace) + plan.seed_slot));
    BBBP_CHECK_ARG(P && G && fingerprint && image && dout && workspace, "mixed_backward: null pointer");
    BBBP_CHECK_ARG(!plan.inference, "mixed_backward: the forward call used an inference workspace (desc.inference = 1)");
    (void)bbbp_param_wait(static_cast<hipStream_t>(stream), nullptr);      // (a deferred optimizer slice: normally consumed by the forward pass already)
    if (workspace_bytes < plan.total) {
        bbbp_set_error("mixed_backward: workspace %zu < %zu bytes", workspace_bytes, plan.total);
        return BBBP_ERR_WORKSPACE;
    }
    // Three streams: `c` (caller's stream) carries the head, the fusion block and the image branch; `ce` carries the
    // fingerprint branch's dependency chain (dy -> dx through the encoder layers); `cl` carries the LEAVES -- weight
    // and bias gradients, LayerNorm parameter gradients -- which nothing downstream waits for, so the chain's
    // critical path is half as long.  With overlap disabled all three are the caller's stream.
    Ctx c{static_cast<hipStream_t>(stream), static_cast<char*>(workspace), &plan};
    Ctx ce = c, cl = c;
    SideStream* ss = nullptr;
    if (overlap_enabled()) {
        TRY(get_side(&ss));
        TRY(fork_side(c.st, ss));                         // chain stream starts after the caller's prior work
        ce.st = ss->s; ce.side = 1;
        TRY(after(ss, c.st, ss->leaf));
        cl.st = ss->leaf; cl.side = 2;
    }
    auto leaf_after = [&](const Ctx& producer) -> int { return ss ? after(ss, producer.st, cl.st) : BBBP_OK; };
    // both LayerNorms' weight / bias gradients of layer l in ONE leaf launch (dgamma = sum_rows dy * xhat, dbeta = sum_rows dy),
    // the last leaf of the layer: its gradient bucket (all twelve tensors, one contiguous slice) is final after it
    auto layer_norm_grads = [&](int l) -> int {
        const LayerOff& o = plan.layer[l]; const LayerGrad& g = plan.lgrad[l];
        const PIdx ixl(d);
        const float* dy[2] = {c.f(g.dyout), c.f(g.dy1)}; const float* zz[2] = {c.f(o.z2), c.f(o.z1)};
        const float* mm[2] = {c.f(o.mean2), c.f(o.mean1)}; const float* rr[2] = {c.f(o.rstd2), c.f(o.rstd1)};
        float* dg[2] = {G[ixl.layer(l, L_N2W)], G[ixl.layer(l, L_N1W)]}; float* db[2] = {G[ixl.layer(l, L_N2B)], G[ixl.layer(l, L_N1B)]};
        return bbbp_ln_param_grad_multi(cl.st, 2, dy, zz, mm, rr, dg, db, plan.B, plan.F);
    };
    auto layer_norm_leaves = [&](int l) -> int {
        TRY(layer_norm_grads(l));
        return record_layer_bucket(cl.st, l);
    };
    const PIdx ix(d);
    const int B = plan.B, F = plan.F, NH = plan.NH, D = plan.D, DFF = plan.DFF;
    const int Bk = (int)plan.Bg;            // attention keys: this rank's rows, or every rank's in exact-global-batch mode
    const float p_drop = plan.drop ? d->dropout_p : 0.f;
    const float inv_keep = plan.drop ? 1.f / (1.f - p_drop) : 1.f;
    const float scale = 1.0f / sqrtf((float)D);

    float* comb = c.f(plan.combined); float* hid = c.f(plan.hid); float* fused = c.f(plan.fused);
    float* h = c.f(plan.h); float* hb = c.f(plan.hb); float* h2 = c.f(plan.h2); float* h3 = c.f(plan.h3);
    float* dh3 = c.f(plan.dh3); float* dh2 = c.f(plan.dh2); float* dhb = c.f(plan.dhb); float* dh = c.f(plan.dh);
    float* dfused = c.f(plan.dfused); float* dcomb = c.f(plan.dcomb);

    // ---- head (chain on the caller's stream, leaves on the leaf stream) ---------------------------
    std::optional<Section> sec;
    sec.emplace(c.st, SEC_HEAD_BWD);
    auto next_section = [&](int id) { sec.reset(); sec.emplace(c.st, id); };
    // default ON since round 2 (bbbp_set_fused_head_bwd / BBBP_FUSED_HEAD_BWD=0 select the launch-per-op chain): with the bias
    // gradients folded into the weight-gradient GEMMs the leaves it feeds are short enough that the shorter chain shows --
    // B = 512 3.32 -> 3.27 ms, B = 256 2.12 -> 2.08, B = 128 2.36 -> 2.23 (round 1, with 38 separate column-sum leaves: neutral)
    if (g_fused_head_bwd < 0) { const char* e = getenv("BBBP_FUSED_HEAD_BWD"); g_fused_head_bwd = e ? atoi(e) != 0 : 1; }
    const int fused_head_bwd = g_fused_head_bwd && !plan.concat;
    float* dlogit = c.f(plan.dlogit); float* dpre = c.f(plan.dpre);
    bool head_leaves_pending = false;
    auto head_leaves = [&]() -> int {
        TRY(linear_bwd_weight_bias(cl, dout, 1, h3, H3, G[ix.fc7_w()], G[ix.fc7_b()], B, 1, H3));
        TRY(linear_bwd_weight_bias(cl, dh3, H3, h2, H2, G[ix.fc5_w()], G[ix.fc5_b()], B, H3, H2));
        TRY(linear_bwd_weight_bias(cl, dh2, H2, hb, H1, G[ix.fc3_w()], G[ix.fc3_b()], B, H2, H1));
        TRY(linear_bwd_weight_bias(cl, dh, H1, fused, COMB, G[ix.fc0_w()], G[ix.fc0_b()], B, H1, COMB));
        for (int hh = 0; hh < (plan.concat ? 0 : NHEADS_FUSION); ++hh) {
            float* dl = dlogit + (size_t)hh * B;
            float* dp = dpre + (size_t)hh * B * FUS_HID;
            const float* hd = hid + (size_t)hh * B * FUS_HID;
            TRY(linear_bwd_weight_bias(cl, dl, 1, hd, FUS_HID, G[ix.fus(hh, 2)], G[ix.fus(hh, 3)], B, 1, FUS_HID));
            TRY(linear_bwd_weight_bias(cl, dp, FUS_HID, comb, COMB, G[ix.fus(hh, 0)], G[ix.fus(hh, 1)], B, FUS_HID, COMB));
        }
        TRY(bbbp_bias_act_bwd(cl.st, dcomb + FC, COMB, nullptr, 0, G[ix.ifc_b()], B, FC, 0, 1.f));
        return BBBP_OK;
    };
    BBBP_CHECK_ARG(!plan.exact || fused_head_bwd, "the exact-global-batch mode needs the fused head backward (bbbp_set_fused_head_bwd(1))");
    if (fused_head_bwd) {
        // the whole input-gradient chain of the head and the fusion block in two launches (head.hip); every weight / bias
        // gradient below is a leaf that reads what those wrote
        const float* fw1[NHEADS_FUSION]; const float* fw2[NHEADS_FUSION];
        for (int hh = 0; hh < NHEADS_FUSION; ++hh) { fw1[hh] = P[ix.fus(hh, 0)]; fw2[hh] = P[ix.fus(hh, 2)]; }
        // exact-global-batch mode: the BatchNorm's two backward sums span every rank's rows
        bbbp_head_sync sync;
        sync.world = plan.world; sync.rank = plan.rank;
        const size_t pcount = (size_t)((B + 15) / 16) * 2 * H1;
        if (plan.exact)
            sync.between = [&]() -> int {
                return run_collective(d, BBBP_COLL_ALLGATHER, BBBP_COLL_BN_BWD, -1, plan.head_partial + (size_t)plan.rank * pcount * sizeof(float),
                                      plan.head_partial, pcount, c.st);
            };
        TRY(bbbp_head_backward_fused_sync(c.st, dout, comb, hid, c.f(plan.attn), h, h2, h3, c.f(plan.bn_mean), c.f(plan.bn_rstd),
                                          P[ix.bn_w()], fw1, fw2, P[ix.fc0_w()], P[ix.fc3_w()], P[ix.fc5_w()], P[ix.fc7_w()], dh3, dh2, dhb,
                                          dh, dlogit, dpre, dcomb, G[ix.bn_w()], G[ix.bn_b()], c.f(plan.head_partial), B, d->training,
                                          plan.exact ? &sync : nullptr));
        TRY(leaf_after(c));
        head_leaves_pending = true;          // enqueued after the image branch's kernels: the host reaches those sooner
    } else {
    // fc.7: out = h3 W7^T + b7
    TRY(linear_bwd_weight_bias(cl, dout, 1, h3, H3, G[ix.fc7_w()], G[ix.fc7_b()], B, 1, H3));
    // the ReLU masks ride in the input-gradient GEMMs' epilogues; the bias gradients (column sums) are leaves
    {
        bbbp_gemm_desc g = gemm_desc(0, 0, B, H3, 1, 1.f, dout, 1, P[ix.fc7_w()], H3, dh3, H3);
        g.gate = h3; g.ldg = H3;
        TRY(bbbp_gemm_f32_grouped(c.st, &g, 1, c.scratch(), c.scratch_bytes()));
    }
    TRY(leaf_after(c));
    TRY(linear_bwd_weight_bias(cl, dh3, H3, h2, H2, G[ix.fc5_w()], G[ix.fc5_b()], B, H3, H2));
    {
        bbbp_gemm_desc g = gemm_desc(0, 0, B, H2, H3, 1.f, dh3, H3, P[ix.fc5_w()], H2, dh2, H2);
        g.gate = h2; g.ldg = H2;
        TRY(bbbp_gemm_f32_grouped(c.st, &g, 1, c.scratch(), c.scratch_bytes()));
    }
    TRY(leaf_after(c));
    TRY(linear_bwd_weight_bias(cl, dh2, H2, hb, H1, G[ix.fc3_w()], G[ix.fc3_b()], B, H2, H1));
    TRY(linear_bwd_input(c, dh2, H2, P[ix.fc3_w()], dhb, H1, B, H2, H1));
    TRY(bbbp_batchnorm1d_bwd_relu(c.st, dhb, h, P[ix.bn_w()], c.f(plan.bn_mean), c.f(plan.bn_rstd), dh, G[ix.bn_w()], G[ix.bn_b()],
                                  B, H1, d->training));
    TRY(leaf_after(c));
    TRY(linear_bwd_weight_bias(cl, dh, H1, fused, COMB, G[ix.fc0_w()], G[ix.fc0_b()], B, H1, COMB));
    if (plan.concat) {
        // torch.cat fusion: dcomb = dh W0, masked by the ReLUs that produced combined = [fp_out | img_out]
        bbbp_gemm_desc g = gemm_desc(0, 0, B, COMB, H1, 1.f, dh, H1, P[ix.fc0_w()], COMB, dcomb, COMB);
        g.gate = comb; g.ldg = COMB;
        TRY(bbbp_gemm_f32_grouped(c.st, &g, 1, c.scratch(), c.scratch_bytes()));
    } else {
    TRY(linear_bwd_input(c, dh, H1, P[ix.fc0_w()], dfused, COMB, B, H1, COMB));

    // ---- attention fusion ------------------------------------------------------------------------
    const float* w2[NHEADS_FUSION];
    for (int hh = 0; hh < NHEADS_FUSION; ++hh) w2[hh] = P[ix.fus(hh, 2)];
    TRY(bbbp_fusion_combine_bwd(c.st, dfused, comb, hid, c.f(plan.attn), w2, dcomb, dlogit, dpre, B, COMB, FUS_HID, NHEADS_FUSION));
    TRY(leaf_after(c));
    for (int hh = 0; hh < NHEADS_FUSION; ++hh) {
        float* dl = dlogit + (size_t)hh * B;
        float* dp = dpre + (size_t)hh * B * FUS_HID;
        const float* hd = hid + (size_t)hh * B * FUS_HID;
        TRY(linear_bwd_weight_bias(cl, dl, 1, hd, FUS_HID, G[ix.fus(hh, 2)], G[ix.fus(hh, 3)], B, 1, FUS_HID));
        TRY(linear_bwd_weight_bias(cl, dp, FUS_HID, comb, COMB, G[ix.fus(hh, 0)], G[ix.fus(hh, 1)], B, FUS_HID, COMB));
        // dcomb += dp W1_h; the last of the four also applies the ReLU mask of both branch outputs (combined = [fp_out | img_out])
        bbbp_gemm_desc g = gemm_desc(0, 0, B, COMB, FUS_HID, 1.f, dp, FUS_HID, P[ix.fus(hh, 0)], COMB, dcomb, COMB);
        g.residual = dcomb; g.ldr = COMB;
        if (hh == NHEADS_FUSION - 1) { g.gate = comb; g.ldg = COMB; g.gate_after_residual = 1; }
        TRY(bbbp_gemm_f32_grouped(c.st, &g, 1, c.scratch(), c.scratch_bytes()));
    }
    }
    // bias gradients of the two branch outputs: column sums of the masked dcomb, leaves
    TRY(leaf_after(c));
    TRY(bbbp_bias_act_bwd(cl.st, dcomb + FC, COMB, nullptr, 0, G[ix.ifc_b()], B, FC, 0, 1.f));

    }

    // both branches only READ dcomb from here on
    if (ss) { TRY(after(ss, c.st, ce.st)); TRY(after(ss, c.st, cl.st)); }
    Partition part(ss != nullptr);

    // ---- image branch (caller's stream): enqueued first, five long MFMA-bound kernels -------------------
    float* pool1 = c.f(plan.pool1); float* pool2 = c.f(plan.pool2);
    float* dpool2 = c.f(plan.dpool2); float* dpool1 = c.f(plan.dpool1);
    next_section(SEC_IMGFC_BWD);
    TRY(linear_bwd_weight(c, dcomb + FC, COMB, pool2, IMG_FLAT, G[ix.ifc_w()], B, FC, IMG_FLAT));
    {
        int dev = 0;
        BBBP_CHECK_HIP(hipGetDevice(&dev));
        if (dev >= 0 && dev < 64) {
            if (!g_bucket_event[dev]) BBBP_CHECK_HIP(hipEventCreateWithFlags(&g_bucket_event[dev], hipEventDisableTiming));
            hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
            (void)hipStreamIsCapturing(c.st, &cap);
            if (cap == hipStreamCaptureStatusNone) {        // an event recorded inside a graph capture means nothing outside it
                BBBP_CHECK_HIP(hipEventRecord(g_bucket_event[dev], c.st));
                g_bucket_recorded[dev] = true;
            } else {
                g_bucket_recorded[dev] = false;
            }
        }
    }
    TRY(linear_bwd_input(c, dcomb + FC, COMB, P[ix.ifc_w()], dpool2, IMG_FLAT, B, FC, IMG_FLAT));
    {
        // ... and from here on the image-FC weight itself is no longer read by this pass (bbbp_mixed_backward_wait_released, bucket 0)
        int dev = 0;
        BBBP_CHECK_HIP(hipGetDevice(&dev));
        if (g_release_events && dev >= 0 && dev < 64 && g_bucket_recorded[dev]) {
            if (!g_bucket0_released[dev]) BBBP_CHECK_HIP(hipEventCreateWithFlags(&g_bucket0_released[dev], hipEventDisableTiming));
            BBBP_CHECK_HIP(hipEventRecord(g_bucket0_released[dev], c.st));
            g_bucket0_released_recorded[dev] = true;
        } else if (dev >= 0 && dev < 64) {
            g_bucket0_released_recorded[dev] = false;
        }
    }
    next_section(SEC_CONV2_WGRAD);
    {
        // beside the encoder's backward chain the sparse weight-gradient kernel runs one wave per SIMD (common.h); the rule looks at the
        // plan only, not at the stream mode: one stream or three give bit-identical steps
        g_bbbp_conv_wgrad_beside_encoder = plan.L > 0 ? 1 : 0;
        const int rcw = bbbp_conv3x3_relu_pool_bwd_weight(c.st, pool1, dpool2, c.u8(plan.mask2), G[ix.c2_w()], G[ix.c2_b()], B, C1, C2, IMG / 2,
                                                          IMG / 2, c.scratch(), c.scratch_bytes());
        g_bbbp_conv_wgrad_beside_encoder = 0;
        TRY(rcw);
    }
    next_section(SEC_CONV2_DGRAD);
    TRY(bbbp_conv3x3_relu_pool_bwd_data(c.st, dpool2, c.u8(plan.mask2), P[ix.c2_w()], dpool1, B, C1, C2, IMG / 2, IMG / 2,
                                        c.scratch(), c.scratch_bytes()));
    next_section(SEC_CONV1_WGRAD);
    TRY(bbbp_conv3x3_relu_pool_bwd_weight(c.st, image, dpool1, c.u8(plan.mask1), G[ix.c1_w()], G[ix.c1_b()], B, 3, C1, IMG, IMG,
                                          c.scratch(), c.scratch_bytes()));
    sec.reset();
    if (head_leaves_pending) TRY(head_leaves());

    // ---- fingerprint branch: chain on `ce`, leaves on `cl` --------------------------------------------
    Section sec_encb(ce.st, SEC_ENCODER_BWD);
    const float* enc_out = plan.L > 0 ? c.f(plan.layer[plan.L - 1].y2) : fingerprint;
    TRY(linear_bwd_weight_bias(cl, dcomb, COMB, enc_out, F, G[ix.fpfc_w()], G[ix.fpfc_b()], B, FC, F));
    float* dy = plan.L > 0 ? c.f(plan.lgrad[plan.L - 1].dyout) : c.f(plan.dA);
    const bool sliced = sliced_encoder(plan, 4) && plan.sl_sync && plan.sl_kvpart;
    if (!sliced && (plan.L > 0 || d->need_input_grad)) TRY(linear_bwd_input(ce, dcomb, COMB, P[ix.fpfc_w()], dy, F, B, FC, F));
    float* dprob = c.f(plan.dprob); float* dctx = c.f(plan.dctx);
    if (sliced) {
        // the whole input-gradient chain in one persistent launch; every weight gradient is a leaf of its buffers afterwards
        bbbp_enc_sliced_bwd_args a;
        memset(&a, 0, sizeof(a));
        a.dcomb = dcomb; a.ldcomb = COMB; a.nfc = FC; a.wfc = P[ix.fpfc_w()];
        a.L = plan.L; a.B = B; a.F = F; a.DFF = DFF; a.p = p_drop; a.scale = scale;
        a.sync = c.u8(plan.sl_sync); a.part = c.f(plan.sl_part); a.kvpart = c.f(plan.sl_kvpart);
        for (int l = 0; l < plan.L; ++l) {
            const LayerOff& o = plan.layer[l]; const LayerGrad& g = plan.lgrad[l];
            bbbp_enc_sliced_bwd_layer& y = a.lay[l];
            y.win = P[ix.layer(l, L_INW)]; y.wo = P[ix.layer(l, L_OUTW)]; y.g1 = P[ix.layer(l, L_N1W)];
            y.w1 = P[ix.layer(l, L_W1)]; y.w2 = P[ix.layer(l, L_W2)]; y.g2 = P[ix.layer(l, L_N2W)];
            y.qkv = c.f(o.qkv); y.prob = c.f(o.prob); y.pd = c.f(o.pd); y.z1 = c.f(o.z1); y.hff = c.f(o.hff); y.z2 = c.f(o.z2);
            y.mean1 = c.f(o.mean1); y.rstd1 = c.f(o.rstd1); y.mean2 = c.f(o.mean2); y.rstd2 = c.f(o.rstd2);
            y.dyout = c.f(g.dyout); y.dz2 = c.f(g.dz2); y.dff = c.f(g.dz2d); y.dhff = c.f(g.dhff); y.dy1 = c.f(g.dy1);
            y.dz1 = c.f(g.dz1); y.dsa = c.f(g.dz1d); y.dqkv = c.f(g.dqkv);
            y.seed0 = site_seed(d->seed, l, 0); y.seed1 = site_seed(d->seed, l, 1); y.seed3 = site_seed(d->seed, l, 3);
        }
        TRY(bbbp_enc_sliced_bwd(ce.st, &a));
        if (d->need_input_grad)
            TRY(linear_bwd_input(ce, c.f(plan.lgrad[0].dqkv), 3 * F, P[ix.layer(0, L_INW)], c.f(plan.dA), F, B, 3 * F, F, c.f(plan.lgrad[0].dz1), F));
        TRY(leaf_after(ce));
        for (int l = plan.L - 1; l >= 0; --l) {
            const LayerOff& o = plan.layer[l]; const LayerGrad& g = plan.lgrad[l];
            const float* xin = l > 0 ? c.f(plan.layer[l - 1].y2) : fingerprint;
            TRY(linear_bwd_weight_bias(cl, c.f(g.dz2d), F, c.f(o.hff), DFF, G[ix.layer(l, L_W2)], G[ix.layer(l, L_B2)], B, F, DFF));
            TRY(linear_bwd_weight_bias(cl, c.f(g.dhff), DFF, c.f(o.y1), F, G[ix.layer(l, L_W1)], G[ix.layer(l, L_B1)], B, DFF, F));
            TRY(linear_bwd_weight_bias(cl, c.f(g.dz1d), F, c.f(o.ctx), F, G[ix.layer(l, L_OUTW)], G[ix.layer(l, L_OUTB)], B, F, F));
            TRY(linear_bwd_weight_bias(cl, c.f(g.dqkv), 3 * F, xin, F, G[ix.layer(l, L_INW)], G[ix.layer(l, L_INB)], B, 3 * F, F));
            TRY(layer_norm_leaves(l));
        }
    }
    const bool fused_rows = !sliced && fused_encoder(plan);
    for (int l = fused_rows ? plan.L - 1 : -1; l >= 0; --l) {
        const LayerOff& o = plan.layer[l];
        const LayerGrad& g = plan.lgrad[l];
        const float* xin = l > 0 ? c.f(plan.layer[l - 1].y2) : fingerprint;
        float* qkv = c.f(o.qkv); float* prob = c.f(o.prob); float* ctx = c.f(o.ctx);
        float* dff = c.f(g.dz2d); float* dhff = c.f(g.dhff); float* dsa = c.f(g.dz1d); float* dqkv = c.f(g.dqkv);
        // one launch: (in_proj input gradient of the layer above + residual ->) norm2 bwd -> linear2 dgrad (.) gate -> linear1
        // dgrad + residual -> norm1 bwd -> out_proj dgrad
        bbbp_enc_row_bwd_args a;
        const bool top = l + 1 == plan.L;
        a.dqkv_up = top ? nullptr : c.f(plan.lgrad[l + 1].dqkv); a.win_up = top ? nullptr : P[ix.layer(l + 1, L_INW)];
        a.dz1_up = top ? nullptr : c.f(plan.lgrad[l + 1].dz1); a.dyout = c.f(g.dyout);
        a.z2 = c.f(o.z2); a.mean2 = c.f(o.mean2); a.rstd2 = c.f(o.rstd2); a.g2 = P[ix.layer(l, L_N2W)]; a.w2 = P[ix.layer(l, L_W2)];
        a.hff = c.f(o.hff); a.w1 = P[ix.layer(l, L_W1)]; a.z1 = c.f(o.z1); a.mean1 = c.f(o.mean1); a.rstd1 = c.f(o.rstd1);
        a.g1 = P[ix.layer(l, L_N1W)]; a.wo = P[ix.layer(l, L_OUTW)];
        a.dz2 = c.f(g.dz2); a.dff = dff; a.dhff = dhff; a.dy1 = c.f(g.dy1); a.dz1 = c.f(g.dz1); a.dsa = dsa; a.dctx = dctx;
        a.B = B; a.F = F; a.DFF = DFF; a.p = p_drop; a.seed1 = site_seed(d->seed, l, 1); a.seed3 = site_seed(d->seed, l, 3);
        TRY(bbbp_enc_row_bwd(ce.st, &a));
        TRY(leaf_after(ce));
        TRY(linear_bwd_weight_bias(cl, dff, F, c.f(o.hff), DFF, G[ix.layer(l, L_W2)], G[ix.layer(l, L_B2)], B, F, DFF));
        TRY(linear_bwd_weight_bias(cl, dhff, DFF, c.f(o.y1), F, G[ix.layer(l, L_W1)], G[ix.layer(l, L_B1)], B, DFF, F));
        TRY(linear_bwd_weight_bias(cl, dsa, F, ctx, F, G[ix.layer(l, L_OUTW)], G[ix.layer(l, L_OUTB)], B, F, F));
        // attention backward (products that become ready together share a launch)
        const float* pdp = c.f(o.pd);
        {
            bbbp_gemm_desc gg[2] = {
                gemm_desc(1, 0, B, D, B, 1.f, pdp, B, dctx, F, dqkv + 2 * F, 3 * F, NH, (long)B * B, D, D),
                gemm_desc(0, 1, B, B, D, 1.f, dctx, F, qkv + 2 * F, 3 * F, dprob, B, NH, D, D, (long)B * B)};
            TRY(bbbp_gemm_f32_grouped(ce.st, gg, 2, ce.scratch(), ce.scratch_bytes()));
        }
        TRY(bbbp_softmax_bwd(ce.st, dprob, prob, (long)NH * B, B, p_drop, site_seed(d->seed, l, 0)));
        {
            bbbp_gemm_desc gg[2] = {
                gemm_desc(0, 0, B, D, B, scale, dprob, B, qkv + F, 3 * F, dqkv, 3 * F, NH, (long)B * B, D, D),
                gemm_desc(1, 0, B, D, B, scale, dprob, B, qkv, 3 * F, dqkv + F, 3 * F, NH, (long)B * B, D, D)};
            TRY(bbbp_gemm_f32_grouped(ce.st, gg, 2, ce.scratch(), ce.scratch_bytes()));
        }
        TRY(leaf_after(ce));
        TRY(linear_bwd_weight_bias(cl, dqkv, 3 * F, xin, F, G[ix.layer(l, L_INW)], G[ix.layer(l, L_INB)], B, 3 * F, F));
        TRY(layer_norm_leaves(l));
        if (l == 0 && d->need_input_grad)
            TRY(linear_bwd_input(ce, dqkv, 3 * F, P[ix.layer(0, L_INW)], c.f(plan.dA), F, B, 3 * F, F, c.f(g.dz1), F));
    }
    for (int l = (fused_rows || sliced) ? -1 : plan.L - 1; l >= 0; --l) {
        const LayerOff& o = plan.layer[l];
        const LayerGrad& g = plan.lgrad[l];
        const float* xin = l > 0 ? c.f(plan.layer[l - 1].y2) : fingerprint;
        float* qkv = c.f(o.qkv); float* prob = c.f(o.prob); float* ctx = c.f(o.ctx);
        float* z1 = c.f(o.z1); float* y1 = c.f(o.y1); float* hff = c.f(o.hff); float* z2 = c.f(o.z2);
        float* dyout = c.f(g.dyout); float* dz2 = c.f(g.dz2); float* dff = c.f(g.dz2d); float* dhff = c.f(g.dhff);
        float* dy1 = c.f(g.dy1); float* dz1 = c.f(g.dz1); float* dsa = c.f(g.dz1d); float* dqkv = c.f(g.dqkv);
        // norm2: dz2 (residual gradient, flows to y1) and its dropped copy dff (gradient of the FFN output)
        {
            Section sl(ce.st, SEC_LN_BWD);
            TRY(bbbp_layernorm_bwd(ce.st, dyout, z2, P[ix.layer(l, L_N2W)], c.f(o.mean2), c.f(o.rstd2), dz2, plan.drop ? dff : nullptr,
                                   nullptr, nullptr, B, F, p_drop, site_seed(d->seed, l, 3)));
        }
        // linear2 input gradient, then ReLU (+ dropout: hff is the post-dropout value, hff > 0 <=> active and kept)
        // the ReLU / dropout mask rides in the GEMM epilogue; the bias gradient (a column sum) is a leaf
        {
            Section sg(ce.st, SEC_FFN2_DGRAD);
            bbbp_gemm_desc g = gemm_desc(0, 0, B, DFF, F, 1.f, dff, F, P[ix.layer(l, L_W2)], DFF, dhff, DFF);
            g.gate = hff; g.ldg = DFF; g.gate_scale = inv_keep;
            TRY(bbbp_gemm_f32_grouped(ce.st, &g, 1, ce.scratch(), ce.scratch_bytes()));
        }
        // leaves of this half layer (they only read per-layer buffers, so ONE event per half layer orders them all)
        TRY(leaf_after(ce));
        {
            Section sw(cl.st, SEC_FFN2_WGRAD);
            TRY(linear_bwd_weight_bias(cl, dff, F, hff, DFF, G[ix.layer(l, L_W2)], G[ix.layer(l, L_B2)], B, F, DFF));
        }
        {
            Section sw(cl.st, SEC_FFN1_WGRAD);
            TRY(linear_bwd_weight_bias(cl, dhff, DFF, y1, F, G[ix.layer(l, L_W1)], G[ix.layer(l, L_B1)], B, DFF, F));
        }
        // dy1 = dhff W1 + dz2
        {
            Section sg(ce.st, SEC_FFN1_DGRAD);
            TRY(linear_bwd_input(ce, dhff, DFF, P[ix.layer(l, L_W1)], dy1, F, B, DFF, F, dz2, F));
        }
        // Layer 0 is the end of the pass: whatever its leaves still hold after the chain's last kernel is the step's tail.  Its LayerNorm
        // parameter gradients only need dyout and dy1, so they start here (one more event) instead of after the attention block.
        const bool ln_grads_early = l == 0 && ss != nullptr;
        if (ln_grads_early) { TRY(leaf_after(ce)); TRY(layer_norm_grads(l)); }
        // norm1
        {
            Section sl(ce.st, SEC_LN_BWD);
            TRY(bbbp_layernorm_bwd(ce.st, dy1, z1, P[ix.layer(l, L_N1W)], c.f(o.mean1), c.f(o.rstd1), dz1, plan.drop ? dsa : nullptr,
                                   nullptr, nullptr, B, F, p_drop, site_seed(d->seed, l, 1)));
        }
        // out_proj input gradient (folded plan: dVW = Pd^T dsa and dPd = dsa VW^T take the out_proj output's gradient itself)
        if (!plan.fold) {
            Section sg(ce.st, SEC_OUTPROJ_DGRAD);
            TRY(linear_bwd_input(ce, dsa, F, P[ix.layer(l, L_OUTW)], dctx, F, B, F, F));
        }
        const float* dattn = plan.fold ? dsa : dctx;
        // attention backward.  Products that become ready together share a launch (bbbp_gemm_f32_grouped):
        //   dV_h = Pd_h^T dctx_h -> dqkv[:, 2F + hD]   |   dPd_h = dctx_h V_h^T
        const float* pdp = c.f(o.pd);
        std::optional<Section> sec_attn;
        sec_attn.emplace(ce.st, SEC_ATTN_BWD);
        if (plan.flash) {
            TRY(bbbp_attn_small_bwd(ce.st, qkv, ctx, c.f(o.lse), dctx, dqkv, B, F, NH, scale, p_drop, site_seed(d->seed, l, 0),
                                    o.keep ? c.u8(o.keep) : nullptr));
        } else {
        // exact-global-batch mode: keys / values are the gathered rows of every rank; dK | dV of ALL keys (this rank's queries' share)
        // go to dkvg and come back reduce-scattered
        const float* kmat = plan.exact ? c.f(o.kvg) : qkv + F; const float* vmat = plan.exact ? c.f(o.kvg) + F : qkv + 2 * F;
        const int ldkv = plan.exact ? 2 * F : 3 * F;
        float* dkmat = plan.exact ? c.f(plan.dkvg) : dqkv + F; float* dvmat = plan.exact ? c.f(plan.dkvg) + F : dqkv + 2 * F;
        {
            bbbp_gemm_desc g[2] = {
                gemm_desc(1, 0, Bk, D, B, 1.f, pdp, Bk, dattn, F, dvmat, ldkv, NH, (long)B * Bk, D, D),
                gemm_desc(0, 1, B, Bk, D, 1.f, dattn, F, vmat, ldkv, dprob, Bk, NH, D, D, (long)B * Bk)};
            TRY(bbbp_gemm_f32_grouped(ce.st, g, 2, ce.scratch(), ce.scratch_bytes()));
        }
        TRY(bbbp_softmax_bwd(ce.st, dprob, prob, (long)NH * B, Bk, p_drop, site_seed(d->seed, l, 0)));
        //   dQ_h = scale dS_h K_h   |   dK_h = scale dS_h^T Q_h
        {
            bbbp_gemm_desc g[2] = {
                gemm_desc(0, 0, B, D, Bk, scale, dprob, Bk, kmat, ldkv, dqkv, 3 * F, NH, (long)B * Bk, D, D),
                gemm_desc(1, 0, Bk, D, B, scale, dprob, Bk, qkv, 3 * F, dkmat, ldkv, NH, (long)B * Bk, D, D)};
            TRY(bbbp_gemm_f32_grouped(ce.st, g, 2, ce.scratch(), ce.scratch_bytes()));
        }
        if (plan.exact) {
            const long n = (long)B * 2 * F;
            TRY(run_collective(d, BBBP_COLL_REDUCE_SCATTER, BBBP_COLL_DKV, l, plan.dkvg, plan.dkvl, (size_t)n, ce.st));
            hipLaunchKernelGGL(kv_unpack_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), g_bbbp_small_lds_pad, ce.st, c.f(plan.dkvl), dqkv, n, F);
            BBBP_CHECK_LAUNCH();
        }
        }
        sec_attn.reset();
        TRY(leaf_after(ce));
        if (!plan.fold) {
            Section sw(cl.st, SEC_OUTPROJ_WGRAD);
            TRY(linear_bwd_weight_bias(cl, dsa, F, ctx, F, G[ix.layer(l, L_OUTW)], G[ix.layer(l, L_OUTB)], B, F, F));
        }
        if (plan.fold) {
            // [dWq; dWk | dbq; dbk] straight into the gradient, the folded block dW' | db' = dVW^T [x | 1] into its scratch (one launch), then
            // dWo = dW' Wv^T + db' bv^T, d[Wv | bv] = Wo^T [dW' | db'] and dbo = column sums of dsa in one more (fold.hip)
            Section sw(cl.st, SEC_QKV_WGRAD);
            float* tdw = c.f(plan.ftdw[l]); float* tdb = c.f(plan.ftdb[l]);
            if (bbbp_gemm_folds_asum(2 * F, F, B, 1) && bbbp_gemm_folds_asum(F, F, B, 1)) {
                bbbp_gemm_desc gg[2] = {gemm_desc(1, 0, 2 * F, F, B, 1.f, dqkv, 3 * F, xin, F, G[ix.layer(l, L_INW)], F, 1, 0, 0, 0),
                                        gemm_desc(1, 0, F, F, B, 1.f, dqkv + 2 * F, 3 * F, xin, F, tdw, F, 1, 0, 0, 0)};
                gg[0].asum = G[ix.layer(l, L_INB)]; gg[1].asum = tdb;
                TRY(bbbp_gemm_f32_grouped(cl.st, gg, 2, cl.scratch(), cl.scratch_bytes()));
            } else {
                TRY(linear_bwd_weight_bias(cl, dqkv, 3 * F, xin, F, G[ix.layer(l, L_INW)], G[ix.layer(l, L_INB)], B, 2 * F, F));
                TRY(linear_bwd_weight_bias(cl, dqkv + 2 * F, 3 * F, xin, F, tdw, tdb, B, F, F));
            }
            TRY(bbbp_outproj_unfold(cl.st, F, B, tdw, tdb, P[ix.layer(l, L_OUTW)], c.f(plan.fwvt[l]), dsa, F, G[ix.layer(l, L_OUTW)],
                                    G[ix.layer(l, L_OUTB)], G[ix.layer(l, L_INW)] + (size_t)2 * F * F, G[ix.layer(l, L_INB)] + 2 * F));
        } else {
            Section sw(cl.st, SEC_QKV_WGRAD);
            TRY(linear_bwd_weight_bias(cl, dqkv, 3 * F, xin, F, G[ix.layer(l, L_INW)], G[ix.layer(l, L_INB)], B, 3 * F, F));
        }
        if (ln_grads_early) TRY(record_layer_bucket(cl.st, l));
        else TRY(layer_norm_leaves(l));
        if (l > 0 || d->need_input_grad) {
            Section sg(ce.st, SEC_QKV_DGRAD);
            TRY(linear_bwd_input(ce, dqkv, 3 * F, plan.fold ? c.f(plan.fwf[l]) : P[ix.layer(l, L_INW)], l > 0 ? c.f(plan.lgrad[l - 1].dyout) : c.f(plan.dA), F, B, 3 * F, F, dz1, F));
        }
    }
    {
        // bucket 1 is final when the chain AND the leaves are: make the leaf stream wait for the chain's tail, record there
        int dev = 0;
        BBBP_CHECK_HIP(hipGetDevice(&dev));
        if (dev >= 0 && dev < 64) {
            if (!g_bucket1_event[dev]) BBBP_CHECK_HIP(hipEventCreateWithFlags(&g_bucket1_event[dev], hipEventDisableTiming));
            if (ss) TRY(after(ss, ce.st, cl.st));
            hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
            (void)hipStreamIsCapturing(cl.st, &cap);
            g_bucket1_recorded[dev] = cap == hipStreamCaptureStatusNone;
            if (g_bucket1_recorded[dev]) BBBP_CHECK_HIP(hipEventRecord(g_bucket1_event[dev], cl.st));
        }
    }
    if (ss) {
        TRY(join_side(c.st, ss));
        BBBP_CHECK_HIP(hipEventRecord(ss->join2, ss->leaf));
        BBBP_CHECK_HIP(hipStreamWaitEvent(c.st, ss->join2, 0));
    }
    return BBBP_OK;
}

namespace {

hipStream_t capture_stream() {
    static hipStream_t cs = nullptr;       // captures never run on the caller's stream: the legacy default stream cannot be captured
    if (!cs && hipStreamCreateWithFlags(&cs, hipStreamNonBlocking) != hipSuccess) cs = nullptr;
    return cs;
}

// Run `enqueue(stream)` eagerly, or capture it on the library's capture stream and replay the graph on `st`.
template <typename F>
int run_or_replay(const GraphKey& key, hipStream_t st, F&& enqueue) {
    GraphEntry* slot = nullptr;
    const int mode = graph_lookup(key, &slot);
    if (mode == 2) {
        BBBP_CHECK_HIP(hipGraphLaunch(slot->exec, st));
        ++g_graph_replays;
        return BBBP_OK;
    }
    hipStream_t cs = mode == 1 ? capture_stream() : nullptr;
    if (cs && hipStreamBeginCapture(cs, hipStreamCaptureModeRelaxed) == hipSuccess) {
        const int rc = enqueue(cs);
        hipGraph_t graph = nullptr;
        const hipError_t e = hipStreamEndCapture(cs, &graph);
        bool ok = rc == BBBP_OK && e == hipSuccess && graph != nullptr;
        if (ok) ok = hipGraphInstantiate(&slot->exec, graph, nullptr, nullptr, 0) == hipSuccess;
        if (graph) (void)hipGraphDestroy(graph);
        if (ok) {
            slot->seen_only = false;
            ++g_graph_captures;
            BBBP_CHECK_HIP(hipGraphLaunch(slot->exec, st));
            return BBBP_OK;
        }
        (void)hipGetLastError();
        slot->exec = nullptr; slot->stamp = 0;       // do not try this key again soon; run it eagerly below
        if (rc) return rc;
    }
    return enqueue(st);
}

GraphKey make_key(int kind, const bbbp_mixed_desc* d, const void* a, const void* b, const void* c, const void* ws, uint64_t phash) {
    GraphKey k{};
    k.kind = kind; k.d = *d; k.d.seed = 0;
    k.ptr[0] = a; k.ptr[1] = b; k.ptr[2] = c; k.ptr[3] = ws;
    k.phash = phash; k.overlap = overlap_enabled() ? 1 : 0; k.st = nullptr;
    return k;
}

}  // namespace

// Make `stream` wait until gradient bucket `bucket` of the most recent bbbp_mixed_backward on this device is final.
// bucket 0 = the image-FC weight (parameter index bbbp_mixed_bucket_param(d, 0)); bucket 1 = every parameter except
// that weight and the four conv tensors.  Returns BBBP_ERR_ARG when no such
// event exists (no backward yet, or the backward was replayed from a graph): the caller then waits for the whole stream.
extern "C" int bbbp_mixed_backward_wait_bucket(void* stream, int bucket) {
    BBBP_CHECK_ARG(bucket >= 0 && bucket < 2 + 32, "wait_bucket: unknown bucket %d", bucket);
    int dev = 0;
    BBBP_CHECK_HIP(hipGetDevice(&dev));
    BBBP_CHECK_ARG(dev >= 0 && dev < 64, "wait_bucket: device %d", dev);
    hipEvent_t ev = bucket == 0 ? g_bucket_event[dev] : bucket == 1 ? g_bucket1_event[dev] : g_layer_event[dev][bucket - 2];
    const bool ok = bucket == 0 ? g_bucket_recorded[dev] : bucket == 1 ? g_bucket1_recorded[dev] : g_layer_recorded[dev][bucket - 2];
    BBBP_CHECK_ARG(ev && ok, "wait_bucket: no event for bucket %d on device %d", bucket, dev);
    BBBP_CHECK_HIP(hipStreamWaitEvent(static_cast<hipStream_t>(stream), ev, 0));
    return BBBP_OK;
}

// Make `stream` wait until the PARAMETERS of gradient bucket `bucket` may be overwritten: its gradient slice is final and this backward
// pass no longer reads those parameters (an optimizer step pipelined into the pass).  Bucket 0: after the image FC's input-gradient
// GEMM.  Layer l >= 1 (bucket 2 + l): when layer l - 1's bucket is final -- its leaves read what layer l's last kernel (the in_proj
// input gradient, the last reader of layer l's weights) wrote, and the leaf stream is in order, so layer l's own leaves are done too.
// Layer 0 and bucket 1 (everything but the conv tensors): the end of the fingerprint branch and its leaves (bucket 1's event).
extern "C" int bbbp_set_release_events(int on) {
    const int prev = g_release_events ? 1 : 0;
    g_release_events = on != 0;
    return prev;
}

extern "C" int bbbp_mixed_backward_wait_released(void* stream, int bucket) {
    BBBP_CHECK_ARG(bucket >= 0 && bucket < 2 + 32, "wait_released: unknown bucket %d", bucket);
    if (bucket == 0) {
        int dev = 0;
        BBBP_CHECK_HIP(hipGetDevice(&dev));
        BBBP_CHECK_ARG(dev >= 0 && dev < 64 && g_bucket_recorded[dev] && g_bucket0_released[dev] && g_bucket0_released_recorded[dev],
                       "wait_released: no event for bucket 0 on device %d (bbbp_set_release_events(1) before the backward pass)", dev);
        BBBP_CHECK_HIP(hipStreamWaitEvent(static_cast<hipStream_t>(stream), g_bucket0_released[dev], 0));
        return BBBP_OK;
    }
    return bbbp_mixed_backward_wait_bucket(stream, bucket >= 3 ? bucket - 1 : 1);
}
__global__ void positive_gate_kernel(const float* x, uint8_t* gate, long n) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) gate[i] = x[i] > 0.f ? 1 : 0;
}
extern "C" int bbbp_mixed_debug_ffn_gate(void* stream, const bbbp_mixed_desc* d, const void* workspace, int layer, uint8_t* gate) {
    Plan plan;
    TRY(make_plan(d, &plan));
    BBBP_CHECK_ARG(workspace && gate, "debug_ffn_gate: null pointer");
    BBBP_CHECK_ARG(layer >= 0 && layer < plan.L, "debug_ffn_gate: layer %d of %d", layer, plan.L);
    BBBP_CHECK_ARG(!plan.inference, "debug_ffn_gate: an inference workspace keeps no per-layer activations");
    const long n = (long)plan.B * plan.DFF;
    const float* hff = reinterpret_cast<const float*>(static_cast<const char*>(workspace) + plan.layer[layer].hff);
    hipLaunchKernelGGL(positive_gate_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), hff, gate, n);
    BBBP_CHECK_LAUNCH();
    return BBBP_OK;
}
extern "C" int bbbp_mixed_debug_pool_mask(void* stream, const bbbp_mixed_desc* d, const void* workspace, int stage, uint8_t* mask) {
    Plan plan;
    TRY(make_plan(d, &plan));
    BBBP_CHECK_ARG(workspace && mask, "debug_pool_mask: null pointer");
    BBBP_CHECK_ARG(stage == 1 || stage == 2, "debug_pool_mask: stage %d (1 = conv1, 2 = conv2)", stage);
    BBBP_CHECK_ARG(!plan.inference, "debug_pool_mask: an inference workspace keeps no masks");
    const size_t n = stage == 1 ? (size_t)plan.B * C1 * (IMG / 2) * (IMG / 2) : (size_t)plan.B * IMG_FLAT;
    const char* src = static_cast<const char*>(workspace) + (stage == 1 ? plan.mask1 : plan.mask2);
    BBBP_CHECK_HIP(hipMemcpyAsync(mask, src, n, hipMemcpyDeviceToDevice, static_cast<hipStream_t>(stream)));
    return BBBP_OK;
}
extern "C" int bbbp_mixed_bucket_param(const bbbp_mixed_desc* d, int bucket) {
    if (!d || bucket != 0) return -1;
    return PIdx(d).ifc_w();
}
// first parameter index and tensor count of a bucket whose tensors are consecutive in params[] order (0: the image-FC weight;
// 2 + l: the twelve tensors of encoder layer l).  Bucket 1 (everything else) is not one range: returns -1.
extern "C" int bbbp_mixed_bucket_range(const bbbp_mixed_desc* d, int bucket, int* first, int* count) {
    if (!d || !first || !count || d->num_layers < 0 || d->num_layers > 32) return -1;
    const PIdx ix(d);
    if (bucket == 0) { *first = ix.ifc_w(); *count = 1; return 0; }
    if (bucket >= 2 && bucket < 2 + d->num_layers) { *first = ix.layer(bucket - 2, L_INW); *count = L_COUNT; return 0; }
    return -1;
}

extern "C" int bbbp_set_graphs(int on) { const int old = graphs_mode(); g_graphs_mode = on ? 1 : 0; return old; }

extern "C" int bbbp_graph_stats(long* captures, long* replays) {
    if (captures) *captures = g_graph_captures;
    if (replays) *replays = g_graph_replays;
    return BBBP_OK;
}

// The engine's side streams, events, profiling slots and graph cache are per-process state: enqueueing a forward or backward call is
// serialised (two host threads driving two models take turns ENQUEUEING, ~1 ms each; the GPU work itself overlaps as the streams allow).
static std::mutex g_engine_mutex;

extern "C" int bbbp_mixed_forward(void* stream, const bbbp_mixed_desc* d, const float* const* P, float* const* bn_running,
                                  const float* fingerprint, const float* image, float* out, void* workspace,
                                  size_t workspace_bytes) {
    std::lock_guard<std::mutex> engine_lock(g_engine_mutex);
    Plan plan;
    TRY(make_plan(d, &plan));
    BBBP_CHECK_ARG(P && fingerprint && image && out && workspace && bn_running, "mixed_forward: null pointer");
    if (workspace_bytes < plan.total) {
        bbbp_set_error("mixed_forward: workspace %zu < %zu bytes", workspace_bytes, plan.total);
        return BBBP_ERR_WORKSPACE;
    }
    hipStream_t st = static_cast<hipStream_t>(stream);
    // the call's dropout seed goes to device memory first (never part of a graph: it changes every call)
    if (plan.drop) {
        hipLaunchKernelGGL(set_seed_kernel, dim3(1), dim3(1), 0, st,
                           reinterpret_cast<unsigned long long*>(static_cast<char*>(workspace) + plan.seed_slot), (unsigned long long)d->seed);
        BBBP_CHECK_LAUNCH();
    }
    const int np = PIdx(d).count();
    uint64_t h = hash_ptrs(reinterpret_cast<const void* const*>(P), np);
    h = hash_ptrs(reinterpret_cast<const void* const*>(bn_running), 2, h);
    const GraphKey key = make_key(0, d, fingerprint, image, out, workspace, h);
    if (d->collective)              // host callbacks between the launches: nothing to capture
        return forward_enqueue(st, d, P, bn_running, fingerprint, image, out, workspace, workspace_bytes);
    return run_or_replay(key, st, [&](hipStream_t s) {
        return forward_enqueue(s, d, P, bn_running, fingerprint, image, out, workspace, workspace_bytes);
    });
}

extern "C" int bbbp_mixed_backward(void* stream, const bbbp_mixed_desc* d, const float* const* P, float* const* G,
                                   const float* fingerprint, const float* image, const float* dout, void* workspace,
                                   size_t workspace_bytes) {
    std::lock_guard<std::mutex> engine_lock(g_engine_mutex);
    Plan plan;
    TRY(make_plan(d, &plan));
    BBBP_CHECK_ARG(P && G && fingerprint && image && dout && workspace, "mixed_backward: null pointer");
    if (workspace_bytes < plan.total) {
        bbbp_set_error("mixed_backward: workspace %zu < %zu bytes", workspace_bytes, plan.total);
        return BBBP_ERR_WORKSPACE;
    }
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int np = PIdx(d).count();
    uint64_t h = hash_ptrs(reinterpret_cast<const void* const*>(P), np);
    h = hash_ptrs(reinterpret_cast<const void* const*>(G), np, h);
    const GraphKey key = make_key(1, d, fingerprint, image, dout, workspace, h);
    if (d->collective)
        return backward_enqueue(st, d, P, G, fingerprint, image, dout, workspace, workspace_bytes);
    return run_or_replay(key, st, [&](hipStream_t s) {
        return backward_enqueue(s, d, P, G, fingerprint, image, dout, workspace, workspace_bytes);
    });
}
