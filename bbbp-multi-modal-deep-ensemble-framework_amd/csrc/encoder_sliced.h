// encoder.hip: the fingerprint encoder's forward chain as ONE persistent launch for small batches (B <= 128, one attention head,
// d_model <= 192): every 16-row block of the batch is shared by S work-groups that split the COLUMNS of each product (so the layer's
// weights are streamed by S x ceil(B / 16) work-groups instead of ceil(B / 16)), synchronised by counters in device memory -- a
// row-block barrier between the row-local stages, a grid barrier only where attention couples the rows.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>

constexpr int BBBP_SLICED_MAX_LAYERS = 8;
struct bbbp_enc_sliced_layer {
    const float *win, *bin, *wo, *bo, *g1, *be1, *w1, *b1, *w2, *b2, *g2, *be2;
    float *qkv, *prob, *pd, *ctx, *z1, *y1, *hff, *z2, *y2, *mean1, *rstd1, *mean2, *rstd2;
    uint64_t seed0, seed1, seed2, seed3;        // dropout sites: attention weights, after out_proj, after linear1, after linear2
};
struct bbbp_enc_sliced_fwd_args {
    const float* x0;                            // [B][F] encoder input
    bbbp_enc_sliced_layer lay[BBBP_SLICED_MAX_LAYERS];
    int L;
    const float* wfc; const float* bfc; float* comb; int nfc, ldcomb;      // fingerprint_fc + ReLU into combined[:, :nfc]
    int B, F, DFF;
    float p, scale;
    void* sync;                                 // bbbp_enc_sliced_sync_bytes(): barrier counters, zeroed by the launcher
    float* part;                                // bbbp_enc_sliced_part_bytes(B, F): split-K partials of linear2
};
// backward chain (input gradients) of the same encoder as one persistent launch; the weight / bias / LayerNorm parameter gradients
// stay leaf launches that read the per-layer gradient buffers this kernel fills (dyout, dz2, dff, dhff, dy1, dz1, dsa, dqkv)
struct bbbp_enc_sliced_bwd_layer {
    const float *win, *wo, *g1, *w1, *w2, *g2;
    const float *qkv, *prob, *pd, *z1, *hff, *z2, *mean1, *rstd1, *mean2, *rstd2;
    float *dyout, *dz2, *dff, *dhff, *dy1, *dz1, *dsa, *dqkv;
    uint64_t seed0, seed1, seed3;
};
struct bbbp_enc_sliced_bwd_args {
    const float* dcomb; int ldcomb, nfc; const float* wfc;      // gradient of combined[:, :nfc] (ReLU mask applied), fingerprint_fc weight [nfc][F]
    bbbp_enc_sliced_bwd_layer lay[BBBP_SLICED_MAX_LAYERS];
    int L;
    int B, F, DFF;
    float p, scale;
    void* sync;
    float* part;                                // bbbp_enc_sliced_part_bytes(B, F) -- split-K partials of linear1's input gradient
    float* kvpart;                              // bbbp_enc_sliced_kvpart_bytes(B, F): per row block, its queries' share of dK | dV
};
size_t bbbp_enc_sliced_kvpart_bytes(int B, int F);
int bbbp_enc_sliced_bwd(hipStream_t st, const bbbp_enc_sliced_bwd_args* a);
bool bbbp_enc_sliced_supported(int B, int F, int nhead, int dff, int layers);
size_t bbbp_enc_sliced_sync_bytes();
size_t bbbp_enc_sliced_part_bytes(int B, int F);
int bbbp_enc_sliced_fwd(hipStream_t st, const bbbp_enc_sliced_fwd_args* a);
// 1 when a barrier of the most recent launches gave up waiting (a work-group never became resident): results are invalid
int bbbp_enc_sliced_aborted(hipStream_t st, const void* sync, int* aborted);
