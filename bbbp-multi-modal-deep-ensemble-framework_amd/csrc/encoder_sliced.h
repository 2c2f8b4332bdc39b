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
bool bbbp_enc_sliced_supported(int B, int F, int nhead, int dff, int layers);
size_t bbbp_enc_sliced_sync_bytes();
size_t bbbp_enc_sliced_part_bytes(int B, int F);
int bbbp_enc_sliced_fwd(hipStream_t st, const bbbp_enc_sliced_fwd_args* a);
// 1 when a barrier of the most recent launches gave up waiting (a work-group never became resident): results are invalid
int bbbp_enc_sliced_aborted(hipStream_t st, const void* sync, int* aborted);
