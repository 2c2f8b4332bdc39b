// Batch-sharded (exact-global-batch) form of the fused head: the BatchNorm statistics and its two backward sums are merged over the
// 16-row blocks of ALL ranks.  `between` is called after the first launch of each direction has written this rank's block partials into
// its slot of `partial` ([world][blocks][2][256] floats) and must make the other ranks' slots arrive (all-gather) before it returns
// control of the stream.  world = 1, between = nullptr: the single-rank head.
#pragma once
#include <functional>
#include <hip/hip_runtime.h>

struct bbbp_head_sync {
    int world = 1, rank = 0;
    std::function<int()> between;
};

int bbbp_head_forward_fused_sync(hipStream_t st, const float* comb, const float* const* fw1, const float* const* fb1,
                                 const float* const* fw2, const float* const* fb2, const float* w0, const float* b0, const float* gamma,
                                 const float* beta, float* running_mean, float* running_var, const float* w3, const float* b3,
                                 const float* w5, const float* b5, const float* w7, const float* b7, float* hid, float* attn, float* fused,
                                 float* h, float* hb, float* bn_mean, float* bn_rstd, float* h2, float* h3, float* out, float* partial,
                                 int B, int training, int concat, const bbbp_head_sync* sync);
int bbbp_head_backward_fused_sync(hipStream_t st, const float* dout, const float* comb, const float* hid, const float* attn, const float* h,
                                  const float* h2, const float* h3, const float* bn_mean, const float* bn_rstd, const float* gamma,
                                  const float* const* fw1, const float* const* fw2, const float* w0, const float* w3, const float* w5,
                                  const float* w7, float* dh3, float* dh2, float* dhb, float* dh, float* dlogit, float* dpre, float* dcomb,
                                  float* dgamma, float* dbeta, float* partial, int B, int training, const bbbp_head_sync* sync);
