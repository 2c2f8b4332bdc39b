// attention_b3.hip: forward-only self-attention for one (or a few) wide heads (96 < head_dim <= 192: the MACCS encoder's single head of
// 167) at screening batch sizes, split-bf16 on the bf16 matrix pipe (float32 accuracy), eval mode (no dropout, nothing saved).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
bool bbbp_attn_b3_supported(int B, int nhead, int head_dim);
// bytes of the partial (O, max, sum) triples when the key axis is split over work-groups (0: one range, no merge pass)
size_t bbbp_attn_b3_workspace_bytes(int B, int nhead, int head_dim);
int bbbp_attn_b3_fwd(hipStream_t st, const float* qkv, float* ctx, int B, int F, int nhead, float scale, float* workspace, size_t workspace_bytes);
