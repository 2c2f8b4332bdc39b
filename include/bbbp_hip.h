/* bbbp_hip.h -- C ABI of libbbbp_hip.so, the MI355X (gfx950) implementation of the neural hot path of
 * FengDushuo/BBBP-Multi-Modal-Deep-Ensemble-Framework.
 *
 * The reference has no FFI of its own: its hot path is stock torch.nn modules called from Python
 * (SURVEY.md 8b).  These entry points are therefore what a binding for that path would call in
 * place of ATen; each one cites the reference statement(s) it replaces, relative to /root/reference/,
 * with R = Models/multi_input_data_regression_opt_transformer_cnn_20250113.py.
 *
 * Conventions: every pointer is a DEVICE pointer to fp32 data unless typed otherwise; tensors are
 * row-major contiguous in the reference's own layouts (NCHW images, [out,in] Linear weights);
 * `stream` is a hipStream_t (NULL = default stream); work is enqueued, never synchronised;
 * scratch memory is caller-provided (`workspace`, sized by the *_workspace_bytes functions);
 * every function returns 0 on success or a BBBP_ERR_* code, with text in bbbp_last_error();
 * nothing throws across the ABI.  No torch types appear anywhere.
 */
#ifndef BBBP_HIP_H
#define BBBP_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BBBP_OK 0
#define BBBP_ERR_ARG 1
#define BBBP_ERR_HIP 2
#define BBBP_ERR_WORKSPACE 3

#define BBBP_ACT_NONE 0
#define BBBP_ACT_RELU 1
#define BBBP_ACT_TANH 2

int bbbp_abi_version(void);
const char* bbbp_last_error(void);

/* ---- dense GEMM + bias + activation (+ residual) -------------------------------------------
 * C[M,N] = act(alpha * op(A) op(B) + bias[n]) + residual[M,N]; op by transA/transB:
 *   (0,1) A[M][K] B[N][K]: nn.Linear forward           R:79-82, 92, 98-107; nn.MultiheadAttention in/out proj R:75-78
 *   (0,0) A[M][K] B[K][N]: input gradients, P.V        autograd of the above (loss.backward(), R:190)
 *   (1,0) A[K][M] B[K][N]: weight gradients            idem
 * batch > 1 runs `batch` independent products with the given element strides (attention heads). */
size_t bbbp_gemm_workspace_bytes(int M, int N, int K, int batch);
int bbbp_gemm_f32(void* stream, int transA, int transB, int M, int N, int K, float alpha,
                  const float* A, int lda, const float* B, int ldb, float* C, int ldc,
                  const float* bias, const float* residual, int ldr, int act,
                  int batch, long strideA, long strideB, long strideC, long strideR,
                  void* workspace, size_t workspace_bytes);

/* One product of a grouped launch: the bbbp_gemm_f32 arguments as a struct, plus an optional gate: the result is
 * multiplied by gate_scale where gate[m][n] > 0 and by 0 elsewhere (after bias/act, before the residual) -- the
 * ReLU(+dropout) backward of linear1 (R:75-78 under loss.backward(), R:190) folded into the input-gradient GEMM. */
typedef struct bbbp_gemm_desc {
    int transA, transB, M, N, K;
    float alpha;
    const float* A; int lda;
    const float* B; int ldb;
    float* C; int ldc;
    const float* bias;
    const float* residual; int ldr;
    int act;
    const float* gate; int ldg; float gate_scale;
    int batch;
    long strideA, strideB, strideC, strideR, strideG;
    int gate_after_residual;      /* 0: gate(act(..)) + residual;  1: gate(act(..) + residual) */
    float* asum;                  /* optional, layout A^T B only (a Linear's weight gradient dW = dY^T X): asum[m] = sum_k A[k][m], the
                                     matching bias gradient, produced by the same MFMAs through a virtual all-ones column N of B.
                                     Honoured by the small-product path only: ask bbbp_gemm_folds_asum first. */
    float drop_p;                 /* optional dropout of the OUTPUT, applied after bias/act (before gate/residual): element (m, n) is
                                     scaled by the keep-scale of element m * N + n of Philox stream drop_seed -- the stream bbbp_dropout
                                     draws for a contiguous [M][N] tensor, so act + dropout of linear1 (R:75-78, train mode) is one launch.
                                     0 = off.  Small-product path only (bbbp_gemm_folds_asum(M, N, K, batch) == 1). */
    unsigned long long drop_seed;
} bbbp_gemm_desc;
/* 1 when a product of this shape (layout A^T B) takes the path that honours bbbp_gemm_desc.asum. */
int bbbp_gemm_folds_asum(int M, int N, int K, int batch);
/* `count` independent products (outputs must not alias another product's operands).  Products that become ready
 * together -- dV | dP and dQ | dK of the attention backward -- go out as ONE launch when both are small; anything
 * else runs back to back on `stream`.  Results are identical to `count` bbbp_gemm_f32 calls. */
int bbbp_gemm_f32_grouped(void* stream, const bbbp_gemm_desc* problems, int count, void* workspace, size_t workspace_bytes);

/* ---- Conv2d(k3,s1,p1) + ReLU + MaxPool2d(2,2), NCHW ------------------------------------------
 * forward: R:85-87 (3->32, 128x128) and R:88-90 (32->64, 64x64).  y is the pooled output,
 * mask[B][cout][H/2][W/2] (u8) records the arg-max of each 2x2 window (0..3, PyTorch first-max order)
 * or 4 where the ReLU is inactive; backward consumes it instead of the pre-pool activation.  The forward's `mask` may be NULL
 * (round 4): a forward-only call (eval loop R:195-203, screening) keeps no decisions -- the engine's inference plans pass NULL.
 * bwd_data / bwd_weight: autograd of the same statements under loss.backward() (R:190). */
size_t bbbp_conv3x3_workspace_bytes(int B, int cin, int cout, int H, int W);
/* Measurement aid (bench.py): shader-clock cycles and 100 MHz wall ticks that work-group 0 of the most recent
 * forward / data-gradient conv launch ran for (synchronises the device). */
int bbbp_conv_last_clock(unsigned long long* shader_cycles, unsigned long long* ticks_100mhz);
/* Selects the algorithm of the 32->64 @ 64x64 stage (R:88-90).  bit 0 = forward, bit 1 = data gradient run as Winograd
 * F(2x2,3x3) in float32 (2.25x fewer multiplies, same tensors and mask, results within 1e-6 of the direct form).
 * bit 2 = forward, bit 3 = data gradient, bit 4 = weight gradient run as the direct implicit GEMM on the bf16 matrix pipe with
 * every float32 operand split into three bf16 pieces (six bf16 products per float32 product, float32 accumulate: float32
 * accuracy, csrc/conv_b3.hip); these bits take precedence over the Winograd bits.  bit 5 = the weight gradient of the 3->32 @
 * 128x128 stage (R:85-87) in the same split-bf16 form; bit 6 = that stage's forward (csrc/conv_b3c1.hip: channel-innermost LDS strip,
 * in-lane pooling windows).  Inside bbbp_mixed_forward the engine keeps the first stage's forward on the f32 kernel while a training
 * step's encoder chain runs beside it (the split-bf16 kernel leaves that chain no wave slots: measured slower for the step), so bit 6
 * acts on forward-only (inference-plan) passes -- eval loops, screening --, the encoder-less two-branch model and the op-level entry point.
 * 0 = direct implicit GEMM on the f32 MFMA everywhere.  Initial value: environment BBBP_CONV_WINOGRAD, else 252.
 * bit 7 (128, round 3): conv2's split-bf16 weight gradient on the 2:4 structured-sparse MFMA (v_smfmac_f32_32x32x32_bf16): the pooled
 * gradient expanded through the arg-max mask has at most one nonzero per pixel pair, so its compressed form is the pooled tensor itself
 * -- half the matrix-pipe time, same six piece products, same float32 accuracy; bit 8 (256): its 4-wave form wherever it runs (tests). */
int bbbp_set_conv_winograd(int mask);
int bbbp_get_conv_winograd(void);
/* Measurement aid: with BBBP_WINO_PROBE=1 in the environment the Winograd kernels stamp the shader clock at phase boundaries;
 * phases4 = cycles work-group 0 spent in {accumulator init, k-steps, stage hand-over, output transform} of the last launch. */
int bbbp_conv_winograd_phases(unsigned long long* phases4);
/* Same for the split-bf16 kernels (BBBP_B3_PROBE=1): [0] global-load issue, [1] MFMA block, [2] split + LDS writes + barrier, [3] epilogue. */
int bbbp_conv_b3_phases(unsigned long long* phases4);
int bbbp_conv3x3_relu_pool_fwd(void* stream, const float* x, const float* w, const float* bias,
                               float* y, uint8_t* mask, int B, int cin, int cout, int H, int W,
                               void* workspace, size_t workspace_bytes);
int bbbp_conv3x3_relu_pool_bwd_data(void* stream, const float* gy, const uint8_t* mask, const float* w,
                                    float* dx, int B, int cin, int cout, int H, int W,
                                    void* workspace, size_t workspace_bytes);
int bbbp_conv3x3_relu_pool_bwd_weight(void* stream, const float* x, const float* gy, const uint8_t* mask,
                                      float* dw, float* db, int B, int cin, int cout, int H, int W,
                                      void* workspace, size_t workspace_bytes);

/* ---- LayerNorm(x_dropped + residual): nn.TransformerEncoderLayer.norm1/norm2 (R:75-78) ---------
 * forward overwrites x with z = dropout(x) + residual (kept for backward) and writes y, mean, rstd.
 * backward: dz (gradient of the residual input), dx (optional; dz times the dropout keep-scale). */
int bbbp_layernorm_fwd(void* stream, float* x_inout_z, const float* residual, float* y, const float* gamma,
                       const float* beta, float* mean, float* rstd, int rows, int cols, float eps,
                       float dropout_p, uint64_t seed);
/* Linear + dropout + residual + LayerNorm as ONE launch for narrow outputs (the encoder's out_proj -> norm1 and linear2 -> norm2 at
 * F = 167, R:75-78): z = dropout(x W^T + bias) + residual (written, kept for backward), y = LayerNorm(z), mean / rstd per row.  The
 * dropout draws the elements bbbp_layernorm_fwd draws.  bbbp_linear_layernorm_supported: N <= 256, K <= 8192.  The engine uses it for
 * out_proj -> norm1 only with BBBP_FUSED_LINEAR_LN=1 in the environment (measured slower inside the B = 512 step). */
int bbbp_linear_layernorm_supported(int M, int N, int K);
int bbbp_linear_layernorm_fwd(void* stream, const float* x, int ldx, const float* W, const float* bias, const float* residual, int ldr,
                              float* z, int ldz, float* y, int ldy, const float* gamma, const float* beta, float* mean, float* rstd,
                              int M, int N, int K, float eps, float dropout_p, uint64_t seed);
/* LayerNorm ABSORBED by the Linear that consumes it (round 4; the encoder's norm1 -> linear1, norm2 -> the next in_proj / fingerprint_fc,
 * R:75-82): ONE launch computes out = dropout(act(LayerNorm(z) W^T + bias)) [M][N] and writes y = LayerNorm(z) [M][K], mean, rstd (what
 * bbbp_layernorm_fwd writes for a z that already holds dropout(x) + residual -- the producing GEMM's epilogue does that now).  Per output
 * element out = rstd_m (sum_k (z - x0_m) gamma_k W[n][k] - (mu_m - x0_m) c_n) + d_n + bias_n with c_n = sum_k gamma_k W[n][k],
 * d_n = sum_k beta_k W[n][k], all taken from the product's own operand stream (csrc/gemm.hip: gemm_direct_lna_kernel); y / mean / rstd come
 * from one extra column of work-groups of the same launch.  K <= 192 (bbbp_layernorm_linear_supported).  The dropout draws element
 * m * N + n of `seed`, the stream bbbp_dropout draws for a contiguous [M][N] tensor. */
int bbbp_layernorm_linear_supported(int M, int N, int K);
int bbbp_layernorm_linear_fwd(void* stream, const float* z, int ldz, const float* gamma, const float* beta, float eps,
                              const float* W, const float* bias, float* out, int ldo, int act, float dropout_p, uint64_t seed,
                              float* y, int ldy, float* mean, float* rstd, int M, int N, int K);
int bbbp_layernorm_bwd(void* stream, const float* dy, const float* z, const float* gamma, const float* mean,
                       const float* rstd, float* dz, float* dx, float* dgamma, float* dbeta, int rows, int cols,
                       float dropout_p, uint64_t seed);

/* ---- row softmax (attention probabilities, R:75-78 via F.scaled_dot_product_attention) --------- */
int bbbp_softmax_fwd(void* stream, float* x_inout, float* dropped_out, long rows, int cols, float dropout_p,
                     uint64_t seed);
int bbbp_softmax_bwd(void* stream, float* dprob_inout, const float* prob, long rows, int cols, float dropout_p,
                     uint64_t seed);

/* ---- nn.Dropout: y = x * keep-scale; the same call with the same seed is its backward ---------- */
int bbbp_dropout(void* stream, const float* x, float* y, long n, float p, uint64_t seed);

/* ---- nn.BatchNorm1d (R:101): batch statistics + running-stat update when training ------------- */
int bbbp_batchnorm1d_fwd(void* stream, const float* x, float* y, const float* gamma, const float* beta,
                         float* running_mean, float* running_var, float* save_mean, float* save_rstd, int rows,
                         int cols, float eps, float momentum, int training);
int bbbp_batchnorm1d_bwd(void* stream, const float* dy, const float* x, const float* gamma, const float* save_mean,
                         const float* save_rstd, float* dx, float* dgamma, float* dbeta, int rows, int cols,
                         int training);
/* ... followed by the backward of the ReLU that produced x (fc.1 -> fc.2 of the head, R:99-100): dx = 0 where x <= 0 */
int bbbp_batchnorm1d_bwd_relu(void* stream, const float* dy, const float* x, const float* gamma, const float* save_mean,
                         const float* save_rstd, float* dx, float* dgamma, float* dbeta, int rows, int cols,
                         int training);

/* Pieces for a BatchNorm1d whose batch is sharded over ranks (exact-global-batch data parallelism): local column sums of
 * (x - centre) and (x - centre)^2, and dx from sums taken over the global batch of n_global rows. */
int bbbp_column_moments(void* stream, const float* x, const float* centre, float* sum_out, float* sumsq_out, int rows, int cols);
int bbbp_batchnorm1d_bwd_apply(void* stream, const float* dy, const float* x, const float* gamma, const float* mean,
                               const float* rstd, const float* sum_dy, const float* sum_dy_xhat, float* dx, int rows,
                               int cols, long n_global);

/* ---- backward of "+ bias, activation": dy *= act'(y) * scale in place, dbias = column sums ------ */
int bbbp_bias_act_bwd(void* stream, float* dy_inout, int lddy, const float* y, int ldy, float* dbias, int rows,
                      int cols, int act, float scale);

/* ---- MultiHeadAttentionFusion.forward combine step (R:60-65) and its backward ------------------ */
int bbbp_fusion_combine_fwd(void* stream, const float* combined, const float* hid, const float* const* w2,
                            const float* const* b2, float* out, float* attn, int rows, int dim, int hidden,
                            int num_heads);
int bbbp_fusion_combine_bwd(void* stream, const float* dout, const float* combined, const float* hid,
                            const float* attn, const float* const* w2, float* dcombined, float* dlogit,
                            float* dpre, int rows, int dim, int hidden, int num_heads);

/* ---- nn.MSELoss (R:143,189) and its gradient; optim.AdamW.step (R:172,191) -------------------- */
int bbbp_mse(void* stream, const float* pred, const float* target, float* loss, float* dpred, int n, float grad_scale);
/* AdamW: the hyper-parameters are DOUBLES, as torch.optim.AdamW holds them (Python floats): the step's float32 constants are derived in
 * double the way torch derives them and rounded once, and the kernel executes torch's float32 op sequence with every rounding pinned --
 * the moments are torch's bits, the parameters torch's bits on > 99 % of the elements (tests/test_gpu_round4.py). */
int bbbp_adamw_step(void* stream, float* param, const float* grad, float* exp_avg, float* exp_avg_sq, long n,
                    double lr, double beta1, double beta2, double eps, double weight_decay, int step, double grad_scale);
/* The same step with the slice [lo, hi) of the flat buffers updated on a library-owned side stream (round 4).  Everything else is updated on
 * `stream`; the slice starts once the gradients are final (an event on `stream`) and runs beside whatever `stream` does next.  Readers are
 * ordered behind it by the library: bbbp_mixed_forward waits right before its first read of a tensor inside the slice (the image-FC weight:
 * 62 % of the optimizer's bytes, first read 0.8 ms into the next forward pass), every other entry point that reads parameters
 * (bbbp_adamw_step*, bbbp_mixed_backward) waits at its start, and bbbp_param_sync(stream) orders ANY stream behind it (call it before
 * reading parameters outside this library: state_dict(), checkpoints).  Element-wise arithmetic: the result is bit-identical to
 * bbbp_adamw_step's.  Inside a stream capture, or with an empty slice, it IS bbbp_adamw_step. */
int bbbp_adamw_step_deferred(void* stream, float* param, const float* grad, float* exp_avg, float* exp_avg_sq, long n, long lo, long hi,
                             double lr, double beta1, double beta2, double eps, double weight_decay, int step, double grad_scale);
/* Multi-tensor form (round 4): parameters and moments are ONE flat buffer of n elements, the gradients are n_tensors separate device tensors
 * (what autograd leaves behind for a per-op model: torch.optim.AdamW's foreach path, here one launch).  `table_dev`: device memory owned by the
 * caller and read on `stream` -- long offsets[n_tensors + 1] (element offsets into the flat buffer, offsets[0] = 0, offsets[n_tensors] = n)
 * followed by const float* grads[n_tensors].  `hyper_dev` (nullable): eight floats in device memory as bbbp_adamw_hyper_store writes them; then
 * lr ... grad_scale are ignored -- a step captured into a HIP graph replays with whatever the caller stored there before the launch.
 * Same per-element expressions as bbbp_adamw_step: bit-identical to n_tensors single launches. */
int bbbp_adamw_step_multi(void* stream, float* param, float* exp_avg, float* exp_avg_sq, long n, const void* table_dev, int n_tensors,
                          double lr, double beta1, double beta2, double eps, double weight_decay, int step, double grad_scale, const float* hyper_dev);
/* stores those eight floats (derived on the host in double, as bbbp_adamw_step derives them) to device memory in stream order */
int bbbp_adamw_hyper_store(void* stream, float* hyper_dev, double lr, double beta1, double beta2, double eps, double weight_decay, int step, double grad_scale);
/* The structured-sparse weight gradient of the 32->64 / 64->128 / 128->256 stages has an 8-wave form (fastest alone) and a 4-wave form (one
 * wave per SIMD: faster when another branch's small kernels share the GPU).  bbbp_mixed_backward chooses by itself; a caller composing the
 * model op by op with its branches on two streams sets this for the CALLING THREAD around bbbp_conv3x3_relu_pool_bwd_weight.  Returns the
 * previous setting. */
int bbbp_set_conv_wgrad_beside_encoder(int on);
/* Forward of the 32->64 / 64->128 stages on 64 x 64 maps: 1 selects, for the CALLING THREAD, the software-pipelined kernel that runs one
 * work-group per CU (slower alone, faster for a step whose encoder chain runs beside it; bbbp_mixed_forward chooses by itself).  Outputs and
 * decisions are bit-identical.  Returns the previous setting. */
int bbbp_set_conv2_fwd_pipe(int on);
/* Op-level dropout streams keyed from device memory (round 4).  With a base set for the CALLING THREAD, every seeded entry point of this
 * header (dropout, softmax, layernorm, linear with output dropout, attention) uses the stream  *base * 0x9E3779B97F4A7C15 + seed  instead of
 * `seed`: a training step captured into a HIP graph draws new masks on each replay when the caller bumps the 64-bit integer at `base_dev`
 * before it.  NULL restores the default.  Forward and backward of one step must see the same base and the same *base. */
int bbbp_set_seed_base(const void* base_dev);
int bbbp_param_sync(void* stream);
void* bbbp_param_stream(void);     /* the side stream of the current device (NULL before the first deferred step) */
int bbbp_scale(void* stream, float* x, long n, float s);

/* ---- input pipeline at the tensor boundary (Descriptors/multi_input_data_preprocess_maccs_opt_IsolationForest_fixed_1.py) ----
 * :56-71  PIL convert('RGB') + torchvision Resize((128,128)) + ToTensor: Pillow's fixed-point two-pass bilinear
 *         resampling, bit-exact; coefficient tables (bounds [out][2] = {first, count}, kk [out][ksize] int32) come from
 *         the host (preprocess.py: pil_resample_coeffs).  src [N][Hs][Ws][3] u8; tmp [N][Hs][Wo][3] u8 scratch;
 *         dst_u8 [N][Ho][Wo][3] (optional); dst_chw [N][3][Ho][Wo] f32 = byte / 255.
 * :86-101 StandardScaler().fit_transform over one chunk of rows of hstack([fingerprint u8, image f32]), float64
 *         statistics, float32 output; mean_out / scale_out [F+I] optional. */
int bbbp_resize_bilinear_totensor(void* stream, const uint8_t* src, uint8_t* tmp, uint8_t* dst_u8, float* dst_chw,
                                  const int* bounds_x, const int* kk_x, int ksize_x, const int* bounds_y,
                                  const int* kk_y, int ksize_y, int N, int Hs, int Ws, int Ho, int Wo);
int bbbp_standardize_chunk(void* stream, const uint8_t* fingerprint_u8, const float* image, float* fingerprint_out,
                           float* image_out, double* mean_out, double* scale_out, int rows, int F, int I);

/* ---- whole-model entry points: MixedInputModel.forward (R:109-119) and its autograd ------------
 * params[]: device pointers in the reference's named_parameters() order:
 *   per encoder layer l (12): self_attn.in_proj_weight, in_proj_bias, out_proj.weight, out_proj.bias,
 *        linear1.weight, linear1.bias, linear2.weight, linear2.bias, norm1.weight, norm1.bias, norm2.weight, norm2.bias
 *   fingerprint_fc.0.{weight,bias}; image_cnn.{0,3,7}.{weight,bias};
 *   attention_fusion.attention_heads.h.{0.weight,0.bias,2.weight,2.bias} for h in 0..3;
 *   fc.0.{w,b}, fc.2.{w,b} (BatchNorm affine), fc.3.{w,b}, fc.5.{w,b}, fc.7.{w,b}
 * num_layers = 0 drops the encoder (fingerprint_fc reads the fingerprint itself): with fusion = 1 that is BASELINE
 * config 2, the two-branch MACCS-Linear + image-CNN + concat + BatchNorm-head model.
 * grads[]: same order, written (not accumulated).  bn_running: {running_mean, running_var} of fc.2.
 * The workspace carries the saved activations from forward to backward. */
/* Collective hook of the exact-global-batch mode.  Called on the HOST while bbbp_mixed_forward / _backward enqueue their work; the
 * callee must enqueue the collective so that it is ordered after everything enqueued on `stream` so far and before anything enqueued on
 * it later (e.g. torch.distributed under torch.cuda.stream(ExternalStream(stream))).  Buffers are byte offsets into the call's
 * workspace; `count` is in floats PER RANK.  Return 0 on success.
 *   BBBP_COLL_ALLGATHER        recv_off: [world][count] floats, this rank's slot (= send_off) already filled: all-gather in place
 *   BBBP_COLL_REDUCE_SCATTER   send_off: [world][count] floats; recv_off: [count] floats = sum over ranks of their slot `rank`
 * `what`: BBBP_COLL_KV / _DKV (layer = encoder layer) or BBBP_COLL_BN_FWD / _BN_BWD (layer = -1). */
enum { BBBP_COLL_ALLGATHER = 0, BBBP_COLL_REDUCE_SCATTER = 1 };
enum { BBBP_COLL_KV = 0, BBBP_COLL_DKV = 1, BBBP_COLL_BN_FWD = 2, BBBP_COLL_BN_BWD = 3 };
typedef int (*bbbp_collective_fn)(void* ctx, int op, int what, int layer, size_t send_off, size_t recv_off, size_t count, void* stream);

typedef struct {
    int batch;            /* B (also the attention sequence length: R:110-111, batch_first=False) */
    int fingerprint_size; /* F = d_model */
    int nhead;            /* R:71-73 */
    int num_layers;       /* 6 */
    int dim_feedforward;  /* 2048 */
    int training;         /* BatchNorm batch statistics + dropout */
    float dropout_p;      /* 0.1 in the encoder when training */
    uint64_t seed;        /* dropout stream of this step */
    int need_input_grad;  /* unused by the reference (inputs do not require grad) */
    int fusion;           /* 0: MultiHeadAttentionFusion(256, 4 heads) (R:48-65); 1: plain torch.cat of the two branch outputs
                             (Descriptors/multi_input_data_regression_opt_round_2_transformer_cnn.py:99) -- the 16
                             attention_fusion.* entries are then absent from params[] / grads[] */
    int inference;        /* 1: forward only (torch.no_grad()): the workspace omits every backward temporary and the encoder
                             layers share one set of activation buffers; bbbp_mixed_backward refuses such a workspace */
    /* Exact-global-batch data parallelism (SURVEY 8e mode 2; round 3): `batch` is THIS rank's shard of a mini-batch of world * batch
     * molecules.  The two places where the reference couples the molecules of a mini-batch are made global through `collective`:
     * every encoder layer attends over the keys / values of all ranks (the engine packs K | V of its rows into its slot of a
     * [world][batch][2F] buffer and asks for an in-place all-gather; backward, its queries' share of dK | dV for ALL keys is
     * reduce-scattered), and the head's BatchNorm1d merges the per-16-row (mean, M2) blocks / backward sums of all ranks (all-gather of
     * [blocks][2][256] floats).  With gradients averaged over ranks an N-rank step equals the single-GPU step at world * batch.
     * collective == NULL (and world <= 1): the replica engine, nothing changes. */
    int world;            /* ranks sharing the mini-batch (0 / 1 with a callback: the same code path with no-op collectives) */
    int rank;             /* this rank's position: its rows are global rows rank * batch .. */
    bbbp_collective_fn collective;
    void* collective_ctx;
} bbbp_mixed_desc;

int bbbp_mixed_num_params(const bbbp_mixed_desc* d);
size_t bbbp_mixed_workspace_bytes(const bbbp_mixed_desc* d);
int bbbp_mixed_forward(void* stream, const bbbp_mixed_desc* d, const float* const* params, float* const* bn_running,
                       const float* fingerprint, const float* image, float* out, void* workspace, size_t workspace_bytes);
int bbbp_mixed_backward(void* stream, const bbbp_mixed_desc* d, const float* const* params, float* const* grads,
                        const float* fingerprint, const float* image, const float* dout, void* workspace,
                        size_t workspace_bytes);

/* ---- batched trainer for scikit-learn style MLP classifiers (Models/model_opt_maccs.py:133,170-181: MLPClassifier under
 * GridSearchCV).  One persistent work-group trains one model; float64, scikit-learn's
 * arithmetic (binary log-loss + L2, Adam, training-loss stopping rule).  The array of models lives in DEVICE memory; the
 * host fills it (including the device pointers), supplies `order` = the global row ids of each of the next `epochs`
 * epochs in visiting order, launches, and reads back n_iter / done / loss_curve.  parameter layout: [W0|b0|W1|b1|...],
 * W[l] is [units[l]][units[l+1]] row-major (scikit-learn's coefs_). */
typedef struct bbbp_mlp_model {
    int n_layers;                 /* weight layers: hidden layers + 1 (2 or 3 for the reference grid; at most 4) */
    int units[5];                 /* units[0] = n_features ... units[n_layers] = 1 */
    int activation;               /* hidden activation: 0 relu, 1 tanh */
    int batch_size;
    int n_train;                  /* rows visited per epoch */
    int n_iter_no_change, max_iter;
    double lr_init, alpha, beta1, beta2, eps, tol;
    double* params; double* adam_m; double* adam_v; double* grads;
    double* act; double* delta;   /* scratch: batch_size * (sum of units[1..n_layers]) each */
    const int* order;             /* [epochs][n_train] */
    double* loss_curve;           /* [max_iter] */
    long t;                       /* Adam step count */
    double best_loss;             /* +inf at start */
    int no_improve, n_iter, done;
} bbbp_mlp_model;
int bbbp_mlp_train_epochs(void* stream, bbbp_mlp_model* models_dev, int n_models, const double* X, const double* y,
                          int n_features, int epochs);
/* out[i] = P(class 1 | X[i]) under models_dev[model]; hidden layers up to 256 units */
int bbbp_mlp_predict_proba(void* stream, const bbbp_mlp_model* models_dev, int model, const double* X, int n, int n_features,
                           int max_units, double* out);
/* Phase profile of the trainer's persistent kernel: on != 0 selects an instrumented build of the kernel for later bbbp_mlp_train_epochs calls
 * (work-group 0 adds the shader-clock cycles of each phase of every mini-batch to 16 counters) and clears them; cycles16 (host, nullable) gets
 * the counters accumulated so far: 0-2 forward layer l, 3 loss, 4 + 2 l / 5 + 2 l delta / weight gradient + Adam of layer l, 10 mini-batch
 * tail, 11 epoch tail, 15 mini-batches. */
int bbbp_mlp_profile(int on, unsigned long long* cycles16);
/* ... and per work-group (= per fit, in device-array order) of the last instrumented launch: out[3 g] / out[3 g + 1] = 100 MHz wall ticks at its
 * start / end, out[3 g + 2] = shader cycles in between (who finishes last, and at which clock) */
int bbbp_mlp_profile_groups(unsigned long long* out, int n_groups);

/* ---- random-forest regression inference (rf base learner of the stack, ...20250113.py:262-266, 394-403) -------------
 * scikit-learn's semantics: float32 X, go left when (double)x[feature] <= threshold, leaf value in float64, mean over
 * trees in float64.  Node arrays are the trees' arrays concatenated (child indices rebased), root[t] = first node of
 * tree t (n_trees + 1 entries); partial: bbbp_forest_groups(n_trees) * n doubles of scratch. */
int bbbp_forest_groups(int n_trees);
int bbbp_forest_predict(void* stream, const float* X, long n, int n_features, const int* left, const int* right,
                        const int* feature, const double* threshold, const double* value, const int* root, int n_trees,
                        double* partial, double* out);

/* ---- gradient-boosted regression trees, prediction (the xgb base learner of the stack, ...20250108.py:186-189; fitted model
 * Models/xgb_model_maccs.pkl) -- XGBoost's predict rule: left when x[feature] < split_condition (float32), the default child when
 * the feature is NaN, leaf value in split_condition[leaf], out = base_score + float32 sum of the leaves in tree order.  Node arrays
 * concatenated over trees (child indices rebased, -1 = leaf), root[t] = first node of tree t; leaf_scratch: n_trees * n floats.
 * The caller validates the arrays (boosters.validate_gbt: children inside their tree, one parent per node, split indices below
 * n_features); the device walk stops after 1024 levels all the same and then yields NaN for that (tree, row). */
int bbbp_gbt_predict(void* stream, const float* X, long n, int n_features, const int* left, const int* right, const int* feature,
                     const float* split_condition, const uint8_t* default_left, const int* root, int n_trees, float base_score,
                     float* leaf_scratch, float* out);

/* ---- oblivious (symmetric) trees, prediction (the cat base learner of the stack, ...20250108.py:192-195) -- CatBoost's rule for float
 * features: bit i of a tree's leaf index is x[split_feature] > split_border of its level i (NaN: false, or true where nan_true[feature]),
 * out = scale * (float64 sum of leaf_values[tree_first_leaf[t] + index] in tree order) + bias.  Splits concatenated over trees
 * (tree_first_split: n_trees + 1 entries, at most 31 levels per tree); leaf_scratch: n_trees * n doubles. */
int bbbp_oblivious_predict(void* stream, const float* X, long n, int n_features, const int* split_feature, const float* split_border,
                           const uint8_t* nan_true, const int* tree_first_split, const long* tree_first_leaf, const double* leaf_values,
                           int n_trees, double scale, double bias, double* leaf_scratch, double* out);

/* ---- optional per-section timing (HIP events on the launch stream; used by bench.py's roofline leg) ----
 * enable(1), run steps, synchronise the stream, collect(ms_sum[n], count[n]) with n = num_sections(). */
int bbbp_set_partition(int reserved_cus, size_t small_lds_pad);   /* CU partition knob, see csrc/common.h */
/* CUs kept out of every persistent grid (conv / GEMM grids are sized from the CU count): room for the collective library's kernels
 * beside the persistent work-groups in a multi-GPU run.  Initial value BBBP_COMM_CUS (default 0); ignored when fewer than 64 CUs would
 * remain.  bench.py's N-rank diagnostic pass measures 0 against 8 and keeps the faster one.  Returns the previous value. */
int bbbp_set_comm_cus(int n);
/* The head / fusion-block input-gradient chain of bbbp_mixed_backward as two fused launches instead of ten (default on since
 * round 2: 3.32 -> 3.27 ms per step at B = 512).  Returns the previous setting.  Initial value: BBBP_FUSED_HEAD_BWD. */
int bbbp_set_fused_head_bwd(int on);
/* Alternative schedules of the fingerprint encoder, a bit mask (default 0: every one measured slower than the launch-per-op chain, see
 * DESIGN.md section 3; initial value BBBP_FUSED_ENCODER).  All of them fill the same workspace and draw the same dropout masks.
 * bit 0: the row-local stretches of a layer (out_proj .. LayerNorm2 + the next in_proj forward; LayerNorm2 backward .. out_proj input
 *        gradient backward) as ONE launch each (csrc/encoder.hip: enc_row_*_kernel) instead of 6 + 6, for d_model <= 192;
 * bit 1: batches of at most 128 molecules, one head, d_model <= 176: the whole FORWARD chain of the encoder (all layers + fingerprint_fc)
 *        as one persistent launch (enc_sliced_fwd_kernel: 16-row blocks shared by column-slicing work-groups, barriers through counters
 *        in the workspace);
 * bit 2: the same for the BACKWARD input-gradient chain (enc_sliced_bwd_kernel); the weight gradients stay leaf launches.
 * Returns the previous mask. */
int bbbp_set_fused_encoder(int mode);
/* One-head encoder layers (nhead 1: a prime fingerprint width such as 167) on the launch-per-op schedule: out_proj folded into the value
 * projection (csrc/fold.hip).  out_proj(Pd V) = Pd (x (Wo Wv)^T + 1 (Wo bv)^T) + bo, so in_proj produces [Q | K | VW] with the folded
 * weight block W' = Wo Wv (one launch per step for all layers), the out_proj GEMM leaves the forward chain, its input-gradient GEMM leaves
 * the backward chain, and its weight gradient is unfolded from the in_proj weight gradient's V block (dWo = dW' Wv^T + db' bv^T,
 * d[Wv | bv] = Wo^T [dW' | db'], dbo = column sums).  The same function as nn.TransformerEncoderLayer (...20250113.py:75-78) up to
 * float32 rounding of the reassociated products.  A bit mask, default 3 (initial value BBBP_FOLD_OUTPROJ): bit 0 the launch-per-op
 * schedule with materialised probabilities (training steps, eval batches below 2048 rows); bit 1 also the forward-only split-bf16
 * attention kernel of 2048+ row plans (no dropout there: bo rides in b'; those plans run on one stream); 0 keeps
 * the reference's operation order.  Never applied with the small-head fused attention, the exact-global-batch mode or
 * bbbp_set_fused_encoder != 0.  Changes the workspace layout: set it before bbbp_mixed_workspace_bytes / the forward call of a step.
 * Returns the previous mask. */
int bbbp_set_fold_outproj(int mask);
/* Fused flash-style self-attention (csrc/attention.hip, csrc/attention_b3.hip), a bit mask (default 13, initial value BBBP_FLASH_ATTENTION):
 * bit 0: many heads of head_dim 8 / 16 (F = 2048: 256 x 8), one work-group per head, scores in registers, no [nhead, B, B] tensors;
 * bit 1: one wide head of 161 .. 176 columns (F = 167, nhead = 1), operands straight from global memory, everywhere -- correct but
 *        measured slower than the batched-GEMM + softmax schedule in the B = 512 .. 2048 training steps, hence opt-in;
 * bit 2: that kernel only where it is faster: forward-only (inference) plans of 2048 rows and more (screening batches);
 * bit 3: forward-only plans of 2048 rows and more (bit 4: from 256 rows, for tests) with a wide head (96 < head_dim <= 192) on the bf16 matrix pipe with split operands
 *        (attention_b3.hip: K / V^T tiles staged once per 128 queries, scores fed back as the second product's operand, key axis split
 *        over work-groups + a merge pass); takes precedence over bit 2.
 * 0 selects the batched GEMM + softmax schedule everywhere.  Returns the previous mask.  Changes the workspace layout: set it
 * before the forward call, not between forward and backward. */
int bbbp_set_flash_attention(int on);
/* Products that take the 128 x 128 tile plan (the F = 2048 encoder's GEMMs, the 65536-wide image FC): 1 (default; initial value
 * BBBP_GEMM_SPLIT_BF16) runs them on the bf16 matrix pipe with every float32 operand split into three bf16 pieces (six MFMAs per
 * k-step, f32 accumulate, float32 accuracy; csrc/gemm.hip: gemm_b3_kernel), 0 on the f32 MFMA.  Returns the previous setting. */
int bbbp_set_gemm_split_bf16(int on);
/* Split-K launches of the split-bf16 GEMM: 1 lets the K range that arrives last at an output tile sum the tile's slabs in split order
 * and apply the epilogue inside the GEMM launch (per-tile arrival counters, slabs written through to the coherence point), 0 (default;
 * initial value BBBP_GEMM_FOLD_REDUCE) runs the separate reduce kernel.  Both produce the same bits; the in-kernel form is measured
 * slower on MI355X.  Returns the previous setting. */
int bbbp_set_gemm_fold_reduce(int on);
/* Debug: shader cycles of work-group 0 / wave 0 of the last split-bf16 GEMM launched with BBBP_GEMM_B3_PROBE=1 in the environment:
 * [0] global-load issue, [1] LDS reads + MFMA block, [2] barrier after it, [3] split + LDS writes, [4] barrier after them, [5] all shader
 * cycles of that wave's K loop and [6] the same span in 100 MHz wall ticks (their ratio is the sustained shader clock). */
int bbbp_gemm_split_bf16_phases(unsigned long long* phases7);
/* LayerNorm absorbed by the Linear that consumes it inside bbbp_mixed_forward (bbbp_layernorm_linear_fwd for norm1 -> linear1 and norm2 -> the
 * next in_proj / fingerprint_fc; dropout + residual move into the producing GEMM's epilogue).  Default 0 (initial value BBBP_LN_ABSORB):
 * built, parity-tested and measured slower inside the B = 512 step.  Returns the previous setting. */
int bbbp_set_ln_absorb(int on);
int bbbp_set_overlap(int on);   /* two/three-stream branch overlap inside bbbp_mixed_forward/backward (default on) */
/* Data-parallel overlap: the gradient of the image-FC weight (62 % of all gradient bytes at F = 167) is final after the
 * first GEMM of the image branch's backward.  wait_bucket(stream, 0) makes `stream` wait for exactly that point of the
 * most recent bbbp_mixed_backward on the current device, so an all-reduce of that bucket can run under the remaining
 * ~2 ms of the backward pass; bucket_param gives the bucket's index in the parameter order.  Bucket 1 = every other
 * gradient except the four conv tensors: final when the fingerprint branch and all weight-gradient leaves are (the image
 * branch's last kernel is then still running); the conv tensors are final only at the end of the pass. */
int bbbp_mixed_backward_wait_bucket(void* stream, int bucket);
/* The same buckets for an OPTIMIZER step pipelined into the pass: `stream` waits until the bucket's gradient slice is final AND the
 * bucket's parameters are no longer read by the most recent bbbp_mixed_backward (bucket 0: after the image FC's input-gradient GEMM;
 * encoder layer l >= 1: when layer l - 1's bucket is final; layer 0 and bucket 1: the end of the fingerprint branch).  The four conv
 * tensors have no bucket: conv2's weight is read by conv2's data gradient, update them after the pass. */
int bbbp_mixed_backward_wait_released(void* stream, int bucket);
/* Bucket 0's release point needs one more event on the image branch's stream: recorded only while this is on (default off).  Returns the
 * previous setting. */
int bbbp_set_release_events(int on);
/* Test hook: the ReLU decisions of encoder layer `layer`'s linear1 (R:75-78; nn.TransformerEncoderLayer.linear1 + ReLU) as
 * the forward pass that filled `workspace` took them -- gate[b * dim_feedforward + j] = 1 where the (post-dropout) hidden
 * activation is > 0.  Parity tests at B = 512 hand these to the float64 oracle: a pre-activation within float32 rounding of
 * zero may legitimately fall on either side, and ONE such element changes that unit's weight-gradient row by percents. */
int bbbp_mixed_debug_ffn_gate(void* stream, const bbbp_mixed_desc* d, const void* workspace, int layer, uint8_t* gate);
/* Test hook, same purpose for the conv stages: the u8 pooling / ReLU decision the forward call saved per POOLED element of stage 1
 * (Conv2d(3,32)+ReLU+MaxPool2d: [B,32,64,64]) or stage 2 (Conv2d(32,64)+...: [B,64,32,32]): 0..3 = position of the first maximum of the
 * 2x2 window in PyTorch's scan order (0,0),(0,1),(1,0),(1,1); 4 = the maximum is <= 0 (ReLU inactive, no gradient).  A window whose
 * two largest pre-activations agree to float32 rounding may legitimately route its gradient to either, and the conv weight / bias
 * gradients are 10^7-term sums over such decisions: B = 512 parity tests hand these to the float64 oracle after checking that they
 * differ from float64's own only at such near-ties. */
int bbbp_mixed_debug_pool_mask(void* stream, const bbbp_mixed_desc* d, const void* workspace, int stage, uint8_t* mask);
int bbbp_mixed_bucket_param(const bbbp_mixed_desc* d, int bucket);
/* Buckets 2 + l (l = encoder layer): the layer's twelve tensors, final when its weight-gradient leaves are (layer L-1 first,
 * layer 0 last).  bucket_range gives first parameter index and tensor count of bucket 0 or 2 + l (consecutive in params[]
 * order, hence one slice of a flat gradient buffer); -1 otherwise. */
int bbbp_mixed_bucket_range(const bbbp_mixed_desc* d, int bucket, int* first, int* count);
/* HIP-graph replay of bbbp_mixed_forward / bbbp_mixed_backward: the second call with identical arguments is captured,
 * later ones are replayed with one hipGraphLaunch.  Opt-in (env BBBP_GRAPHS=1 or bbbp_set_graphs(1), which returns the
 * previous setting): measured slower than the eager three-stream enqueue on ROCm 7.2.  Counters since load. */
int bbbp_set_graphs(int on);
int bbbp_graph_stats(long* captures, long* replays);
int bbbp_profile_enable(int on);
int bbbp_profile_select(unsigned section_mask);   /* bit i = section i records events; 0 = all (the default) */
int bbbp_profile_num_sections(void);
const char* bbbp_profile_section_name(int i);
int bbbp_profile_collect(float* ms_sum, int* count);
/* Before collect(): section id, start and end (ms after the first recorded section's start) of up to max_entries recorded
 * section instances, in recording order; returns how many were written (or a negative error). */
int bbbp_profile_timeline(int* section, float* start_ms, float* end_ms, int max_entries);

#ifdef __cplusplus
}
#endif
#endif /* BBBP_HIP_H */
