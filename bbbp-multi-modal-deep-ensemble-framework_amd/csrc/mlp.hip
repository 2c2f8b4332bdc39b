// Batched trainer for the small scikit-learn style MLP classifiers of the reference's model-selection stage
// (Models/model_opt_maccs.py:133,170-181: MLPClassifier(max_iter=2000) under a 54-point GridSearchCV x 5 folds on
// [n,100] PCA features).  The arithmetic is scikit-learn's (1.3.2 pinned by the
// reference's pickles; _multilayer_perceptron.py / _stochastic_optimizers.py), restated:
//   forward   a[l+1] = act(a[l] W[l] + b[l]), hidden act relu | tanh, output logistic
//   loss      binary log-loss with probabilities clipped to [eps, 1-eps]  +  0.5 alpha sum ||W||^2 / n_batch
//   backward  delta[last] = a[last] - y;  dW[l] = (a[l]^T delta[l] + alpha W[l]) / n_batch;  db[l] = sum_rows delta[l] / n_batch
//             delta[l-1] = delta[l] W[l]^T, times act'(a[l])  (relu: zero where a == 0; tanh: 1 - a^2)
//   Adam      t += 1; m = b1 m + (1-b1) g; v = b2 v + (1-b2) g^2; lr_t = lr sqrt(1-b2^t)/(1-b1^t); p -= lr_t m/(sqrt(v)+eps)
//   epoch end loss_ = sum(batch_loss * batch_rows) / n; stop when the loss failed to improve by tol for more than
//             n_iter_no_change consecutive epochs (training-loss criterion, early_stopping=False)
// in float64 like scikit-learn on float64 input.  The grid is embarrassingly parallel and every fit is tiny (a
// 100x100 weight matrix), so ONE persistent work-group trains one model from start to finish and the whole grid
// (270 fits) runs as one launch per chunk of epochs; the host only supplies the visiting order of the rows (the
// estimator's own RandomState stream, so fits are reproducible against scikit-learn) and reads the stop flags.
// Every output element is an ordered sum over k by one thread => deterministic.
#include "common.h"
#include "bbbp_hip.h"
#include <stdlib.h>

namespace {

constexpr int NT = 1024;      // 16 waves per model: the loops are latency-bound chains of f64 FMAs over L2-resident operands

__device__ __forceinline__ double block_sum(double v, double* red) {
    // fixed-order tree over the threads of the work-group
    red[threadIdx.x] = v;
    __syncthreads();
    for (int s = NT / 2; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    const double r = red[0];
    __syncthreads();
    return r;
}

__device__ __forceinline__ double xlogy(double x, double y) { return x == 0.0 ? 0.0 : x * log(y); }

__global__ __launch_bounds__(NT) void mlp_train_scalar_kernel(bbbp_mlp_model* models, const double* X, const double* y, int n_features,
                                                      int epochs) {
    __shared__ double red[NT];
    bbbp_mlp_model& M = models[blockIdx.x];
    if (M.done) return;
    const int t = threadIdx.x;
    const int L = M.n_layers;
    // parameter layout: [W0 | b0 | W1 | b1 | ...]
    int woff[4], boff[4], total = 0;
    for (int l = 0; l < L; ++l) { woff[l] = total; total += M.units[l] * M.units[l + 1]; boff[l] = total; total += M.units[l + 1]; }
    // activation / delta layout: layer l (1..L) at aoff[l], [batch][units[l]]
    int aoff[5]; aoff[0] = 0; aoff[1] = 0;
    for (int l = 1; l < L; ++l) aoff[l + 1] = aoff[l] + M.batch_size * M.units[l];
    const int n = M.n_train, bs = M.batch_size;
    double* P = M.params; double* G = M.grads; double* A = M.act; double* D = M.delta;
    const double eps_clip = 2.220446049250313e-16;

    for (int e = 0; e < epochs && !M.done; ++e) {
        const int* order = M.order + (long)e * n;
        double accumulated = 0.0;                         // kept by every thread identically (all take the same reductions)
        for (int b0 = 0; b0 < n; b0 += bs) {
            const int nb = min(bs, n - b0);
            // ---- forward ----
            for (int l = 0; l < L; ++l) {
                const int fin = M.units[l], fout = M.units[l + 1];
                const double* W = P + woff[l]; const double* bias = P + boff[l];
                double* out = A + aoff[l + 1];
                for (int idx = t; idx < nb * fout; idx += NT) {
                    const int r = idx / fout, j = idx % fout;
                    const double* in = l == 0 ? X + (long)order[b0 + r] * n_features : A + aoff[l] + (long)r * fin;
                    double s = 0.0;
                    for (int k = 0; k < fin; ++k) s += in[k] * W[(long)k * fout + j];
                    s += bias[j];
                    if (l + 1 < L) s = M.activation == 0 ? (s > 0.0 ? s : 0.0) : tanh(s);
                    else s = 1.0 / (1.0 + exp(-s));
                    out[idx] = s;
                }
                __syncthreads();
            }
            // ---- loss ----
            const double* prob = A + aoff[L];
            double part = 0.0;
            for (int r = t; r < nb; r += NT) {
                const double yt = y[order[b0 + r]];
                const double pc = fmin(fmax(prob[r], eps_clip), 1.0 - eps_clip);
                part += xlogy(yt, pc) + xlogy(1.0 - yt, 1.0 - pc);
            }
            double loss = -block_sum(part, red) / nb;
            part = 0.0;
            for (int l = 0; l < L; ++l) {
                const double* W = P + woff[l];
                const int cnt = M.units[l] * M.units[l + 1];
                for (int i = t; i < cnt; i += NT) part += W[i] * W[i];
            }
            loss += 0.5 * M.alpha * block_sum(part, red) / nb;
            accumulated += loss * nb;
            // ---- backward ----
            for (int r = t; r < nb; r += NT) D[aoff[L] + r] = prob[r] - y[order[b0 + r]];
            __syncthreads();
            for (int l = L - 1; l >= 0; --l) {
                const int fin = M.units[l], fout = M.units[l + 1];
                const double* W = P + woff[l];
                const double* dl = D + aoff[l + 1];
                // dW[k][j] and db[j]
                for (int idx = t; idx < fin * fout; idx += NT) {
                    const int k = idx / fout, j = idx % fout;
                    double s = 0.0;
                    if (l == 0) { for (int r = 0; r < nb; ++r) s += X[(long)order[b0 + r] * n_features + k] * dl[(long)r * fout + j]; }
                    else { const double* al = A + aoff[l]; for (int r = 0; r < nb; ++r) s += al[(long)r * fin + k] * dl[(long)r * fout + j]; }
                    G[woff[l] + idx] = (s + M.alpha * W[idx]) / nb;
                }
                for (int j = t; j < fout; j += NT) {
                    double s = 0.0;
                    for (int r = 0; r < nb; ++r) s += dl[(long)r * fout + j];
                    G[boff[l] + j] = s / nb;
                }
                // delta of the layer below
                if (l > 0) {
                    const double* al = A + aoff[l];
                    double* dn = D + aoff[l];
                    for (int idx = t; idx < nb * fin; idx += NT) {
                        const int r = idx / fin, k = idx % fin;
                        double s = 0.0;
                        for (int j = 0; j < fout; ++j) s += dl[(long)r * fout + j] * W[(long)k * fout + j];
                        const double a = al[idx];
                        s = M.activation == 0 ? (a == 0.0 ? 0.0 : s) : s * (1.0 - a * a);
                        dn[idx] = s;
                    }
                }
                __syncthreads();
            }
            // ---- Adam ----
            const long step = M.t + 1;
            const double lr_t = M.lr_init * sqrt(1.0 - pow(M.beta2, (double)step)) / (1.0 - pow(M.beta1, (double)step));
            for (int i = t; i < total; i += NT) {
                const double g = G[i];
                const double m = M.beta1 * M.adam_m[i] + (1.0 - M.beta1) * g;
                const double v = M.beta2 * M.adam_v[i] + (1.0 - M.beta2) * (g * g);
                M.adam_m[i] = m; M.adam_v[i] = v;
                P[i] += -lr_t * m / (sqrt(v) + M.eps);
            }
            __syncthreads();
            if (t == 0) M.t = step;
            __syncthreads();
        }
        // ---- epoch end: scikit-learn's training-loss stopping rule ----
        if (t == 0) {
            const double loss_epoch = accumulated / n;
            M.loss_curve[M.n_iter] = loss_epoch;
            M.n_iter += 1;
            if (loss_epoch > M.best_loss - M.tol) M.no_improve += 1; else M.no_improve = 0;
            if (loss_epoch < M.best_loss) M.best_loss = loss_epoch;
            if (M.no_improve > M.n_iter_no_change || M.n_iter >= M.max_iter) M.done = 1;
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------------------------------------
// Round 4: the same trainer on the float64 MATRIX pipe (v_mfma_f64_16x16x4_f64), one persistent 512-thread work-group per fit.
// The scalar kernel above spends ~1.4 ms per 32-row mini-batch of the 100 -> 200 -> 100 -> 1 net (213 ms per epoch of the grid): every
// output element is one thread's chain of K dependent FMAs on operands fetched one by one from L2.  A mini-batch is five small GEMMs
// (3.4 k MFMAs for that net: ~25 us of matrix time on one CU) plus an Adam update whose float64 sqrt / divide cost about as much again.
//   * every product runs as 16 x 16 x 4 MFMAs with operands fetched STRAIGHT from global memory (all of a fit's state -- parameters,
//     moments, activations: < 1 MB -- is L2-resident) into the instruction's register layout, the way gemm.hip's small-product kernel
//     does: a k-contiguous operand costs two 16-byte loads per lane and 16 k (lane (i, kq) takes k = 16 c + 4 kq .. + 3 and feeds element
//     jj to MFMA jj: a permutation of K applied to both operands), a k-major operand four 8-byte loads whose 16 lanes cover 128-byte rows;
//   * forward / delta products: a wave owns a 16-column tile and up to four 16-row tiles (the weight fragment is loaded once per chunk);
//     weight gradients: 32 x 32 wave tiles, K = the mini-batch rows;
//   * the weight-gradient epilogue IS the Adam step: g = (acc + alpha W) / n_b, the moments and the parameter are read, updated and
//     written where the accumulator sits -- no gradient buffer round trip -- and sum W^2 of the L2 term falls out of the same pass;
//   * K order is fixed (chunk by chunk, MFMA by MFMA), partial sums are combined in wave / lane order: results are deterministic.
// scikit-learn's own BLAS sums in yet another order: tests/test_gpu_mlp.py holds both kernels to its tolerances.
// ------------------------------------------------------------------------------------------------------------------------------
typedef double f64x4 __attribute__((ext_vector_type(4)));
typedef double f64x2u __attribute__((ext_vector_type(2), aligned(8)));

// An operand as the MFMA sees it: value(idx, k) for idx = the tile's row (A) / column (B) index and k the reduction index.
//   KCONT: element (idx, k) at base[row(idx) * ld + k]            (k contiguous; `gather` maps idx to a row when not null)
//   KMAJ : element (idx, k) at base[row(k) * ld + idx]            (idx contiguous; `gather` maps k to a row when not null)
struct Opnd { const double* base; long ld; int extent; const int* gather; };

template <bool KMAJ>
__device__ __forceinline__ void opnd_fetch(const Opnd& o, int idx, int k0, int K, double (&v)[4]) {
    // idx: this lane's row / column (may lie beyond extent: clamped, never stored); k0 = 16 c + 4 kq
    const int ic = idx < o.extent ? idx : o.extent - 1;
    if (KMAJ) {
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            const int k = k0 + jj, kc = k < K ? k : K - 1;
            const long row = o.gather ? o.gather[kc] : kc;
            const double x = o.base[row * o.ld + ic];
            v[jj] = k < K ? x : 0.0;
        }
    } else {
        const long row = o.gather ? o.gather[ic] : ic;
        const double* p = o.base + row * o.ld;
        if (k0 + 3 < K) {
            const f64x2u a = *reinterpret_cast<const f64x2u*>(p + k0), b = *reinterpret_cast<const f64x2u*>(p + k0 + 2);
            v[0] = a[0]; v[1] = a[1]; v[2] = b[0]; v[3] = b[1];
        } else {
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) { const int k = k0 + jj; const double x = p[k < K ? k : K - 1]; v[jj] = k < K ? x : 0.0; }
        }
    }
}

// acc[um][un] += A(m0 + 16 um .., :) B(:, n0 + 16 un ..) over all K; TM x TN tiles of 16 x 16 per wave.  The fragments of chunk c + 1 are
// in flight while chunk c's MFMAs run (two register sets, addressed with compile-time indices: the loop is unrolled by two).
template <int TM, int TN, bool AKMAJ, bool BKMAJ>
__device__ __forceinline__ void mfma_product(const Opnd& A, const Opnd& B, int m0, int n0, int K, f64x4 (&acc)[TM][TN]) {
    const int lane = threadIdx.x & 63, q = lane & 15, kq = lane >> 4;
    const int nch = (K + 15) >> 4;
    double a0[TM][4], b0[TN][4], a1[TM][4], b1[TN][4];
    auto fetch = [&](int c, double (&a)[TM][4], double (&b)[TN][4]) __attribute__((always_inline)) {
#pragma unroll
        for (int um = 0; um < TM; ++um) opnd_fetch<AKMAJ>(A, m0 + 16 * um + q, 16 * c + 4 * kq, K, a[um]);
#pragma unroll
        for (int un = 0; un < TN; ++un) opnd_fetch<BKMAJ>(B, n0 + 16 * un + q, 16 * c + 4 * kq, K, b[un]);
    };
    auto mma = [&](const double (&a)[TM][4], const double (&b)[TN][4]) __attribute__((always_inline)) {
#pragma unroll
        for (int jj = 0; jj < 4; ++jj)
#pragma unroll
            for (int um = 0; um < TM; ++um)
#pragma unroll
                for (int un = 0; un < TN; ++un)
                    acc[um][un] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[um][jj], b[un][jj], acc[um][un], 0, 0, 0);
    };
    fetch(0, a0, b0);
    for (int c = 0; c < nch; c += 2) {
        if (c + 1 < nch) fetch(c + 1, a1, b1);
        mma(a0, b0);
        if (c + 1 < nch) {
            if (c + 2 < nch) fetch(c + 2, a0, b0);
            mma(a1, b1);
        }
    }
}

constexpr int MT = 512, MW = MT / 64;           // threads / waves of a fit's work-group: two waves per SIMD, up to 256 registers each (1024 or 768 threads spill: the float64 transcendentals and the fragment sets need ~265)

// fixed-order sum over the work-group: lanes by shuffles, then the 16 wave partials in wave order (every thread returns the same value)
__device__ __forceinline__ double block_sum16(double v, double* red) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    __syncthreads();                                    // the previous use of `red` has been read
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    double s = 0.0;
#pragma unroll
    for (int w = 0; w < MW; ++w) s += red[w];
    return s;
}

// PROF: work-group 0's thread 0 adds the shader-clock cycles of every phase of a mini-batch to g_mlp_prof (bbbp_mlp_profile reads them): the
// only way to see inside one persistent launch.  Slots: 0-2 forward layer l; 3 loss; 4 + 2 l delta below layer l, 5 + 2 l weight gradient + Adam of
// layer l; 10 mini-batch tail; 11 epoch tail; 15 mini-batches counted.
__device__ unsigned long long g_mlp_prof[16];
__device__ unsigned long long g_mlp_wall[3 * 1024];          // per work-group (PROF): 100 MHz wall ticks at start and end, shader cycles in between

template <bool PROF>
__global__ __launch_bounds__(MT) void mlp_train_mfma_kernel(bbbp_mlp_model* models, const double* X, const double* y, int n_features, int epochs) {
    __shared__ double red[MW];
    unsigned long long prof_last = PROF ? clock64() : 0ull;
    const unsigned long long prof_c0 = prof_last;
    if (PROF && threadIdx.x == 0 && blockIdx.x < 1024) g_mlp_wall[3 * blockIdx.x] = wall_clock64();
    auto mark = [&](int slot) __attribute__((always_inline)) {
        if (PROF && blockIdx.x == 0 && threadIdx.x == 0) {
            const unsigned long long now = clock64();
            g_mlp_prof[slot] += now - prof_last;
            prof_last = now;
        }
    };
    __shared__ double lr_shared;
    bbbp_mlp_model& M = models[blockIdx.x];
    if (M.done) return;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, q = lane & 15, kq = lane >> 4;
    const int L = M.n_layers;
    int woff[4], boff[4], total = 0;
    for (int l = 0; l < L; ++l) { woff[l] = total; total += M.units[l] * M.units[l + 1]; boff[l] = total; total += M.units[l + 1]; }
    int aoff[5]; aoff[0] = 0; aoff[1] = 0;
    for (int l = 1; l < L; ++l) aoff[l + 1] = aoff[l] + M.batch_size * M.units[l];
    const int n = M.n_train, bs = M.batch_size, act_kind = M.activation;
    double* P = M.params; double* A = M.act; double* D = M.delta; double* Am = M.adam_m; double* Av = M.adam_v;
    const double alpha = M.alpha, beta1 = M.beta1, beta2 = M.beta2, adam_eps = M.eps;
    const double eps_clip = 2.220446049250313e-16;

    for (int e = 0; e < epochs && !M.done; ++e) {
        const int* order = M.order + (long)e * n;
        double accumulated = 0.0;                         // identical in every thread
        for (int b0 = 0; b0 < n; b0 += bs) {
            const int nb = min(bs, n - b0);
            const int mtiles = (nb + 15) >> 4;
            // gradient / n_b: for a power of two (every full batch of the reference grid: 32 / 64 / 128) the product with 1 / n_b IS the
            // quotient, bit for bit, and saves one float64 division (~12 instructions) per parameter and update
            const bool nb_pow2 = (nb & (nb - 1)) == 0;
            const double inv_nb = 1.0 / nb;
            // Adam's step size for this update: one lane, while the forward pass runs (read after several barriers)
            if (t == MT - 1) {
                const double step = (double)(M.t + 1);
                lr_shared = M.lr_init * sqrt(1.0 - pow(beta2, step)) / (1.0 - pow(beta1, step));
            }
            // ---- forward: a[l+1] = act(a[l] W[l] + b[l]); A operand rows k-contiguous (layer 0: gathered rows of X), B = W k-major ----
            for (int l = 0; l < L; ++l) {
                const int fin = M.units[l], fout = M.units[l + 1];
                const Opnd Aop = l == 0 ? Opnd{X, (long)n_features, nb, order + b0} : Opnd{A + aoff[l], (long)fin, nb, nullptr};
                const Opnd Bop = {P + woff[l], (long)fout, fout, nullptr};
                const double* bias = P + boff[l];
                double* out = A + aoff[l + 1];
                const int ntiles = (fout + 15) >> 4, mgroups = (mtiles + 3) >> 2;
                for (int u = wave; u < ntiles * mgroups; u += MW) {
                    const int nt = u % ntiles, mg = u / ntiles;
                    f64x4 acc[4][1];
#pragma unroll
                    for (int um = 0; um < 4; ++um) acc[um][0] = f64x4{0.0, 0.0, 0.0, 0.0};
                    if (mtiles <= 2) {                       // batches of at most 32 rows: two row tiles (no MFMAs on clamped rows)
                        f64x4 (&acc2)[2][1] = reinterpret_cast<f64x4 (&)[2][1]>(acc);
                        mfma_product<2, 1, false, true>(Aop, Bop, mg * 64, nt * 16, fin, acc2);
                    } else mfma_product<4, 1, false, true>(Aop, Bop, mg * 64, nt * 16, fin, acc);
                    const int j = nt * 16 + q;
                    const double bj = j < fout ? bias[j] : 0.0;
#pragma unroll
                    for (int um = 0; um < 4; ++um)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int row = mg * 64 + um * 16 + kq + 4 * r;
                            if (row < nb && j < fout) {
                                double s = acc[um][0][r] + bj;
                                if (l + 1 < L) s = act_kind == 0 ? (s > 0.0 ? s : 0.0) : tanh(s);
                                else s = 1.0 / (1.0 + exp(-s));
                                out[(long)row * fout + j] = s;
                            }
                        }
                }
                __syncthreads();
                mark(l);
            }
            // ---- loss: log-loss of the batch; the L2 term's sum W^2 comes out of the weight-gradient pass below (same weights) ----
            const double* prob = A + aoff[L];
            double part = 0.0;
            for (int r = t; r < nb; r += MT) {
                const double yt = y[order[b0 + r]];
                const double pc = fmin(fmax(prob[r], eps_clip), 1.0 - eps_clip);
                part += xlogy(yt, pc) + xlogy(1.0 - yt, 1.0 - pc);
                D[aoff[L] + r] = prob[r] - yt;            // delta of the output layer
            }
            double loss = -block_sum16(part, red) / nb;   // (its barriers also publish the output delta)
            const double lr_t = lr_shared;
            double wsq = 0.0;
            mark(3);
            // ---- backward: per layer, top down: delta of the layer below (old weights), then weight gradient + Adam ----
            for (int l = L - 1; l >= 0; --l) {
                const int fin = M.units[l], fout = M.units[l + 1];
                double* W = P + woff[l];
                const double* dl = D + aoff[l + 1];
                if (l > 0) {
                    // delta[l][r][k] = (sum_j delta[l+1][r][j] W[k][j]) act'(a[l][r][k]): both operands k(= j)-contiguous
                    const Opnd Aop = {dl, (long)fout, nb, nullptr};
                    const Opnd Bop = {W, (long)fout, fin, nullptr};
                    const double* al = A + aoff[l];
                    double* dn = D + aoff[l];
                    const int ntiles = (fin + 15) >> 4, mgroups = (mtiles + 3) >> 2;
                    for (int u = wave; u < ntiles * mgroups; u += MW) {
                        const int nt = u % ntiles, mg = u / ntiles;
                        f64x4 acc[4][1];
#pragma unroll
                        for (int um = 0; um < 4; ++um) acc[um][0] = f64x4{0.0, 0.0, 0.0, 0.0};
                        if (mtiles <= 2) {
                            f64x4 (&acc2)[2][1] = reinterpret_cast<f64x4 (&)[2][1]>(acc);
                            mfma_product<2, 1, false, false>(Aop, Bop, mg * 64, nt * 16, fout, acc2);
                        } else mfma_product<4, 1, false, false>(Aop, Bop, mg * 64, nt * 16, fout, acc);
                        const int k = nt * 16 + q;
#pragma unroll
                        for (int um = 0; um < 4; ++um)
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                const int row = mg * 64 + um * 16 + kq + 4 * r;
                                if (row < nb && k < fin) {
                                    const double a = al[(long)row * fin + k];
                                    const double s = acc[um][0][r];
                                    dn[(long)row * fin + k] = act_kind == 0 ? (a == 0.0 ? 0.0 : s) : s * (1.0 - a * a);
                                }
                            }
                    }
                    __syncthreads();      // every wave is done READING W[l] (and the delta below is complete) before anybody updates W[l]
                    mark(4 + 2 * l);
                }
                // dW[k][j] = (sum_r a[l][r][k] delta[l+1][r][j] + alpha W[k][j]) / nb, then Adam in place; 32 x 32 wave tiles, K = batch rows
                {
                    const Opnd Aop = l == 0 ? Opnd{X, (long)n_features, fin, order + b0} : Opnd{A + aoff[l], (long)fin, fin, nullptr};
                    const Opnd Bop = {dl, (long)fout, fout, nullptr};
                    double* mW = Am + woff[l]; double* vW = Av + woff[l];
                    const int nt2 = (fout + 31) >> 5, mt2 = (fin + 31) >> 5;
                    for (int u = wave; u < nt2 * mt2; u += MW) {
                        const int nt = u % nt2, mt = u / nt2;
                        f64x4 acc[2][2];
#pragma unroll
                        for (int um = 0; um < 2; ++um)
#pragma unroll
                            for (int un = 0; un < 2; ++un) acc[um][un] = f64x4{0.0, 0.0, 0.0, 0.0};
                        // Adam where the accumulators sit.  The state of this lane's 16 elements (parameter + two moments) comes in two batches
                        // of 24 loads issued ahead of their stores -- element by element, every load waited a full round trip to the memory-side
                        // cache behind the previous element's stores (bbbp_mlp_profile: 64 % of a mini-batch was this epilogue).
                        double wv[2][2][4], mv[2][2][4], vv[2][2][4];
                        auto load_state = [&](int um) __attribute__((always_inline)) {
#pragma unroll
                            for (int un = 0; un < 2; ++un)
#pragma unroll
                                for (int r = 0; r < 4; ++r) {
                                    const int k = mt * 32 + um * 16 + kq + 4 * r, j = nt * 32 + un * 16 + q;
                                    const long i = (long)min(k, fin - 1) * fout + min(j, fout - 1);
                                    wv[um][un][r] = W[i]; mv[um][un][r] = mW[i]; vv[um][un][r] = vW[i];
                                }
                        };
                        auto update = [&](int um) __attribute__((always_inline)) {
#pragma unroll
                            for (int un = 0; un < 2; ++un)
#pragma unroll
                                for (int r = 0; r < 4; ++r) {
                                    const int k = mt * 32 + um * 16 + kq + 4 * r, j = nt * 32 + un * 16 + q;
                                    if (k < fin && j < fout) {
                                        const long i = (long)k * fout + j;
                                        const double w = wv[um][un][r];
                                        wsq += w * w;
                                        const double gs = acc[um][un][r] + alpha * w;
                                        const double g = nb_pow2 ? gs * inv_nb : gs / nb;
                                        const double m = beta1 * mv[um][un][r] + (1.0 - beta1) * g;
                                        const double v = beta2 * vv[um][un][r] + (1.0 - beta2) * (g * g);
                                        mW[i] = m; vW[i] = v;
                                        W[i] = w + -lr_t * m / (sqrt(v) + adam_eps);
                                    }
                                }
                        };
                        // (issuing batch 0 before the product, or both batches at once, measured the same 525 k cycles per mini-batch with 38 - 60
                        // spilled registers instead of 10: profiles/r04_mlp_phases.txt)
                        mfma_product<2, 2, true, true>(Aop, Bop, mt * 32, nt * 32, nb, acc);
                        load_state(0);
                        update(0);
                        load_state(1);
                        update(1);
                    }
                    // bias gradient db[j] = sum_r delta[l+1][r][j] / nb (rows in order) + Adam
                    double* bp = P + boff[l]; double* mb = Am + boff[l]; double* vb = Av + boff[l];
                    for (int j = t; j < fout; j += MT) {
                        double s = 0.0;
                        for (int r = 0; r < nb; ++r) s += dl[(long)r * fout + j];
                        const double g = nb_pow2 ? s * inv_nb : s / nb;
                        const double m = beta1 * mb[j] + (1.0 - beta1) * g;
                        const double v = beta2 * vb[j] + (1.0 - beta2) * (g * g);
                        mb[j] = m; vb[j] = v;
                        bp[j] += -lr_t * m / (sqrt(v) + adam_eps);
                    }
                }
                mark(5 + 2 * l);
            }
            loss += 0.5 * alpha * block_sum16(wsq, red) / nb;      // sum over all layers of ||W||^2 (the weights the forward pass used)
            accumulated += loss * nb;
            if (t == 0) M.t = M.t + 1;
            __syncthreads();              // the update is complete before the next mini-batch reads the parameters
            mark(10);
            if (PROF && blockIdx.x == 0 && t == 0) g_mlp_prof[15] += 1;
        }
        if (t == 0) {
            const double loss_epoch = accumulated / n;
            M.loss_curve[M.n_iter] = loss_epoch;
            M.n_iter += 1;
            if (loss_epoch > M.best_loss - M.tol) M.no_improve += 1; else M.no_improve = 0;
            if (loss_epoch < M.best_loss) M.best_loss = loss_epoch;
            if (M.no_improve > M.n_iter_no_change || M.n_iter >= M.max_iter) M.done = 1;
        }
        __syncthreads();
        mark(11);
    }
    if (PROF && threadIdx.x == 0 && blockIdx.x < 1024) {
        g_mlp_wall[3 * blockIdx.x + 1] = wall_clock64();
        g_mlp_wall[3 * blockIdx.x + 2] = clock64() - prof_c0;
    }
}

// probabilities of the positive class for rows [0, n) of X under model `m` (one work-group per 8 rows)
__global__ __launch_bounds__(NT) void mlp_predict_kernel(const bbbp_mlp_model* models, int model, const double* X, int n, int n_features,
                                                        double* out) {
    __shared__ double buf[2][8 * 256];           // activations of 8 rows, up to 256 units (reference grid: <= 200)
    const bbbp_mlp_model& M = models[model];
    const int t = threadIdx.x, r0 = blockIdx.x * 8, nb = min(8, n - r0), L = M.n_layers;
    int off = 0;
    for (int l = 0; l < L; ++l) {
        const int fin = M.units[l], fout = M.units[l + 1];
        const double* W = M.params + off; const double* bias = W + fin * fout;
        off += fin * fout + fout;
        double* o = buf[l & 1];
        for (int idx = t; idx < nb * fout; idx += NT) {
            const int r = idx / fout, j = idx % fout;
            const double* in = l == 0 ? X + (long)(r0 + r) * n_features : buf[(l + 1) & 1] + r * fin;
            double s = 0.0;
            for (int k = 0; k < fin; ++k) s += in[k] * W[(long)k * fout + j];
            s += bias[j];
            if (l + 1 < L) s = M.activation == 0 ? (s > 0.0 ? s : 0.0) : tanh(s);
            else s = 1.0 / (1.0 + exp(-s));
            o[idx] = s;
        }
        __syncthreads();
    }
    for (int r = t; r < nb; r += NT) out[r0 + r] = buf[(L - 1) & 1][r];
}

bool g_mlp_profile_on = false;

}  // namespace

extern "C" int bbbp_mlp_train_epochs(void* stream, bbbp_mlp_model* models_dev, int n_models, const double* X, const double* y,
                                     int n_features, int epochs) {
    BBBP_CHECK_ARG(n_models >= 0 && epochs >= 0 && n_features >= 1, "mlp_train: bad sizes");
    if (n_models == 0 || epochs == 0) return BBBP_OK;
    BBBP_CHECK_ARG(models_dev && X && y, "mlp_train: null pointer");
    // round 4: the float64-MFMA trainer; BBBP_MLP_SCALAR=1 selects the scalar kernel of rounds 1-3 (the cross-check in tests/test_gpu_mlp.py)
    const char* e = getenv("BBBP_MLP_SCALAR");
    if (e && atoi(e) != 0)
        hipLaunchKernelGGL(mlp_train_scalar_kernel, dim3(n_models), dim3(NT), 0, static_cast<hipStream_t>(stream), models_dev, X, y, n_features, epochs);
    else if (g_mlp_profile_on)
        hipLaunchKernelGGL(mlp_train_mfma_kernel<true>, dim3(n_models), dim3(MT), 0, static_cast<hipStream_t>(stream), models_dev, X, y, n_features, epochs);
    else
        hipLaunchKernelGGL(mlp_train_mfma_kernel<false>, dim3(n_models), dim3(MT), 0, static_cast<hipStream_t>(stream), models_dev, X, y, n_features, epochs);
    BBBP_CHECK_LAUNCH();
    return BBBP_OK;
}

// Phase profile of the trainer (see g_mlp_prof): on != 0 selects the instrumented kernel for later bbbp_mlp_train_epochs calls and clears the
// counters; cycles16 (host, nullable) receives the 16 counters accumulated so far (synchronises the device).
extern "C" int bbbp_mlp_profile_groups(unsigned long long* out, int n_groups) {
    BBBP_CHECK_ARG(out && n_groups >= 0 && n_groups <= 1024, "mlp_profile_groups: up to 1024 work-groups");
    if (n_groups) BBBP_CHECK_HIP(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_mlp_wall), sizeof(unsigned long long) * 3 * n_groups));
    return BBBP_OK;
}

extern "C" int bbbp_mlp_profile(int on, unsigned long long* cycles16) {
    if (cycles16) BBBP_CHECK_HIP(hipMemcpyFromSymbol(cycles16, HIP_SYMBOL(g_mlp_prof), sizeof(unsigned long long) * 16));
    if (on) {
        static const unsigned long long zeros[16] = {};
        BBBP_CHECK_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_mlp_prof), zeros, sizeof(zeros)));
    }
    g_mlp_profile_on = on != 0;
    return BBBP_OK;
}

extern "C" int bbbp_mlp_predict_proba(void* stream, const bbbp_mlp_model* models_dev, int model, const double* X, int n,
                                      int n_features, int max_units, double* out) {
    BBBP_CHECK_ARG(n >= 0 && n_features >= 1 && model >= 0, "mlp_predict: bad sizes");
    BBBP_CHECK_ARG(max_units <= 256, "mlp_predict: hidden layers wider than 256 units are not supported (got %d)", max_units);
    if (n == 0) return BBBP_OK;
    BBBP_CHECK_ARG(models_dev && X && out, "mlp_predict: null pointer");
    hipLaunchKernelGGL(mlp_predict_kernel, dim3(cdiv(n, 8)), dim3(NT), 0, static_cast<hipStream_t>(stream), models_dev, model, X, n,
                       n_features, out);
    BBBP_CHECK_LAUNCH();
    return BBBP_OK;
}
