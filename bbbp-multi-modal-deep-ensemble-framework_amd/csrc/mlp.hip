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

__global__ __launch_bounds__(NT) void mlp_train_kernel(bbbp_mlp_model* models, const double* X, const double* y, int n_features,
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

}  // namespace

extern "C" int bbbp_mlp_train_epochs(void* stream, bbbp_mlp_model* models_dev, int n_models, const double* X, const double* y,
                                     int n_features, int epochs) {
    BBBP_CHECK_ARG(n_models >= 0 && epochs >= 0 && n_features >= 1, "mlp_train: bad sizes");
    if (n_models == 0 || epochs == 0) return BBBP_OK;
    BBBP_CHECK_ARG(models_dev && X && y, "mlp_train: null pointer");
    hipLaunchKernelGGL(mlp_train_kernel, dim3(n_models), dim3(NT), 0, static_cast<hipStream_t>(stream), models_dev, X, y, n_features,
                       epochs);
    BBBP_CHECK_LAUNCH();
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
