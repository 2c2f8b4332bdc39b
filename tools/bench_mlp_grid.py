"""BASELINE config 1 on the GPU: the reference's MLP grid (Models/model_opt_maccs.py:170-181: 54 parameter points x 5 folds
= 270 fits of MLPClassifier on [6245, 100] float64 PCA-like features) as ONE batched run, with scikit-learn timed on a
sample of the same fits on the host for scale.  Synthetic data."""
import sys, os, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from bbbp_amd.mlp import MLPConfig, grid_search_cv

rs = np.random.RandomState(0)
n, f = 6245, 100
X = rs.randn(n, f)
y = ((X @ rs.randn(f) + 2.0 * rs.randn(n)) > 0).astype(np.float64)
grid = {"hidden_layer_sizes": [(100,), (100, 50), (200, 100)], "activation": ["relu", "tanh"], "learning_rate_init": [0.001, 0.01, 0.1],
        "batch_size": [32, 64, 128]}
max_iter = int(os.environ.get("MLP_MAX_ITER", "60"))
t0 = time.perf_counter()
best, scores, fitted = grid_search_cv(X, y, grid, cv=5, base=MLPConfig(max_iter=max_iter), random_state=0)
t_gpu = time.perf_counter() - t0
epochs = sum(m.n_iter_ for m in fitted)
print(f"GPU: {len(fitted)} fits, {epochs} epochs in total, {t_gpu:.1f} s  ->  {epochs * n * 0.8 / t_gpu / 1e6:.2f} M sample-visits/s; best {best}, mean f1 {max(scores):.4f}")
from sklearn.neural_network import MLPClassifier
from sklearn.model_selection import StratifiedKFold
folds = list(StratifiedKFold(5).split(X, y))
sample = [0, 100, 200, 269]
t_cpu, ep_cpu = 0.0, 0
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    for i in sample:
        c = fitted[i].config
        t1 = time.perf_counter()
        m = MLPClassifier(hidden_layer_sizes=tuple(c.hidden_layer_sizes), activation=c.activation, learning_rate_init=c.learning_rate_init,
                          batch_size=c.batch_size, max_iter=max_iter, random_state=0).fit(X[c.train_rows], y[c.train_rows])
        t_cpu += time.perf_counter() - t1; ep_cpu += m.n_iter_
print(f"scikit-learn on the host, {len(sample)} of the fits: {ep_cpu} epochs in {t_cpu:.1f} s -> {ep_cpu * n * 0.8 / t_cpu / 1e6:.2f} M sample-visits/s per process")
