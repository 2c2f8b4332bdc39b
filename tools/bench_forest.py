"""Screening-scale random-forest inference: ForestGPU vs scikit-learn's predict on the host (synthetic data)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from sklearn.ensemble import RandomForestRegressor
from bbbp_amd.trees import ForestGPU

rs = np.random.RandomState(0)
n_train, f = 1058, 2000                       # B3DB-sized training set; 2000 features keep the CPU fit short
X = rs.randn(n_train, f).astype(np.float32); X[:, :167] = (X[:, :167] > 0.6)
y = X[:, :50] @ rs.randn(50) + 0.3 * rs.randn(n_train)
t0 = time.perf_counter()
rf = RandomForestRegressor(n_estimators=300, max_depth=30, random_state=42, n_jobs=-1).fit(X, y)
print(f"fit (scikit-learn, host): {time.perf_counter() - t0:.1f} s, {sum(e.tree_.node_count for e in rf.estimators_)} nodes")
g = ForestGPU.from_sklearn(rf)
n = 1 << 20
Xd = torch.randn(n, f, device="cuda")
g.predict(Xd[:1000])
torch.cuda.synchronize(); t0 = time.perf_counter()
p = g.predict(Xd)
dt = time.perf_counter() - t0
print(f"GPU: {n} rows x 300 trees in {dt * 1e3:.1f} ms -> {n / dt / 1e6:.2f} M molecules/s (includes the copy of the predictions to the host)")
ns = 20000
Xh = Xd[:ns].cpu().numpy()
t0 = time.perf_counter(); q = rf.predict(Xh); dc = time.perf_counter() - t0
print(f"scikit-learn (host, n_jobs=-1): {ns} rows in {dc * 1e3:.0f} ms -> {ns / dc / 1e6:.3f} M molecules/s; max |diff| {np.abs(q - p[:ns]).max():.2e}")
