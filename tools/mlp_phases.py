"""Phase profile of the MLP trainer's persistent kernel on BASELINE config 1 (bbbp_mlp_profile): cycles per mini-batch phase of work-group 0
(= the most expensive fit of the grid, slot 0 after the cost sort: hidden (200, 100), batch 32).  usage: python tools/mlp_phases.py [epochs]"""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from itertools import product
from sklearn.model_selection import StratifiedKFold
from bbbp_amd import _lib
from bbbp_amd.mlp import GridMLPTrainer, MLPConfig

epochs = int(sys.argv[1]) if len(sys.argv) > 1 else 4
dev = torch.device("cuda:0")
rs = np.random.RandomState(0)
n, f = 6245, 100
X = rs.randn(n, f)
y = ((X @ rs.randn(f) + 2.0 * rs.randn(n)) > 0).astype(np.float64)
grid = {"hidden_layer_sizes": [(100,), (100, 50), (200, 100)], "activation": ["relu", "tanh"],
        "learning_rate_init": [0.001, 0.01, 0.1], "batch_size": [32, 64, 128]}
keys = sorted(grid)
points = [dict(zip(keys, vals)) for vals in product(*(grid[k] for k in keys))]
folds = list(StratifiedKFold(5).split(X, y))
cfgs = [MLPConfig(max_iter=epochs, tol=0.0, n_iter_no_change=10 ** 9, train_rows=tr, random_state=0, **pt) for pt in points for tr, _ in folds]
only = os.environ.get("MLP_ONLY")          # e.g. "32": keep the fits of one batch size
if only:
    cfgs = [c for c in cfgs if c.batch_size == int(only)]
trainer = GridMLPTrainer(X, y, device=dev)
trainer.fit(cfgs[:8], epochs_per_launch=1)        # warm-up
L = _lib.lib()
_lib.check(L.bbbp_mlp_profile(1, None), "bbbp_mlp_profile")
torch.cuda.synchronize(); t0 = time.perf_counter()
trainer.fit(cfgs, epochs_per_launch=epochs)
torch.cuda.synchronize(); dt = time.perf_counter() - t0
out = (ctypes.c_ulonglong * 16)()
_lib.check(L.bbbp_mlp_profile(0, ctypes.cast(out, ctypes.c_void_p)), "bbbp_mlp_profile")
c = list(out)
mb = max(c[15], 1)
names = {0: "forward layer 0", 1: "forward layer 1", 2: "forward layer 2", 3: "loss", 4: "delta below layer 0 (never)", 5: "weight gradient + Adam layer 0",
         6: "delta below layer 1", 7: "weight gradient + Adam layer 1", 8: "delta below layer 2", 9: "weight gradient + Adam layer 2",
         10: "mini-batch tail (sum W^2, barrier)", 11: "epoch tail"}
tot = sum(c[:12])
print(f"{len(cfgs)} fits x {epochs} epochs with the instrumented kernel: {dt * 1e3 / epochs:.1f} ms per epoch; work-group 0: {mb} mini-batches, {tot / mb:.0f} cycles each")
for k in range(12):
    if c[k]:
        print(f"  {names[k]:38s} {c[k] / mb:9.0f} cycles / mini-batch  {100.0 * c[k] / tot:5.1f} %")
ng = len(cfgs)
w = (ctypes.c_ulonglong * (3 * ng))()
_lib.check(L.bbbp_mlp_profile_groups(ctypes.cast(w, ctypes.c_void_p), ng), "bbbp_mlp_profile_groups")
w = np.array(list(w), dtype=np.float64).reshape(ng, 3)
t_first = w[:, 0].min()
start, end, cyc = (w[:, 0] - t_first) / 100.0, (w[:, 1] - t_first) / 100.0, w[:, 2]          # us
ghz = cyc / np.maximum(end - start, 1e-9) / 1e3
order_end = np.argsort(-end)
print(f"work-groups: launch span {end.max() / 1e3:.1f} ms; started late (> 1 ms): {(start > 1000).sum()}; clock while running: median {np.median(ghz):.2f} GHz, min {ghz.min():.2f}, max {ghz.max():.2f}")
print("  last to finish (slot: start ms -> end ms, GHz):", ", ".join(f"{g}: {start[g] / 1e3:.1f} -> {end[g] / 1e3:.1f}, {ghz[g]:.2f}" for g in order_end[:8]))
print("  slot 0:", f"{start[0] / 1e3:.1f} -> {end[0] / 1e3:.1f} ms, {ghz[0]:.2f} GHz;  quartiles of end (ms):", np.round(np.percentile(end, [25, 50, 75, 100]) / 1e3, 1))
_lib.check(L.bbbp_mlp_profile(0, None), "bbbp_mlp_profile")
torch.cuda.synchronize(); t0 = time.perf_counter()
trainer.fit(cfgs, epochs_per_launch=epochs)
torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(f"plain kernel: {dt * 1e3 / epochs:.1f} ms per epoch")
