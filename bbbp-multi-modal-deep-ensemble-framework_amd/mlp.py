"""GPU batched trainer for the scikit-learn style MLP classifiers of the reference's model-selection stage.

Reference: ``Models/model_opt_maccs.py:133`` (``MLPClassifier(max_iter=2000, learning_rate_init=0.001, batch_size=32)``)
and ``:170-181`` (GridSearchCV over hidden_layer_sizes x activation x learning_rate_init x batch_size, cv=5: 270 fits on
``[n, 100]`` PCA features).  Every fit is tiny, the grid is embarrassingly parallel: one persistent work-group trains
one model (``csrc/mlp.hip``), all fits of a grid run in one launch per chunk of epochs.

The arithmetic is scikit-learn's, in float64; the row visiting order and the initial weights come from the estimator's
own ``RandomState`` stream (``_init_coef`` draws, then one ``sklearn.utils.shuffle`` of the index vector per epoch), so a
fit here follows the scikit-learn fit with the same ``random_state`` step for step (up to summation order inside the
matrix products).  Only the training-loss stopping rule (``early_stopping=False``, the reference's setting) and the
``adam`` solver are implemented -- that is what the reference uses.
"""
from __future__ import annotations

import ctypes
from dataclasses import dataclass, field
from typing import List, Optional, Sequence

import numpy as np
import torch

from . import _lib, ops

_ACT = {"relu": 0, "tanh": 1}


@dataclass
class MLPConfig:
    """One fit: scikit-learn ``MLPClassifier`` hyper-parameters + the training rows (a CV fold) it sees."""
    hidden_layer_sizes: Sequence[int] = (100,)
    activation: str = "relu"
    learning_rate_init: float = 0.001
    batch_size: int = 200
    alpha: float = 1e-4
    max_iter: int = 200
    tol: float = 1e-4
    n_iter_no_change: int = 10
    beta_1: float = 0.9
    beta_2: float = 0.999
    epsilon: float = 1e-8
    random_state: int = 0
    train_rows: Optional[np.ndarray] = None          # row ids into X; None = all rows


@dataclass
class FittedMLP:
    config: MLPConfig
    coefs_: List[np.ndarray] = field(default_factory=list)
    intercepts_: List[np.ndarray] = field(default_factory=list)
    n_iter_: int = 0
    loss_: float = float("nan")
    best_loss_: float = float("inf")
    loss_curve_: List[float] = field(default_factory=list)
    converged_: bool = False                        # stopped by the tol rule (not by max_iter)
    _trainer: "GridMLPTrainer" = None
    _index: int = -1

    def predict_proba(self, X) -> np.ndarray:
        p1 = self._trainer._predict(self._index, X)
        return np.stack([1.0 - p1, p1], axis=1)

    def predict(self, X) -> np.ndarray:
        return (self._trainer._predict(self._index, X) > 0.5).astype(np.int64)


class GridMLPTrainer:
    """Trains many small MLP classifiers at once on one GPU.  ``X``: [n, n_features] float64, ``y``: [n] in {0, 1}."""

    def __init__(self, X, y, device="cuda"):
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("GridMLPTrainer needs a GPU (no CPU fallback)")
        X = np.ascontiguousarray(X, dtype=np.float64)
        y = np.ascontiguousarray(y, dtype=np.float64)
        if X.ndim != 2 or y.shape != (X.shape[0],):
            raise ValueError(f"X must be [n, features] and y [n]; got {X.shape} and {y.shape}")
        if not np.isin(y, (0.0, 1.0)).all():
            raise ValueError("y must be binary {0, 1} (binarise labels first, as MLPClassifier does)")
        self.n, self.n_features = X.shape
        self.X = torch.from_numpy(X).to(self.device)
        self.y = torch.from_numpy(y).to(self.device)
        self._models_dev = None
        self._keep = []

    # ---- host replica of the estimator's RandomState protocol -------------------------------------------------------
    @staticmethod
    def _init_params(cfg: MLPConfig, units, rs):
        parts = []
        for fan_in, fan_out in zip(units[:-1], units[1:]):
            bound = np.sqrt(6.0 / (fan_in + fan_out))            # relu / tanh: factor 6 (sklearn _init_coef)
            parts.append(rs.uniform(-bound, bound, (fan_in, fan_out)).ravel())
            parts.append(rs.uniform(-bound, bound, fan_out))
        return np.concatenate(parts)

    def fit(self, configs: Sequence[MLPConfig], epochs_per_launch: int = 8) -> List[FittedMLP]:
        """Round 4 host side: (i) the fits sit in the device array in order of DECREASING cost (mini-batches per epoch x parameters): the
        work-groups of a launch are handed out in array order and a work-group is a whole CU, so the few fits that do not get a CU at once
        (270 fits, 256 CUs) are the cheapest ones instead of whatever the grid's order put last; (ii) the visiting orders of chunk k + 1 --
        every estimator's own RandomState stream, 270 x epochs shuffles of ~5000 rows on the host -- are drawn WHILE the GPU trains chunk k
        and travel as one pinned copy on a second stream (they used to sit between two launches: ~20 ms per epoch of the reference grid)."""
        L = _lib.lib()
        nm = len(configs)
        if nm == 0:
            return []
        dev = self.device
        E = int(epochs_per_launch)
        host, streams = [], {}
        for i, cfg in enumerate(configs):
            if cfg.activation not in _ACT:
                raise ValueError(f"activation {cfg.activation!r}: the reference grid uses relu and tanh")
            hidden = list(cfg.hidden_layer_sizes)
            if not 1 <= len(hidden) <= 3 or max(hidden) > 256:
                raise ValueError("1 to 3 hidden layers of at most 256 units")
            rows = np.arange(self.n) if cfg.train_rows is None else np.asarray(cfg.train_rows, dtype=np.int64)
            n_train = len(rows)
            units = [self.n_features] + hidden + [1]
            # Fits that share (integer random_state, layer sizes, number of training rows) -- in the reference grid all folds, activations,
            # learning rates and batch sizes of one hidden_layer_sizes value: 90 of 270 -- consume ONE and the same RandomState stream:
            # identical initial parameters and identical shuffles.  The stream is run once per group (init: 270 -> 3 draws of up to 40 k
            # uniforms, 68 ms -> 1 ms on the bench box; shuffles: 270 -> 3 per epoch) and every member indexes its own rows with it.
            shared = isinstance(cfg.random_state, (int, np.integer))
            skey = (int(cfg.random_state), tuple(units), n_train) if shared else ("own", i)
            st = streams.get(skey)
            if st is None:
                rs = np.random.RandomState(cfg.random_state)
                st = streams[skey] = dict(rs=rs, p0=self._init_params(cfg, units, rs), idx=np.arange(n_train), orders={}, made=0)
            p0 = st["p0"]
            bs = min(int(cfg.batch_size), n_train)
            cost = -(-n_train // bs) * (len(p0) + 4000)          # mini-batches per epoch x (parameters + a per-update constant)
            host.append(dict(cfg=cfg, rows=rows, stream=st, units=units, p0=p0, bs=bs, n_train=n_train, cost=cost))
        slot_of = np.argsort([-h["cost"] for h in host], kind="stable")      # slot k of the device array holds fit slot_of[k]
        # one device buffer for every fit's visiting orders, two halves (the chunk being trained / the chunk being drawn)
        offs, tot = [], 0
        for k in range(nm):
            offs.append(tot); tot += E * host[slot_of[k]]["n_train"]
        orders_dev = torch.empty((2, tot), dtype=torch.int32, device=dev)
        orders_host = [torch.empty(tot, dtype=torch.int32).pin_memory() for _ in range(2)]
        # one device buffer per kind for ALL fits (270 fits x 7 small tensors were 1900 allocations and 270 host-to-device copies per fit() call)
        p_off, a_off, c_off, ptot, atot, ctot = [], [], [], 0, 0, 0
        for k in range(nm):
            h = host[slot_of[k]]
            p_off.append(ptot); ptot += len(h["p0"])
            a_off.append(atot); atot += h["bs"] * sum(h["units"][1:])
            c_off.append(ctot); ctot += h["cfg"].max_iter
        params_all = torch.from_numpy(np.concatenate([host[slot_of[k]]["p0"] for k in range(nm)])).to(dev)
        m_all, v_all, g_all = (torch.zeros(ptot, dtype=torch.float64, device=dev) for _ in range(3))
        act_all, delta_all = (torch.zeros(atot, dtype=torch.float64, device=dev) for _ in range(2))
        curve_all = torch.zeros(ctot, dtype=torch.float64, device=dev)
        self._buffers = (params_all, m_all, v_all, g_all, act_all, delta_all, curve_all)
        structs = (_lib.MlpModel * nm)()
        for k in range(nm):
            h = host[slot_of[k]]
            cfg, units, p0, bs, n_train = h["cfg"], h["units"], h["p0"], h["bs"], h["n_train"]
            np_, na = len(p0), bs * sum(units[1:])
            t = dict(params=params_all[p_off[k]:p_off[k] + np_], m=m_all[p_off[k]:p_off[k] + np_], v=v_all[p_off[k]:p_off[k] + np_],
                     g=g_all[p_off[k]:p_off[k] + np_], act=act_all[a_off[k]:a_off[k] + na], delta=delta_all[a_off[k]:a_off[k] + na],
                     curve=curve_all[c_off[k]:c_off[k] + cfg.max_iter])
            h["t"], h["slot"] = t, k
            s = structs[k]
            s.n_layers = len(units) - 1
            for q, u in enumerate(units):
                s.units[q] = u
            s.activation, s.batch_size, s.n_train = _ACT[cfg.activation], bs, n_train
            s.n_iter_no_change, s.max_iter = cfg.n_iter_no_change, cfg.max_iter
            s.lr_init, s.alpha, s.beta1, s.beta2, s.eps, s.tol = (cfg.learning_rate_init, cfg.alpha, cfg.beta_1, cfg.beta_2,
                                                                    cfg.epsilon, cfg.tol)
            s.params, s.adam_m, s.adam_v, s.grads = (t["params"].data_ptr(), t["m"].data_ptr(), t["v"].data_ptr(),
                                                     t["g"].data_ptr())
            s.act, s.delta, s.loss_curve = t["act"].data_ptr(), t["delta"].data_ptr(), t["curve"].data_ptr()
            s.order = orders_dev[0].data_ptr() + 4 * offs[k]
            s.t, s.best_loss, s.no_improve, s.n_iter, s.done = 0, float("inf"), 0, 0, 0
        nbytes = ctypes.sizeof(structs)
        models_dev = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        staging = torch.empty(nbytes, dtype=torch.uint8).pin_memory()
        copy_stream = torch.cuda.Stream(device=dev)

        def draw(half, live_slots, n_epochs):
            """the next n_epochs epochs' visiting orders of the live fits, from each estimator's own RandomState stream, into half `half`
            (never more than a fit has left to run)"""
            buf = orders_host[half].numpy()
            for k in live_slots:
                h = host[slot_of[k]]
                n_train = h["n_train"]
                todo = min(n_epochs, h["cfg"].max_iter - h["drawn"])
                st = h["stream"]
                for e in range(todo):
                    ae = h["drawn"] + e                       # absolute epoch of this fit = position in its group's stream
                    while st["made"] <= ae:
                        # == sample_idx = sklearn.utils.shuffle(sample_idx, random_state=rs): resample() shuffles arange(n)
                        # with the RandomState and indexes the array with it
                        perm = np.arange(n_train)
                        st["rs"].shuffle(perm)
                        st["idx"] = st["idx"][perm]
                        st["orders"][st["made"]] = st["idx"]
                        st["made"] += 1
                    buf[offs[k] + e * n_train: offs[k] + (e + 1) * n_train] = h["rows"][st["orders"][ae]]
                h["drawn"] += max(todo, 0)
            floors = {}
            for k in live_slots:                              # drop the epochs every live member of a group has passed
                h = host[slot_of[k]]
                floors[id(h["stream"])] = min(floors.get(id(h["stream"]), h["drawn"]), h["drawn"])
            for st in streams.values():
                floor = floors.get(id(st), st["made"])
                for ae in [a for a in st["orders"] if a < floor]:
                    del st["orders"][ae]
            with torch.cuda.stream(copy_stream):
                orders_dev[half].copy_(orders_host[half], non_blocking=True)

        # chunk sizes 1, 2, 4, ... E epochs: only the FIRST chunk's orders are drawn with the GPU idle (16 ms per epoch of the reference grid on
        # the host against 45 ms on the GPU: a chunk twice as long as the one being trained is still drawn in its shadow)
        live = list(range(nm))
        for h in host:
            h["drawn"] = 0
        chunk = 1
        draw(0, live, chunk)
        copy_stream.synchronize()
        half = 0
        while live:
            for k in live:
                structs[k].order = orders_dev[half].data_ptr() + 4 * offs[k]
            ctypes.memmove(staging.data_ptr(), ctypes.addressof(structs), nbytes)
            models_dev.copy_(staging, non_blocking=False)
            torch.cuda.current_stream(dev).wait_stream(copy_stream)
            _lib.check(L.bbbp_mlp_train_epochs(ops._stream(), models_dev.data_ptr(), nm, self.X.data_ptr(), self.y.data_ptr(),
                                               self.n_features, chunk), "bbbp_mlp_train_epochs")
            chunk = min(E, 2 * chunk)
            draw(half ^ 1, live, chunk)                    # host work + the copy run beside the kernel (fits that finish in this chunk: wasted, harmless)
            staging.copy_(models_dev)                      # synchronises with the kernel
            ctypes.memmove(ctypes.addressof(structs), staging.data_ptr(), nbytes)
            live = [k for k in live if not structs[k].done]
            half ^= 1
        copy_stream.synchronize()
        self._models_dev, self._structs = models_dev, structs
        self._keep = host
        self._orders = (orders_dev, orders_host)
        out = []
        params_host, curve_host = params_all.cpu().numpy(), curve_all.cpu().numpy()
        for i, h in enumerate(host):
            s, cfg, units = structs[h["slot"]], h["cfg"], h["units"]
            p = params_host[p_off[h["slot"]]:p_off[h["slot"]] + len(h["p0"])]
            coefs, inter, off = [], [], 0
            for fan_in, fan_out in zip(units[:-1], units[1:]):
                coefs.append(p[off:off + fan_in * fan_out].reshape(fan_in, fan_out).copy()); off += fan_in * fan_out
                inter.append(p[off:off + fan_out].copy()); off += fan_out
            curve = curve_host[c_off[h["slot"]]:c_off[h["slot"]] + s.n_iter].tolist()
            out.append(FittedMLP(config=cfg, coefs_=coefs, intercepts_=inter, n_iter_=int(s.n_iter), loss_=curve[-1] if curve else float("nan"),
                                 best_loss_=float(s.best_loss), loss_curve_=curve, converged_=bool(s.no_improve > cfg.n_iter_no_change),
                                 _trainer=self, _index=i))
        return out

    def _predict(self, index: int, X) -> np.ndarray:
        if self._models_dev is None:
            raise RuntimeError("predict before fit")
        Xd = torch.from_numpy(np.ascontiguousarray(X, dtype=np.float64)).to(self.device)
        if Xd.dim() != 2 or Xd.shape[1] != self.n_features:
            raise ValueError(f"X must be [n, {self.n_features}]")
        out = torch.empty(Xd.shape[0], dtype=torch.float64, device=self.device)
        max_units = max(self._keep[index]["units"][1:])
        _lib.check(_lib.lib().bbbp_mlp_predict_proba(ops._stream(), self._models_dev.data_ptr(), self._keep[index]["slot"], Xd.data_ptr(), Xd.shape[0],
                                                     self.n_features, max_units, out.data_ptr()), "bbbp_mlp_predict_proba")
        return out.cpu().numpy()


def grid_search_cv(X, y, param_grid: dict, cv: int = 5, base: Optional[MLPConfig] = None, device="cuda", random_state: int = 0):
    """The reference's ``GridSearchCV(MLPClassifier(...), param_grid, cv=5, scoring='f1')`` (model_opt_maccs.py:183-184) as
    one batched GPU run: every (parameter point, fold) pair is one model.  Folds are scikit-learn's StratifiedKFold (what
    GridSearchCV uses for classifiers).  Returns (best_params, mean_f1 per point, all fitted models)."""
    from itertools import product
    from sklearn.metrics import f1_score
    from sklearn.model_selection import StratifiedKFold
    base = base or MLPConfig()
    keys = sorted(param_grid)
    points = [dict(zip(keys, vals)) for vals in product(*(param_grid[k] for k in keys))]
    folds = list(StratifiedKFold(n_splits=cv).split(X, y))
    configs = []
    for pt in points:
        for tr, _ in folds:
            kw = {**base.__dict__, **{k: v for k, v in pt.items() if k != "solver"}, "train_rows": tr, "random_state": random_state}
            configs.append(MLPConfig(**kw))
    trainer = GridMLPTrainer(X, y, device=device)
    fitted = trainer.fit(configs)
    scores = []
    X = np.asarray(X, dtype=np.float64); y = np.asarray(y)
    for pi in range(len(points)):
        f1 = [f1_score(y[te], fitted[pi * len(folds) + fi].predict(X[te])) for fi, (_, te) in enumerate(folds)]
        scores.append(float(np.mean(f1)))
    best = int(np.argmax(scores))
    return points[best], scores, fitted
