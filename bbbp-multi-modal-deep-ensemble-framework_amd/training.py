"""The reference's per-fold training / evaluation loop for the NN member of the ensemble, on device-resident data.

Mirrors Models/multi_input_data_regression_opt_transformer_cnn_20250113.py:165-241: AdamW(lr 1e-4, wd 1e-5), MSELoss, 50
epochs of batch 32 with a shuffled train loader, a validation pass after every epoch, test predictions at the end.
``faithful_mode=True`` reproduces the published script's quirk: ``model.train()`` is called ONCE before the
epoch loop and the per-epoch validation leaves the model in ``eval()``, so epochs 2..N train with dropout off and BatchNorm
on running statistics.  ``faithful_mode=False`` re-enters train mode every epoch (what the authors presumably meant).

Differences from the reference loop, none of which change results: the dataset lives on the GPU (no per-item
``torch.tensor`` copies, no DataLoader workers), the per-step ``loss.item()`` sync is dropped (losses are accumulated on
the device and read once per epoch), and the optimizer is the fused single-launch AdamW.
"""
from __future__ import annotations

from typing import Dict, List, Optional

import numpy as np
import torch

from .models import MSELoss

from .optim import AdamW


def r2_score(y_true, y_pred) -> float:
    y_true = np.asarray(y_true, dtype=np.float64); y_pred = np.asarray(y_pred, dtype=np.float64)
    ss_res = float(((y_true - y_pred) ** 2).sum()); ss_tot = float(((y_true - y_true.mean()) ** 2).sum())
    return 1.0 - ss_res / ss_tot if ss_tot > 0 else 0.0


def mean_squared_error(y_true, y_pred) -> float:
    return float(((np.asarray(y_true, dtype=np.float64) - np.asarray(y_pred, dtype=np.float64)) ** 2).mean())


class GraphedTrainStep:
    """One training step -- forward, loss, backward, optimizer -- captured ONCE into a HIP graph and replayed (round 4).

    For the models that are compositions of this package's ops at the Python level (``variants.WideDeepMixedInputModel``: ~600 launches
    per step, each costing ~20 us of interpreter + autograd time: 12.5-16.5 ms per step eagerly, 6.3 replayed).  The flagship
    ``MixedInputModel`` does not need it: its whole pass is one C call.  Each call is exactly one step of the reference loop's body
    (Models/multi_input_data_regression_opt_transformer_cnn_20250113.py:186-193: ``zero_grad``, forward, ``criterion``, ``backward``,
    ``optimizer.step``):

    * a graph belongs to a (batch shapes, ``model.training``) pair -- the published loop trains in eval mode from epoch 2 on, and its last
      batch of an epoch is ragged: each pair gets its own capture; the first ``eager_steps`` calls of a pair run eagerly (they are real
      steps, and they let every kernel set its attributes and every lazily allocated buffer exist before the capture), the next call
      captures, copies the batch into the graph's static input tensors and replays, as does every later one;
    * dropout: the recorded per-site seeds are frozen, so the streams are keyed from a step counter in device memory that is bumped
      before every replay (``bbbp_set_seed_base``): new masks every step, the same masks in a step's forward and backward;
    * the optimizer must be ``AdamW(..., capturable=True)``: its step count / learning rate reach the kernel through device memory
      (``advance()`` before every replay), so schedulers that rewrite ``group["lr"]`` keep working.

    Returns the loss as a 0-d device tensor (replays of one graph return the same tensor: read or accumulate it before the next call)."""

    class _Entry:
        __slots__ = ("calls", "graph", "static", "loss")

        def __init__(self):
            self.calls, self.graph, self.static, self.loss = 0, None, None, None

    def __init__(self, model, optimizer, loss_fn=None, eager_steps: int = 2):
        if not getattr(optimizer, "_capturable", False):
            raise ValueError("GraphedTrainStep needs optim.AdamW(..., capturable=True)")
        self.model, self.optimizer = model, optimizer
        self.loss_fn = loss_fn if loss_fn is not None else torch.nn.MSELoss()
        self.eager_steps = int(eager_steps)
        self.calls = 0
        self.entries = {}
        self._seed = None

    @property
    def graph(self):
        """The most recently captured graph (None before the first capture)."""
        got = [e.graph for e in self.entries.values() if e.graph is not None]
        return got[-1] if got else None

    def _body(self, inputs, target):
        loss = self.loss_fn(self.model(*inputs).squeeze(), target)
        loss.backward()
        self.optimizer.step()
        return loss

    def _capture(self, entry, inputs, target):
        from . import ops
        entry.static = [t.clone() for t in (*inputs, target)]
        if self._seed is None:
            self._seed = torch.zeros(1, dtype=torch.int64, device=target.device)
        self.optimizer.zero_grad(set_to_none=True)          # the captured backward then WRITES the gradients (no accumulate kernels)
        self.optimizer.prepare_capture()
        graph = torch.cuda.CUDAGraph()
        prev = ops.set_seed_base(self._seed.data_ptr())
        try:
            with torch.cuda.graph(graph):
                loss = self._body(entry.static[:-1], entry.static[-1])
        finally:
            ops.set_seed_base(prev)
        self.optimizer.zero_grad(set_to_none=True)          # the graph keeps its own gradient buffers; eager steps allocate theirs
        entry.graph, entry.loss = graph, loss.detach()

    def __call__(self, *batch):
        *inputs, target = batch
        self.calls += 1
        key = (bool(self.model.training),) + tuple(tuple(t.shape) for t in batch)
        entry = self.entries.get(key)
        if entry is None:
            entry = self.entries[key] = GraphedTrainStep._Entry()
        entry.calls += 1
        if entry.graph is None:
            if entry.calls <= self.eager_steps:
                self.optimizer.zero_grad(set_to_none=True)
                return self._body(inputs, target).detach()
            self._capture(entry, inputs, target)
        for s, t in zip(entry.static, batch):
            if t.data_ptr() != s.data_ptr():
                s.copy_(t, non_blocking=True)
        self._seed.add_(1)
        self.optimizer.advance()
        entry.graph.replay()
        return entry.loss


@torch.no_grad()
def predict(model, fingerprints: torch.Tensor, images: torch.Tensor, batch_size: int = 32) -> torch.Tensor:
    """Eval-mode predictions in loader order (reference :229-237).  NOTE: the encoder attends across each mini-batch, so the
    batch composition is part of the function -- use the reference's batch size to reproduce its numbers."""
    was_training = model.training
    model.eval()
    outs = []
    for i in range(0, fingerprints.shape[0], batch_size):
        o = model(fingerprints[i:i + batch_size], images[i:i + batch_size])
        outs.append(o.reshape(-1))          # .squeeze() of a single-row batch gives a 0-d tensor in the reference
    model.train(was_training)
    return torch.cat(outs)


def train_fold(model, train, test=None, *, epochs: int = 50, batch_size: int = 32, lr: float = 1e-4, weight_decay: float = 1e-5,
               faithful_mode: bool = True, shuffle: bool = True, generator: Optional[torch.Generator] = None,
               optimizer=None, batch_orders: Optional[List[np.ndarray]] = None, scheduler=None,
               early_stopping_patience: Optional[int] = None, graph_steps: bool = False) -> Dict[str, list]:
    """Train ``model`` on ``train = (fingerprints[N,F], images[N,49152], labels[N])`` (device tensors).  Returns the per-epoch
    mean training / validation losses like the reference's ``train_losses`` / ``val_losses`` lists.  ``batch_orders``
    (one permutation per epoch) overrides the shuffling for reproducible comparisons.

    ``scheduler``: the canonical script anneals the learning rate with ``CosineAnnealingWarmRestarts(optimizer, T_0=10, T_mult=2)``
    stepped once per EPOCH after the epoch's batches (Models/multi_input_data_regression_opt_transformer_cnn.py:161,177; the
    published ...20250113.py has no scheduler).  Pass ``"cosine_warm_restarts"`` for exactly that, or a callable
    ``optimizer -> torch.optim.lr_scheduler.LRScheduler``; the fused AdamW reads ``group["lr"]`` at every step.  The learning
    rate used in each epoch is recorded in ``hist["lr"]``.

    ``early_stopping_patience``: the dense-MLP scripts stop on the TRAINING loss (Descriptors/multi_input_data_nn.py:114-143,
    ``patience = 10``): after each epoch, ``avg_loss < best_loss`` resets a counter, anything else increments it, and the loop breaks
    once the counter EXCEEDS the patience (i.e. after patience + 1 epochs without a new best).  ``hist["stopped_epoch"]`` is the
    0-based epoch the loop broke at, or None."""
    fp, img, y = train
    if not (fp.is_cuda and img.is_cuda and y.is_cuda):
        raise RuntimeError("train_fold expects device-resident tensors")
    y = y.to(torch.float32)
    if optimizer is not None:
        opt = optimizer
    else:
        opt = AdamW(model.parameters(), lr=lr, weight_decay=weight_decay, capturable=graph_steps)      # (optim.AdamW(defer=...) is opt-in: measured slower on one GPU)
    crit = MSELoss()                      # nn.MSELoss semantics, fused value + gradient kernel
    # graph_steps (round 4): every step replays a HIP graph captured per (batch shape, train / eval mode) -- for the variants that are per-op
    # compositions (GraphedTrainStep); pointless for MixedInputModel, whose pass is one C call
    stepper = GraphedTrainStep(model, opt, crit) if graph_steps else None
    N = fp.shape[0]
    if scheduler == "cosine_warm_restarts":
        scheduler = torch.optim.lr_scheduler.CosineAnnealingWarmRestarts(opt, T_0=10, T_mult=2)
    elif callable(scheduler) and not hasattr(scheduler, "step"):
        scheduler = scheduler(opt)
    hist = {"train_loss": [], "val_loss": [], "lr": [], "stopped_epoch": None}
    best_loss, patience_counter = float("inf"), 0
    model.train()                                   # reference :179 -- once, outside the epoch loop
    for epoch in range(epochs):
        if not faithful_mode:
            model.train()
        if batch_orders is not None:
            order = torch.as_tensor(batch_orders[epoch], device=fp.device, dtype=torch.long)
        elif shuffle:
            order = torch.randperm(N, generator=generator).to(fp.device)
        else:
            order = torch.arange(N, device=fp.device)
        total = torch.zeros((), device=fp.device)
        nb = 0
        for i in range(0, N, batch_size):
            idx = order[i:i + batch_size]
            if stepper is not None:
                total += stepper(fp[idx], img[idx], y[idx])
                nb += 1
                continue
            opt.zero_grad(set_to_none=True)
            pred = model(fp[idx], img[idx]).squeeze()
            loss = crit(pred, y[idx])
            loss.backward()
            opt.step()
            total += loss.detach()
            nb += 1
        hist["train_loss"].append(float(total) / nb)
        hist["lr"].append(float(opt.param_groups[0]["lr"]))
        if scheduler is not None:
            scheduler.step()                        # canonical script :177 -- per epoch, after the batches
        if test is not None:
            tfp, timg, ty = test
            model.eval()                            # reference :195 -- and it stays in eval mode when faithful
            with torch.no_grad():
                vl, vb = 0.0, 0
                for i in range(0, tfp.shape[0], batch_size):
                    p = model(tfp[i:i + batch_size], timg[i:i + batch_size]).squeeze()
                    vl += float(crit(p, ty[i:i + batch_size].to(torch.float32))); vb += 1
            hist["val_loss"].append(vl / vb)
        if early_stopping_patience is not None:     # multi_input_data_nn.py:134-141
            if hist["train_loss"][-1] < best_loss:
                best_loss, patience_counter = hist["train_loss"][-1], 0
            else:
                patience_counter += 1
            if patience_counter > early_stopping_patience:
                hist["stopped_epoch"] = epoch
                break
    if hasattr(opt, "synchronize"):
        opt.synchronize()                           # a deferred slice of the last step: the caller may read the parameters any way it likes
    return hist



def kfold_indices(n: int, n_splits: int = 10, seed: int = 42):
    """``KFold(n_splits, shuffle=True, random_state=42).split`` (...20250113.py:146; scikit-learn is the reference's own
    third-party dependency, so its splitter is used as is)."""
    from sklearn.model_selection import KFold
    return list(KFold(n_splits=n_splits, shuffle=True, random_state=seed).split(np.zeros((n, 1))))


def cross_validate_oof(fingerprints, images, labels, *, model_factory, n_splits: int = 10, epochs: int = 50, batch_size: int = 32,
                       faithful_mode: bool = True, scheduler=None, rf_params: Optional[dict] = None, extra_columns: Optional[dict] = None,
                       init_seed: Optional[int] = None, device=None, folds=None, batch_orders=None) -> Dict[str, np.ndarray]:
    """The published script's fold loop (Models/multi_input_data_regression_opt_transformer_cnn_20250113.py:147-266): for each of
    the ``KFold(10, shuffle=True, random_state=42)`` splits train a fresh network on the training rows (``train_fold``), write
    its eval-mode predictions of the held-out rows into ``nn[test_idx]`` (:229-241), fit the random forest on
    ``hstack([fingerprint, image])`` of the training rows (:262-266; scikit-learn on the host, third-party CPU code exactly
    as in the reference) and write its held-out predictions -- walked on the GPU (``trees.ForestGPU``) -- into ``rf[test_idx]``.
    XGBoost / CatBoost are not installed here: their out-of-fold columns enter precomputed through ``extra_columns``
    (name -> [N]).  Returns the out-of-fold matrix the stack is fitted on: ``{"nn", "rf", <extra...>, "actuals", "X"}`` with
    ``X = [nn, rf, *extra]`` column-stacked in the reference's order.

    ``model_factory()`` builds the (CPU) model of a fold; ``init_seed`` (+ fold index) seeds torch before each call so runs are
    reproducible (the reference sets no seeds).  ``rf_params`` defaults to the reference's (300 trees, depth 30, seed 42);
    ``rf_params=False`` skips the forest.  ``folds`` overrides the splits, ``batch_orders[fold]`` the shuffling (tests)."""
    from .trees import ForestGPU
    fp_all = torch.as_tensor(fingerprints, dtype=torch.float32)
    img_all = torch.as_tensor(images, dtype=torch.float32)
    y_all = np.asarray(labels, dtype=np.float64)
    N = fp_all.shape[0]
    dev = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
    fp_d, img_d = fp_all.to(dev), img_all.to(dev)
    y_d = torch.as_tensor(y_all, dtype=torch.float32, device=dev)          # MixedDataset casts labels to float32 (:44)
    folds = kfold_indices(N, n_splits) if folds is None else folds
    out = {"nn": np.zeros(N), "actuals": np.zeros(N), "train_loss": [], "val_loss": []}
    if rf_params is not False:
        out["rf"] = np.zeros(N)
        rfp = dict(n_estimators=300, max_depth=30, random_state=42)
        rfp.update(rf_params or {})
    feats = None
    for k, (train_idx, test_idx) in enumerate(folds):
        tr = torch.as_tensor(train_idx, device=dev); te = torch.as_tensor(test_idx, device=dev)
        if init_seed is not None:
            torch.manual_seed(init_seed + k)
        model = model_factory().to(dev)
        hist = train_fold(model, (fp_d[tr], img_d[tr], y_d[tr]), (fp_d[te], img_d[te], y_d[te]), epochs=epochs, batch_size=batch_size,
                          faithful_mode=faithful_mode, scheduler=scheduler,
                          batch_orders=None if batch_orders is None else batch_orders[k])
        out["train_loss"].append(hist["train_loss"]); out["val_loss"].append(hist["val_loss"])
        out["nn"][test_idx] = predict(model, fp_d[te], img_d[te], batch_size=batch_size).double().cpu().numpy()
        out["actuals"][test_idx] = y_all[test_idx]
        if rf_params is not False:
            from sklearn.ensemble import RandomForestRegressor
            if feats is None:
                feats = np.hstack([fp_all.numpy(), img_all.numpy()])
            rf = RandomForestRegressor(**rfp).fit(feats[train_idx], y_all[train_idx])
            out["rf"][test_idx] = ForestGPU.from_sklearn(rf, device=dev).predict(feats[test_idx])
    cols = [out["nn"]] + ([out["rf"]] if "rf" in out else [])
    for name, col in (extra_columns or {}).items():
        out[name] = np.asarray(col, dtype=np.float64)
        cols.append(out[name])
    out["X"] = np.stack(cols, axis=1)
    return out
