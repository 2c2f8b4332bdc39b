"""Stacked-ensemble predict surface of the reference (float64, host side: k <= 4 columns, N ~ 1e3 rows).

The reference stacks out-of-fold predictions ``[nn, rf, xgb(, cat)]`` with a linear meta-learner --
``Ridge(alpha=1.0)`` (Models/multi_input_data_regression_opt_transformer_cnn_opt.py:173-176), ``LinearRegression``
(..._morgan.py, same lines; final estimator of the StackingRegressor at ...20250113.py:394-403) -- or a fixed weighted
sum 0.4/0.3/0.3 (Models/multi_input_data_regression_opt_transformer_cnn.py:216-218), and later calls
``loaded_stacked_model.predict([[nn_out, rf_pred, xgb_pred]])`` (..._opt.py:202-203).  Tree base learners are third-party
CPU code and enter as precomputed columns.  The meta-learner is a 3-4 parameter least-squares problem, so it stays on the
host in float64 like sklearn's; ``predict_device`` applies it to device-resident NN predictions during screening.
"""
from __future__ import annotations

import numpy as np


def weighted_ensemble(nn_pred, rf_pred, xgb_pred, weights=(0.4, 0.3, 0.3)):
    """0.4*nn + 0.3*rf + 0.3*xgb (..._transformer_cnn.py:216-218)."""
    a, b, c = (np.asarray(v, dtype=np.float64) for v in (nn_pred, rf_pred, xgb_pred))
    return weights[0] * a + weights[1] * b + weights[2] * c


class StackedEnsemble:
    """Linear meta-learner with sklearn's attribute names: ``alpha=0`` is LinearRegression, ``alpha=1`` the
    reference's Ridge.  The intercept is not penalised (sklearn semantics)."""

    def __init__(self, alpha: float = 0.0):
        if alpha < 0:
            raise ValueError("alpha must be >= 0")
        self.alpha = float(alpha)
        self.coef_ = None
        self.intercept_ = None

    @classmethod
    def from_coefficients(cls, coef, intercept, alpha: float = 0.0):
        """Rebuild a fitted meta-learner, e.g. from the values stored in the reference's stacked_model*.pkl."""
        self = cls(alpha)
        self.coef_ = np.asarray(coef, dtype=np.float64).copy()
        self.intercept_ = float(intercept)
        return self

    def fit(self, X, y):
        X = np.asarray(X, dtype=np.float64)
        y = np.asarray(y, dtype=np.float64).reshape(-1)
        if X.ndim != 2 or X.shape[0] != y.shape[0]:
            raise ValueError(f"X {X.shape} and y {y.shape} do not match")
        xm, ym = X.mean(axis=0), y.mean()
        Xc, yc = X - xm, y - ym
        if self.alpha == 0.0:
            self.coef_ = np.linalg.lstsq(Xc, yc, rcond=None)[0]
        else:
            self.coef_ = np.linalg.solve(Xc.T @ Xc + self.alpha * np.eye(X.shape[1]), Xc.T @ yc)
        self.intercept_ = float(ym - xm @ self.coef_)
        return self

    def predict(self, X):
        if self.coef_ is None:
            raise RuntimeError("StackedEnsemble is not fitted")
        X = np.asarray(X, dtype=np.float64)
        if X.ndim != 2 or X.shape[1] != self.coef_.shape[0]:
            raise ValueError(f"X has shape {X.shape}, expected [N, {self.coef_.shape[0]}]")
        return X @ self.coef_ + self.intercept_

    def predict_device(self, nn_pred, *other_columns):
        """Same arithmetic on torch tensors (float64 on the tensors' device): screening keeps NN outputs on the GPU."""
        import torch
        cols = [nn_pred.reshape(-1).to(torch.float64)] + [torch.as_tensor(c, device=nn_pred.device).reshape(-1).to(torch.float64)
                                                          for c in other_columns]
        if len(cols) != self.coef_.shape[0]:
            raise ValueError(f"{len(cols)} columns given, meta-learner has {self.coef_.shape[0]} coefficients")
        out = torch.full_like(cols[0], self.intercept_)
        for c, w in zip(cols, self.coef_):
            out = out + float(w) * c
        return out


class StackingRegressor:
    """``sklearn.ensemble.StackingRegressor(estimators, final_estimator=LinearRegression())`` as the published script builds
    and uses it (Models/multi_input_data_regression_opt_transformer_cnn_20250113.py:394-403) -- NOT ``X c + b`` on the input
    columns: ``fit(X, y)`` (X = the [N,4] out-of-fold matrix ``[nn, rf, xgb, cat]``) fits every base learner on ALL of X,
    builds the meta-features from 5-fold ``cross_val_predict`` of clones (``KFold(5)`` without shuffling, scikit-learn's default
    for regressors) and fits the final linear model on those [N, n_estimators] columns; ``predict(X)`` runs the full-data base
    learners and the final model on their outputs.  Base learners are anything with scikit-learn's ``fit`` / ``predict``
    (third-party CPU code, as in the reference); fitted random forests / extra-trees predict through ``trees.ForestGPU`` when
    ``device`` is given.  The final estimator is this module's ``StackedEnsemble`` (OLS by default)."""

    def __init__(self, estimators, final_estimator=None, cv: int = 5, device=None):
        self.estimators = list(estimators)
        self.final_estimator = final_estimator if final_estimator is not None else StackedEnsemble(0.0)
        self.cv, self.device = int(cv), device
        self.estimators_ = None
        self.final_estimator_ = None

    def _predict_one(self, est, X):
        if self.device is not None and hasattr(est, "estimators_") and hasattr(est.estimators_[0], "tree_"):
            from .trees import ForestGPU
            return ForestGPU.from_sklearn(est, device=self.device).predict(X)
        return np.asarray(est.predict(X), dtype=np.float64)

    def fit(self, X, y):
        from sklearn.base import clone
        X = np.asarray(X, dtype=np.float64)
        y = np.asarray(y, dtype=np.float64).reshape(-1)
        n = X.shape[0]
        if X.ndim != 2 or n != y.shape[0]:
            raise ValueError(f"X {X.shape} and y {y.shape} do not match")
        if n < self.cv:
            raise ValueError(f"cv={self.cv} folds need at least as many rows, got {n}")
        self.estimators_ = [(name, clone(est).fit(X, y)) for name, est in self.estimators]
        # KFold(cv) without shuffling: the first n % cv folds have one row more
        sizes = np.full(self.cv, n // self.cv); sizes[: n % self.cv] += 1
        bounds = np.concatenate([[0], np.cumsum(sizes)])
        meta = np.zeros((n, len(self.estimators)))
        for f in range(self.cv):
            test = np.arange(bounds[f], bounds[f + 1])
            train = np.concatenate([np.arange(0, bounds[f]), np.arange(bounds[f + 1], n)])
            for j, (_, est) in enumerate(self.estimators):
                meta[test, j] = self._predict_one(clone(est).fit(X[train], y[train]), X[test])
        self.final_estimator_ = StackedEnsemble(getattr(self.final_estimator, "alpha", 0.0)).fit(meta, y)
        return self

    def transform(self, X):
        if self.estimators_ is None:
            raise RuntimeError("StackingRegressor is not fitted")
        X = np.asarray(X, dtype=np.float64)
        return np.stack([self._predict_one(est, X) for _, est in self.estimators_], axis=1)

    def predict(self, X):
        return self.final_estimator_.predict(self.transform(X))


def screen(model, forest, stack, fingerprints, images, extra_columns=(), batch_size: int = 4096, boosters=()):
    """Stacked prediction over a library (BASELINE config 5; ``Descriptors/virtualscreening.py`` + ...20250113.py:394-403):
    per batch, the multi-modal network in eval mode, the random forest on ``hstack([fingerprint, image])`` (``trees.ForestGPU``),
    fitted gradient-boosted learners (``boosters``: ``boosters.XGBTrees`` / ``boosters.CatBoostTrees``, the xgb and cat columns of
    ...20250108.py:186-207, on the same hstack) and any precomputed columns go through the linear meta-learner, all on the GPU.  Column order:
    nn, rf, boosters..., extra_columns...
    Returns float64 predictions on the device.  Like the reference, the network attends ACROSS the batch, so its
    column depends on ``batch_size`` and on the order of the library."""
    import torch
    n = fingerprints.shape[0]
    if images.shape[0] != n or any(len(c) != n for c in extra_columns):
        raise ValueError("fingerprints, images and extra columns must have the same number of rows")
    was_training = model.training
    model.eval()
    out = []
    try:
        with torch.no_grad():
            for i in range(0, n, batch_size):
                fp, im = fingerprints[i:i + batch_size], images[i:i + batch_size]
                nn_col = model(fp, im).reshape(-1)
                cols = []
                # the tree learners all read hstack([fingerprint, image]) (805 MB per 4096 molecules): built once per batch
                feats = torch.cat([fp, im], dim=1) if (forest is not None or boosters) else None
                if forest is not None:
                    cols.append(forest.predict_device(feats))
                cols += [bst.predict_device(feats).double() for bst in boosters]
                del feats
                cols += [torch.as_tensor(c[i:i + batch_size], device=nn_col.device) for c in extra_columns]
                out.append(stack.predict_device(nn_col, *cols))
    finally:
        model.train(was_training)
    return torch.cat(out) if out else torch.empty(0, dtype=torch.float64, device=fingerprints.device)
