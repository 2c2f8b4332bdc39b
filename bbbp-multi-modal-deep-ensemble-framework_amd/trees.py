"""GPU inference for the random-forest base learner of the reference's stacked ensemble.

Reference: ``Models/multi_input_data_regression_opt_transformer_cnn_20250113.py:262-266`` (``RandomForestRegressor(
n_estimators=300, max_depth=30, random_state=42)`` fitted on ``hstack([fingerprints, images])``) and ``:394-403`` (the same
learner inside ``StackingRegressor``).  Fitting stays with scikit-learn (CPU, a few thousand molecules); prediction over
screening-scale libraries (ZINC) is what needs the GPU: ``ForestGPU.from_sklearn(rf).predict(X)`` reproduces
``rf.predict(X)`` (float32 features, float64 thresholds/values, mean over trees in float64).  The XGBoost learner predicts
through ``boosters.XGBTrees`` (parity unpinned: package absent); CatBoost stays a precomputed input column of
``ensemble.StackedEnsemble``.
"""
from __future__ import annotations

import numpy as np
import torch

from . import _lib, ops


class ForestGPU:
    def __init__(self, left, right, feature, threshold, value, root, n_features, device="cuda"):
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("ForestGPU needs a GPU (no CPU fallback)")
        to = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a, dtype=dt)).to(self.device)
        self.left, self.right, self.feature = to(left, np.int32), to(right, np.int32), to(feature, np.int32)
        self.threshold, self.value, self.root = to(threshold, np.float64), to(value, np.float64), to(root, np.int32)
        self.n_trees, self.n_features = len(root) - 1, int(n_features)

    @staticmethod
    def flatten_sklearn(forest):
        """Concatenate the trees of a fitted single-output forest: (left, right, feature, threshold, value, root, n_features).
        Child indices are rebased to the concatenated arrays, -1 marks a leaf, root[t] is the first node of tree t."""
        if getattr(forest, "n_outputs_", 1) != 1:
            raise ValueError("single-output regressors only")
        left, right, feature, threshold, value, root = [], [], [], [], [], [0]
        for est in forest.estimators_:
            t, off = est.tree_, root[-1]
            cl, cr = t.children_left.astype(np.int64), t.children_right.astype(np.int64)
            left.append(np.where(cl >= 0, cl + off, -1)); right.append(np.where(cr >= 0, cr + off, -1))
            feature.append(np.where(t.feature >= 0, t.feature, 0)); threshold.append(t.threshold)
            value.append(t.value.reshape(t.node_count))
            root.append(off + t.node_count)
        if root[-1] >= 2 ** 31:
            raise ValueError("forest too large for 32-bit node indices")
        return (np.concatenate(left), np.concatenate(right), np.concatenate(feature), np.concatenate(threshold),
                np.concatenate(value), np.asarray(root), int(forest.n_features_in_))

    @classmethod
    def from_sklearn(cls, forest, device="cuda") -> "ForestGPU":
        """From a fitted ``RandomForestRegressor`` / ``ExtraTreesRegressor`` (single output)."""
        return cls(*cls.flatten_sklearn(forest), device=device)

    def predict_device(self, X) -> torch.Tensor:
        """``X``: [n, n_features] (numpy or a CUDA float32 tensor).  Returns float64 predictions as a tensor on the GPU."""
        if isinstance(X, torch.Tensor):
            Xd = X.to(self.device, torch.float32).contiguous()
        else:
            Xd = torch.from_numpy(np.ascontiguousarray(X, dtype=np.float32)).to(self.device)       # sklearn casts X to float32 too
        if Xd.dim() != 2 or Xd.shape[1] != self.n_features:
            raise ValueError(f"X must be [n, {self.n_features}]")
        n = Xd.shape[0]
        L = _lib.lib()
        groups = L.bbbp_forest_groups(self.n_trees)
        partial = torch.empty(max(groups * n, 1), dtype=torch.float64, device=self.device)
        out = torch.empty(n, dtype=torch.float64, device=self.device)
        _lib.check(L.bbbp_forest_predict(ops._stream(), Xd.data_ptr(), n, self.n_features, self.left.data_ptr(), self.right.data_ptr(),
                                         self.feature.data_ptr(), self.threshold.data_ptr(), self.value.data_ptr(), self.root.data_ptr(),
                                         self.n_trees, partial.data_ptr(), out.data_ptr()), "bbbp_forest_predict")
        return out

    def predict(self, X) -> np.ndarray:
        """As ``predict_device``, predictions copied to the host (what ``rf.predict`` returns)."""
        return self.predict_device(X).cpu().numpy()
