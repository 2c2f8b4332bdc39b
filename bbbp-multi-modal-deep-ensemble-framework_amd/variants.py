"""The reference's MLP-only variants of ``MixedInputModel`` (same class name in every script; here one class per
script, same constructor signature ``(fingerprint_size, image_feature_size)`` and ``state_dict`` keys), running on
the HIP GEMM / BatchNorm / dropout / fusion ops through per-op autograd nodes.

* ``PCAFusionModel``  -- Models/multi_input_data_regression_opt_transformer_cnn_opt.py:72-105 (also _morgan.py):
  PCA-reduced fingerprint and image each through Linear+ReLU, attention fusion, 256->128->64->1.  This is the
  architecture of the shipped ``best_nn_model*.pth``.
* ``DenseMLPModel``   -- Models/multi_input_data_regression_opt.py:41-85: raw fingerprint F->512->256->128 and raw image
  49152->1024->256->128 with ReLU -> BatchNorm1d -> Dropout(0.2), concat, BatchNorm head.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from .functional import run_sequential
from .models import MultiHeadAttentionFusion, flatten_parameters


def _need_cuda(x):
    if not x.is_cuda:
        raise RuntimeError("this model runs on MI355X only (HIP kernels); there is no CPU fallback")


class PCAFusionModel(nn.Module):
    def __init__(self, fingerprint_size, image_feature_size):
        super().__init__()
        self.fingerprint_fc = nn.Sequential(nn.Linear(fingerprint_size, 128), nn.ReLU())
        self.image_fc = nn.Sequential(nn.Linear(image_feature_size, 128), nn.ReLU())
        self.attention_fusion = MultiHeadAttentionFusion(256)
        self.fc = nn.Sequential(nn.Linear(256, 128), nn.ReLU(), nn.Linear(128, 64), nn.ReLU(), nn.Linear(64, 1))
        flatten_parameters(self)

    def _apply(self, fn, *a, **k):
        out = super()._apply(fn, *a, **k)
        flatten_parameters(self)
        return out

    def forward(self, fingerprint, image):
        _need_cuda(fingerprint)
        a = run_sequential(self.fingerprint_fc, fingerprint.float().contiguous())
        b = run_sequential(self.image_fc, image.float().contiguous())
        return run_sequential(self.fc, self.attention_fusion(a, b))


class DenseMLPModel(nn.Module):
    def __init__(self, fingerprint_size, image_feature_size):
        super().__init__()
        def branch(n_in, wide):
            return nn.Sequential(nn.Linear(n_in, wide), nn.ReLU(), nn.BatchNorm1d(wide), nn.Dropout(0.2),
                                 nn.Linear(wide, 256), nn.ReLU(), nn.BatchNorm1d(256), nn.Linear(256, 128), nn.ReLU())
        self.fingerprint_fc = branch(fingerprint_size, 512)
        self.image_fc = branch(image_feature_size, 1024)
        self.fc = nn.Sequential(nn.Linear(128 + 128, 256), nn.ReLU(), nn.BatchNorm1d(256), nn.Linear(256, 128), nn.ReLU(),
                                nn.Linear(128, 64), nn.ReLU(), nn.Linear(64, 1))
        flatten_parameters(self)

    def _apply(self, fn, *a, **k):
        out = super()._apply(fn, *a, **k)
        flatten_parameters(self)
        return out

    def forward(self, fingerprint, image):
        _need_cuda(fingerprint)
        a = run_sequential(self.fingerprint_fc, fingerprint.float().contiguous())
        b = run_sequential(self.image_fc, image.float().contiguous())
        return run_sequential(self.fc, torch.cat((a, b), dim=1))
