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
               optimizer=None, batch_orders: Optional[List[np.ndarray]] = None) -> Dict[str, list]:
    """Train ``model`` on ``train = (fingerprints[N,F], images[N,49152], labels[N])`` (device tensors).  Returns the per-epoch
    mean training / validation losses like the reference's ``train_losses`` / ``val_losses`` lists.  ``batch_orders``
    (one permutation per epoch) overrides the shuffling for reproducible comparisons."""
    fp, img, y = train
    if not (fp.is_cuda and img.is_cuda and y.is_cuda):
        raise RuntimeError("train_fold expects device-resident tensors")
    y = y.to(torch.float32)
    opt = optimizer if optimizer is not None else AdamW(model.parameters(), lr=lr, weight_decay=weight_decay)
    crit = MSELoss()                      # nn.MSELoss semantics, fused value + gradient kernel
    N = fp.shape[0]
    hist = {"train_loss": [], "val_loss": []}
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
            opt.zero_grad(set_to_none=True)
            pred = model(fp[idx], img[idx]).squeeze()
            loss = crit(pred, y[idx])
            loss.backward()
            opt.step()
            total += loss.detach()
            nb += 1
        hist["train_loss"].append(float(total) / nb)
        if test is not None:
            tfp, timg, ty = test
            model.eval()                            # reference :195 -- and it stays in eval mode when faithful
            with torch.no_grad():
                vl, vb = 0.0, 0
                for i in range(0, tfp.shape[0], batch_size):
                    p = model(tfp[i:i + batch_size], timg[i:i + batch_size]).squeeze()
                    vl += float(crit(p, ty[i:i + batch_size].to(torch.float32))); vb += 1
            hist["val_loss"].append(vl / vb)
    return hist
