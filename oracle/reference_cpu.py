"""CPU oracle for the BBBP multi-modal hot path.  TEST INFRASTRUCTURE ONLY.

This file restates, with plain PyTorch CPU tensor ops, the arithmetic of the reference's
``MixedInputModel`` family so that the HIP path can be checked against it.  It is imported
only by ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py``;
the product package never imports it and fails loudly when its HIP library is missing.

Pinning: the restatement is pinned by golden vectors produced from the reference's OWN classes
(``tools/make_golden.py`` extracts the class definitions from the files under /root/reference
as text and runs them on seeded inputs; the fixtures live in ``tests/golden/*.npz``), by the two
shipped ``best_nn_model*.pth`` state_dicts and by the five decoded ``stacked_model*.pkl``
coefficient vectors.  ``tests/test_oracle_golden.py`` checks all of them.

Everything is functional: parameters arrive as a ``dict`` keyed by the reference's own
``state_dict`` names, so the oracle shares no module code with the product.

Reference citations are relative to /root/reference/.
"""
from __future__ import annotations

import math
from typing import Dict, Optional, Tuple

import numpy as np
import torch
import torch.nn.functional as F

Params = Dict[str, torch.Tensor]


# --------------------------------------------------------------------------------------------
# a2: head-count rule
# --------------------------------------------------------------------------------------------
def nhead_rule(fingerprint_size: int, start: Optional[int] = None) -> int:
    """Models/multi_input_data_regression_opt_transformer_cnn_20250113.py:71-73 (start=None):
    nhead = max(1, F // 8), decremented until it divides F.
    Models/..._opt_20250107_network.py:112-117 (start=8): start at 8, decrement while > 1."""
    if start is None:
        nhead = max(1, fingerprint_size // 8)
        while fingerprint_size % nhead != 0:
            nhead -= 1
        return nhead
    nhead = start
    while fingerprint_size % nhead != 0 and nhead > 1:
        nhead -= 1
    if fingerprint_size % nhead != 0:
        raise ValueError(f"fingerprint_size={fingerprint_size} must be divisible by nhead={nhead}.")
    return nhead


# --------------------------------------------------------------------------------------------
# a3: one post-norm encoder layer applied to x[S, E] (sequence axis = the mini-batch, N = 1)
# --------------------------------------------------------------------------------------------
def encoder_layer(x: torch.Tensor, p: Params, prefix: str, nhead: int,
                  eps: float = 1e-5, ffn_gate: Optional[torch.Tensor] = None,
                  ffn_pre: Optional[list] = None) -> torch.Tensor:
    """nn.TransformerEncoderLayer(d_model=F, nhead) with defaults (post-norm, ReLU,
    dim_feedforward=2048, batch_first=False) as built at ...20250113.py:75-78 and called at
    :110-111 with a [B,1,F] tensor, i.e. sequence length S=B and batch 1.  Dropout is the
    identity here (eval mode / p=0): train-mode dropout has no cross-implementation parity."""
    S, E = x.shape
    d = E // nhead
    qkv = F.linear(x, p[prefix + "self_attn.in_proj_weight"], p[prefix + "self_attn.in_proj_bias"])
    q, k, v = qkv[:, :E], qkv[:, E:2 * E], qkv[:, 2 * E:]
    # heads: [S, E] -> [h, S, d]
    q = q.reshape(S, nhead, d).transpose(0, 1)
    k = k.reshape(S, nhead, d).transpose(0, 1)
    v = v.reshape(S, nhead, d).transpose(0, 1)
    scores = torch.matmul(q, k.transpose(1, 2)) * (1.0 / math.sqrt(d))      # [h, S, S]
    probs = torch.softmax(scores, dim=-1)
    ctx = torch.matmul(probs, v).transpose(0, 1).reshape(S, E)              # concat heads
    sa = F.linear(ctx, p[prefix + "self_attn.out_proj.weight"], p[prefix + "self_attn.out_proj.bias"])
    x = F.layer_norm(x + sa, (E,), p[prefix + "norm1.weight"], p[prefix + "norm1.bias"], eps)
    pre = F.linear(x, p[prefix + "linear1.weight"], p[prefix + "linear1.bias"])
    if ffn_pre is not None:
        ffn_pre.append(pre.detach())
    # ``ffn_gate`` (a 0/1 tensor) replaces the ReLU's own decision pre > 0: parity tests at large batch hand in the decisions
    # of the implementation under test, so that a pre-activation within rounding of zero (where float32 and float64 may
    # legitimately disagree about the sign) does not turn into a gradient difference; the test checks separately that the
    # decisions differ only at such near-zero elements
    h = F.relu(pre) if ffn_gate is None else pre * ffn_gate.to(pre.dtype)
    ff = F.linear(h, p[prefix + "linear2.weight"], p[prefix + "linear2.bias"])
    x = F.layer_norm(x + ff, (E,), p[prefix + "norm2.weight"], p[prefix + "norm2.bias"], eps)
    return x


def encoder(x: torch.Tensor, p: Params, prefix: str, nhead: int, num_layers: int,
            ffn_gates: Optional[list] = None, ffn_pre: Optional[list] = None) -> torch.Tensor:
    for i in range(num_layers):
        x = encoder_layer(x, p, f"{prefix}layers.{i}.", nhead, ffn_gate=None if ffn_gates is None else ffn_gates[i], ffn_pre=ffn_pre)
    return x


# --------------------------------------------------------------------------------------------
# a6/a7: conv3x3(pad 1) + ReLU + maxpool 2x2
# --------------------------------------------------------------------------------------------
def pool_windows(c: torch.Tensor) -> torch.Tensor:
    """[B,C,H,W] -> [B,C,H/2,W/2,4]: the 2x2 windows of MaxPool2d(2,2), last axis in scan order (0,0),(0,1),(1,0),(1,1)."""
    B, C, H, W = c.shape
    return c.reshape(B, C, H // 2, 2, W // 2, 2).permute(0, 1, 2, 4, 3, 5).reshape(B, C, H // 2, W // 2, 4)


def pool_decisions(c: torch.Tensor) -> torch.Tensor:
    """The decision ReLU + MaxPool2d(2,2) take per pooled element of pre-activations ``c``, in the encoding of the HIP path's saved
    mask (include/bbbp_hip.h: bbbp_mixed_debug_pool_mask): 0..3 = position of the FIRST maximum of the window (PyTorch's rule on
    ties), 4 = the maximum is <= 0 (ReLU passes nothing)."""
    win = pool_windows(c)
    best, idx = win.max(dim=-1)                                   # torch.max returns the first index among equal maxima on CPU
    first = (win == best.unsqueeze(-1)).to(torch.uint8).argmax(dim=-1)
    return torch.where(best > 0, first, torch.full_like(first, 4)).to(torch.uint8)


def conv3x3_relu_pool(x: torch.Tensor, w: torch.Tensor, b: torch.Tensor, pool_mask: Optional[torch.Tensor] = None,
                      pre: Optional[list] = None) -> torch.Tensor:
    """...20250113.py:85-90: Conv2d(k=3,s=1,p=1) -> ReLU -> MaxPool2d(2,2).
    ``pool_mask`` (u8 per pooled element, encoding of ``pool_decisions``) replaces the stage's own ReLU / arg-max decisions: parity
    tests at large batch hand in the decisions of the implementation under test, so that a window whose two largest
    pre-activations agree to float32 rounding (where float32 and float64 may legitimately pick different elements, or disagree
    about the sign of a pre-activation at zero) does not turn into a gradient difference; the test checks separately that the
    decisions differ only at such near-ties.  ``pre`` (a list) receives the pre-activations."""
    c = F.conv2d(x, w, b, stride=1, padding=1)
    if pre is not None:
        pre.append(c.detach())
    if pool_mask is None:
        return F.max_pool2d(F.relu(c), 2, 2)
    m = pool_mask.to(torch.int64)
    picked = pool_windows(c).gather(-1, m.clamp(max=3).unsqueeze(-1)).squeeze(-1)
    return picked * (m < 4).to(c.dtype)


# --------------------------------------------------------------------------------------------
# a9: attention fusion (softmax over the heads)
# --------------------------------------------------------------------------------------------
def attention_fusion(x1: torch.Tensor, x2: torch.Tensor, p: Params, prefix: str,
                     num_heads: int = 4) -> torch.Tensor:
    """...20250113.py:60-65.  cat -> per head Linear/Tanh/Linear -> softmax over heads ->
    sum_h w_h * combined."""
    combined = torch.cat((x1, x2), dim=1)
    logits = []
    for h in range(num_heads):
        hid = torch.tanh(F.linear(combined, p[f"{prefix}attention_heads.{h}.0.weight"],
                                  p[f"{prefix}attention_heads.{h}.0.bias"]))
        logits.append(F.linear(hid, p[f"{prefix}attention_heads.{h}.2.weight"],
                               p[f"{prefix}attention_heads.{h}.2.bias"]).unsqueeze(1))
    w = torch.softmax(torch.cat(logits, dim=1), dim=1)                       # [B, heads, 1]
    return torch.sum(w * combined.unsqueeze(1), dim=1)


def attention_fusion_single(x1: torch.Tensor, x2: torch.Tensor, p: Params, prefix: str) -> torch.Tensor:
    """Single-head AttentionFusion, Models/multi_input_data_regression_opt_transformer_cnn_rdkit.py:53-66:
    Linear(256,128) -> Tanh -> Linear(128,1) -> Softmax(dim=1) over a size-1 dimension (weights == 1), ``weights * combined``."""
    combined = torch.cat((x1, x2), dim=1)
    hid = torch.tanh(F.linear(combined, p[prefix + "attention.0.weight"], p[prefix + "attention.0.bias"]))
    w = torch.softmax(F.linear(hid, p[prefix + "attention.2.weight"], p[prefix + "attention.2.bias"]), dim=1)   # [B, 1]
    return w * combined


def batchnorm1d(x: torch.Tensor, p: Params, prefix: str, training: bool,
                bn_state: Optional[Dict[str, torch.Tensor]] = None,
                momentum: float = 0.1, eps: float = 1e-5) -> torch.Tensor:
    """nn.BatchNorm1d defaults (...20250113.py:101).  train: batch mean / biased variance for the
    normalisation, running stats updated with the UNBIASED variance; eval: running stats.
    ``bn_state`` (if given) receives the updated running_mean / running_var / num_batches_tracked
    under the same keys instead of mutating ``p``."""
    if training:
        if x.shape[0] == 1:
            raise ValueError("Expected more than 1 value per channel when training, got input size "
                             + str(list(x.shape)))
        mean = x.mean(dim=0)
        var = x.var(dim=0, unbiased=False)
        if bn_state is not None:
            n = x.shape[0]
            with torch.no_grad():
                bn_state[prefix + "running_mean"] = (1 - momentum) * p[prefix + "running_mean"] + momentum * mean
                bn_state[prefix + "running_var"] = (1 - momentum) * p[prefix + "running_var"] + momentum * var * n / (n - 1)
                bn_state[prefix + "num_batches_tracked"] = p[prefix + "num_batches_tracked"] + 1
    else:
        mean, var = p[prefix + "running_mean"], p[prefix + "running_var"]
    return (x - mean) / torch.sqrt(var + eps) * p[prefix + "weight"] + p[prefix + "bias"]


# --------------------------------------------------------------------------------------------
# the flagship model: ...20250113.py:68-119 (== ...transformer_cnn.py:71-135)
# --------------------------------------------------------------------------------------------
def mixed_input_forward(p: Params, fingerprint: torch.Tensor, image: torch.Tensor, *,
                        training: bool = False, num_layers: int = 6,
                        bn_state: Optional[Dict[str, torch.Tensor]] = None,
                        parts: Optional[dict] = None, fusion: str = "attention",
                        ffn_gates: Optional[list] = None, pool_masks: Optional[tuple] = None) -> torch.Tensor:
    """MixedInputModel.forward(fingerprint[B,F], image[B,49152]) -> [B,1].
    ``fusion="concat"`` is the earliest variant, Descriptors/multi_input_data_regression_opt_round_2_transformer_cnn.py:89-102
    (plain torch.cat, no attention_fusion parameters); with ``num_layers=0`` on top, BASELINE config 2 (that class without
    its encoder; no exact reference script).
    ``training`` only selects the BatchNorm1d statistics (dropout is the identity in the oracle).
    ``parts`` (optional dict) receives the intermediate activations for per-op parity tests."""
    Fdim = fingerprint.shape[1]
    nhead = nhead_rule(Fdim)
    ffn_pre = [] if parts is not None else None
    x = encoder(fingerprint, p, "fingerprint_transformer.", nhead, num_layers, ffn_gates=ffn_gates, ffn_pre=ffn_pre)
    fp_out = F.relu(F.linear(x, p["fingerprint_fc.0.weight"], p["fingerprint_fc.0.bias"]))
    img = image.reshape(-1, 3, 128, 128)
    conv_pre = [] if parts is not None else None
    pm1, pm2 = pool_masks if pool_masks is not None else (None, None)
    p1 = conv3x3_relu_pool(img, p["image_cnn.0.weight"], p["image_cnn.0.bias"], pm1, conv_pre)
    p2 = conv3x3_relu_pool(p1, p["image_cnn.3.weight"], p["image_cnn.3.bias"], pm2, conv_pre)
    img_out = F.relu(F.linear(p2.flatten(1), p["image_cnn.7.weight"], p["image_cnn.7.bias"]))
    if fusion == "attention":
        fused = attention_fusion(fp_out, img_out, p, "attention_fusion.")
    elif fusion == "concat":
        fused = torch.cat((fp_out, img_out), dim=1)
    else:
        raise ValueError(fusion)
    h = F.relu(F.linear(fused, p["fc.0.weight"], p["fc.0.bias"]))
    hb = batchnorm1d(h, p, "fc.2.", training, bn_state)
    h2 = F.relu(F.linear(hb, p["fc.3.weight"], p["fc.3.bias"]))
    h3 = F.relu(F.linear(h2, p["fc.5.weight"], p["fc.5.bias"]))
    out = F.linear(h3, p["fc.7.weight"], p["fc.7.bias"])
    if parts is not None:
        parts.update(ffn_pre=ffn_pre, conv_pre=conv_pre, enc=x, fp_out=fp_out, pool1=p1, pool2=p2, img_out=img_out, fused=fused,
                     h=h, hb=hb, h2=h2, h3=h3)
    return out


def mse_loss(pred: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
    """nn.MSELoss() on pred.squeeze() (...20250113.py:143,188-189)."""
    return torch.mean((pred.squeeze() - target) ** 2)


# --------------------------------------------------------------------------------------------
# a17: PCA-MLP fusion model (Models/multi_input_data_regression_opt_transformer_cnn_opt.py:72-105);
# this is the architecture of the shipped best_nn_model*.pth
# --------------------------------------------------------------------------------------------
def pca_mlp_forward(p: Params, fingerprint: torch.Tensor, image: torch.Tensor, single_head: bool = False) -> torch.Tensor:
    """``single_head=True``: the _rdkit.py variant (Models/multi_input_data_regression_opt_transformer_cnn_rdkit.py:69-105),
    same layers around the single-head AttentionFusion."""
    a = F.relu(F.linear(fingerprint, p["fingerprint_fc.0.weight"], p["fingerprint_fc.0.bias"]))
    b = F.relu(F.linear(image, p["image_fc.0.weight"], p["image_fc.0.bias"]))
    fused = attention_fusion_single(a, b, p, "attention_fusion.") if single_head else attention_fusion(a, b, p, "attention_fusion.")
    h = F.relu(F.linear(fused, p["fc.0.weight"], p["fc.0.bias"]))
    h = F.relu(F.linear(h, p["fc.2.weight"], p["fc.2.bias"]))
    return F.linear(h, p["fc.4.weight"], p["fc.4.bias"])


def opt_more_forward(p: Params, fingerprint: torch.Tensor, image: torch.Tensor, *, training: bool = False,
                     bn_state: Optional[Dict[str, torch.Tensor]] = None) -> torch.Tensor:
    """Models/multi_input_data_regression_opt_transformer_cnn_opt_more.py:80-107: 256-wide branches Linear -> ReLU -> BatchNorm1d
    (-> Dropout(0.3): identity here), MultiHeadAttentionFusion(512), head Linear(512,256) -> ReLU -> BatchNorm1d -> Linear(256,128)
    -> ReLU -> Linear(128,1)."""
    a = batchnorm1d(F.relu(F.linear(fingerprint, p["fingerprint_fc.0.weight"], p["fingerprint_fc.0.bias"])), p, "fingerprint_fc.2.", training, bn_state)
    b = batchnorm1d(F.relu(F.linear(image, p["image_fc.0.weight"], p["image_fc.0.bias"])), p, "image_fc.2.", training, bn_state)
    fused = attention_fusion(a, b, p, "attention_fusion.")
    h = batchnorm1d(F.relu(F.linear(fused, p["fc.0.weight"], p["fc.0.bias"])), p, "fc.2.", training, bn_state)
    h = F.relu(F.linear(h, p["fc.3.weight"], p["fc.3.bias"]))
    return F.linear(h, p["fc.5.weight"], p["fc.5.bias"])


# --------------------------------------------------------------------------------------------
# a16: dense raw-feature MLP (Models/multi_input_data_regression_opt.py:41-85), dropout = identity
# --------------------------------------------------------------------------------------------
def dense_mlp_forward(p: Params, fingerprint: torch.Tensor, image: torch.Tensor, *,
                      training: bool = False,
                      bn_state: Optional[Dict[str, torch.Tensor]] = None) -> torch.Tensor:
    def branch(x, pre):
        x = F.relu(F.linear(x, p[pre + "0.weight"], p[pre + "0.bias"]))
        x = batchnorm1d(x, p, pre + "2.", training, bn_state)
        x = F.relu(F.linear(x, p[pre + "4.weight"], p[pre + "4.bias"]))
        x = batchnorm1d(x, p, pre + "6.", training, bn_state)
        return F.relu(F.linear(x, p[pre + "7.weight"], p[pre + "7.bias"]))
    a = branch(fingerprint, "fingerprint_fc.")
    b = branch(image, "image_fc.")
    h = F.relu(F.linear(torch.cat((a, b), dim=1), p["fc.0.weight"], p["fc.0.bias"]))
    h = batchnorm1d(h, p, "fc.2.", training, bn_state)
    h = F.relu(F.linear(h, p["fc.3.weight"], p["fc.3.bias"]))
    h = F.relu(F.linear(h, p["fc.5.weight"], p["fc.5.bias"]))
    return F.linear(h, p["fc.7.weight"], p["fc.7.bias"])


# --------------------------------------------------------------------------------------------
# wide/deep variant: Models/multi_input_data_regression_opt_transformer_cnn_opt_20250107_network.py:51-174 (dropout = identity)
# --------------------------------------------------------------------------------------------
def _seq_tanh_head(x, p, pre):
    return F.linear(torch.tanh(F.linear(x, p[pre + "0.weight"], p[pre + "0.bias"])), p[pre + "2.weight"], p[pre + "2.bias"])


def multimodal_attention_fusion(fp: torch.Tensor, img: torch.Tensor, p: Params, prefix: str) -> torch.Tensor:
    """:71-105 statement by statement, INCLUDING the [B,1,1] * [B,512] -> [B,B,512] broadcast and the mean over dim 1."""
    fw = _seq_tanh_head(fp, p, prefix + "fingerprint_attention.").unsqueeze(1)        # [B,1,1]
    iw = _seq_tanh_head(img, p, prefix + "image_attention.").unsqueeze(1)
    cross = _seq_tanh_head(torch.cat((fp, img), dim=1), p, prefix + "cross_modal_attention.")
    aw = torch.softmax(torch.cat([fw, iw], dim=1), dim=1)                              # [B,2,1]
    fpw = (aw[:, 0:1] * fp).mean(dim=1)                                                # [B,B,512] -> [B,512]
    imgw = (aw[:, 1:2] * img).mean(dim=1)
    return torch.cat((fpw, imgw, cross), dim=1)


def wide_deep_forward(p: Params, fingerprint: torch.Tensor, image: torch.Tensor, *, training: bool = False,
                      bn_state: Optional[Dict[str, torch.Tensor]] = None) -> torch.Tensor:
    Fdim = fingerprint.shape[1]
    nhead = nhead_rule(Fdim, start=8)
    x = encoder(fingerprint, p, "fingerprint_transformer.", nhead, 12)
    fp_out = F.relu(F.linear(x, p["fingerprint_fc.0.weight"], p["fingerprint_fc.0.bias"]))
    h = image.reshape(-1, 3, 128, 128)
    for i in (0, 3, 6):
        h = conv3x3_relu_pool(h, p[f"image_cnn.{i}.weight"], p[f"image_cnn.{i}.bias"])
    img_out = F.relu(F.linear(h.flatten(1), p["image_cnn.10.weight"], p["image_cnn.10.bias"]))
    fused = multimodal_attention_fusion(fp_out, img_out, p, "attention_fusion.")
    h = F.relu(F.linear(fused, p["fc.0.weight"], p["fc.0.bias"]))
    h = batchnorm1d(h, p, "fc.2.", training, bn_state)
    h = F.relu(F.linear(h, p["fc.3.weight"], p["fc.3.bias"]))
    h = F.relu(F.linear(h, p["fc.6.weight"], p["fc.6.bias"]))
    h = F.relu(F.linear(h, p["fc.8.weight"], p["fc.8.bias"]))
    h = F.relu(F.linear(h, p["fc.10.weight"], p["fc.10.bias"]))
    return F.linear(h, p["fc.12.weight"], p["fc.12.bias"])


# --------------------------------------------------------------------------------------------
# a13: AdamW (torch.optim.AdamW defaults used at ...20250113.py:172: lr 1e-4, wd 1e-5)
# --------------------------------------------------------------------------------------------
def adamw_step(param: torch.Tensor, grad: torch.Tensor, exp_avg: torch.Tensor,
               exp_avg_sq: torch.Tensor, step: int, *, lr: float = 1e-4,
               betas: Tuple[float, float] = (0.9, 0.999), eps: float = 1e-8,
               weight_decay: float = 1e-5) -> None:
    """In-place single-tensor AdamW step, ``step`` is the 1-based step count.
    p *= 1 - lr*wd ; m = b1 m + (1-b1) g ; v = b2 v + (1-b2) g^2 ;
    p -= lr/(1-b1^t) * m / (sqrt(v)/sqrt(1-b2^t) + eps).
    The op sequence is torch.optim.adamw._single_tensor_adamw's (torch 2.x: the first moment is a lerp_), so that in float32 the
    roundings are torch's own -- tests/test_oracle_golden.py holds it bit for bit against torch.optim.AdamW."""
    b1, b2 = betas
    param.mul_(1 - lr * weight_decay)
    exp_avg.lerp_(grad, 1 - b1)
    exp_avg_sq.mul_(b2).addcmul_(grad, grad, value=1 - b2)
    bc1 = 1 - b1 ** step
    bc2 = 1 - b2 ** step
    denom = (exp_avg_sq.sqrt() / math.sqrt(bc2)).add_(eps)
    param.addcdiv_(exp_avg, denom, value=-lr / bc1)


# --------------------------------------------------------------------------------------------
# a19: ensemble combiners (float64, numpy)
# --------------------------------------------------------------------------------------------
def weighted_ensemble(nn_pred, rf_pred, xgb_pred, weights=(0.4, 0.3, 0.3)) -> np.ndarray:
    """Models/multi_input_data_regression_opt_transformer_cnn.py:216-218."""
    return (weights[0] * np.asarray(nn_pred, dtype=np.float64)
            + weights[1] * np.asarray(rf_pred, dtype=np.float64)
            + weights[2] * np.asarray(xgb_pred, dtype=np.float64))


def linear_fit(X: np.ndarray, y: np.ndarray, alpha: float = 0.0) -> Tuple[np.ndarray, float]:
    """LinearRegression (alpha=0) / Ridge(alpha) with an unpenalised intercept, as used at
    ..._opt.py:173-176 and ...20250113.py:398: centre, solve, recover the intercept."""
    X = np.asarray(X, dtype=np.float64)
    y = np.asarray(y, dtype=np.float64)
    xm, ym = X.mean(axis=0), y.mean()
    Xc, yc = X - xm, y - ym
    if alpha == 0.0:
        coef = np.linalg.lstsq(Xc, yc, rcond=None)[0]
    else:
        coef = np.linalg.solve(Xc.T @ Xc + alpha * np.eye(X.shape[1]), Xc.T @ yc)
    return coef, float(ym - xm @ coef)


def linear_predict(X: np.ndarray, coef: np.ndarray, intercept: float) -> np.ndarray:
    """sklearn LinearModel.predict: X @ coef_ + intercept_ (..._opt.py:202-203)."""
    return np.asarray(X, dtype=np.float64) @ np.asarray(coef, dtype=np.float64) + intercept


# Known-answer vectors: the fitted meta-learners shipped in Models/stacked_model*.pkl, decoded from
# the raw pickle bytes without unpickling.  (coef for [nn, rf, xgb], intercept)
STACKED_KNOWN = {
    "stacked_model.pkl": ((0.19813994153864287, 0.8730076113813537, 0.16470120078934247), 0.019492486407121146),
    "stacked_model_maccs_opt.pkl": ((0.1407031509307508, 0.947918903501133, 0.07758759534350594), 0.01774997745987686),
    "stacked_model_morgan.pkl": ((0.11566863194671459, 1.1419519445995372, 0.06362825077495388), 0.02280488814967495),
    "stacked_model_rdkit.pkl": ((0.15119656041952942, 0.8976983917102106, 0.20171374717530366), 0.02521309973011923),
    "stacked_model_maccs_multiattention.pkl": ((0.2134672413252459, 0.6878011884620018, 0.27686389616560697), 0.016440310778525757),
}


def xgb_predict(left, right, feature, cond, default_left, root, base_score, X):
    """XGBoost's gbtree predict rule for reg:squarederror, restated (src/predictor/cpu_predictor.cc: PredValue; include/xgboost/
    tree_model.h: RegTree::GetNext / GetLeafIndex): float32 rows, missing = NaN takes the default child, otherwise left when
    x[split_index] < split_condition; a leaf's weight is split_conditions[leaf]; margin = base_score + float32 sum of the leaf
    weights in tree order.  The package is absent from the build image: this restatement is the only checker (parity unpinned)."""
    import numpy as np
    X = np.asarray(X, np.float32)
    n = X.shape[0]
    psum = np.zeros(n, np.float32)
    rows = np.arange(n)
    for t in range(len(root) - 1):
        node = np.full(n, root[t], np.int64)
        while True:
            l = left[node]
            active = l >= 0
            if not active.any():
                break
            v = X[rows, np.where(active, feature[node], 0)]
            go_left = np.where(np.isnan(v), default_left[node] != 0, v < cond[node])
            node = np.where(active, np.where(go_left, l, right[node]), node)
        psum = (psum + cond[node]).astype(np.float32)
    return (np.float32(base_score) + psum).astype(np.float32)


def catboost_predict(split_feature, split_border, nan_true, tree_first_split, tree_first_leaf, leaf_values, scale, bias, X):
    """CatBoost's oblivious-tree rule for float features, restated from the model format's documentation (JSON export:
    "oblivious_trees"[t]["splits"][i] = {float_feature_index, border}, "leaf_values", "scale_and_bias"): bit i of a tree's leaf index
    is x[feature_i] > border_i (NaN: false, or true for nan_value_treatment "AsTrue"); prediction = scale * sum_t leaf_values[t][index_t]
    + bias in float64.  Package absent, no fitted model in the reference: this restatement is the only checker (parity unpinned)."""
    import numpy as np
    X = np.asarray(X, np.float32)
    n = X.shape[0]
    acc = np.zeros(n, np.float64)
    for t in range(len(tree_first_split) - 1):
        idx = np.zeros(n, np.int64)
        for lvl, s in enumerate(range(tree_first_split[t], tree_first_split[t + 1])):
            v = X[:, split_feature[s]]
            bit = np.where(np.isnan(v), nan_true[split_feature[s]] != 0, v > split_border[s])
            idx |= bit.astype(np.int64) << lvl
        acc += leaf_values[tree_first_leaf[t] + idx]
    return scale * acc + bias
