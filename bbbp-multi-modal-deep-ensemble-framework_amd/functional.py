"""Autograd-aware building blocks on the HIP ops, for the modules that are used outside the fused
whole-model path (standalone fusion block, the reference's MLP-only model variants)."""
from __future__ import annotations

import ctypes

import torch

from . import _lib, ops


class _Linear(torch.autograd.Function):
    """y = act(x W^T + b) on the MFMA GEMM (nn.Linear + ReLU/Tanh of the reference's Sequential blocks)."""

    @staticmethod
    def forward(ctx, x, weight, bias, act):
        y = ops.linear(x, weight, bias, act=act)
        ctx.act = act
        ctx.save_for_backward(x, weight, y)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight, y = ctx.saved_tensors
        dy = dy.contiguous().clone()
        db = ops.bias_act_bwd(dy, y if ctx.act else None, act=ctx.act)
        dx = ops.gemm(dy, weight) if ctx.needs_input_grad[0] else None
        dw = ops.gemm(dy, x, trans_a=True)
        return dx, dw, db, None


def linear(x, weight, bias, act=None):
    return _Linear.apply(x.contiguous(), weight, bias, act)


class _FusionCombine(torch.autograd.Function):
    @staticmethod
    def forward(ctx, combined, hid, *w2b2):
        nh = len(w2b2) // 2
        w2, b2 = w2b2[:nh], w2b2[nh:]
        rows, dim = combined.shape
        hidden = hid.shape[2]
        out = torch.empty_like(combined)
        attn = torch.empty((rows, nh), device=combined.device, dtype=torch.float32)
        L = _lib.lib()
        _lib.check(L.bbbp_fusion_combine_fwd(ops._stream(), combined.data_ptr(), hid.data_ptr(),
                                             _lib.ptr_array([w.data_ptr() for w in w2]),
                                             _lib.ptr_array([b.data_ptr() for b in b2]), out.data_ptr(), attn.data_ptr(),
                                             rows, dim, hidden, nh), "bbbp_fusion_combine_fwd")
        ctx.nh = nh
        ctx.save_for_backward(combined, hid, attn, *w2)
        return out

    @staticmethod
    def backward(ctx, dout):
        combined, hid, attn, *w2 = ctx.saved_tensors
        nh = ctx.nh
        rows, dim = combined.shape
        hidden = hid.shape[2]
        dout = dout.contiguous()
        dcomb = torch.empty_like(combined)
        dlogit = torch.empty((nh, rows), device=dout.device, dtype=torch.float32)
        dpre = torch.empty_like(hid)
        L = _lib.lib()
        _lib.check(L.bbbp_fusion_combine_bwd(ops._stream(), dout.data_ptr(), combined.data_ptr(), hid.data_ptr(),
                                             attn.data_ptr(), _lib.ptr_array([w.data_ptr() for w in w2]), dcomb.data_ptr(),
                                             dlogit.data_ptr(), dpre.data_ptr(), rows, dim, hidden, nh),
                   "bbbp_fusion_combine_bwd")
        # dpre is the gradient wrt the PRE-tanh activations; hid receives it through _TanhPre below
        dw2 = [ops.gemm(dlogit[h].view(rows, 1), hid[h], trans_a=True) for h in range(nh)]
        db2 = [dlogit[h].sum().view(1) for h in range(nh)]
        ctx.dpre = dpre
        # hand d(pre-tanh) to the producer of hid through its grad: hid = tanh(pre) => dhid = dpre / (1 - hid^2)
        # is ill-conditioned, so the heads' first Linear is differentiated here instead (see attention_fusion).
        return (dcomb, dpre, *dw2, *db2)


class _HeadHidden(torch.autograd.Function):
    """hid[h] = tanh(combined W1_h^T + b1_h) for all heads; its backward receives d(pre-tanh) directly."""

    @staticmethod
    def forward(ctx, combined, *w1b1):
        nh = len(w1b1) // 2
        w1, b1 = w1b1[:nh], w1b1[nh:]
        rows = combined.shape[0]
        hidden = w1[0].shape[0]
        hid = torch.empty((nh, rows, hidden), device=combined.device, dtype=torch.float32)
        for h in range(nh):
            ops.linear(combined, w1[h], b1[h], act="tanh", out=hid[h])
        ctx.nh = nh
        ctx.save_for_backward(combined, *w1)
        return hid

    @staticmethod
    def backward(ctx, dpre):
        combined, *w1 = ctx.saved_tensors
        nh = ctx.nh
        dpre = dpre.contiguous()
        dcomb = None
        dw1, db1 = [], []
        for h in range(nh):
            dw1.append(ops.gemm(dpre[h], combined, trans_a=True))
            db1.append(ops.bias_act_bwd(dpre[h], None))
            dcomb = ops.gemm(dpre[h], w1[h], residual=dcomb)
        return (dcomb, *dw1, *db1)


def attention_fusion(module, x1, x2):
    """MultiHeadAttentionFusion.forward (Models/...20250113.py:60-65) on the HIP ops."""
    combined = torch.cat((x1, x2), dim=1).contiguous()
    heads = module.attention_heads
    w1 = [h[0].weight for h in heads]; b1 = [h[0].bias for h in heads]
    w2 = [h[2].weight for h in heads]; b2 = [h[2].bias for h in heads]
    hid = _HeadHidden.apply(combined, *w1, *b1)
    return _FusionCombine.apply(combined, hid, *w2, *b2)


class _BatchNorm1d(torch.autograd.Function):
    """nn.BatchNorm1d forward/backward on the HIP kernels; running statistics are updated in place when training."""

    @staticmethod
    def forward(ctx, x, weight, bias, running_mean, running_var, training, eps, momentum):
        y, save_mean, save_rstd = ops.batchnorm1d_fwd(x, weight, bias, running_mean, running_var, training, eps, momentum)
        ctx.training = training
        ctx.save_for_backward(x, weight, save_mean, save_rstd)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight, save_mean, save_rstd = ctx.saved_tensors
        dx, dg, db = ops.batchnorm1d_bwd(dy, x, weight, save_mean, save_rstd, ctx.training)
        return dx, dg, db, None, None, None, None, None


def batchnorm1d(x, bn: "torch.nn.BatchNorm1d"):
    y = _BatchNorm1d.apply(x.contiguous(), bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.training, bn.eps,
                           0.1 if bn.momentum is None else bn.momentum)
    if bn.training and bn.num_batches_tracked is not None:
        bn.num_batches_tracked += 1
    return y


class _Dropout(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, p, seed):
        ctx.p, ctx.seed = p, seed
        return ops.dropout(x, p, seed)

    @staticmethod
    def backward(ctx, dy):
        return ops.dropout(dy.contiguous(), ctx.p, ctx.seed), None, None


def dropout(x, p: float, training: bool):
    """nn.Dropout: identity in eval mode; the mask stream is seeded from torch's CPU generator."""
    if not training or p == 0.0:
        return x
    seed = int(torch.randint(0, 2 ** 62, (1,)).item())
    return _Dropout.apply(x.contiguous(), float(p), seed)


def run_sequential(seq, x):
    """Execute an nn.Sequential of Linear / ReLU / Tanh / BatchNorm1d / Dropout on the HIP ops, fusing each
    Linear with the activation that follows it (the reference's MLP branches are built this way)."""
    mods = list(seq)
    i = 0
    while i < len(mods):
        m = mods[i]
        if isinstance(m, torch.nn.Linear):
            act = None
            if i + 1 < len(mods) and isinstance(mods[i + 1], torch.nn.ReLU):
                act, i = "relu", i + 1
            elif i + 1 < len(mods) and isinstance(mods[i + 1], torch.nn.Tanh):
                act, i = "tanh", i + 1
            x = linear(x, m.weight, m.bias, act)
        elif isinstance(m, torch.nn.BatchNorm1d):
            x = batchnorm1d(x, m)
        elif isinstance(m, torch.nn.Dropout):
            x = dropout(x, m.p, m.training)
        elif isinstance(m, (torch.nn.ReLU, torch.nn.Tanh)):
            raise RuntimeError("activation without a preceding Linear is not on the reference's path")
        else:
            raise RuntimeError(f"unsupported module on the HIP path: {type(m).__name__}")
        i += 1
    return x
