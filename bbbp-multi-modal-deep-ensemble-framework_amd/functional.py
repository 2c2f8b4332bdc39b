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
