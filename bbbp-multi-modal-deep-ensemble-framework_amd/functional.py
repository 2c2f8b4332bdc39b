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


def attention_fusion(module, x1, x2, heads=None):
    """MultiHeadAttentionFusion.forward (Models/...20250113.py:60-65) on the HIP ops.  ``heads`` overrides the list of
    Linear/Tanh/Linear scorers (the single-head AttentionFusion of ..._rdkit.py:53-66 is the one-head case: its softmax runs
    over a size-1 dimension, so the weight is exactly 1 and its scorer's gradients exactly 0)."""
    combined = torch.cat((x1, x2), dim=1).contiguous()
    heads = module.attention_heads if heads is None else heads
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


class _MatMul(torch.autograd.Function):
    """op(a) @ op(b) on the MFMA GEMM with its two gradient GEMMs (2-D operands, unit inner stride)."""

    @staticmethod
    def forward(ctx, a, b, trans_a, trans_b):
        ctx.ta, ctx.tb = trans_a, trans_b
        ctx.save_for_backward(a, b)
        return ops.gemm(a, b, trans_a=trans_a, trans_b=trans_b)

    @staticmethod
    def backward(ctx, dc):
        a, b = ctx.saved_tensors
        dc = dc.contiguous()
        ta, tb = ctx.ta, ctx.tb
        da = db = None
        if ctx.needs_input_grad[0]:
            if not ta:   # C = A op(B): dA = dC op(B)^T
                da = ops.gemm(dc, b, trans_b=not tb)
            else:        # C = A^T op(B): dA = op(B) dC^T
                da = ops.gemm(b, dc, trans_a=tb, trans_b=True)
        if ctx.needs_input_grad[1]:
            if not tb:   # C = op(A) B: dB = op(A)^T dC
                db = ops.gemm(a, dc, trans_a=not ta)
            else:        # C = op(A) B^T: dB = dC^T op(A)
                db = ops.gemm(dc, a, trans_a=True, trans_b=ta)
        return da, db, None, None


def matmul(a, b, trans_a=False, trans_b=False):
    return _MatMul.apply(a.contiguous(), b.contiguous(), trans_a, trans_b)


class _Softmax(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        p, _ = ops.softmax_fwd(x)
        ctx.save_for_backward(p)
        return p

    @staticmethod
    def backward(ctx, dp):
        (p,) = ctx.saved_tensors
        return ops.softmax_bwd(dp, p)


def softmax_lastdim(x):
    return _Softmax.apply(x.contiguous())


class _ConvReluPool(torch.autograd.Function):
    """Conv2d(k3,s1,p1) + ReLU + MaxPool2d(2,2) as one fused HIP op with its data / weight gradients."""

    @staticmethod
    def forward(ctx, x, weight, bias, beside):
        if beside:
            # another branch runs beside this one: the one-work-group-per-CU forward kernel where there is one (64 x 64 maps)
            L = _lib.lib()
            prev = L.bbbp_set_conv2_fwd_pipe(1)
            try:
                y, mask = ops.conv3x3_relu_pool_fwd(x, weight, bias)
            finally:
                L.bbbp_set_conv2_fwd_pipe(prev)
        else:
            y, mask = ops.conv3x3_relu_pool_fwd(x, weight, bias)
        ctx.save_for_backward(x, weight, mask)
        ctx.beside = beside
        return y

    @staticmethod
    def backward(ctx, gy):
        x, weight, mask = ctx.saved_tensors
        gy = gy.contiguous()
        if ctx.beside:
            # another branch's kernels run beside this one on a second stream: the one-wave-per-SIMD form of the weight gradient (set for
            # this thread -- autograd's -- around the call)
            L = _lib.lib()
            prev = L.bbbp_set_conv_wgrad_beside_encoder(1)
            try:
                dw, db = ops.conv3x3_relu_pool_bwd_weight(x, gy, mask)
            finally:
                L.bbbp_set_conv_wgrad_beside_encoder(prev)
        else:
            dw, db = ops.conv3x3_relu_pool_bwd_weight(x, gy, mask)
        dx = ops.conv3x3_relu_pool_bwd_data(gy, mask, weight) if ctx.needs_input_grad[0] else None
        return dx, dw, db, None


def conv3x3_relu_pool(x, conv: "torch.nn.Conv2d", beside_other_branch: bool = False):
    """``beside_other_branch``: the caller runs another branch of the model on a second stream at the same time (variants.py)."""
    return _ConvReluPool.apply(x.contiguous(), conv.weight, conv.bias, bool(beside_other_branch))


class _EncoderLayer(torch.autograd.Function):
    """One post-norm nn.TransformerEncoderLayer applied to x[S, E] (sequence = the mini-batch, batch 1), op by op on the HIP
    kernels -- the same schedule as csrc/engine.hip, orchestrated from Python for the model variants that do not have a
    fused engine.  params: in_w, in_b, out_w, out_b, w1, b1, w2, b2, n1w, n1b, n2w, n2b."""

    @staticmethod
    def forward(ctx, x, nhead, p_drop, seed, group, *P):
        in_w, in_b, out_w, out_b, w1, b1, w2, b2, n1w, n1b, n2w, n2b = P
        S, E = x.shape
        d = E // nhead
        scale = 1.0 / (d ** 0.5)
        qkv = ops.linear(x, in_w, in_b)
        # exact-global-batch mode: the sequence IS the batch, sharded over ranks; keys and values of all ranks are gathered
        # (one collective per layer, K|V in one buffer), queries stay local
        kv = _gather_rows(qkv[:, E:].contiguous(), group) if group is not None else None
        probs, pds = [], []
        ctxv = torch.empty((S, E), device=x.device, dtype=torch.float32)
        for h in range(nhead):
            q = qkv[:, h * d:(h + 1) * d]
            if kv is None:
                k, v = qkv[:, E + h * d:E + (h + 1) * d], qkv[:, 2 * E + h * d:2 * E + (h + 1) * d]
            else:
                k, v = kv[:, h * d:(h + 1) * d], kv[:, E + h * d:E + (h + 1) * d]
            s = ops.gemm(q, k, trans_b=True, alpha=scale)
            pr, pd = ops.softmax_fwd(s, p_drop, seed + 16 * h)
            ops.gemm(pd, v, out=ctxv[:, h * d:(h + 1) * d])
            probs.append(pr); pds.append(pd if p_drop > 0 else None)
        sa = ops.linear(ctxv, out_w, out_b)
        y1, z1, mean1, rstd1 = ops.layernorm_fwd(sa, x, n1w, n1b, 1e-5, p_drop, seed + 1)
        hff = ops.linear(y1, w1, b1, act="relu")
        if p_drop > 0:
            hff = ops.dropout(hff, p_drop, seed + 2)
        ff = ops.linear(hff, w2, b2)
        y2, z2, mean2, rstd2 = ops.layernorm_fwd(ff, y1, n2w, n2b, 1e-5, p_drop, seed + 3)
        ctx.cfg = (nhead, p_drop, seed, group)
        ctx.pds = pds
        ctx.kv = kv
        ctx.save_for_backward(x, qkv, ctxv, z1, mean1, rstd1, y1, hff, z2, mean2, rstd2, *probs, *P)
        return y2

    @staticmethod
    def backward(ctx, dy):
        nhead, p_drop, seed, group = ctx.cfg
        kv = ctx.kv
        saved = ctx.saved_tensors
        x, qkv, ctxv, z1, mean1, rstd1, y1, hff, z2, mean2, rstd2 = saved[:11]
        probs = saved[11:11 + nhead]
        in_w, in_b, out_w, out_b, w1, b1, w2, b2, n1w, n1b, n2w, n2b = saved[11 + nhead:]
        S, E = x.shape
        d = E // nhead
        scale = 1.0 / (d ** 0.5)
        inv_keep = 1.0 / (1.0 - p_drop) if p_drop > 0 else 1.0
        dy = dy.contiguous()
        dz2, dff, dn2w, dn2b = ops.layernorm_bwd(dy, z2, n2w, mean2, rstd2, p_drop, seed + 3)
        dw2 = ops.gemm(dff, hff, trans_a=True)
        db2 = ops.bias_act_bwd(dff, None)
        dh = ops.gemm(dff, w2)
        db1 = ops.bias_act_bwd(dh, hff, act="relu", scale=inv_keep)
        dw1 = ops.gemm(dh, y1, trans_a=True)
        dy1 = ops.gemm(dh, w1, residual=dz2)
        dz1, dsa, dn1w, dn1b = ops.layernorm_bwd(dy1, z1, n1w, mean1, rstd1, p_drop, seed + 1)
        dwo = ops.gemm(dsa, ctxv, trans_a=True)
        dbo = ops.bias_act_bwd(dsa, None)
        dctx = ops.gemm(dsa, out_w)
        dqkv = torch.empty_like(qkv)
        dkv = torch.empty_like(kv) if kv is not None else None      # [S_global, 2E]: this rank's queries' share of dK | dV
        for h in range(nhead):
            q = qkv[:, h * d:(h + 1) * d]
            if kv is None:
                k, v = qkv[:, E + h * d:E + (h + 1) * d], qkv[:, 2 * E + h * d:2 * E + (h + 1) * d]
                dk_out, dv_out = dqkv[:, E + h * d:E + (h + 1) * d], dqkv[:, 2 * E + h * d:2 * E + (h + 1) * d]
            else:
                k, v = kv[:, h * d:(h + 1) * d], kv[:, E + h * d:E + (h + 1) * d]
                dk_out, dv_out = dkv[:, h * d:(h + 1) * d], dkv[:, E + h * d:E + (h + 1) * d]
            dch = dctx[:, h * d:(h + 1) * d]
            pd = ctx.pds[h] if ctx.pds[h] is not None else probs[h]
            ops.gemm(pd, dch, trans_a=True, out=dv_out)
            dpd = ops.gemm(dch, v, trans_b=True)
            ds = ops.softmax_bwd(dpd, probs[h], p_drop, seed + 16 * h)
            ops.gemm(ds, k, alpha=scale, out=dqkv[:, h * d:(h + 1) * d])
            ops.gemm(ds, q, trans_a=True, alpha=scale, out=dk_out)
        if dkv is not None:
            dqkv[:, E:] = _reduce_scatter_rows(dkv, group)          # sum over ranks, keep the local rows
        dwin = ops.gemm(dqkv, x, trans_a=True)
        dbin = ops.bias_act_bwd(dqkv, None)
        dx = ops.gemm(dqkv, in_w, residual=dz1) if ctx.needs_input_grad[0] else None
        return (dx, None, None, None, None, dwin, dbin, dwo, dbo, dw1, db1, dw2, db2, dn1w, dn1b, dn2w, dn2b)


def _gather_rows(t, group):
    """Concatenate the row shards of all ranks (rank order = global row order)."""
    import torch.distributed as dist
    world = dist.get_world_size(group)
    out = torch.empty((world * t.shape[0],) + tuple(t.shape[1:]), device=t.device, dtype=t.dtype)
    if dist.get_backend(group) == "gloo":          # rehearsal backend: list form
        dist.all_gather(list(out.chunk(world, dim=0)), t.contiguous(), group=group)
    else:
        dist.all_gather_into_tensor(out, t.contiguous(), group=group)
    return out


def _reduce_scatter_rows(t, group):
    """Sum a [world * n, ...] tensor over ranks and return this rank's n rows."""
    import torch.distributed as dist
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    n = t.shape[0] // world
    if dist.get_backend(group) == "gloo":          # no reduce_scatter in gloo (CPU / rehearsal): all-reduce and slice
        t = t.contiguous()
        dist.all_reduce(t, group=group)
        return t[rank * n:(rank + 1) * n]
    out = torch.empty((n,) + tuple(t.shape[1:]), device=t.device, dtype=t.dtype)
    dist.reduce_scatter_tensor(out, t.contiguous(), group=group)
    return out


class _SyncBatchNorm1d(torch.autograd.Function):
    """BatchNorm1d over a batch that is sharded across ranks: statistics and the two backward sums are all-reduced
    ([2 x C] floats per collective), the element-wise work runs in the HIP kernels."""

    @staticmethod
    def forward(ctx, x, weight, bias, running_mean, running_var, eps, momentum, group):
        import torch.distributed as dist
        rows, cols = x.shape
        world = dist.get_world_size(group)
        n = rows * world
        L = _lib.lib()
        st = torch.empty((2, cols), device=x.device, dtype=torch.float32)
        _lib.check(L.bbbp_column_moments(ops._stream(), x.data_ptr(), None, st[0].data_ptr(), st[1].data_ptr(), rows, cols), "column_moments")
        dist.all_reduce(st[0], group=group)
        mean = st[0] / n
        _lib.check(L.bbbp_column_moments(ops._stream(), x.data_ptr(), mean.data_ptr(), st[0].data_ptr(), st[1].data_ptr(), rows, cols), "column_moments")
        dist.all_reduce(st[1], group=group)
        var = st[1] / n                                            # biased, two-pass about the global mean
        y, save_mean, save_rstd = ops.batchnorm1d_fwd(x, weight, bias, mean, var, False, eps, momentum)   # normalise with the given stats
        with torch.no_grad():
            running_mean.mul_(1 - momentum).add_(mean, alpha=momentum)
            running_var.mul_(1 - momentum).add_(var * (n / (n - 1)), alpha=momentum)
        ctx.group, ctx.n = group, n
        ctx.save_for_backward(x, weight, save_mean, save_rstd)
        return y

    @staticmethod
    def backward(ctx, dy):
        import torch.distributed as dist
        x, weight, save_mean, save_rstd = ctx.saved_tensors
        dy = dy.contiguous()
        rows, cols = x.shape
        _, dg, db = ops.batchnorm1d_bwd(dy, x, weight, save_mean, save_rstd, True)      # local sums (its dx is not used)
        sums = torch.stack((db, dg))
        dist.all_reduce(sums, group=ctx.group)
        dx = torch.empty_like(x)
        _lib.check(_lib.lib().bbbp_batchnorm1d_bwd_apply(ops._stream(), dy.data_ptr(), x.data_ptr(), weight.data_ptr(),
                                                         save_mean.data_ptr(), save_rstd.data_ptr(), sums[0].data_ptr(),
                                                         sums[1].data_ptr(), dx.data_ptr(), rows, cols, ctx.n), "batchnorm1d_bwd_apply")
        return dx, dg, db, None, None, None, None, None


def sync_batchnorm1d(x, bn: "torch.nn.BatchNorm1d", group):
    if not bn.training:
        return batchnorm1d(x, bn)
    y = _SyncBatchNorm1d.apply(x.contiguous(), bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.eps,
                               0.1 if bn.momentum is None else bn.momentum, group)
    bn.num_batches_tracked += 1
    return y


def transformer_encoder(x, encoder: "torch.nn.TransformerEncoder", nhead: int, training: bool, group=None):
    """nn.TransformerEncoder over x[S, E] (sequence axis = the mini-batch, as the reference calls it with [B,1,F]).
    With ``group`` the rows of x are this rank's shard of the global batch (sequence-parallel attention)."""
    x = x.contiguous()
    for layer in encoder.layers:
        p = float(layer.dropout.p) if training else 0.0
        seed = int(torch.randint(0, 2 ** 60, (1,)).item()) if p > 0 else 0
        a = layer.self_attn
        x = _EncoderLayer.apply(x, nhead, p, seed, group, a.in_proj_weight, a.in_proj_bias, a.out_proj.weight, a.out_proj.bias,
                                layer.linear1.weight, layer.linear1.bias, layer.linear2.weight, layer.linear2.bias,
                                layer.norm1.weight, layer.norm1.bias, layer.norm2.weight, layer.norm2.bias)
    return x
