"""torch-tensor front end of the C ABI: one thin function per entry point of include/bbbp_hip.h.

torch supplies device memory and the current HIP stream; all arithmetic happens in libbbbp_hip.so.
Every function raises on non-CUDA / non-fp32 / non-contiguous operands, exactly like a torch op
would, and there is no CPU path.
"""
from __future__ import annotations

from typing import Optional, Tuple

import threading

import torch

from . import _lib

ACT = {None: 0, "none": 0, "relu": 1, "tanh": 2, 0: 0, 1: 1, 2: 2}


_SEED_BASE = 0                  # device address of a 64-bit step counter the dropout streams are keyed from (training.GraphedTrainStep), or 0
_tls = threading.local()


def set_seed_base(ptr: int) -> int:
    """Key every seeded op's dropout stream from the 64-bit integer at device address ``ptr`` (``bbbp_set_seed_base``); 0 restores the
    default.  Process-wide on the Python side -- the library's setting is per thread and autograd runs backward nodes on its own
    threads, so every op call re-applies it for the thread it runs on (``_stream``).  Returns the previous address."""
    global _SEED_BASE
    prev, _SEED_BASE = _SEED_BASE, int(ptr or 0)
    return prev


def _stream() -> int:
    # every op passes here exactly once before it launches: the place to bring the calling thread's seed base up to date
    if _SEED_BASE or getattr(_tls, "applied", 0):
        _lib.check(_lib.lib().bbbp_set_seed_base(_SEED_BASE or None), "bbbp_set_seed_base")
        _tls.applied = _SEED_BASE
    return torch.cuda.current_stream().cuda_stream


def _chk(t: torch.Tensor, name: str, dtype=torch.float32) -> torch.Tensor:
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"{name}: expected a torch.Tensor, got {type(t).__name__}")
    if not t.is_cuda:
        raise RuntimeError(f"{name}: expected a CUDA (HIP) tensor, got device {t.device}; the BBBP hot path has no CPU fallback")
    if t.dtype != dtype:
        raise RuntimeError(f"{name}: expected dtype {dtype}, got {t.dtype}")
    return t


def _p(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def _rowmajor_2d(t: torch.Tensor, name: str) -> Tuple[int, int, int]:
    """(rows, cols, ld) of a 2-D tensor whose last dim is contiguous (column slices allowed)."""
    if t.dim() != 2 or (t.shape[1] > 1 and t.stride(1) != 1):
        raise RuntimeError(f"{name}: expected a 2-D tensor with unit inner stride, got shape {tuple(t.shape)} strides {t.stride()}")
    ld = t.stride(0) if t.shape[0] > 1 else max(t.shape[1], t.stride(0))
    return t.shape[0], t.shape[1], ld


def _gemm_desc(a, b, trans_a, trans_b, alpha, bias, residual, act, out, gate, gate_scale, gate_after_residual=False, dropout_p=0.0,
               dropout_seed=0):
    """Validate one product and describe it as a bbbp_gemm_desc; returns (desc, out, split-K workspace bytes)."""
    _chk(a, "a"); _chk(b, "b")
    batched = a.dim() == 3
    if batched:
        if not (a.is_contiguous() and b.is_contiguous()):
            raise RuntimeError("batched gemm: operands must be contiguous")
        nb = a.shape[0]
        ar, ac = a.shape[1], a.shape[2]
        br, bc = b.shape[1], b.shape[2]
        lda, ldb = ac, bc
        sa, sb = ar * ac, br * bc
    else:
        nb = 1
        ar, ac, lda = _rowmajor_2d(a, "a")
        br, bc, ldb = _rowmajor_2d(b, "b")
        sa = sb = 0
    M, K = (ac, ar) if trans_a else (ar, ac)
    K2, N = (bc, br) if trans_b else (br, bc)
    if K != K2:
        raise RuntimeError(f"gemm: inner dimensions differ ({K} vs {K2})")
    if out is None:
        out = torch.empty((nb, M, N) if batched else (M, N), device=a.device, dtype=torch.float32)
    _chk(out, "out")
    ldc = N if batched else _rowmajor_2d(out, "out")[2]
    sc = M * N if batched else 0
    ldr, sr = 0, 0
    if residual is not None:
        _chk(residual, "residual")
        ldr = N if batched else _rowmajor_2d(residual, "residual")[2]
        sr = M * N if batched else 0
    ldg, sg = 0, 0
    if gate is not None:
        _chk(gate, "gate")
        if tuple(gate.shape) != tuple(out.shape):
            raise RuntimeError(f"gemm: gate shape {tuple(gate.shape)} != output shape {tuple(out.shape)}")
        ldg = N if batched else _rowmajor_2d(gate, "gate")[2]
        sg = M * N if batched else 0
    if bias is not None:
        _chk(bias, "bias")
        if bias.numel() != N:
            raise RuntimeError(f"gemm: bias has {bias.numel()} elements, expected {N}")
    d = _lib.GemmDesc(int(trans_a), int(trans_b), M, N, K, float(alpha), a.data_ptr(), lda, b.data_ptr(), ldb,
                      out.data_ptr(), ldc, _p(bias), _p(residual), ldr, ACT[act], _p(gate), ldg, float(gate_scale), nb,
                      sa, sb, sc, sr, sg, int(bool(gate_after_residual)), None, float(dropout_p), int(dropout_seed))
    return d, out, _lib.lib().bbbp_gemm_workspace_bytes(M, N, K, nb)


def gemm(a: torch.Tensor, b: torch.Tensor, *, trans_a: bool = False, trans_b: bool = False, alpha: float = 1.0,
         bias: Optional[torch.Tensor] = None, residual: Optional[torch.Tensor] = None, act=None,
         out: Optional[torch.Tensor] = None, gate: Optional[torch.Tensor] = None, gate_scale: float = 1.0,
         gate_after_residual: bool = False) -> torch.Tensor:
    """out = gate_mask(act(alpha * op(a) @ op(b) + bias)) + residual  (2-D; or 3-D batched with equal batch dims).

    gate: optional tensor of the output's shape; the result is multiplied by gate_scale where gate > 0, by 0 elsewhere
    (before the residual is added, or after it with gate_after_residual)."""
    if gate is not None:
        return gemm_grouped([dict(a=a, b=b, trans_a=trans_a, trans_b=trans_b, alpha=alpha, bias=bias, residual=residual,
                                  act=act, out=out, gate=gate, gate_scale=gate_scale, gate_after_residual=gate_after_residual)])[0]
    d, out, wsb = _gemm_desc(a, b, trans_a, trans_b, alpha, bias, residual, act, out, None, 1.0)
    ws = torch.empty(max(wsb, 1), dtype=torch.uint8, device=a.device)
    _lib.check(_lib.lib().bbbp_gemm_f32(_stream(), d.transA, d.transB, d.M, d.N, d.K, d.alpha, d.A, d.lda, d.B, d.ldb,
                                        d.C, d.ldc, d.bias, d.residual, d.ldr, d.act, d.batch, d.strideA, d.strideB,
                                        d.strideC, d.strideR, ws.data_ptr(), wsb), "bbbp_gemm_f32")
    return out


def gemm_grouped(problems) -> list:
    """Independent products (dicts of gemm()'s arguments); neighbours that are small enough share one launch."""
    keys = ("trans_a", "trans_b", "alpha", "bias", "residual", "act", "out", "gate", "gate_scale", "gate_after_residual", "dropout_p",
            "dropout_seed")
    defaults = dict(trans_a=False, trans_b=False, alpha=1.0, bias=None, residual=None, act=None, out=None, gate=None,
                    gate_scale=1.0, gate_after_residual=False, dropout_p=0.0, dropout_seed=0)
    descs, outs, wsb = [], [], 0
    for pr in problems:
        kw = {k: pr.get(k, defaults[k]) for k in keys}
        d, out, w = _gemm_desc(pr["a"], pr["b"], kw["trans_a"], kw["trans_b"], kw["alpha"], kw["bias"], kw["residual"],
                               kw["act"], kw["out"], kw["gate"], kw["gate_scale"], kw["gate_after_residual"], kw["dropout_p"],
                               kw["dropout_seed"])
        descs.append(d); outs.append(out); wsb = max(wsb, w)
    if not descs:
        return []
    arr = (_lib.GemmDesc * len(descs))(*descs)
    ws = torch.empty(max(wsb, 1), dtype=torch.uint8, device=outs[0].device)
    _lib.check(_lib.lib().bbbp_gemm_f32_grouped(_stream(), arr, len(descs), ws.data_ptr(), wsb), "bbbp_gemm_f32_grouped")
    return outs


def linear_weight_bias_grad(dy: torch.Tensor, x: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """dW = dy^T x and db = column sums of dy for a Linear (2-D).  When the product takes the small-GEMM path both come out of
    ONE launch: the bias gradient rides through the same MFMAs as a virtual all-ones column of x (bbbp_gemm_desc.asum)."""
    M, N = dy.shape
    K = x.shape[1]
    L = _lib.lib()
    if not L.bbbp_gemm_folds_asum(N, K, M, 1):
        return gemm(dy, x, trans_a=True), dy.sum(dim=0)
    d, dw, wsb = _gemm_desc(dy, x, True, False, 1.0, None, None, None, None, None, 1.0)
    db = torch.empty(N, device=dy.device, dtype=torch.float32)
    d.asum = db.data_ptr()
    arr = (_lib.GemmDesc * 1)(d)
    ws = torch.empty(max(wsb, 1), dtype=torch.uint8, device=dy.device)
    _lib.check(L.bbbp_gemm_f32_grouped(_stream(), arr, 1, ws.data_ptr(), wsb), "bbbp_gemm_f32_grouped")
    return dw, db


def linear(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor] = None, act=None,
           residual: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """F.linear (+activation, +residual) on the MFMA GEMM."""
    return gemm(x, weight, trans_b=True, bias=bias, act=act, residual=residual, out=out)


def conv3x3_relu_pool_fwd(x: torch.Tensor, w: torch.Tensor, b: torch.Tensor, keep_mask: bool = True) -> Tuple[torch.Tensor, Optional[torch.Tensor]]:
    """``keep_mask=False``: a forward-only call -- no pooling decisions are written (the C entry takes a null mask) and None is returned
    in their place."""
    _chk(x, "x"); _chk(w, "weight"); _chk(b, "bias")
    if not (x.is_contiguous() and w.is_contiguous() and b.is_contiguous()):
        raise RuntimeError("conv3x3_relu_pool: operands must be contiguous (NCHW)")
    B, cin, H, W = x.shape
    cout = w.shape[0]
    if tuple(w.shape) != (cout, cin, 3, 3):
        raise RuntimeError(f"conv3x3_relu_pool: weight shape {tuple(w.shape)} does not match input channels {cin}")
    y = torch.empty((B, cout, H // 2, W // 2), device=x.device, dtype=torch.float32)
    mask = torch.empty((B, cout, H // 2, W // 2), device=x.device, dtype=torch.uint8) if keep_mask else None
    L = _lib.lib()
    wsb = L.bbbp_conv3x3_workspace_bytes(B, cin, cout, H, W)
    ws = torch.empty(wsb, dtype=torch.uint8, device=x.device)
    _lib.check(L.bbbp_conv3x3_relu_pool_fwd(_stream(), x.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(),
                                            mask.data_ptr() if keep_mask else None, B, cin, cout, H, W, ws.data_ptr(), wsb),
               "bbbp_conv3x3_relu_pool_fwd")
    return y, mask


def conv3x3_relu_pool_bwd_data(gy: torch.Tensor, mask: torch.Tensor, w: torch.Tensor) -> torch.Tensor:
    _chk(gy, "gy"); _chk(mask, "mask", torch.uint8); _chk(w, "weight")
    gy = gy.contiguous()
    B, cout, Hp, Wp = gy.shape
    cin = w.shape[1]
    H, W = 2 * Hp, 2 * Wp
    dx = torch.empty((B, cin, H, W), device=gy.device, dtype=torch.float32)
    L = _lib.lib()
    wsb = L.bbbp_conv3x3_workspace_bytes(B, cin, cout, H, W)
    ws = torch.empty(wsb, dtype=torch.uint8, device=gy.device)
    _lib.check(L.bbbp_conv3x3_relu_pool_bwd_data(_stream(), gy.data_ptr(), mask.data_ptr(), w.data_ptr(), dx.data_ptr(),
                                                 B, cin, cout, H, W, ws.data_ptr(), wsb), "bbbp_conv3x3_relu_pool_bwd_data")
    return dx


def conv3x3_relu_pool_bwd_weight(x: torch.Tensor, gy: torch.Tensor, mask: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    _chk(x, "x"); _chk(gy, "gy"); _chk(mask, "mask", torch.uint8)
    gy = gy.contiguous()
    B, cin, H, W = x.shape
    cout = gy.shape[1]
    dw = torch.empty((cout, cin, 3, 3), device=x.device, dtype=torch.float32)
    db = torch.empty((cout,), device=x.device, dtype=torch.float32)
    L = _lib.lib()
    wsb = L.bbbp_conv3x3_workspace_bytes(B, cin, cout, H, W)
    ws = torch.empty(wsb, dtype=torch.uint8, device=x.device)
    _lib.check(L.bbbp_conv3x3_relu_pool_bwd_weight(_stream(), x.data_ptr(), gy.data_ptr(), mask.data_ptr(), dw.data_ptr(),
                                                   db.data_ptr(), B, cin, cout, H, W, ws.data_ptr(), wsb),
               "bbbp_conv3x3_relu_pool_bwd_weight")
    return dw, db


def layernorm_fwd(x: torch.Tensor, residual: Optional[torch.Tensor], gamma: torch.Tensor, beta: torch.Tensor,
                  eps: float = 1e-5, dropout_p: float = 0.0, seed: int = 0):
    """Returns (y, z, mean, rstd) with z = dropout(x) + residual (x is NOT modified: it is cloned)."""
    _chk(x, "x")
    z = x.contiguous().clone()
    rows, cols = z.shape
    y = torch.empty_like(z)
    mean = torch.empty(rows, device=x.device, dtype=torch.float32)
    rstd = torch.empty(rows, device=x.device, dtype=torch.float32)
    _lib.check(_lib.lib().bbbp_layernorm_fwd(_stream(), z.data_ptr(), _p(residual), y.data_ptr(), gamma.data_ptr(),
                                             beta.data_ptr(), mean.data_ptr(), rstd.data_ptr(), rows, cols, eps,
                                             dropout_p, seed), "bbbp_layernorm_fwd")
    return y, z, mean, rstd


def linear_layernorm_fwd(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor], residual: Optional[torch.Tensor],
                         gamma: torch.Tensor, beta: torch.Tensor, eps: float = 1e-5, dropout_p: float = 0.0, seed: int = 0):
    """y = LayerNorm(dropout(x @ weight.T + bias) + residual) in ONE launch (narrow outputs: weight [N <= 256, K]).
    Returns (y, z, mean, rstd) like ``layernorm_fwd``."""
    _chk(x, "x"); _chk(weight, "weight")
    x = x.contiguous(); weight = weight.contiguous()
    M, K = x.shape
    N = weight.shape[0]
    if weight.shape[1] != K:
        raise RuntimeError("linear_layernorm_fwd: inner dimensions differ")
    z = torch.empty((M, N), device=x.device, dtype=torch.float32)
    y = torch.empty_like(z)
    mean = torch.empty(M, device=x.device, dtype=torch.float32)
    rstd = torch.empty(M, device=x.device, dtype=torch.float32)
    res = residual.contiguous() if residual is not None else None
    _lib.check(_lib.lib().bbbp_linear_layernorm_fwd(_stream(), x.data_ptr(), K, weight.data_ptr(), _p(bias), _p(res), N, z.data_ptr(), N,
                                                    y.data_ptr(), N, gamma.data_ptr(), beta.data_ptr(), mean.data_ptr(), rstd.data_ptr(),
                                                    M, N, K, eps, dropout_p, seed), "bbbp_linear_layernorm_fwd")
    return y, z, mean, rstd


def layernorm_linear_fwd(z: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor],
                         act: int = 0, eps: float = 1e-5, dropout_p: float = 0.0, seed: int = 0):
    """out = dropout(act(LayerNorm(z) @ weight.T + bias)) with the LayerNorm absorbed by the product (ONE launch; K = z.shape[1] <= 192).
    Returns (out [M, N], y = LayerNorm(z) [M, K], mean [M], rstd [M])."""
    _chk(z, "z"); _chk(weight, "weight"); _chk(gamma, "gamma"); _chk(beta, "beta")
    z = z.contiguous(); weight = weight.contiguous()
    M, K = z.shape
    N = weight.shape[0]
    if weight.shape[1] != K or gamma.numel() != K or beta.numel() != K:
        raise RuntimeError("layernorm_linear_fwd: inner dimensions differ")
    L = _lib.lib()
    if not L.bbbp_layernorm_linear_supported(M, N, K):
        raise RuntimeError(f"layernorm_linear_fwd: shape M={M} N={N} K={K} is not supported (K <= 192)")
    out = torch.empty((M, N), device=z.device, dtype=torch.float32)
    y = torch.empty_like(z)
    mean = torch.empty(M, device=z.device, dtype=torch.float32)
    rstd = torch.empty(M, device=z.device, dtype=torch.float32)
    _lib.check(L.bbbp_layernorm_linear_fwd(_stream(), z.data_ptr(), K, gamma.data_ptr(), beta.data_ptr(), eps, weight.data_ptr(), _p(bias),
                                           out.data_ptr(), N, act, dropout_p, seed, y.data_ptr(), K, mean.data_ptr(), rstd.data_ptr(), M, N, K),
               "bbbp_layernorm_linear_fwd")
    return out, y, mean, rstd


def layernorm_bwd(dy, z, gamma, mean, rstd, dropout_p: float = 0.0, seed: int = 0):
    """Returns (dz, dx, dgamma, dbeta); dx is dz when dropout_p == 0."""
    rows, cols = z.shape
    dz = torch.empty_like(z)
    dx = torch.empty_like(z) if dropout_p > 0 else None
    dg = torch.empty(cols, device=z.device, dtype=torch.float32)
    db = torch.empty(cols, device=z.device, dtype=torch.float32)
    _lib.check(_lib.lib().bbbp_layernorm_bwd(_stream(), dy.contiguous().data_ptr(), z.data_ptr(), gamma.data_ptr(),
                                             mean.data_ptr(), rstd.data_ptr(), dz.data_ptr(), _p(dx), dg.data_ptr(),
                                             db.data_ptr(), rows, cols, dropout_p, seed), "bbbp_layernorm_bwd")
    return dz, (dx if dx is not None else dz), dg, db


def softmax_fwd(x: torch.Tensor, dropout_p: float = 0.0, seed: int = 0):
    """Row softmax over the last dim of a contiguous tensor. Returns (prob, dropped_prob)."""
    p = _chk(x, "x").contiguous().clone()
    cols = p.shape[-1]
    rows = p.numel() // cols
    pd = torch.empty_like(p) if dropout_p > 0 else None
    _lib.check(_lib.lib().bbbp_softmax_fwd(_stream(), p.data_ptr(), _p(pd), rows, cols, dropout_p, seed), "bbbp_softmax_fwd")
    return p, (pd if pd is not None else p)


def softmax_bwd(dprob: torch.Tensor, prob: torch.Tensor, dropout_p: float = 0.0, seed: int = 0) -> torch.Tensor:
    d = _chk(dprob, "dprob").contiguous().clone()
    cols = d.shape[-1]
    _lib.check(_lib.lib().bbbp_softmax_bwd(_stream(), d.data_ptr(), prob.data_ptr(), d.numel() // cols, cols, dropout_p, seed),
               "bbbp_softmax_bwd")
    return d


def dropout(x: torch.Tensor, p: float, seed: int) -> torch.Tensor:
    x = _chk(x, "x").contiguous()
    y = torch.empty_like(x)
    _lib.check(_lib.lib().bbbp_dropout(_stream(), x.data_ptr(), y.data_ptr(), x.numel(), p, seed), "bbbp_dropout")
    return y


def batchnorm1d_fwd(x, gamma, beta, running_mean, running_var, training: bool, eps: float = 1e-5, momentum: float = 0.1):
    """Returns (y, save_mean, save_rstd); running stats are updated in place when training."""
    x = _chk(x, "x").contiguous()
    rows, cols = x.shape
    y = torch.empty_like(x)
    sm = torch.empty(cols, device=x.device, dtype=torch.float32)
    sr = torch.empty(cols, device=x.device, dtype=torch.float32)
    _lib.check(_lib.lib().bbbp_batchnorm1d_fwd(_stream(), x.data_ptr(), y.data_ptr(), gamma.data_ptr(), beta.data_ptr(),
                                               running_mean.data_ptr(), running_var.data_ptr(), sm.data_ptr(), sr.data_ptr(),
                                               rows, cols, eps, momentum, int(training)), "bbbp_batchnorm1d_fwd")
    return y, sm, sr


def batchnorm1d_bwd(dy, x, gamma, save_mean, save_rstd, training: bool):
    x = x.contiguous(); dy = dy.contiguous()
    rows, cols = x.shape
    dx = torch.empty_like(x)
    dg = torch.empty(cols, device=x.device, dtype=torch.float32)
    db = torch.empty(cols, device=x.device, dtype=torch.float32)
    _lib.check(_lib.lib().bbbp_batchnorm1d_bwd(_stream(), dy.data_ptr(), x.data_ptr(), gamma.data_ptr(), save_mean.data_ptr(),
                                               save_rstd.data_ptr(), dx.data_ptr(), dg.data_ptr(), db.data_ptr(), rows, cols,
                                               int(training)), "bbbp_batchnorm1d_bwd")
    return dx, dg, db


def bias_act_bwd(dy: torch.Tensor, y: Optional[torch.Tensor], act=None, scale: float = 1.0):
    """In place: dy *= act'(y) * scale.  Returns the column sums (bias gradient)."""
    rows, cols, lddy = _rowmajor_2d(_chk(dy, "dy"), "dy")
    ldy = 0
    if y is not None:
        ldy = _rowmajor_2d(_chk(y, "y"), "y")[2]
    db = torch.empty(cols, device=dy.device, dtype=torch.float32)
    _lib.check(_lib.lib().bbbp_bias_act_bwd(_stream(), dy.data_ptr(), lddy, _p(y), ldy, db.data_ptr(), rows, cols, ACT[act],
                                            scale), "bbbp_bias_act_bwd")
    return db


def mse(pred: torch.Tensor, target: torch.Tensor, grad_scale: float = 1.0):
    """Returns (loss[1], dpred) for loss = mean((pred - target)^2)."""
    pred = _chk(pred, "pred").contiguous().view(-1)
    target = _chk(target, "target").contiguous().view(-1)
    loss = torch.empty(1, device=pred.device, dtype=torch.float32)
    dpred = torch.empty_like(pred)
    _lib.check(_lib.lib().bbbp_mse(_stream(), pred.data_ptr(), target.data_ptr(), loss.data_ptr(), dpred.data_ptr(),
                                   pred.numel(), grad_scale), "bbbp_mse")
    return loss, dpred


def adamw_step_(param, grad, exp_avg, exp_avg_sq, step: int, lr=1e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-5,
                grad_scale: float = 1.0) -> None:
    for t, n in ((param, "param"), (grad, "grad"), (exp_avg, "exp_avg"), (exp_avg_sq, "exp_avg_sq")):
        _chk(t, n)
        if not t.is_contiguous():
            raise RuntimeError(f"adamw: {n} must be contiguous")
    _lib.check(_lib.lib().bbbp_adamw_step(_stream(), param.data_ptr(), grad.data_ptr(), exp_avg.data_ptr(),
                                          exp_avg_sq.data_ptr(), param.numel(), lr, betas[0], betas[1], eps, weight_decay,
                                          step, grad_scale), "bbbp_adamw_step")


def adamw_step_deferred_(param, grad, exp_avg, exp_avg_sq, step: int, lo: int, hi: int, lr=1e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-5,
                         grad_scale: float = 1.0) -> None:
    """``adamw_step_`` with elements [lo, hi) of the flat buffers updated on a library-owned side stream (``bbbp_adamw_step_deferred``):
    the next ``MixedInputModel`` forward waits for that slice right before it reads it; ``param_sync()`` orders anything else."""
    for t, n in ((param, "param"), (grad, "grad"), (exp_avg, "exp_avg"), (exp_avg_sq, "exp_avg_sq")):
        _chk(t, n)
        if not t.is_contiguous():
            raise RuntimeError(f"adamw: {n} must be contiguous")
    _lib.check(_lib.lib().bbbp_adamw_step_deferred(_stream(), param.data_ptr(), grad.data_ptr(), exp_avg.data_ptr(), exp_avg_sq.data_ptr(),
                                                   param.numel(), lo, hi, lr, betas[0], betas[1], eps, weight_decay, step, grad_scale),
               "bbbp_adamw_step_deferred")
    # torch's caching allocator is stream-ordered: the gradient buffer is usually freed (zero_grad(set_to_none=True)) while the side stream may
    # still read it -- tell the allocator, or the block could be handed to the next allocation on the current stream too early
    h = _lib.lib().bbbp_param_stream()
    if h:
        side = torch.cuda.ExternalStream(int(h), device=param.device)
        grad.record_stream(side)                 # (parameters and moments live as long as the optimizer)


def adamw_hyper_store_(hyper, step: int, lr=1e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-5, grad_scale: float = 1.0) -> None:
    """Store the eight derived floats the AdamW kernels compute with into the 8-float device tensor ``hyper``, in stream order
    (``bbbp_adamw_hyper_store``)."""
    _chk(hyper, "hyper")
    if hyper.numel() != 8 or not hyper.is_contiguous():
        raise RuntimeError("adamw_hyper_store: hyper must be 8 contiguous float32 values")
    _lib.check(_lib.lib().bbbp_adamw_hyper_store(_stream(), hyper.data_ptr(), lr, betas[0], betas[1], eps, weight_decay, step, grad_scale),
               "bbbp_adamw_hyper_store")


def adamw_step_multi_(param, exp_avg, exp_avg_sq, table, n_tensors: int, step: int, lr=1e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-5,
                      grad_scale: float = 1.0, hyper=None) -> None:
    """One launch over a flat parameter / moment buffer whose gradients are ``n_tensors`` separate tensors (``bbbp_adamw_step_multi``).
    ``table``: int64 device tensor, ``n_tensors + 1`` element offsets followed by ``n_tensors`` gradient addresses.  ``hyper``: optional
    8-float device tensor (``adamw_hyper_store_``) read by the kernel instead of the scalar arguments."""
    for t, n in ((param, "param"), (exp_avg, "exp_avg"), (exp_avg_sq, "exp_avg_sq")):
        _chk(t, n)
        if not t.is_contiguous():
            raise RuntimeError(f"adamw: {n} must be contiguous")
    if table.dtype != torch.int64 or not table.is_cuda or table.numel() != 2 * n_tensors + 1 or not table.is_contiguous():
        raise RuntimeError("adamw_multi: table must be a contiguous int64 CUDA tensor of 2 * n_tensors + 1 entries")
    if hyper is not None and (hyper.dtype != torch.float32 or not hyper.is_cuda or hyper.numel() != 8):
        raise RuntimeError("adamw_multi: hyper must be 8 float32 values on the device")
    _lib.check(_lib.lib().bbbp_adamw_step_multi(_stream(), param.data_ptr(), exp_avg.data_ptr(), exp_avg_sq.data_ptr(), param.numel(),
                                                table.data_ptr(), n_tensors, lr, betas[0], betas[1], eps, weight_decay, step, grad_scale,
                                                hyper.data_ptr() if hyper is not None else None), "bbbp_adamw_step_multi")


def param_sync() -> None:
    """Order the current stream behind a deferred optimizer slice (no-op when none is pending)."""
    _lib.check(_lib.lib().bbbp_param_sync(_stream()), "bbbp_param_sync")
