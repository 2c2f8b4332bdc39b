"""Round 4 engine features against the schedules they replace (GPU): the optimizer slice deferred to a side stream, the LayerNorm-absorbing
GEMMs inside the whole-model forward, the software-pipelined conv1 forward inside training plans."""
import os

import numpy as np
import pytest
import torch

import bbbp_amd
from bbbp_amd import _lib, ops
from bbbp_amd.optim import AdamW
from helpers import synth_inputs

pytestmark = pytest.mark.gpu


@pytest.fixture
def dev():
    return torch.device("cuda:0")


def _steps(dev, defer, steps=6, B=48, F=167, seed=3, read_between=False):
    torch.manual_seed(seed)
    model = bbbp_amd.MixedInputModel(F, 128).to(dev).train()
    params = list(model.parameters())
    opt = AdamW(params, lr=1e-3, weight_decay=1e-5, defer=model.image_cnn[7].weight if defer else None)
    fp, img, y = (t.to(dev) for t in synth_inputs(11, 2 * B, F, 49152))
    crit = bbbp_amd.MSELoss()
    torch.manual_seed(99)                                   # the dropout seeds of both runs
    losses = []
    for i in range(steps):
        s = (i % 2) * B
        loss = crit(model(fp[s:s + B], img[s:s + B]).squeeze(), y[s:s + B])
        loss.backward()
        opt.step()
        opt.zero_grad(set_to_none=True)
        if read_between:
            # a reader outside the library between two steps: allocations on the current stream that could take the freed gradient block,
            # and a read of the deferred tensor after synchronize()
            junk = torch.randn(B, 49152, device=dev)
            opt.synchronize()
            losses.append(float(model.image_cnn[7].weight.abs().sum()) + 0.0 * float(junk[0, 0]))
        else:
            losses.append(float(loss.detach()))
    opt.synchronize()
    torch.cuda.synchronize()
    st = opt.state[model.image_cnn[7].weight]
    return (torch.cat([p.detach().flatten() for p in params]).cpu(), st["exp_avg"].flatten().cpu().clone(), st["exp_avg_sq"].flatten().cpu().clone(),
            int(st["step"]), losses)


@pytest.mark.parametrize("read_between", [False, True])
def test_deferred_image_fc_slice_is_bit_identical_to_the_one_launch_step(dev, read_between):
    """optim.AdamW(defer=image-FC weight) / bbbp_adamw_step_deferred: 62 % of the optimizer's bytes updated on a side stream beside the next
    forward pass, which waits for the slice right before its image FC.  Parameters, both moments and every loss after six steps (dropout on)
    are bit-identical to the one-launch step's -- a forward pass that read the weight too early, or a gradient buffer recycled under the
    side stream, would change them; with `read_between` the test also reads the tensor between steps through synchronize()."""
    a = _steps(dev, defer=False, read_between=read_between)
    b = _steps(dev, defer=True, read_between=read_between)
    assert a[3] == b[3] == 6
    assert a[4] == b[4]
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and torch.equal(a[2], b[2])


def test_deferred_step_then_eval_forward_and_state_dict_see_the_update(dev):
    """After a deferred step, an eval-mode forward of the SAME model and of ANOTHER model, state_dict() of the optimizer and a plain
    adamw_step_ are all ordered behind the slice (bbbp_param_wait at the library's entry points)."""
    torch.manual_seed(1)
    model = bbbp_amd.MixedInputModel(64, 128).to(dev).train()
    other = bbbp_amd.MixedInputModel(64, 128).to(dev).eval()
    w = model.image_cnn[7].weight
    opt = AdamW(model.parameters(), lr=1e-2, defer=w)
    ref = AdamW([torch.nn.Parameter(p.detach().clone()) for p in model.parameters()], lr=1e-2)
    fp, img, y = (t.to(dev) for t in synth_inputs(5, 16, 64, 49152))
    bbbp_amd.MSELoss()(model(fp, img).squeeze(), y).backward()
    for q, p in zip(ref.param_groups[0]["params"], model.parameters()):
        q.grad = p.grad.detach().clone()
    ref.step()
    opt.step()
    with torch.no_grad():
        other(fp, img)                                      # another model's forward: waits at its start
        model.eval()
        out = model(fp, img)                                # this model's forward: waits before the image FC
    sd = opt.state_dict()
    assert sd["state"]
    torch.cuda.synchronize()
    want = [q.detach() for q in ref.param_groups[0]["params"]]
    for p, q in zip(model.parameters(), want):
        assert torch.equal(p.detach(), q)
    assert torch.isfinite(out).all()


def test_layernorm_absorbing_engine_matches_the_standalone_layernorm_schedule(dev):
    """bbbp_set_ln_absorb(1): norm1 -> linear1 and norm2 -> the next in_proj / fingerprint_fc absorbed by the consuming GEMM, the dropout +
    residual in the producing GEMM's epilogue (same Philox elements).  Outputs, loss and every gradient of one training step with dropout ON
    agree with the default schedule to rounding, at the folded one-head width (167) and a multi-head width (64: the out_proj epilogue)."""
    L = _lib.lib()
    for F, B in ((167, 96), (64, 40)):
        fp, img, y = (t.to(dev) for t in synth_inputs(21, B, F, 49152))
        res = []
        for mode in (0, 1):
            old = L.bbbp_set_ln_absorb(mode)
            try:
                torch.manual_seed(7)
                model = bbbp_amd.MixedInputModel(F, 128).to(dev).train()
                torch.manual_seed(123)
                out = model(fp, img)
                loss = bbbp_amd.MSELoss()(out.squeeze(), y)
                loss.backward()
                res.append((out.detach().clone(), float(loss), {n: p.grad.detach().clone() for n, p in model.named_parameters()}))
            finally:
                L.bbbp_set_ln_absorb(old)
        (o0, l0, g0), (o1, l1, g1) = res
        assert float((o0 - o1).abs().max()) <= 2e-5 * float(o0.abs().max()) + 1e-6
        assert abs(l0 - l1) <= 1e-5 * abs(l0)
        for n in g0:
            if n.startswith("attention_fusion."):
                continue                                    # exact gradient 0: rounding noise in any implementation
            scale = float(g0[n].abs().max())
            assert float((g0[n] - g1[n]).abs().max()) <= 2e-4 * scale + 1e-9, (F, n, float((g0[n] - g1[n]).abs().max()), scale)


def test_pipelined_conv1_forward_in_training_plans_matches_f32_kernel(dev):
    """BBBP_C1_TRAIN (default on since round 4): the software-pipelined split-bf16 conv1 forward inside a training step, one work-group per
    CU.  Against the same step with conv mask bit 6 cleared (the f32 kernel of rounds 1-3): pooled activations agree to 2 ulp-ish, so output,
    loss and every gradient agree to rounding; the pooling decisions differ only at near-ties."""
    L = _lib.lib()
    F, B = 167, 64
    fp, img, y = (t.to(dev) for t in synth_inputs(33, B, F, 49152))
    res = []
    old = L.bbbp_get_conv_winograd()
    for mask in (old & ~64, old | 64):
        L.bbbp_set_conv_winograd(mask)
        try:
            torch.manual_seed(7)
            model = bbbp_amd.MixedInputModel(F, 128).to(dev).train()
            model.keep_workspace = True
            torch.manual_seed(5)
            out = model(fp, img)
            loss = bbbp_amd.MSELoss()(out.squeeze(), y)
            loss.backward()
            res.append((out.detach().clone(), float(loss), {n: p.grad.detach().clone() for n, p in model.named_parameters()}, model.debug_pool_masks()[0].clone()))
        finally:
            L.bbbp_set_conv_winograd(old)
    (o0, l0, g0, m0), (o1, l1, g1, m1) = res
    assert float((m0 != m1).float().mean()) < 1e-4
    assert float((o0 - o1).abs().max()) <= 2e-5 * float(o0.abs().max()) + 1e-6
    for n in g0:
        if n.startswith("attention_fusion."):
            continue
        scale = float(g0[n].abs().max())
        tol = 1e-2 if n.startswith("image_cnn.") else 5e-4           # the conv gradients: 10^6-term sums behind pooling decisions that flip at near-ties
        assert float((g0[n] - g1[n]).abs().max()) <= tol * scale + 1e-9, (n, float((g0[n] - g1[n]).abs().max()), scale)
