"""GPU: out_proj folded into the value projection of a one-head encoder layer (csrc/fold.hip, bbbp_set_fold_outproj; the default schedule
for F = 167) against the reference's operation order (in_proj -> attention -> out_proj, nn.TransformerEncoderLayer as built at
...20250113.py:75-78) and against the float64 oracle."""
import pytest
import torch

import bbbp_amd
from oracle import reference_cpu as oracle
from helpers import assert_close, synth_inputs

pytestmark = pytest.mark.gpu


def build(F, seed, dev):
    torch.manual_seed(seed)
    return bbbp_amd.MixedInputModel(F, 128).to(dev)


@pytest.mark.parametrize("F,B,training", [(167, 37, True), (167, 7, True), (173, 21, True), (167, 512, True), (167, 130, False), (167, 1, False)])
def test_folded_out_proj_matches_the_reference_operation_order(dev, F, B, training):
    """Same workspace contents otherwise, same Philox streams (the out_proj output's dropout draws the same elements): outputs and every
    gradient -- in particular out_proj.weight / bias and the V rows of in_proj, which the folded plan unfolds from dW' | db' -- agree to the
    rounding of the reassociated products.  173: a second prime width (nhead 1); 7 / 37: ragged tiles; 130 / 1: eval plans."""
    from bbbp_amd import _lib
    L = _lib.lib()
    assert bbbp_amd.models.reference_nhead(F) == 1
    fp, img, y = synth_inputs(2900 + B, max(B, 2), F, 49152)
    fp, img, y = fp[:B], img[:B], y[:B]
    m = build(F, 41, dev).train(training)
    res = []
    for fold in (0, 1):
        old = L.bbbp_set_fold_outproj(fold)
        try:
            m.zero_grad(set_to_none=True)
            m.fc[2].running_mean.zero_(); m.fc[2].running_var.fill_(1.0)
            torch.manual_seed(79)                  # same dropout seeds in both passes
            if training:
                out = m(fp.to(dev), img.to(dev))
                bbbp_amd.MSELoss()(out.reshape(-1), y.to(dev)).backward()
                grads = {k: p.grad.detach().cpu().double() for k, p in m.named_parameters()}
            else:
                with torch.no_grad():
                    out = m(fp.to(dev), img.to(dev))
                grads = {}
            torch.cuda.synchronize()
            res.append((out.detach().cpu().double(), grads))
        finally:
            L.bbbp_set_fold_outproj(old)
    (o0, g0), (o1, g1) = res
    assert torch.isfinite(o1).all()
    assert float((o0 - o1).abs().max()) <= 2e-5 * float(o0.abs().max()) + 1e-7, float((o0 - o1).abs().max())
    assert set(g0) == set(g1)
    for k in g0:
        if k.startswith("attention_fusion."):
            continue
        assert torch.isfinite(g1[k]).all(), k
        err = float((g0[k] - g1[k]).abs().max())
        if B <= 64:
            assert err <= 1e-4 * float(g0[k].abs().max()) + 1e-12, (k, err, float(g0[k].abs().max()))
        elif err > 1e-4 * float(g0[k].abs().max()) + 1e-12:
            # B = 512: an occasional ReLU / dropout gate at a pre-activation within rounding of zero falls differently (test_gpu_model.py)
            rel = float((g0[k] - g1[k]).norm() / g0[k].norm().clamp_min(1e-30))
            assert rel <= 2e-3 and err <= 5e-2 * float(g0[k].abs().max()), (k, rel, err)


def test_fold_under_the_bf16_attention_kernel_of_screening_batches(dev):
    """Bit 1 of the mask (opt-in): forward-only plans of 2048+ rows run csrc/attention_b3.hip, which then reads VW where it read V and whose
    output is the out_proj output itself (softmax rows sum to one, so bo rides in b').  B = 2500: ragged query block and key tile."""
    from bbbp_amd import _lib
    L = _lib.lib()
    m = build(167, 47, dev).eval()
    fp, img, _ = synth_inputs(5400, 2500, 167, 49152)
    outs = []
    for fold in (0, 3):
        old = L.bbbp_set_fold_outproj(fold)
        try:
            with torch.no_grad():
                outs.append(m(fp.to(dev), img.to(dev)).cpu().double())
        finally:
            L.bbbp_set_fold_outproj(old)
    assert torch.isfinite(outs[1]).all()
    assert float((outs[0] - outs[1]).abs().max()) <= 2e-5 * float(outs[0].abs().max()) + 1e-7


@pytest.mark.parametrize("fold", [0, 1])
def test_attention_block_gradients_against_float64_oracle(dev, fold):
    """Dropout off, B = 24: out_proj / in_proj weights and biases of every layer (the tensors the fold touches) and the output against the
    float64 oracle, both schedules at the same tolerance."""
    from bbbp_amd import _lib
    L = _lib.lib()
    F, B = 167, 24
    m = build(F, 43, dev)
    for mod in m.modules():
        if isinstance(mod, torch.nn.Dropout):
            mod.p = 0.0
        if isinstance(mod, torch.nn.MultiheadAttention):
            mod.dropout = 0.0
    m.train()
    fp, img, y = synth_inputs(3100, B, F, 49152)
    p = {k: (v.detach().cpu().double() if v.dtype.is_floating_point else v.detach().cpu()).clone()
         .requires_grad_(v.dtype.is_floating_point and "running" not in k) for k, v in m.state_dict().items()}
    ref = oracle.mixed_input_forward(p, fp.double(), img.double(), training=True, bn_state={})
    oracle.mse_loss(ref, y.double()).backward()
    old = L.bbbp_set_fold_outproj(fold)
    try:
        out = m(fp.to(dev), img.to(dev))
        bbbp_amd.MSELoss()(out.reshape(-1), y.to(dev)).backward()
        torch.cuda.synchronize()
    finally:
        L.bbbp_set_fold_outproj(old)
    assert_close(out.detach().cpu().numpy(), ref.detach().numpy(), rtol=1e-4, atol_frac=2e-5, what="output")
    for k, q in m.named_parameters():
        if ".self_attn." in k:
            assert_close(q.grad.cpu().numpy(), p[k].grad.numpy(), rtol=1e-4, atol_frac=5e-5, what=f"fold={fold} {k}")
