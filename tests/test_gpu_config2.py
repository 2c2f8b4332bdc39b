"""GPU: BASELINE config 2 -- the two-branch MACCS-Linear + image-CNN + torch.cat + BatchNorm-head model -- and the reference
class it is cut from (Descriptors/multi_input_data_regression_opt_round_2_transformer_cnn.py:45-102, torch.cat fusion), both
on the fused engine (bbbp_mixed_desc.fusion = 1, num_layers = 0 / 6), against the goldens of the reference class."""
import pytest
import torch

import bbbp_amd
from oracle import reference_cpu as oracle
from helpers import assert_close, assert_close_or_as_accurate_as_fp32, check_param_checksums, check_summary, check_summary_adam, golden, synth_inputs
from test_gpu_model import grad_atol, zero_dropout

pytestmark = pytest.mark.gpu
SEED = 20250102


def two_branch(dev):
    torch.manual_seed(SEED)
    full = bbbp_amd.ConcatMixedInputModel(167, 128)           # the reference class's seeded init stream
    m = bbbp_amd.TwoBranchConcatModel(167, 128)
    m.load_state_dict({k: v for k, v in full.state_dict().items() if not k.startswith("fingerprint_transformer.")}, strict=True)
    return m.to(dev)


def test_two_branch_eval_outputs(dev):
    g = golden("two_branch_f167")
    m = two_branch(dev).eval()
    check_param_checksums(g, {k: v.cpu() for k, v in m.state_dict().items()})
    for B in (2, 7):
        fp, img, _ = synth_inputs(1000 + B, B, 167, 49152)
        with torch.no_grad():
            out = m(fp.to(dev), img.to(dev))
        assert_close(out.cpu().numpy(), g[f"eval/B{B}/out"], rtol=1e-4, atol_frac=2e-5, what=f"config 2 eval B{B}")


@pytest.mark.parametrize("B", [7, 256])
def test_two_branch_train_step(dev, B):
    """fwd + MSE + bwd in train mode at B = 7 and at the config's own batch 256: output, loss, BatchNorm running statistics
    and gradient summaries against the reference class; at B = 7 additionally every gradient element vs the float64 oracle."""
    g = golden("two_branch_f167")
    m = two_branch(dev).train()
    fp, img, y = synth_inputs(1000 + B, B, 167, 49152)
    out = m(fp.to(dev), img.to(dev))
    loss = bbbp_amd.MSELoss()(out.squeeze(), y.to(dev))
    loss.backward()
    assert_close(out.detach().cpu().numpy(), g[f"train/B{B}/out"], rtol=1e-4, atol_frac=2e-5, what="train out")
    assert abs(float(loss.detach()) - float(g[f"train/B{B}/loss"])) <= 1e-4 * abs(float(g[f"train/B{B}/loss"]))
    sd = m.state_dict()
    for k in ("fc.2.running_mean", "fc.2.running_var"):
        assert_close(sd[k].cpu().numpy(), g[f"train/B{B}/bn/{k}"], rtol=1e-4, what=k)
    assert int(sd["fc.2.num_batches_tracked"]) == 1
    for k, q in m.named_parameters():
        check_summary(g, f"train/B{B}/{k}", q.grad, rtol=5e-4, atol_frac=grad_atol(k))
    refs = []
    for cast in (torch.Tensor.double, torch.Tensor.float):
        p = {k: (cast(v.detach().cpu()) if v.dtype.is_floating_point else v.detach().cpu()).clone()
             .requires_grad_(v.dtype.is_floating_point and "running" not in k) for k, v in two_branch(dev).state_dict().items()}
        oracle.mse_loss(oracle.mixed_input_forward(p, cast(fp), cast(img), training=True, bn_state={}, num_layers=0, fusion="concat"),
                        cast(y)).backward()
        refs.append(p)
    for k, q in m.named_parameters():
        if k.startswith(("image_cnn.0.", "image_cnn.3.")):      # ill-conditioned float32 sums: see the helper
            assert_close_or_as_accurate_as_fp32(q.grad.cpu().numpy(), refs[0][k].grad.numpy(), refs[1][k].grad.numpy(), what=k)
        else:
            assert_close(q.grad.cpu().numpy(), refs[0][k].grad.numpy(), rtol=1e-4, atol_frac=5e-5, what=k)


def test_two_branch_adamw_steps(dev):
    from bbbp_amd.optim import AdamW
    g = golden("two_branch_f167")
    m = two_branch(dev).train()
    opt = AdamW(m.parameters(), lr=1e-4, weight_decay=1e-5)
    fp, img, y = (t.to(dev) for t in synth_inputs(1007, 7, 167, 49152))
    for step in range(1, 4):
        opt.zero_grad(set_to_none=True)
        loss = bbbp_amd.MSELoss()(m(fp, img).squeeze(), y)
        loss.backward()
        opt.step()
        want = float(g[f"adamw/B7/step{step}/loss"])
        assert abs(float(loss.detach()) - want) <= (1e-4 if step == 1 else 3e-3) * abs(want)
        if step in (1, 3):
            for k, q in m.named_parameters():
                if k != "fc.0.bias":
                    check_summary_adam(g, f"adamw/B7/step{step}/{k}", q, lr=1e-4, steps=step,
                                       tight_lr_frac=0.02 if step == 1 else 0.6, min_frac=0.9 if step == 1 else 0.75)


def test_two_branch_inference_workspace_and_errors(dev):
    m = two_branch(dev)
    fp, img, _ = synth_inputs(3, 5, 167, 49152)
    m.eval()
    with torch.no_grad():
        a = m(fp.to(dev), img.to(dev))                 # inference plan (no backward temporaries)
    b = m(fp.to(dev), img.to(dev))                     # grad-enabled eval call: full plan
    assert torch.equal(a, b.detach())
    b.sum().backward()
    assert all(torch.isfinite(q.grad).all() for q in m.parameters())
    with pytest.raises(RuntimeError):
        m.double()(fp.to(dev), img.to(dev))             # float32-only kernels: refuse instead of reading past buffers


def test_concat_fusion_reference_class(dev):
    """The full round-2 class (6-layer encoder + CNN + torch.cat + head) on the fused engine."""
    g = golden("concat_f167")
    torch.manual_seed(SEED)
    m = bbbp_amd.ConcatMixedInputModel(167, 128).to(dev)
    zero_dropout(m)
    m.eval()
    fp, img, _ = synth_inputs(1002, 2, 167, 49152)
    with torch.no_grad():
        assert_close(m(fp.to(dev), img.to(dev)).cpu().numpy(), g["eval/B2/out"], rtol=1e-4, atol_frac=2e-5, what="concat eval")
    m.train()
    fp, img, y = synth_inputs(1007, 7, 167, 49152)
    out = m(fp.to(dev), img.to(dev))
    torch.nn.MSELoss()(out.squeeze(), y.to(dev)).backward()
    assert_close(out.detach().cpu().numpy(), g["train/B7/out"], rtol=1e-4, atol_frac=2e-5, what="concat train")
    for k, q in m.named_parameters():
        check_summary(g, f"train/B7/{k}", q.grad, rtol=5e-4, atol_frac=grad_atol(k))
