"""CPU: the oracle (oracle/reference_cpu.py) against the golden vectors produced from the reference's own
classes (tools/make_golden.py), the shipped .pth state_dicts and the decoded stacked_model*.pkl coefficients."""
import numpy as np
import pytest
import torch

import bbbp_amd
from oracle import reference_cpu as oracle
from helpers import assert_close, check_param_checksums, check_summary, check_summary_adam, golden, synth_inputs

FUSION = "attention_fusion."      # degenerate heads: gradients are rounding noise


def adam_noise_amplified(k):
    """Tensors whose exact gradient is (partly) ZERO, so AdamW divides rounding noise by (|noise| + eps):
    the fusion heads, and the key-bias third of in_proj_bias (softmax is invariant to a per-query constant)."""
    # ... and fc.0.bias: a unit whose ReLU is active for the whole batch feeds BatchNorm, which cancels bias shifts
    return k.startswith(FUSION) or k.endswith("self_attn.in_proj_bias") or k == "fc.0.bias"


def build(F, seed):
    torch.manual_seed(seed)
    m = bbbp_amd.MixedInputModel(F, 128)
    return m


def named(m, requires_grad=True):
    p = {k: v.detach().clone().requires_grad_(requires_grad and v.dtype.is_floating_point and "running" not in k)
         for k, v in m.state_dict().items()}
    return p


@pytest.mark.parametrize("name,F,seed", [("flagship_f167", 167, 20250113), ("flagship_f64", 64, 64),
                                         ("flagship_f128", 128, 128), ("canonical_f167", 167, 20250113)])
def test_init_and_eval_outputs(name, F, seed):
    g = golden(name)
    m = build(F, seed)
    check_param_checksums(g, m.state_dict())          # drop-in init: same seed => same weights as the reference
    p = named(m, False)
    for key in [k for k in g.files if k.startswith("eval/")]:
        B = int(key.split("/")[1][1:])
        fp, img, _ = synth_inputs(1000 + B, B, F, 49152)
        with torch.no_grad():
            out = oracle.mixed_input_forward(p, fp, img, training=False)
        assert_close(out.numpy(), g[key], rtol=1e-5, what=key)


@pytest.mark.parametrize("name,F,seed,B", [("flagship_f167", 167, 20250113, 7), ("flagship_f64", 64, 64, 7),
                                           ("flagship_f128", 128, 128, 5)])
def test_train_mode_grads_and_bn(name, F, seed, B):
    g = golden(name)
    m = build(F, seed)
    p = named(m)
    fp, img, y = synth_inputs(1000 + B, B, F, 49152)
    bn_state = {}
    out = oracle.mixed_input_forward(p, fp, img, training=True, bn_state=bn_state)
    loss = oracle.mse_loss(out, y)
    loss.backward()
    assert_close(out.detach().numpy(), g[f"train/B{B}/out"], rtol=1e-5, what="train out")
    assert abs(float(loss.detach()) - float(g[f"train/B{B}/loss"])) <= 1e-5 * abs(float(g[f"train/B{B}/loss"]))
    for k in ("fc.2.running_mean", "fc.2.running_var"):
        assert_close(bn_state[k].numpy(), g[f"train/B{B}/bn/{k}"], rtol=1e-5, what=k)
    for k, _ in m.named_parameters():
        if k.startswith(FUSION):
            continue
        check_summary(g, f"train/B{B}/{k}", p[k].grad, rtol=2e-4)


def test_eval_mode_grads_flagship():
    g = golden("flagship_f167")
    m = build(167, 20250113)
    p = named(m)
    fp, img, y = synth_inputs(1002, 2, 167, 49152)
    loss = oracle.mse_loss(oracle.mixed_input_forward(p, fp, img, training=False), y)
    loss.backward()
    assert abs(float(loss.detach()) - float(g["evalgrad/B2/loss"])) <= 1e-5 * abs(float(g["evalgrad/B2/loss"]))
    for k, _ in m.named_parameters():
        if not k.startswith(FUSION):
            check_summary(g, f"evalgrad/B2/{k}", p[k].grad, rtol=2e-4)


def test_adamw_three_steps():
    g = golden("flagship_f167")
    m = build(167, 20250113)
    p = named(m)
    B = 7
    fp, img, y = synth_inputs(1000 + B, B, 167, 49152)
    keys = [k for k, _ in m.named_parameters()]
    state = {k: (torch.zeros_like(p[k]), torch.zeros_like(p[k])) for k in keys}
    for step in range(1, 4):
        for k in keys:
            p[k].grad = None
        bn_state = {}
        loss = oracle.mse_loss(oracle.mixed_input_forward(p, fp, img, training=True, bn_state=bn_state), y)
        loss.backward()
        # step 1 sees identical weights; later steps inherit the +-lr moves AdamW makes out of rounding noise
        ltol = 2e-5 if step == 1 else 3e-3
        assert abs(float(loss.detach()) - float(g[f"adamw/B{B}/step{step}/loss"])) <= ltol * abs(float(g[f"adamw/B{B}/step{step}/loss"]))
        with torch.no_grad():
            for k in keys:
                oracle.adamw_step(p[k], p[k].grad, state[k][0], state[k][1], step)
            for k, v in bn_state.items():
                p[k] = v
        if step in (1, 3):
            for k in keys:
                if not adam_noise_amplified(k):
                    check_summary_adam(g, f"adamw/B{B}/step{step}/{k}", p[k], lr=1e-4, steps=step,
                                       tight_lr_frac=0.02 if step == 1 else 0.6, min_frac=0.9 if step == 1 else 0.75)


@pytest.mark.parametrize("name,pth_F,pth_I", [("pca_mlp_maccs_pth", 64, 128), ("pca_mlp_pth", 128, 256)])
def test_pca_mlp_shipped_weights(name, pth_F, pth_I):
    """The shipped best_nn_model*.pth run through the oracle reproduce the reference class's outputs.  The
    golden holds per-tensor checksums and the outputs (the two .pth files, the reference's data, sit beside it under tests/golden/)."""
    g = golden(name)
    keys, shapes = list(g["meta/keys"]), list(g["meta/shapes"])
    assert len(keys) == 26 and keys[0] == "fingerprint_fc.0.weight" and "image_fc.0.weight" in keys
    assert shapes[keys.index("fingerprint_fc.0.weight")] == f"128,{pth_F}"
    assert shapes[keys.index("image_fc.0.weight")] == f"128,{pth_I}"
    import os
    from helpers import GOLDEN
    # tests/golden/*.pth are the reference's own weight files (data, shipped in Models/)
    pth = os.path.join(GOLDEN, {"pca_mlp_maccs_pth": "best_nn_model_maccs.pth", "pca_mlp_pth": "best_nn_model.pth"}[name])
    sd = torch.load(pth, map_location="cpu", weights_only=True)
    check_param_checksums(g, sd)
    for B in (1, 9):
        fp, img, _ = synth_inputs(1000 + B, B, pth_F, pth_I)
        with torch.no_grad():
            out = oracle.pca_mlp_forward(sd, fp, img)
        assert_close(out.numpy(), g[f"eval/B{B}/out"], rtol=1e-5, what=f"{name} B{B}")


def test_ops_goldens():
    g = golden("ops")
    for name in ("conv1", "conv2"):
        x = torch.from_numpy(g[f"{name}/x"]).requires_grad_(True)
        w = torch.from_numpy(g[f"{name}/w"]).requires_grad_(True)
        b = torch.from_numpy(g[f"{name}/b"]).requires_grad_(True)
        y = oracle.conv3x3_relu_pool(x, w, b)
        y.backward(torch.from_numpy(g[f"{name}/gy"]))
        assert_close(y.detach().numpy(), g[f"{name}/y"], rtol=1e-6, what=name)
        assert_close(x.grad.numpy(), g[f"{name}/gx"], rtol=1e-5, what=name + " gx")
        assert_close(w.grad.numpy(), g[f"{name}/gw"], rtol=1e-5, what=name + " gw")
        assert_close(b.grad.numpy(), g[f"{name}/gb"], rtol=1e-5, what=name + " gb")
    for name, nh in (("enc_e12h3", 3), ("enc_e7h1", 1)):
        p = {"l." + k[len(name) + 3:]: torch.from_numpy(g[k]).requires_grad_(True) for k in g.files if k.startswith(name + "/p/")}
        x = torch.from_numpy(g[f"{name}/x"]).requires_grad_(True)
        y = oracle.encoder_layer(x, p, "l.", nh)
        y.backward(torch.from_numpy(g[f"{name}/gy"]))
        assert_close(y.detach().numpy(), g[f"{name}/y"], rtol=1e-5, what=name)
        assert_close(x.grad.numpy(), g[f"{name}/gx"], rtol=1e-4, what=name + " gx")
        for k, v in p.items():
            assert_close(v.grad.numpy(), g[f"{name}/g/{k[2:]}"], rtol=1e-4, atol_frac=1e-4, what=f"{name} grad {k}")
    # BatchNorm1d
    p = {"bn.weight": torch.from_numpy(g["bn/w"]), "bn.bias": torch.from_numpy(g["bn/b"]),
         "bn.running_mean": torch.zeros(6), "bn.running_var": torch.ones(6), "bn.num_batches_tracked": torch.tensor(0)}
    st = {}
    y = oracle.batchnorm1d(torch.from_numpy(g["bn/x"]), p, "bn.", True, st)
    assert_close(y.numpy(), g["bn/y_train"], rtol=1e-5, what="bn train")
    assert_close(st["bn.running_mean"].numpy(), g["bn/running_mean"], rtol=1e-6, what="bn rm")
    assert_close(st["bn.running_var"].numpy(), g["bn/running_var"], rtol=1e-6, what="bn rv")
    p.update(st)
    assert_close(oracle.batchnorm1d(torch.from_numpy(g["bn/x"]), p, "bn.", False).numpy(), g["bn/y_eval"], rtol=1e-5, what="bn eval")


def test_dense_mlp_golden():
    g = golden("dense_mlp")
    keys = list(g["meta/keys"])
    # rebuild the reference init: same module order as Models/multi_input_data_regression_opt.py:45-78
    import torch.nn as nn
    torch.manual_seed(5)
    fpb = nn.Sequential(nn.Linear(167, 512), nn.ReLU(), nn.BatchNorm1d(512), nn.Dropout(0.2), nn.Linear(512, 256), nn.ReLU(),
                        nn.BatchNorm1d(256), nn.Linear(256, 128), nn.ReLU())
    imb = nn.Sequential(nn.Linear(768, 1024), nn.ReLU(), nn.BatchNorm1d(1024), nn.Dropout(0.2), nn.Linear(1024, 256), nn.ReLU(),
                        nn.BatchNorm1d(256), nn.Linear(256, 128), nn.ReLU())
    fc = nn.Sequential(nn.Linear(256, 256), nn.ReLU(), nn.BatchNorm1d(256), nn.Linear(256, 128), nn.ReLU(), nn.Linear(128, 64),
                       nn.ReLU(), nn.Linear(64, 1))
    sd = {}
    for pre, mod in (("fingerprint_fc.", fpb), ("image_fc.", imb), ("fc.", fc)):
        sd.update({pre + k: v for k, v in mod.state_dict().items()})
    assert list(sd.keys()) == keys
    check_param_checksums(g, sd)
    fp, img, _ = synth_inputs(1004, 4, 167, 768)
    with torch.no_grad():
        out = oracle.dense_mlp_forward(sd, fp, img, training=False)
    assert_close(out.numpy(), g["eval/B4/out"], rtol=1e-5, what="dense eval")
    fp, img, y = synth_inputs(1006, 6, 167, 768)
    with torch.no_grad():
        out = oracle.dense_mlp_forward(sd, fp, img, training=True, bn_state={})
    assert_close(out.numpy(), g["train/B6/out"], rtol=1e-5, what="dense train")


def test_stacked_known_answers():
    """Shipped meta-learners (Models/stacked_model*.pkl, decoded without unpickling): predict = X c + b."""
    X = np.array([[-0.5, -0.3, -0.7], [0.2, 0.1, 0.4], [1.1, 0.9, 1.0]])
    for name, (coef, icpt) in oracle.STACKED_KNOWN.items():
        want = X[:, 0] * coef[0] + X[:, 1] * coef[1] + X[:, 2] * coef[2] + icpt
        np.testing.assert_allclose(oracle.linear_predict(X, coef, icpt), want, rtol=1e-15)
    # fit/predict round trip against scikit-learn (third-party arithmetic of the reference's stack surface)
    from sklearn.linear_model import LinearRegression, Ridge
    rng = np.random.default_rng(0)
    Xf = rng.normal(size=(200, 3)); yf = Xf @ np.array([0.2, 0.7, 0.1]) + 0.03 + 0.05 * rng.normal(size=200)
    for alpha, sk in ((0.0, LinearRegression()), (1.0, Ridge(alpha=1.0))):
        sk.fit(Xf, yf)
        coef, icpt = oracle.linear_fit(Xf, yf, alpha)
        np.testing.assert_allclose(coef, sk.coef_, rtol=1e-9)
        np.testing.assert_allclose(icpt, sk.intercept_, rtol=1e-9)
    np.testing.assert_allclose(oracle.weighted_ensemble([1.0], [2.0], [3.0]), [0.4 + 0.6 + 0.9])


def test_nhead_rule():
    assert [oracle.nhead_rule(f) for f in (167, 64, 128, 2048)] == [1, 8, 16, 256]
    assert [bbbp_amd.reference_nhead(f) for f in (167, 64, 128, 2048)] == [1, 8, 16, 256]
    assert oracle.nhead_rule(167, start=8) == 1 and oracle.nhead_rule(2048, start=8) == 8


def test_wide_deep_variant_golden():
    """Oracle restatement of the wide/deep MixedInputModel (incl. the batch-mean broadcast of its fusion block) against
    the reference class; the product's module tree reproduces the reference's seeded initial weights."""
    from bbbp_amd.variants import WideDeepMixedInputModel
    g = golden("wide_deep_f167")
    torch.manual_seed(20250107)
    m = WideDeepMixedInputModel(167, 128)
    check_param_checksums(g, m.state_dict())
    assert m.nhead == 1 and len(m.fingerprint_transformer.layers) == 12
    p = named(m)
    fp, img, _ = synth_inputs(1003, 3, 167, 49152)
    with torch.no_grad():
        assert_close(oracle.wide_deep_forward(p, fp, img, training=False).numpy(), g["eval/B3/out"], rtol=1e-5, what="wide eval")
    fp, img, y = synth_inputs(1005, 5, 167, 49152)
    st = {}
    out = oracle.wide_deep_forward(p, fp, img, training=True, bn_state=st)
    oracle.mse_loss(out, y).backward()
    assert_close(out.detach().numpy(), g["train/B5/out"], rtol=1e-5, what="wide train")
    for k in ("fc.0.weight", "image_cnn.6.weight", "image_cnn.10.weight", "fingerprint_transformer.layers.11.linear1.weight",
              "attention_fusion.cross_modal_attention.2.weight", "fc.12.weight"):
        check_summary(g, f"train/B5/{k}", p[k].grad, rtol=2e-4, atol_frac=1e-4)


def _filtered(sd, drop_prefix="fingerprint_transformer."):
    return {k: v for k, v in sd.items() if not k.startswith(drop_prefix)}


def test_concat_variant_golden():
    """Earliest Transformer+CNN class (Descriptors/multi_input_data_regression_opt_round_2_transformer_cnn.py:45-102: torch.cat
    fusion, no attention_fusion block): seeded init of the product's module tree and the oracle's fusion="concat" path."""
    g = golden("concat_f167")
    torch.manual_seed(20250102)
    m = bbbp_amd.ConcatMixedInputModel(167, 128)
    check_param_checksums(g, m.state_dict())
    assert not any(k.startswith("attention_fusion") for k in m.state_dict())
    p = named(m)
    fp, img, _ = synth_inputs(1002, 2, 167, 49152)
    with torch.no_grad():
        assert_close(oracle.mixed_input_forward(p, fp, img, training=False, fusion="concat").numpy(), g["eval/B2/out"], rtol=1e-5, what="concat eval")
    fp, img, y = synth_inputs(1007, 7, 167, 49152)
    st = {}
    out = oracle.mixed_input_forward(p, fp, img, training=True, bn_state=st, fusion="concat")
    oracle.mse_loss(out, y).backward()
    assert_close(out.detach().numpy(), g["train/B7/out"], rtol=1e-5, what="concat train")
    for k, _ in m.named_parameters():
        check_summary(g, f"train/B7/{k}", p[k].grad, rtol=2e-4)


def test_two_branch_config2_golden():
    """BASELINE config 2 (the class above without its encoder; tools/make_golden.py:drop_encoder): key set, weights carried
    over from the seeded full class, oracle (num_layers=0, fusion="concat") vs the golden up to the config's batch 256,
    and three AdamW steps."""
    g = golden("two_branch_f167")
    torch.manual_seed(20250102)
    full = bbbp_amd.ConcatMixedInputModel(167, 128)
    m = bbbp_amd.TwoBranchConcatModel(167, 128)
    m.load_state_dict(_filtered(full.state_dict()), strict=True)
    check_param_checksums(g, m.state_dict())
    assert sum(q.numel() for q in m.parameters()) == 8_537_153       # 21504 + 896 + 18496 + 8388736 + head 107521
    p = named(m)
    for B in (2, 7):
        fp, img, _ = synth_inputs(1000 + B, B, 167, 49152)
        with torch.no_grad():
            out = oracle.mixed_input_forward(p, fp, img, training=False, num_layers=0, fusion="concat")
        assert_close(out.numpy(), g[f"eval/B{B}/out"], rtol=1e-5, what=f"two-branch eval B{B}")
    for B in (7, 256):
        for k in p:
            p[k].grad = None
        fp, img, y = synth_inputs(1000 + B, B, 167, 49152)
        st = {}
        out = oracle.mixed_input_forward(p, fp, img, training=True, bn_state=st, num_layers=0, fusion="concat")
        loss = oracle.mse_loss(out, y)
        loss.backward()
        assert_close(out.detach().numpy(), g[f"train/B{B}/out"], rtol=1e-5, what=f"two-branch train B{B}")
        assert abs(float(loss.detach()) - float(g[f"train/B{B}/loss"])) <= 1e-5 * abs(float(g[f"train/B{B}/loss"]))
        for k in ("fc.2.running_mean", "fc.2.running_var"):
            assert_close(st[k].numpy(), g[f"train/B{B}/bn/{k}"], rtol=1e-5, what=k)
        for k, _ in m.named_parameters():
            check_summary(g, f"train/B{B}/{k}", p[k].grad, rtol=2e-4, atol_frac=1e-4 if k.startswith("image_cnn.") else 2e-5)
    # AdamW
    keys = [k for k, _ in m.named_parameters()]
    for k in keys:
        p[k].grad = None
    state = {k: (torch.zeros_like(p[k]), torch.zeros_like(p[k])) for k in keys}
    fp, img, y = synth_inputs(1007, 7, 167, 49152)
    for step in range(1, 4):
        for k in keys:
            p[k].grad = None
        st = {}
        oracle.mse_loss(oracle.mixed_input_forward(p, fp, img, training=True, bn_state=st, num_layers=0, fusion="concat"), y).backward()
        with torch.no_grad():
            for k in keys:
                oracle.adamw_step(p[k], p[k].grad, state[k][0], state[k][1], step)
            for k, v in st.items():
                p[k] = v
        if step in (1, 3):
            for k in keys:
                if k != "fc.0.bias":
                    check_summary_adam(g, f"adamw/B7/step{step}/{k}", p[k], lr=1e-4, steps=step,
                                       tight_lr_frac=0.02 if step == 1 else 0.6, min_frac=0.9 if step == 1 else 0.75)


def test_rdkit_single_head_fusion_golden():
    """_rdkit.py variant: PCA-MLP with the single-head AttentionFusion (softmax over a size-1 dim): seeded init of the
    product's module tree, oracle vs the golden, and the scorer's gradients are exact zeros in the reference too."""
    from bbbp_amd.variants import RdkitPCAFusionModel
    g = golden("rdkit_pca")
    torch.manual_seed(3)
    m = RdkitPCAFusionModel(128, 256)
    check_param_checksums(g, m.state_dict())
    assert "attention_fusion.attention.0.weight" in m.state_dict() and "attention_fusion.attention.2.bias" in m.state_dict()
    p = named(m)
    for B in (1, 9):
        fp, img, _ = synth_inputs(1000 + B, B, 128, 256)
        with torch.no_grad():
            assert_close(oracle.pca_mlp_forward(p, fp, img, single_head=True).numpy(), g[f"eval/B{B}/out"], rtol=1e-5, what=f"rdkit B{B}")
    fp, img, y = synth_inputs(1006, 6, 128, 256)
    out = oracle.pca_mlp_forward(p, fp, img, single_head=True)
    oracle.mse_loss(out, y).backward()
    assert_close(out.detach().numpy(), g["train/B6/out"], rtol=1e-5, what="rdkit train")
    for k, _ in m.named_parameters():
        if k.startswith(FUSION):
            assert float(g[f"train/B6/{k}/stats"][1]) == 0.0, k          # reference: exact zero gradient
            assert p[k].grad is None or float(p[k].grad.abs().sum()) == 0.0
        else:
            check_summary(g, f"train/B6/{k}", p[k].grad, rtol=2e-4)


def test_oof_f64_fixture_is_reproduced_by_the_float32_oracle_on_one_fold():
    """tests/golden/oof_f64.npz (tools/make_golden.py:case_oof_f64 -- the reference class's seeded initial weights, float64 oracle
    loop): the drop-in's constructor draws the same weights under the same seed, and the float32 oracle loop lands within
    many-step float32 rounding of the stored float64 predictions on a fold (the GPU suite checks all ten)."""
    import bbbp_amd
    from bbbp_amd import training
    from helpers import oracle_train
    g = golden("oof_f64")
    F, N, BS, EPOCHS, SEED = int(g["meta/F"]), int(g["meta/N"]), int(g["meta/batch_size"]), int(g["meta/epochs"]), int(g["meta/init_seed"])
    fp, img, y = synth_inputs(int(g["meta/input_seed"]), N, F, 49152)
    y = (0.5 * fp[:, 0] - 0.3 * fp[:, 1] + 0.2 * y)
    np.testing.assert_allclose([float(fp.double().sum()), float(img.double().sum()), float(y.double().sum())], g["inputs/checksum"], rtol=1e-12)
    folds = training.kfold_indices(N, 10)
    rng = np.random.default_rng(3)
    orders = [[rng.permutation(len(tr)) for _ in range(EPOCHS)] for tr, _ in folds]
    k = 3
    tr, te = folds[k]
    assert np.array_equal(te, g[f"fold{k}/test_idx"])
    torch.manual_seed(SEED + k)
    model = bbbp_amd.MixedInputModel(F, 128)
    for mod in model.modules():
        if isinstance(mod, torch.nn.Dropout):
            mod.p = 0.0
        if isinstance(mod, torch.nn.MultiheadAttention):
            mod.dropout = 0.0
    sd = model.state_dict()
    np.testing.assert_allclose([sum(float(v.double().sum()) for v in sd.values() if v.dtype.is_floating_point),
                                sum(float(v.double().abs().sum()) for v in sd.values() if v.dtype.is_floating_point)],
                               g[f"fold{k}/param_checksum"], rtol=1e-9)
    losses, preds = oracle_train({kk: v.clone() for kk, v in sd.items()}, fp[tr], img[tr], y[tr], orders[k], BS, True, (fp[te], img[te]))
    want = g["nn_f64"][te]
    np.testing.assert_allclose(losses, g[f"fold{k}/train_loss"], rtol=2e-3)
    assert np.max(np.abs(preds.numpy() - want)) <= 5e-3 * max(1.0, np.max(np.abs(g["nn_f64"])))


def test_opt_more_golden():
    """Models/multi_input_data_regression_opt_transformer_cnn_opt_more.py:80-107 (256-wide BatchNorm + Dropout(0.3) branches, fusion over
    512 columns): the oracle against the reference class's goldens; the drop-in draws the reference's initial weights."""
    from bbbp_amd.variants import OptMoreFusionModel
    g = golden("opt_more")
    torch.manual_seed(11)
    m = OptMoreFusionModel(64, 128)
    check_param_checksums(g, m.state_dict())
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    for B in (1, 9):
        fp, img, _ = synth_inputs(1000 + B, B, 64, 128)
        with torch.no_grad():
            out = oracle.opt_more_forward(sd, fp, img, training=False)
        assert_close(out.numpy(), g[f"eval/B{B}/out"], rtol=1e-5, what=f"opt_more eval B={B}")
    fp, img, y = synth_inputs(1006, 6, 64, 128)
    p = {k: v.clone().requires_grad_(v.dtype.is_floating_point and "running" not in k) for k, v in sd.items()}
    st = {}
    out = oracle.opt_more_forward(p, fp, img, training=True, bn_state=st)
    assert_close(out.detach().numpy(), g["train/B6/out"], rtol=1e-5, what="opt_more train")
    oracle.mse_loss(out, y).backward()
    for k in p:
        if p[k].requires_grad and not k.startswith("attention_fusion."):
            check_summary(g, f"train/B6/{k}", p[k].grad, rtol=5e-4, atol_frac=1e-4)
    for k, v in st.items():
        if "running" in k:
            assert_close(v.numpy(), g[f"train/B6/bn/{k}"], rtol=1e-5, what=k)


def test_oracle_adamw_is_torch_optim_adamw_bit_for_bit():
    """The reference's optimizer is torch.optim.AdamW (...20250113.py:172); the oracle's single-tensor restatement runs the same tensor
    ops in the same order, so in float32 it must produce torch's bits (the HIP kernel is held to <= 1 ulp of the same thing in
    tests/test_gpu_round4.py)."""
    n = 50_000
    g = torch.Generator().manual_seed(3)
    p0 = torch.randn(n, generator=g)
    ref = torch.nn.Parameter(p0.clone())
    ropt = torch.optim.AdamW([ref], lr=3e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, foreach=False)
    p, m, v = p0.clone(), torch.zeros(n), torch.zeros(n)
    for step in range(1, 4):
        grad = torch.randn(n, generator=g) * 10.0 ** torch.randint(-6, 2, (n,), generator=g).float()
        ref.grad = grad.clone()
        ropt.step()
        oracle.adamw_step(p, grad, m, v, step, lr=3e-3, weight_decay=1e-2)
        assert torch.equal(p, ref.detach()), step
        assert torch.equal(m, ropt.state[ref]["exp_avg"]) and torch.equal(v, ropt.state[ref]["exp_avg_sq"]), step
