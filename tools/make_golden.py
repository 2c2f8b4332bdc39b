#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ from the reference's OWN classes.

Runs only in the build container (needs /root/reference).  The reference scripts cannot be
imported (top-level ``pickle.load`` of absent files, ``xgboost``/``catboost`` imports), so the
class definitions (``ClassDef`` nodes only) are pulled out of the source text with ``ast`` and
exec'd in a namespace holding ``torch``/``nn``/``Dataset``.  Nothing of the reference's text is
written anywhere: the fixtures hold seeds, tensor checksums, small slices and outputs only.

Weights are NOT stored (the flagship state_dict is 54 MB): every case records the
``torch.manual_seed`` used before the constructor plus a float64 (sum, abs-sum) per tensor, so a
test rebuilds the identical initial weights from the seed and proves it with the checksums.

Usage:  python tools/make_golden.py            (writes tests/golden/*.npz)
"""
from __future__ import annotations

import ast
import os
import sys

import numpy as np
import torch
import torch.nn as nn
from torch.utils.data import Dataset

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def load_classes(relpath: str, names):
    src = open(os.path.join(REF, relpath), encoding="utf-8").read()
    tree = ast.parse(src)
    ns = {"torch": torch, "nn": nn, "Dataset": Dataset}
    for node in tree.body:
        if isinstance(node, ast.ClassDef) and node.name in names:
            exec(compile(ast.Module([node], []), relpath, "exec"), ns)
    return ns


def zero_dropout(model: nn.Module):
    for m in model.modules():
        if isinstance(m, nn.Dropout):
            m.p = 0.0
        if isinstance(m, nn.MultiheadAttention):
            m.dropout = 0.0


def tensor_summary(t: torch.Tensor):
    d = t.detach().double().flatten()
    n = d.numel()
    idx = torch.linspace(0, n - 1, steps=min(n, 64)).long().clamp_(max=n - 1)     # float32 linspace can round up to n
    return dict(sum=float(d.sum()), abssum=float(d.abs().sum()), l2=float(d.norm()),
                head=d[:64].numpy().copy(), idx=idx.numpy().copy(), samp=d[idx].numpy().copy())


def pack(prefix: str, summ: dict, out: dict):
    out[prefix + "/stats"] = np.array([summ["sum"], summ["abssum"], summ["l2"]], dtype=np.float64)
    out[prefix + "/head"] = summ["head"]
    out[prefix + "/idx"] = summ["idx"]
    out[prefix + "/samp"] = summ["samp"]


def synth_inputs(seed: int, B: int, F: int, I: int):
    g = torch.Generator().manual_seed(seed)
    fp = torch.randn(B, F, generator=g)
    img = torch.randn(B, I, generator=g)
    y = torch.randn(B, generator=g) * 0.8 - 0.1
    return fp, img, y


def drop_encoder(model: nn.Module):
    """BASELINE config 2 (no exact reference script): the reference's round-2 Transformer+CNN class with its
    encoder removed -- the module tree is the reference's own (seeded init included), the encoder is deleted and the two
    statements of its forward that call it are skipped."""
    import types
    del model.fingerprint_transformer

    def forward(self, fingerprint, image):
        fingerprint_out = self.fingerprint_fc(fingerprint)
        image_out = self.image_cnn(image.view(-1, 3, 128, 128))
        return self.fc(torch.cat((fingerprint_out, image_out), dim=1))
    model.forward = types.MethodType(forward, model)


def case_model(tag, relpath, F, I_ctor, I_flat, Bs, init_seed, train_Bs=(), adam_B=None,
               state_dict_path=None, extra_classes=(), mutate=None):
    ns = load_classes(relpath, {"MixedDataset", "MultiHeadAttentionFusion", "MixedInputModel",
                                "AttentionFusion", "MultiModalAttentionFusion", *extra_classes})
    torch.manual_seed(init_seed)
    model = ns["MixedInputModel"](F, I_ctor)
    if state_dict_path is not None:
        sd = torch.load(os.path.join(REF, state_dict_path), map_location="cpu", weights_only=True)
        model.load_state_dict(sd, strict=True)
    if mutate is not None:
        mutate(model)
    zero_dropout(model)
    out = {"meta/init_seed": np.array(init_seed), "meta/F": np.array(F), "meta/I_ctor": np.array(I_ctor),
           "meta/I_flat": np.array(I_flat),
           "meta/keys": np.array(list(model.state_dict().keys())),
           "meta/shapes": np.array([",".join(map(str, v.shape)) for v in model.state_dict().values()])}
    for k, v in model.state_dict().items():
        if v.dtype.is_floating_point:
            out[f"param/{k}"] = np.array([float(v.double().sum()), float(v.double().abs().sum())])
    crit = nn.MSELoss()
    for B in Bs:
        fp, img, y = synth_inputs(1000 + B, B, F, I_flat)
        model.eval()
        with torch.no_grad():
            out[f"eval/B{B}/out"] = model(fp, img).numpy().copy()
        # eval-mode gradients (BN uses running stats; what the published loop does in epochs 2-50)
        model.zero_grad()
        pred = model(fp, img).squeeze()
        loss = crit(pred, y) if B > 1 else ((pred - y.squeeze()) ** 2).mean()
        loss.backward()
        out[f"evalgrad/B{B}/loss"] = np.array(float(loss))
        for k, p in model.named_parameters():
            pack(f"evalgrad/B{B}/{k}", tensor_summary(p.grad), out)
    for B in train_Bs:
        fp, img, y = synth_inputs(1000 + B, B, F, I_flat)
        sd0 = {k: v.clone() for k, v in model.state_dict().items()}
        model.train()
        model.zero_grad()
        o = model(fp, img)
        loss = crit(o.squeeze(), y)
        loss.backward()
        out[f"train/B{B}/out"] = o.detach().numpy().copy()
        out[f"train/B{B}/loss"] = np.array(float(loss))
        for k, p in model.named_parameters():
            pack(f"train/B{B}/{k}", tensor_summary(p.grad), out)
        for k, v in model.state_dict().items():
            if "running_" in k or "num_batches" in k:
                out[f"train/B{B}/bn/{k}"] = v.numpy().copy()
        model.load_state_dict(sd0)
    if adam_B is not None:
        B = adam_B
        fp, img, y = synth_inputs(1000 + B, B, F, I_flat)
        sd0 = {k: v.clone() for k, v in model.state_dict().items()}
        opt = torch.optim.AdamW(model.parameters(), lr=1e-4, weight_decay=1e-5)
        model.train()
        for step in range(1, 4):
            opt.zero_grad()
            loss = crit(model(fp, img).squeeze(), y)
            loss.backward()
            opt.step()
            out[f"adamw/B{B}/step{step}/loss"] = np.array(float(loss))
            if step in (1, 3):
                for k, p in model.named_parameters():
                    pack(f"adamw/B{B}/step{step}/{k}", tensor_summary(p), out)
                for k, v in model.state_dict().items():
                    if "running_" in k:
                        out[f"adamw/B{B}/step{step}/bn/{k}"] = v.numpy().copy()
        model.load_state_dict(sd0)
    path = os.path.join(OUT, tag + ".npz")
    np.savez_compressed(path, **out)
    print(f"wrote {path}: {len(out)} arrays, {os.path.getsize(path) / 1024:.1f} KiB")


def case_ops():
    """Per-op goldens with explicit small tensors (all data stored)."""
    out = {}
    g = torch.Generator().manual_seed(7)
    # conv+relu+pool, both channel configs, small spatial size
    for name, cin, cout, hw in (("conv1", 3, 32, 16), ("conv2", 32, 64, 8)):
        x = torch.randn(2, cin, hw, hw, generator=g)
        conv = nn.Conv2d(cin, cout, 3, 1, 1)
        with torch.no_grad():
            conv.weight.copy_(torch.randn(conv.weight.shape, generator=g) * 0.2)
            conv.bias.copy_(torch.randn(conv.bias.shape, generator=g) * 0.1)
        seq = nn.Sequential(conv, nn.ReLU(), nn.MaxPool2d(2, 2))
        x.requires_grad_(True)
        y = seq(x)
        gy = torch.randn(y.shape, generator=g)
        y.backward(gy)
        out.update({f"{name}/x": x.detach().numpy(), f"{name}/w": conv.weight.detach().numpy(),
                    f"{name}/b": conv.bias.detach().numpy(), f"{name}/y": y.detach().numpy(),
                    f"{name}/gy": gy.numpy(), f"{name}/gx": x.grad.numpy(),
                    f"{name}/gw": conv.weight.grad.numpy(), f"{name}/gb": conv.bias.grad.numpy()})
    # one encoder layer, E=12 nhead=3 and E=7 nhead=1 (prime, like MACCS-167), dropout 0
    for name, E, nh, S in (("enc_e12h3", 12, 3, 9), ("enc_e7h1", 7, 1, 5)):
        torch.manual_seed(11)
        layer = nn.TransformerEncoderLayer(d_model=E, nhead=nh, dropout=0.0)
        with torch.no_grad():
            for p_ in layer.parameters():
                p_.copy_(torch.randn(p_.shape, generator=g) * 0.3)
        x = torch.randn(S, 1, E, generator=g, requires_grad=True)
        layer.train()
        y = layer(x)
        gy = torch.randn(y.shape, generator=g)
        y.backward(gy)
        out[f"{name}/x"] = x.detach().numpy()[:, 0]
        out[f"{name}/y"] = y.detach().numpy()[:, 0]
        out[f"{name}/gy"] = gy.numpy()[:, 0]
        out[f"{name}/gx"] = x.grad.numpy()[:, 0]
        for k, p_ in layer.named_parameters():
            out[f"{name}/p/{k}"] = p_.detach().numpy()
            out[f"{name}/g/{k}"] = p_.grad.numpy()
    # BatchNorm1d train/eval incl. running stats
    bn = nn.BatchNorm1d(6)
    with torch.no_grad():
        bn.weight.copy_(torch.randn(6, generator=g)); bn.bias.copy_(torch.randn(6, generator=g))
    x = torch.randn(5, 6, generator=g, requires_grad=True)
    bn.train(); y = bn(x); gy = torch.randn(5, 6, generator=g); y.backward(gy)
    out.update({"bn/x": x.detach().numpy(), "bn/w": bn.weight.detach().numpy(), "bn/b": bn.bias.detach().numpy(),
                "bn/y_train": y.detach().numpy(), "bn/gy": gy.numpy(), "bn/gx": x.grad.numpy(),
                "bn/gw": bn.weight.grad.numpy(), "bn/gb": bn.bias.grad.numpy(),
                "bn/running_mean": bn.running_mean.numpy().copy(), "bn/running_var": bn.running_var.numpy().copy()})
    bn.eval()
    with torch.no_grad():
        out["bn/y_eval"] = bn(x).numpy()
    path = os.path.join(OUT, "ops.npz")
    np.savez_compressed(path, **out)
    print(f"wrote {path}: {len(out)} arrays, {os.path.getsize(path) / 1024:.1f} KiB")


def case_xgb_head(n_keep=6):
    """The first trees of the XGBoost model the reference ships (Models/xgb_model_maccs.pkl: XGBRegressor, 300 trees, 49 319
    features, reg:squarederror), lifted out of the pickle stream without unpickling: node arrays, base_score, and the raw UBJSON
    bytes of a document cut down to those trees (so that the parser is checked on the library's own byte layout).  No expected
    predictions: xgboost is not installed, the predict rule is pinned only by its published description (parity unpinned)."""
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from bbbp_amd import boosters
    raw = boosters.lift_raw_from_pickle(os.path.join(REF, "Models", "xgb_model_maccs.pkl"))
    doc = boosters.parse_ubjson(raw)
    model = doc["Model"]
    trees = model["learner"]["gradient_booster"]["model"]["trees"]
    left, right, feature, cond, dleft, root, nfeat, base = boosters.XGBTrees.flatten(doc)
    hi = int(root[n_keep])
    out = dict(left=left[:hi].astype(np.int32), right=right[:hi].astype(np.int32), feature=feature[:hi].astype(np.int32), cond=cond[:hi],
               default_left=dleft[:hi], root=root[:n_keep + 1].astype(np.int32), n_features=np.int64(nfeat), base_score=np.float32(base),
               n_trees_total=np.int64(len(trees)), n_nodes_total=np.int64(root[-1]),
               checksum_all=np.array([float(cond.astype(np.float64).sum()), float(np.abs(cond.astype(np.float64)).sum()),
                                      float(feature.astype(np.float64).sum())]))
    # the byte layout of the library's writer: the span of the first tree object inside the raw buffer (tree objects open with the
    # key "base_weights")
    mark = b"{L" + (12).to_bytes(8, "big") + b"base_weights"
    start = raw.index(mark)
    nxt = raw.index(mark, start + 1)
    probe = boosters.parse_ubjson(raw[start:nxt])
    assert np.array_equal(probe["left_children"], trees[0]["left_children"]) and np.array_equal(probe["split_conditions"], trees[0]["split_conditions"])
    out["tree0_ubjson"] = np.frombuffer(raw[start:nxt], dtype=np.uint8)
    np.savez_compressed(os.path.join(OUT, "xgb_maccs_head.npz"), **out)
    print("xgb_maccs_head:", hi, "nodes of", int(root[-1]), "| tree 0 bytes", nxt - start)


def case_oof_f64():
    """The network column of the published fold loop (...20250113.py:147-266), all ten folds, by the FLOAT64 oracle: the yardstick for
    tests/test_gpu_training.py::test_out_of_fold_driver_and_stack_against_oracle_folds.  Per fold the REFERENCE class is constructed
    under torch.manual_seed(40 + k) (its initial weights are what the drop-in's constructor draws under the same seed -- checked by the
    parameter checksums stored here), dropout off, and trained by the oracle's faithful loop in float64 on the seeded 96-molecule set
    the test rebuilds (helpers.synth_inputs(2025, ...), KFold(10, shuffle, random_state=42), batch orders from default_rng(3))."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
    from helpers import oracle_train
    from sklearn.model_selection import KFold
    ns = load_classes("Models/multi_input_data_regression_opt_transformer_cnn_20250113.py",
                      {"MixedDataset", "MultiHeadAttentionFusion", "MixedInputModel"})
    F, N, BS, EPOCHS, SEED = 64, 96, 32, 2, 40
    fp, img, y = synth_inputs(2025, N, F, 49152)
    y = (0.5 * fp[:, 0] - 0.3 * fp[:, 1] + 0.2 * y)
    folds = list(KFold(10, shuffle=True, random_state=42).split(np.arange(N)))
    rng = np.random.default_rng(3)
    orders = [[rng.permutation(len(tr)) for _ in range(EPOCHS)] for tr, _ in folds]
    nn64 = np.zeros(N)
    out = {"meta/F": np.array(F), "meta/N": np.array(N), "meta/batch_size": np.array(BS), "meta/epochs": np.array(EPOCHS),
           "meta/init_seed": np.array(SEED), "meta/input_seed": np.array(2025),
           "inputs/checksum": np.array([float(fp.double().sum()), float(img.double().sum()), float(y.double().sum())])}
    for k, (tr, te) in enumerate(folds):
        torch.manual_seed(SEED + k)
        model = ns["MixedInputModel"](F, 128)
        zero_dropout(model)
        state0 = {kk: v.clone() for kk, v in model.state_dict().items()}
        out[f"fold{k}/param_checksum"] = np.array([sum(float(v.double().sum()) for v in state0.values() if v.dtype.is_floating_point),
                                                   sum(float(v.double().abs().sum()) for v in state0.values() if v.dtype.is_floating_point)])
        losses, preds = oracle_train(state0, fp[tr], img[tr], y[tr], orders[k], BS, True, (fp[te], img[te]), dtype=torch.float64)
        nn64[te] = preds.numpy()
        out[f"fold{k}/test_idx"] = np.asarray(te, dtype=np.int64)
        out[f"fold{k}/train_loss"] = np.asarray(losses)
        print(f"oof_f64 fold {k}: losses {losses}", flush=True)
    out["nn_f64"] = nn64
    path = os.path.join(OUT, "oof_f64.npz")
    np.savez_compressed(path, **out)
    print(f"wrote {path}: {len(out)} arrays, {os.path.getsize(path) / 1024:.1f} KiB")


def b3db_inputs():
    """The B3DB-scale acceptance set: every molecule of B3DB_regression.tsv that has a drawing under
    Descriptors/img_output (1 058), in NO. order; image = the reference's own pipeline (convert('RGB') -> Resize((128,128)) -> ToTensor
    -> flatten, ...fixed_1.py:56-71) kept as the resized uint8 bytes; label = the file's logBB; fingerprint = SYNTHETIC MACCS-shaped bits
    (RDKit is absent): bit 0 always 0, bits 1..24 drawn with a probability that depends on the standardised label (so the encoder
    branch has something to learn), the rest Bernoulli(0.25).  Returns (numbers [N], images_u8 [N,128,128,3], bits_u8 [N,167], logBB [N])."""
    import csv
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    from oracle import preprocess_cpu
    rows = list(csv.DictReader(open(os.path.join(REF, "B3DB/B3DB/B3DB_regression.tsv"), encoding="utf-8"), delimiter="\t"))
    nos, imgs, ys = [], [], []
    for r in rows:
        path = os.path.join(REF, "Descriptors/img_output", f"{int(r['NO.'])}.png")
        if os.path.exists(path):
            nos.append(int(r["NO."])); imgs.append(preprocess_cpu.resized_bytes(path)); ys.append(float(r["logBB"]))
    nos = np.asarray(nos, dtype=np.int64); imgs = np.stack(imgs); ys = np.asarray(ys, dtype=np.float64)
    rng = np.random.default_rng(167)
    z = (ys - ys.mean()) / ys.std()
    p = np.full((len(ys), 167), 0.25)
    w = rng.choice([-1.2, 1.2], size=24)
    p[:, 1:25] = 1.0 / (1.0 + np.exp(-(w[None, :] * z[:, None] - 1.1)))
    bits = (rng.random((len(ys), 167)) < p).astype(np.uint8)
    bits[:, 0] = 0
    return nos, imgs, bits, ys


def case_b3db_oof(epochs=10, batch_size=32, init_seed=4200):
    """R^2 / MSE acceptance at B3DB scale (north_star: "R^2/MSE within +-0.002 of reference"): the published fold loop
    (...20250113.py:146-241: KFold(10, shuffle, random_state=42), a fresh MixedInputModel + AdamW(1e-4, wd 1e-5) per fold, batch 32,
    model.train() ONCE before the epoch loop and the per-epoch validation pass leaving it in eval() -- the faithful quirk) run with the
    REFERENCE'S OWN CLASS and torch.optim.AdamW on the CPU, once in float32 (the reference's precision) and once in float64 (the
    yardstick), 10 epochs, dropout zeroed (no cross-implementation parity for dropout masks), batch orders from default_rng(5).
    Writes tests/golden/b3db_images_u8.npz (the resized bytes + labels + synthetic bits: DATA) and tests/golden/b3db_oof.npz
    (per-fold held-out predictions and losses of both runs, R^2 / MSE)."""
    from sklearn.model_selection import KFold
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    from oracle import preprocess_cpu
    ns = load_classes("Models/multi_input_data_regression_opt_transformer_cnn_20250113.py",
                      {"MixedDataset", "MultiHeadAttentionFusion", "MixedInputModel"})
    nos, imgs_u8, bits, ys = b3db_inputs()
    N = len(ys)
    np.savez_compressed(os.path.join(OUT, "b3db_images_u8.npz"), numbers=nos, images_u8=imgs_u8, bits_u8=bits, logBB=ys)
    flat = (imgs_u8.transpose(0, 3, 1, 2).astype(np.float32) / np.float32(255.0)).reshape(N, -1)
    fp_n, img_n = preprocess_cpu.standardize_features(bits, flat)            # ...fixed_1.py:86-101 (chunks of 100)
    folds = list(KFold(10, shuffle=True, random_state=42).split(np.arange(N)))
    rng = np.random.default_rng(5)
    orders = [[rng.permutation(len(tr)) for _ in range(epochs)] for tr, _ in folds]
    out = {"meta/N": np.array(N), "meta/F": np.array(167), "meta/batch_size": np.array(batch_size), "meta/epochs": np.array(epochs),
           "meta/init_seed": np.array(init_seed), "meta/order_seed": np.array(5),
           "inputs/checksum": np.array([float(fp_n.astype(np.float64).sum()), float(img_n.astype(np.float64).sum()), float(ys.sum())])}
    crit = nn.MSELoss()
    for tag, dt in (("f32", torch.float32), ("f64", torch.float64)):
        fp_t, img_t = torch.from_numpy(fp_n).to(dt), torch.from_numpy(img_n).to(dt)
        y_t = torch.from_numpy(ys).to(torch.float32).to(dt)                   # MixedDataset casts labels to float32 (:44)
        preds_all = np.zeros(N)
        for k, (tr, te) in enumerate(folds):
            torch.manual_seed(init_seed + k)
            model = ns["MixedInputModel"](167, 128)
            zero_dropout(model)
            if tag == "f32":
                sd = model.state_dict()
                out[f"fold{k}/param_checksum"] = np.array([sum(float(v.double().sum()) for v in sd.values() if v.dtype.is_floating_point),
                                                           sum(float(v.double().abs().sum()) for v in sd.values() if v.dtype.is_floating_point)])
                out[f"fold{k}/test_idx"] = np.asarray(te, dtype=np.int64)
            model = model.to(dt)
            opt = torch.optim.AdamW(model.parameters(), lr=1e-4, weight_decay=1e-5)
            tr_t, te_t = torch.as_tensor(tr), torch.as_tensor(te)
            losses, vals = [], []
            model.train()
            for ep in range(epochs):
                tot, nb = 0.0, 0
                for i in range(0, len(tr), batch_size):
                    idx = tr_t[torch.as_tensor(orders[k][ep][i:i + batch_size])]
                    opt.zero_grad()
                    loss = crit(model(fp_t[idx], img_t[idx]).squeeze(), y_t[idx])
                    loss.backward()
                    opt.step()
                    tot += loss.item(); nb += 1
                model.eval()
                with torch.no_grad():
                    vl, vb = 0.0, 0
                    for i in range(0, len(te), batch_size):
                        idx = te_t[i:i + batch_size]
                        vl += crit(model(fp_t[idx], img_t[idx]).squeeze(), y_t[idx]).item(); vb += 1
                losses.append(tot / nb); vals.append(vl / vb)
            with torch.no_grad():
                pr = torch.cat([model(fp_t[te_t[i:i + batch_size]], img_t[te_t[i:i + batch_size]]).reshape(-1) for i in range(0, len(te), batch_size)])
            preds_all[te] = pr.double().numpy()
            out[f"fold{k}/train_loss_{tag}"] = np.asarray(losses); out[f"fold{k}/val_loss_{tag}"] = np.asarray(vals)
            print(f"b3db_oof {tag} fold {k}: train {losses} val {vals}", flush=True)
        out[f"nn_{tag}"] = preds_all
        yt = y_t.double().numpy()
        mse = float(((yt - preds_all) ** 2).mean()); r2 = 1.0 - float(((yt - preds_all) ** 2).sum()) / float(((yt - yt.mean()) ** 2).sum())
        out[f"metrics_{tag}"] = np.array([r2, mse])
        print(f"b3db_oof {tag}: R2 {r2:.6f} MSE {mse:.6f}", flush=True)
        np.savez_compressed(os.path.join(OUT, "b3db_oof.npz"), **out)
    print("wrote b3db_oof.npz")


def main():
    if not os.path.isdir(REF):
        sys.exit("needs /root/reference (build container only)")
    os.makedirs(OUT, exist_ok=True)
    torch.set_num_threads(8)
    M = "Models/"
    only = set(sys.argv[1:])                      # optional: regenerate just the named fixtures
    global case_model, case_ops
    if only:
        all_case_model, all_case_ops = case_model, case_ops
        case_model = lambda tag, *a, **k: all_case_model(tag, *a, **k) if tag in only else None
        case_ops = lambda: all_case_ops() if "ops" in only else None
    # flagship (published variant), MACCS width; nhead = 1, head_dim = 167
    case_model("flagship_f167", M + "multi_input_data_regression_opt_transformer_cnn_20250113.py",
               167, 128, 49152, Bs=(1, 2, 7, 32), init_seed=20250113, train_Bs=(7, 32), adam_B=7)
    # same classes, widths that exercise multi-head attention (nhead 8 / 16, head_dim 8)
    case_model("flagship_f64", M + "multi_input_data_regression_opt_transformer_cnn_20250113.py",
               64, 128, 49152, Bs=(2, 7), init_seed=64, train_Bs=(7,))
    case_model("flagship_f128", M + "multi_input_data_regression_opt_transformer_cnn_20250113.py",
               128, 128, 49152, Bs=(5,), init_seed=128, train_Bs=(5,))
    # canonical variant #1: same architecture, must give the same numbers for the same seed
    case_model("canonical_f167", M + "multi_input_data_regression_opt_transformer_cnn.py",
               167, 128, 49152, Bs=(2,), init_seed=20250113)
    # PCA-MLP fusion model with the two shipped state_dicts
    case_model("pca_mlp_maccs_pth", M + "multi_input_data_regression_opt_transformer_cnn_opt.py",
               64, 128, 128, Bs=(1, 9), init_seed=1, state_dict_path="Models/best_nn_model_maccs.pth")
    case_model("pca_mlp_pth", M + "multi_input_data_regression_opt_transformer_cnn_opt.py",
               128, 256, 256, Bs=(1, 9), init_seed=1, state_dict_path="Models/best_nn_model.pth")
    # dense raw-feature MLP, image width shrunk to 3*16*16 for size
    case_model("dense_mlp", M + "multi_input_data_regression_opt.py",
               167, 768, 768, Bs=(4,), init_seed=5, train_Bs=(6,))
    # wide/deep variant: 12-layer encoder, 3-conv CNN, MultiModalAttentionFusion (batch-mean broadcast), 6-layer head
    case_model("wide_deep_f167", M + "multi_input_data_regression_opt_transformer_cnn_opt_20250107_network.py",
               167, 128, 49152, Bs=(3,), init_seed=20250107, train_Bs=(5,))
    # earliest Transformer+CNN class: torch.cat fusion, no attention_fusion block
    R2 = "Descriptors/multi_input_data_regression_opt_round_2_transformer_cnn.py"
    case_model("concat_f167", R2, 167, 128, 49152, Bs=(2,), init_seed=20250102, train_Bs=(7,))
    # BASELINE config 2: that class with the encoder removed (see drop_encoder), up to the config's batch 256
    case_model("two_branch_f167", R2, 167, 128, 49152, Bs=(2, 7), init_seed=20250102, train_Bs=(7, 256), adam_B=7, mutate=drop_encoder)
    # PCA-MLP fusion with the single-head AttentionFusion (softmax over a size-1 dim => weights == 1)
    case_model("rdkit_pca", M + "multi_input_data_regression_opt_transformer_cnn_rdkit.py",
               128, 256, 256, Bs=(1, 9), init_seed=3, train_Bs=(6,))
    # PCA-MLP fusion with 256-wide BatchNorm + Dropout(0.3) branches, fusion over 512 columns
    case_model("opt_more", M + "multi_input_data_regression_opt_transformer_cnn_opt_more.py",
               64, 128, 128, Bs=(1, 9), init_seed=11, train_Bs=(6,))
    case_ops()
    if not only or "xgb_head" in only:
        case_xgb_head()
    if not only or "oof_f64" in only:
        case_oof_f64()
    if "b3db_oof" in only:                        # ~1 h of CPU: only on request
        case_b3db_oof()


if __name__ == "__main__":
    main()
