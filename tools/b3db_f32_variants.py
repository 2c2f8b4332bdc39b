#!/usr/bin/env python3
"""How far apart are two EQUALLY VALID float32 runs of the B3DB-scale acceptance loop (tools/make_golden.py: case_b3db_oof)?

The float32 run of the reference class committed in tests/golden/b3db_oof.npz sits 0.0008 (R^2) from the float64 run -- but per fold its
training losses differ from float64's by 10-60 % after ten epochs: the loop amplifies float32 rounding chaotically, so ONE float32 run is a
noisy ruler.  This script repeats that float32 run (reference class, torch CPU, same folds, same batch orders) with every initial weight
moved by half an ulp (x (1 +- 2^-24), random signs, one seed per variant): the size of difference any other correct float32 implementation
starts from after its first rounding.  It writes the R^2 / MSE of each variant to tests/golden/b3db_oof_f32_variants.npz -- the spread is
the yardstick the GPU run is held to (tools/exp_b3db_r2.py).  Build container only (needs /root/reference); ~8 min of CPU per variant."""
import os
import sys

import numpy as np
import torch
import torch.nn as nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import make_golden as mg                                        # noqa: E402
from oracle import preprocess_cpu                               # noqa: E402
from sklearn.model_selection import KFold                       # noqa: E402


def main(variants=(1, 2, 3)):
    torch.set_num_threads(int(os.environ.get("OMP_NUM_THREADS", "4")))
    ns = mg.load_classes("Models/multi_input_data_regression_opt_transformer_cnn_20250113.py", {"MixedDataset", "MultiHeadAttentionFusion", "MixedInputModel"})
    g = np.load(os.path.join(ROOT, "tests", "golden", "b3db_oof.npz"))
    d = np.load(os.path.join(ROOT, "tests", "golden", "b3db_images_u8.npz"))
    imgs_u8, bits, ys = d["images_u8"], d["bits_u8"], d["logBB"]
    N, EPOCHS, BS, SEED = int(g["meta/N"]), int(g["meta/epochs"]), int(g["meta/batch_size"]), int(g["meta/init_seed"])
    flat = (imgs_u8.transpose(0, 3, 1, 2).astype(np.float32) / np.float32(255.0)).reshape(N, -1)
    fp_n, img_n = preprocess_cpu.standardize_features(bits, flat)
    folds = list(KFold(10, shuffle=True, random_state=42).split(np.arange(N)))
    rng = np.random.default_rng(int(g["meta/order_seed"]))
    orders = [[rng.permutation(len(tr)) for _ in range(EPOCHS)] for tr, _ in folds]
    fp_t, img_t = torch.from_numpy(fp_n), torch.from_numpy(img_n)
    y_t = torch.from_numpy(ys).to(torch.float32)
    crit = nn.MSELoss()
    out = {}
    path = os.path.join(ROOT, "tests", "golden", "b3db_oof_f32_variants.npz")
    for var in variants:
        preds_all = np.zeros(N)
        for k, (tr, te) in enumerate(folds):
            torch.manual_seed(SEED + k)
            model = ns["MixedInputModel"](167, 128)
            mg.zero_dropout(model)
            gp = torch.Generator().manual_seed(1000 * var + k)
            with torch.no_grad():
                for p in model.parameters():
                    p.mul_(1.0 + (torch.randint(0, 2, p.shape, generator=gp).float() * 2 - 1) * 2.0 ** -24)
            opt = torch.optim.AdamW(model.parameters(), lr=1e-4, weight_decay=1e-5)
            tr_t, te_t = torch.as_tensor(tr), torch.as_tensor(te)
            model.train()
            for ep in range(EPOCHS):
                for i in range(0, len(tr), BS):
                    idx = tr_t[torch.as_tensor(orders[k][ep][i:i + BS])]
                    opt.zero_grad()
                    crit(model(fp_t[idx], img_t[idx]).squeeze(), y_t[idx]).backward()
                    opt.step()
                model.eval()
            with torch.no_grad():
                pr = torch.cat([model(fp_t[te_t[i:i + BS]], img_t[te_t[i:i + BS]]).reshape(-1) for i in range(0, len(te), BS)])
            preds_all[te] = pr.double().numpy()
            print(f"variant {var} fold {k} done", flush=True)
        yt = y_t.double().numpy()
        mse = float(((yt - preds_all) ** 2).mean()); r2 = 1.0 - float(((yt - preds_all) ** 2).sum()) / float(((yt - yt.mean()) ** 2).sum())
        out[f"nn_f32_v{var}"] = preds_all; out[f"metrics_f32_v{var}"] = np.array([r2, mse])
        print(f"variant {var}: R2 {r2:.6f} MSE {mse:.6f}   (unperturbed f32 {g['metrics_f32']}, f64 {g['metrics_f64']})", flush=True)
        np.savez_compressed(path, **out)


if __name__ == "__main__":
    main(tuple(int(v) for v in sys.argv[1:]) or (1, 2, 3))
