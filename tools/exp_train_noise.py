"""How far the 9-step faithful training run of tests/test_gpu_training.py lands from the CPU oracle for each conv2 form
(direct / Winograd): separates rounding-noise amplification (Adam normalises noise-level gradients) from real deviations."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import bbbp_amd
from bbbp_amd import _lib, training
from helpers import synth_inputs
from test_gpu_training import oracle_train

dev = torch.device("cuda")
F, N, NT, BS, EPOCHS = 64, 96, 32, 32, 3
fp, img, y = synth_inputs(31, N + NT, F, 49152)
rng = np.random.default_rng(0)
orders = [rng.permutation(N) for _ in range(EPOCHS)]
ref = None
for mode in (0, 1, 2, 3):
    _lib.lib().bbbp_set_conv_winograd(mode)
    torch.manual_seed(5)
    model = bbbp_amd.MixedInputModel(F, 128)
    for mod in model.modules():
        if isinstance(mod, torch.nn.Dropout): mod.p = 0.0
        if isinstance(mod, torch.nn.MultiheadAttention): mod.dropout = 0.0
    state0 = {k: v.clone() for k, v in model.state_dict().items()}
    model = model.to(dev)
    d = lambda t: t.to(dev)
    hist = training.train_fold(model, (d(fp[:N]), d(img[:N]), d(y[:N])), (d(fp[N:]), d(img[N:]), d(y[N:])), epochs=EPOCHS,
                               batch_size=BS, faithful_mode=True, batch_orders=orders)
    preds = training.predict(model, d(fp[N:]), d(img[N:]), BS).cpu().numpy()
    if ref is None:
        ref = oracle_train(state0, fp[:N], img[:N], y[:N], orders, BS, True, (fp[N:], img[N:]))
    ref_losses, ref_preds = ref[0], ref[1].numpy()
    yt = y[N:].numpy()
    print(f"mode {mode}: r2 {training.r2_score(yt, preds):+.6f} (oracle {training.r2_score(yt, ref_preds):+.6f})  "
          f"mse {training.mean_squared_error(yt, preds):.6f} (oracle {training.mean_squared_error(yt, ref_preds):.6f})  "
          f"max|pred - oracle| {np.max(np.abs(preds - ref_preds)):.2e}  losses {['%.6f' % l for l in hist['train_loss']]} "
          f"oracle {['%.6f' % l for l in ref_losses]}", flush=True)
    if mode == 0: p0 = preds
    else: print(f"         max|pred - direct-form pred| {np.max(np.abs(preds - p0)):.2e}")
