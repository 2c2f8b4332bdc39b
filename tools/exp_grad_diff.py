"""Per-parameter gradient difference between the direct and the Winograd form of conv2 (same weights, same batch)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import bbbp_amd
from bbbp_amd import _lib
from helpers import synth_inputs

dev = torch.device("cuda")
F, B = 64, 32
fp, img, y = synth_inputs(31, B, F, 49152)
torch.manual_seed(5)
model = bbbp_amd.MixedInputModel(F, 128)
for mod in model.modules():
    if isinstance(mod, torch.nn.Dropout): mod.p = 0.0
    if isinstance(mod, torch.nn.MultiheadAttention): mod.dropout = 0.0
model = model.to(dev)
L = _lib.lib()
res = {}
for train in (True, False):
    model.train(train)
    for mode in (0, 1, 2):
        L.bbbp_set_conv_winograd(mode)
        model.zero_grad(set_to_none=True)
        out = model(fp.to(dev), img.to(dev)).squeeze()
        loss = torch.nn.functional.mse_loss(out, y.to(dev))
        loss.backward()
        res[mode] = (out.detach().cpu().double(), {k: p.grad.detach().cpu().double() for k, p in model.named_parameters()})
        if train:
            model.fc[2].running_mean.zero_(); model.fc[2].running_var.fill_(1.0)
    for mode in (1, 2):
        o0, g0 = res[0]; o1, g1 = res[mode]
        print(f"train={train} mode {mode} vs direct: out max|diff| {float((o1 - o0).abs().max()):.2e} (max|out| {float(o0.abs().max()):.2f})")
        rows = []
        for k in g0:
            den = float(g0[k].abs().max()) + 1e-30
            rows.append((float((g1[k] - g0[k]).abs().max()) / den, den, k))
        rows.sort(reverse=True)
        for r in rows[:8]:
            print(f"    {r[2]:60s} rel diff {r[0]:.2e}   max|g| {r[1]:.2e}")
