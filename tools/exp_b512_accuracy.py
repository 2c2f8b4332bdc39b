#!/usr/bin/env python3
"""Where does the B=512 training step lose accuracy?  Per gradient tensor: max |GPU - float64 oracle| and max |torch-CPU
float32 - float64 oracle|, both relative to the tensor's max.  (GPU box; writes to stdout.)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import bbbp_amd
from oracle import reference_cpu as oracle
from helpers import synth_inputs

dev = torch.device("cuda:0")
B = int(os.environ.get("B", "512")); F = 167
torch.manual_seed(20250113)
m = bbbp_amd.MixedInputModel(F, 128).to(dev)
for mod in m.modules():
    if isinstance(mod, torch.nn.Dropout): mod.p = 0.0
    if isinstance(mod, torch.nn.MultiheadAttention): mod.dropout = 0.0
m.train()
fp, img, y = synth_inputs(512512, B, F, 49152)
def params(cast):
    return {k: (cast(v.detach().cpu()) if v.dtype.is_floating_point else v.detach().cpu()).clone()
            .requires_grad_(v.dtype.is_floating_point and "running" not in k) for k, v in m.state_dict().items()}
res = {}
for name, cast in (("f64", torch.Tensor.double), ("f32", torch.Tensor.float)):
    p = params(cast); parts = {}
    out = oracle.mixed_input_forward(p, cast(fp), cast(img), training=True, bn_state={}, parts=parts)
    for v in parts.values(): v.retain_grad()
    oracle.mse_loss(out, cast(y)).backward()
    res[name] = (p, parts, out)
out = m(fp.to(dev), img.to(dev))
bbbp_amd.MSELoss()(out.squeeze(), y.to(dev)).backward()
torch.cuda.synchronize()
p64, parts64, o64 = res["f64"]; p32, parts32, o32 = res["f32"]
print("output: gpu %.2e  cpu32 %.2e (rel. to max)" % (float((out.detach().cpu().double() - o64).abs().max() / o64.abs().max()),
                                                     float((o32.double() - o64).abs().max() / o64.abs().max())))
for k, q in m.named_parameters():
    e = p64[k].grad; s = float(e.abs().max()) + 1e-300
    print("%-62s gpu %.2e  cpu32 %.2e  scale %.2e" % (k, float((q.grad.cpu().double() - e).abs().max()) / s,
                                                      float((p32[k].grad.double() - e).abs().max()) / s, s))
k = "fingerprint_transformer.layers.0.self_attn.in_proj_weight"
err = (dict(m.named_parameters())[k].grad.cpu().double() - p64[k].grad).abs()
print("in_proj_weight L0 err by block Q/K/V:", [float(err[i * F:(i + 1) * F].max()) for i in range(3)])
print("worst rows:", torch.topk(err.max(dim=1).values, 8))
print("worst cols:", torch.topk(err.max(dim=0).values, 8))
