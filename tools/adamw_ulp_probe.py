"""Where the HIP AdamW step differs from torch.optim.AdamW's float32 CPU step, element by element (diagnostic for
tests/test_gpu_round4.py::test_adamw_kernel_follows_torch_optim_adamw_to_the_ulp)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bbbp_amd.optim import AdamW
dev = torch.device("cuda:0")
n = 200_000
g = torch.Generator().manual_seed(12)
p0 = torch.randn(n, generator=g)
ref = torch.nn.Parameter(p0.clone())
ropt = torch.optim.AdamW([ref], lr=3e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, foreach=False)
mine = torch.nn.Parameter(p0.clone().to(dev))
opt = AdamW([mine], lr=3e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2)
grad = torch.randn(n, generator=g) * 10.0 ** torch.randint(-6, 2, (n,), generator=g).float()
ref.grad, mine.grad = grad.clone(), grad.clone().to(dev)
ropt.step(); opt.step()
a, b = mine.detach().cpu(), ref.detach()
st = ropt.state[ref]
print("cpu capability:", torch.backends.cpu.get_cpu_capability())
print("m equal:", torch.equal(opt.state[mine]["exp_avg"].cpu(), st["exp_avg"]), " v equal:", torch.equal(opt.state[mine]["exp_avg_sq"].cpu(), st["exp_avg_sq"]))
mm = (opt.state[mine]["exp_avg"].cpu() != st["exp_avg"]); vv = (opt.state[mine]["exp_avg_sq"].cpu() != st["exp_avg_sq"])
print("m mismatches", int(mm.sum()), "v mismatches", int(vv.sum()), "p mismatches", int((a != b).sum()))
bad = torch.nonzero(a != b).flatten()[:12]
f32 = np.float32
lr, b1, b2, eps, wd = 3e-3, 0.9, 0.999, 1e-8, 1e-2
decay = f32(1 - lr * wd); ss = f32(-(lr / (1 - b1))); bc2s = f32((1 - b2) ** 0.5)
for i in bad.tolist():
    p, gi = f32(p0[i]), f32(grad[i])
    m = f32(st["exp_avg"][i]); v = f32(st["exp_avg_sq"][i])
    pi = f32(p * decay); sq = f32(np.sqrt(v) / bc2s); den = f32(sq + f32(eps)); num = f32(ss * m); q = f32(num / den); emu = f32(pi + q)
    print(f"i={i} p={p!r} g={gi!r} m_eq={not bool(mm[i])} v_eq={not bool(vv[i])} gpu={f32(a[i])!r} torch={f32(b[i])!r} numpy-emulation={emu!r}  pi={pi!r} q={q!r}")
