"""Whole training step (fwd + MSE + bwd + fused AdamW) wall time, with / without the engine's section events."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bbbp_amd
from bbbp_amd import _lib
from bbbp_amd.optim import AdamW
dev = torch.device("cuda:0")
torch.manual_seed(0)
m = bbbp_amd.MixedInputModel(167, 128).to(dev).train()
opt = AdamW(m.parameters(), lr=1e-4, weight_decay=1e-5)
B = 512
fp = torch.randn(2 * B, 167, device=dev); img = torch.rand(2 * B, 49152, device=dev); y = torch.randn(2 * B, device=dev)
crit = torch.nn.MSELoss()
def step(i):
    s = (i % 2) * B
    loss = crit(m(fp[s:s + B], img[s:s + B]).squeeze(), y[s:s + B])
    loss.backward()
    opt.step()
    opt.zero_grad(set_to_none=True)
for i in range(5): step(i)
L = _lib.lib()
for prof in (0, 1, 0, 1):
    L.bbbp_profile_enable(prof)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(20): step(i)
    th = time.perf_counter() - t0
    torch.cuda.synchronize(); tt = time.perf_counter() - t0
    print(f"profiling {prof}: host loop {th / 20 * 1e3:.3f} ms/step, wall {tt / 20 * 1e3:.3f} ms/step", flush=True)
L.bbbp_profile_enable(0)
