"""Host-side cost of the whole-model C calls (enqueue only) vs GPU time."""
import sys, os, time, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bbbp_amd
from bbbp_amd import _lib
dev = torch.device("cuda:0")
torch.manual_seed(0)
m = bbbp_amd.MixedInputModel(167, 128).to(dev).train()
B = 512
fp = torch.randn(B, 167, device=dev); img = torch.randn(B, 49152, device=dev); y = torch.randn(B, device=dev)
crit = torch.nn.MSELoss()
def step(timing=None):
    t0 = time.perf_counter()
    out = m(fp, img)
    t1 = time.perf_counter()
    loss = crit(out.squeeze(), y)
    t2 = time.perf_counter()
    loss.backward()
    t3 = time.perf_counter()
    for p in m.parameters(): p.grad = None
    if timing is not None: timing.append((t1 - t0, t2 - t1, t3 - t2))
for _ in range(3): step()
torch.cuda.synchronize()
for mode in ("overlap", "single"):
    tm = []
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): step(tm)
    th = time.perf_counter() - t0
    torch.cuda.synchronize(); tt = time.perf_counter() - t0
    f = sum(t[0] for t in tm) / 10 * 1e3; l = sum(t[1] for t in tm) / 10 * 1e3; b = sum(t[2] for t in tm) / 10 * 1e3
    print(f"host ms/step: forward call {f:.3f}  loss {l:.3f}  backward {b:.3f}  | host loop {th/10*1e3:.3f}  wall {tt/10*1e3:.3f}", flush=True)
