"""Where the host time of one training step goes (enqueue only; the GPU runs behind)."""
import sys, os, time, cProfile, pstats, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bbbp_amd
from bbbp_amd.optim import AdamW
dev = torch.device("cuda:0")
torch.manual_seed(0)
m = bbbp_amd.MixedInputModel(167, 128).to(dev).train()
opt = AdamW(m.parameters(), lr=1e-4, weight_decay=1e-5)
B = 512
fp = torch.randn(2 * B, 167, device=dev); img = torch.rand(2 * B, 49152, device=dev); y = torch.randn(2 * B, device=dev)
crit = torch.nn.MSELoss()
def step(i, t=None):
    s = (i % 2) * B
    t0 = time.perf_counter(); out = m(fp[s:s + B], img[s:s + B]).squeeze()
    t1 = time.perf_counter(); loss = crit(out, y[s:s + B])
    t2 = time.perf_counter(); loss.backward()
    t3 = time.perf_counter(); opt.step()
    t4 = time.perf_counter(); opt.zero_grad(set_to_none=True)
    t5 = time.perf_counter()
    if t is not None: t.append((t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4))
for i in range(5): step(i)
torch.cuda.synchronize()
tm = []
for i in range(20):
    step(i, tm)
    torch.cuda.synchronize()          # so that the host never waits on a full queue: pure enqueue cost
names = ["forward", "loss", "backward", "opt.step", "zero_grad"]
print("host ms/step:", {n: round(sum(t[j] for t in tm) / len(tm) * 1e3, 3) for j, n in enumerate(names)}, "total", round(sum(sum(t) for t in tm) / len(tm) * 1e3, 3))
pr = cProfile.Profile(); pr.enable()
for i in range(10): step(i)
pr.disable(); torch.cuda.synchronize()
st = io.StringIO(); pstats.Stats(pr, stream=st).sort_stats("cumulative").print_stats(22); print(st.getvalue()[:4500])
