"""Where the engine's sections sit on the device timeline inside the training step (HIP events of all three streams share
one clock): start, end and the gap to the previous section on the same stream group, for a few consecutive steps."""
import ctypes, os, sys
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
crit = bbbp_amd.MSELoss()


def step(i):
    s = (i % 2) * B
    crit(m(fp[s:s + B], img[s:s + B]).squeeze(), y[s:s + B]).backward()
    opt.step(); opt.zero_grad(set_to_none=True)


L = _lib.lib()
for i in range(30): step(i)
torch.cuda.synchronize()
L.bbbp_profile_select((1 << 11) - 1); L.bbbp_profile_enable(1)      # the eleven top-level sections only: per-layer sections (round 3) add ~200 event pairs per step
NSTEP = 4
for i in range(NSTEP): step(i)
torch.cuda.synchronize()
N = 4096
sec = (ctypes.c_int * N)(); a = (ctypes.c_float * N)(); b = (ctypes.c_float * N)()
n = L.bbbp_profile_timeline(sec, a, b, N)
names = [L.bbbp_profile_section_name(i).decode() for i in range(L.bbbp_profile_num_sections())]
rows = sorted((a[i], b[i], names[sec[i]]) for i in range(n))
per = n // NSTEP
t_first = rows[0][0]
for k, (s0, e0, nm) in enumerate(rows):
    if k % per == 0:
        base = s0
        print(f"--- step {k // per}: starts at +{s0 - t_first:.3f} ms")
    print(f"   {nm:12s} {s0 - base:7.3f} -> {e0 - base:7.3f}   ({e0 - s0:.3f} ms)")
ms = (ctypes.c_float * len(names))(); cnt = (ctypes.c_int * len(names))()
L.bbbp_profile_collect(ms, cnt); L.bbbp_profile_enable(0)
