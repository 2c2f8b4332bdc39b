"""Where the exact-global-batch engine's step goes at one rank: host time of the forward / backward calls and GPU time."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bbbp_amd
from bbbp_amd import variants
dev = torch.device("cuda:0")
B, F = int(os.environ.get("B", 512)), 167
fp = torch.randn(B, F, device=dev); img = torch.randn(B, 49152, device=dev); y = torch.randn(B, device=dev)
for name, cls in (("replica", bbbp_amd.MixedInputModel), ("exact", variants.ExactBatchMixedInputModel)):
    torch.manual_seed(0)
    m = cls(F, 128).to(dev).train()
    loss_fn = bbbp_amd.MSELoss()
    for it in range(8):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        out = m(fp, img)
        t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
        loss = loss_fn(out.squeeze(), y)
        loss.backward()
        t3 = time.perf_counter(); torch.cuda.synchronize(); t4 = time.perf_counter()
        m.zero_grad(set_to_none=True)
        if it >= 5:
            print(f"{name}: fwd host {1e3*(t1-t0):.2f} ms, fwd done {1e3*(t2-t0):.2f}; bwd host {1e3*(t3-t2):.2f}, bwd done {1e3*(t4-t2):.2f}", flush=True)

# the same models inside tools/time_configs.py's loop (torch.optim.AdamW, torch's MSELoss)
for name, cls in (("replica", bbbp_amd.MixedInputModel), ("exact", variants.ExactBatchMixedInputModel)):
    torch.manual_seed(0)
    m = cls(F, 128).to(dev).train()
    opt = torch.optim.AdamW(m.parameters(), lr=1e-4, weight_decay=1e-5)
    for it in range(8):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        torch.nn.MSELoss()(m(fp, img).squeeze(), y).backward()
        t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
        opt.step()
        t3 = time.perf_counter(); torch.cuda.synchronize(); t4 = time.perf_counter()
        opt.zero_grad(set_to_none=True)
        t5 = time.perf_counter()
        if it >= 5:
            print(f"{name}: fwd+bwd host {1e3*(t1-t0):.2f} done {1e3*(t2-t0):.2f}; opt host {1e3*(t3-t2):.2f} done {1e3*(t4-t2):.2f}; zero_grad {1e3*(t5-t4):.2f}", flush=True)
