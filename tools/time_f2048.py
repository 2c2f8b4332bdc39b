import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bbbp_amd
from bbbp_amd.optim import AdamW
dev = torch.device("cuda:0")
def run(F, B, steps=10):
    torch.manual_seed(0)
    m = bbbp_amd.MixedInputModel(F, 128).to(dev).train()
    opt = AdamW(m.parameters(), lr=1e-4, weight_decay=1e-5)
    fp = torch.randn(B, F, device=dev); img = torch.randn(B, 49152, device=dev); y = torch.randn(B, device=dev)
    def step():
        bbbp_amd.MSELoss()(m(fp, img).squeeze(), y).backward(); opt.step(); opt.zero_grad(set_to_none=True)
    for _ in range(5): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps): step()
    th = time.perf_counter() - t0
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / steps
    print(f"F={F} B={B}: {dt*1e3:8.2f} ms/step (host loop {th/steps*1e3:.2f}) {B/dt:10.0f} molecules/s", flush=True)
    del m, opt; torch.cuda.empty_cache()
run(2048, 64); run(2048, 128); run(2048, 512); run(167, 64)
