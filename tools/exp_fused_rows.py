"""Row-fused encoder schedule (csrc/encoder.hip) against the launch-per-op schedule: ms per training step by batch size."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bbbp_amd
from bbbp_amd import _lib
from bbbp_amd.optim import AdamW
dev = torch.device("cuda:0")
L = _lib.lib()
def run(F, B, train, fused, steps=30):
    L.bbbp_set_fused_encoder(fused)
    torch.manual_seed(0)
    m = bbbp_amd.MixedInputModel(F, 128).to(dev).train(train)
    opt = AdamW(m.parameters(), lr=1e-4, weight_decay=1e-5)
    fp = torch.randn(B, F, device=dev); img = torch.randn(B, 49152, device=dev); y = torch.randn(B, device=dev)
    crit = bbbp_amd.MSELoss()
    def step():
        if train:
            crit(m(fp, img).squeeze(), y).backward(); opt.step(); opt.zero_grad(set_to_none=True)
        else:
            with torch.no_grad(): m(fp, img)
    for _ in range(5): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps): step()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / steps
    return dt * 1e3
for F, B, train in ((167, 32, True), (167, 64, True), (167, 128, True), (167, 256, True), (167, 512, True), (64, 256, True), (167, 4096, False)):
    a, b = run(F, B, train, 0, 30 if B < 4096 else 5), run(F, B, train, 1, 30 if B < 4096 else 5)
    print(f"F={F} B={B} {'train' if train else 'eval'}: per-op {a:7.3f} ms  row-fused {b:7.3f} ms   ({B / a * 1e3:9.0f} -> {B / b * 1e3:9.0f} molecules/s)", flush=True)
