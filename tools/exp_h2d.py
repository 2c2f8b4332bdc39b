"""Pinned host -> device copy rate of one training batch's images (100.7 MB) alone and beside the training step."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
dev = torch.device("cuda:0")
h = torch.empty(512, 49152, dtype=torch.float32).pin_memory()
d = torch.empty(512, 49152, dtype=torch.float32, device=dev)
s = torch.cuda.Stream()
for _ in range(3):
    d.copy_(h, non_blocking=True)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
with torch.cuda.stream(s):
    e0.record(s)
    for _ in range(20):
        d.copy_(h, non_blocking=True)
    e1.record(s)
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 20
print(f"H2D 100.7 MB pinned: {ms:.3f} ms = {h.numel() * 4 / ms / 1e6:.1f} GB/s")
