"""Micro-experiment: latency of small kernels on stream B while a persistent conv kernel runs on stream A."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bbbp_amd import ops, _lib
dev = torch.device("cuda:0")
L = _lib.lib()
B = 512
x1 = torch.randn(B, 32, 64, 64, device=dev); w = torch.randn(64, 32, 3, 3, device=dev) * 0.1; b = torch.zeros(64, device=dev)
z = torch.randn(512, 167, device=dev); g = torch.ones(167, device=dev); be = torch.zeros(167, device=dev)
a = torch.randn(512, 167, device=dev); wq = torch.randn(501, 167, device=dev)
sA, sB = torch.cuda.Stream(), torch.cuda.Stream()

def small(n):
    for _ in range(n):
        ops.layernorm_fwd(z, None, g, be)
        ops.gemm(a, wq, trans_b=True)

def run(label, with_conv, reserved, pad):
    L.bbbp_set_partition(reserved, pad)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    c0, c1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    if with_conv:
        with torch.cuda.stream(sA):
            c0.record()
            for _ in range(6):
                ops.conv3x3_relu_pool_fwd(x1, w, b)
            c1.record()
    with torch.cuda.stream(sB):
        time.sleep(0.0005)
        e0.record()
        small(40)
        e1.record()
    torch.cuda.synchronize()
    L.bbbp_set_partition(0, 0)
    msg = f"{label:40s} small pair (ln+gemm) avg {e0.elapsed_time(e1)/40*1e3:7.1f} us"
    if with_conv:
        msg += f"   conv avg {c0.elapsed_time(c1)/6*1e3:7.1f} us"
    print(msg, flush=True)

for _ in range(2):
    small(10); ops.conv3x3_relu_pool_fwd(x1, w, b)
torch.cuda.synchronize()
run("alone", False, 0, 0)
run("alone, pad 48K", False, 0, 48 * 1024)
run("beside conv, shared CUs", True, 0, 0)
run("beside conv, reserved 32, pad 48K", True, 32, 48 * 1024)
run("beside conv, reserved 64, pad 48K", True, 64, 48 * 1024)
run("beside conv, reserved 128, pad 48K", True, 128, 48 * 1024)
run("beside conv, reserved 32, no pad", True, 32, 0)
