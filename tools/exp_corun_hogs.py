"""What costs the latency-bound launches their time beside the image branch?  A chain of small kernels (LayerNorm + the in_proj GEMM,
the F = 167 encoder's shapes) on stream B while stream A runs one of several "hogs" that differ in what they do to the chip:
write-heavy streaming (conv forward kernels), read-heavy matrix work with almost no writes (the image FC forward, K = 65536), or nothing.

    python tools/exp_corun_hogs.py
"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bbbp_amd import _lib, ops

dev = torch.device("cuda:0")
L = _lib.lib()
B = 512
x2 = torch.relu(torch.randn(B, 32, 64, 64, device=dev)); w2 = torch.randn(64, 32, 3, 3, device=dev) * 0.1; b2 = torch.zeros(64, device=dev)
x1 = torch.rand(B, 3, 128, 128, device=dev); w1 = torch.randn(32, 3, 3, 3, device=dev) * 0.1; b1 = torch.zeros(32, device=dev)
pool2 = torch.randn(B, 65536, device=dev); wfc = torch.randn(128, 65536, device=dev) * 0.01; outfc = torch.empty(B, 128, device=dev)
gy2 = torch.randn(B, 64, 32, 32, device=dev)
z = torch.randn(512, 167, device=dev); g = torch.ones(167, device=dev); be = torch.zeros(167, device=dev)
a = torch.randn(512, 167, device=dev); wq = torch.randn(501, 167, device=dev)
sA, sB = torch.cuda.Stream(), torch.cuda.Stream()
_, m2 = ops.conv3x3_relu_pool_fwd(x2, w2, b2)


outs = [torch.empty(512, 501, device=dev) for _ in range(4)]


def small(n):
    """n in_proj-shaped GEMMs (512 x 501 x 167, the latency path) enqueued by ONE C call: launch-to-launch time on the GPU, not the
    ~16 us per call that Python + ctypes cost."""
    ops.gemm_grouped([dict(a=a, b=wq, trans_b=True, out=outs[i % 4]) for i in range(n)])


# (label -> (launch, repetitions)): repetitions chosen so that the hog outlasts the 240-launch chain (~5 ms)
HOGS = {
    "nothing": None,
    "conv2 forward, split-bf16 (268 MB in, 168 MB out)": (lambda: ops.conv3x3_relu_pool_fwd(x2, w2, b2), 14),
    "conv2 weight gradient, split-bf16 (436 MB in, 0.3 MB out)": (lambda: ops.conv3x3_relu_pool_bwd_weight(x2, gy2, m2), 14),
    "conv1 forward, f32 MFMA (101 MB in, 336 MB out)": (lambda: ops.conv3x3_relu_pool_fwd(x1, w1, b1), 28),
    "image FC forward, split-bf16 GEMM K=65536 (168 MB in, 0.3 MB out + 33 MB slabs)": (lambda: ops.gemm(pool2, wfc, trans_b=True, out=outfc), 120),
}


def run(label, hog):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    c0, c1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 1
    if hog is not None:
        hog, reps = hog
        with torch.cuda.stream(sA):
            c0.record()
            for _ in range(reps):
                hog()
            c1.record()
    with torch.cuda.stream(sB):
        time.sleep(0.0005)
        e0.record()
        small(240)
        e1.record()
    torch.cuda.synchronize()
    msg = f"{label:82s} in_proj GEMM {e0.elapsed_time(e1) / 240 * 1e3:7.1f} us per launch"
    if hog is not None:
        msg += f"   hog {c0.elapsed_time(c1) / reps * 1e3:7.1f} us per launch"
    print(msg, flush=True)


for h in HOGS.values():
    if h is not None:
        h[0]()
small(10)
torch.cuda.synchronize()
for _ in range(2):
    for label, h in HOGS.items():
        run(label, h)
