"""Forms of the 3 -> 32 @ 128x128 conv stage: agreement and isolated kernel time (one GPU).

    python tools/bench_conv1.py [B]         # default B = 512
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bbbp_amd import _lib, ops
from tools.bench_conv2 import timed


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    dev = torch.device("cuda")
    L = _lib.lib()
    g = torch.Generator().manual_seed(5)
    x = torch.rand(B, 3, 128, 128, generator=g).to(dev)
    w = (0.2 * torch.randn(32, 3, 3, 3, generator=g)).to(dev)
    bias = (0.1 * torch.randn(32, generator=g)).to(dev)
    gy = torch.randn(B, 32, 64, 64, generator=g).to(dev)
    ref = None
    old = L.bbbp_get_conv_winograd()
    for mask, name in ((0, "f32"), (32, "split-bf16 weight gradient"), (96, "split-bf16 forward + weight gradient")):
        L.bbbp_set_conv_winograd(mask)
        y, m = ops.conv3x3_relu_pool_fwd(x, w, bias)
        dw, db = ops.conv3x3_relu_pool_bwd_weight(x, gy, m)
        tf = timed(lambda: ops.conv3x3_relu_pool_fwd(x, w, bias))
        tw = timed(lambda: ops.conv3x3_relu_pool_bwd_weight(x, gy, m))
        if ref is None:
            ref = (y, m, dw, db)
        fb = B * (3 * 128 * 128 * 4 + 32 * 64 * 64 * 5)
        print(f"{name:38s} B={B}: fwd {tf:.3f} ms ({fb / tf / 1e6:.0f} GB/s), wgrad {tw:.3f} ms ({fb / tw / 1e6:.0f} GB/s); "
              f"max|y - f32| {float((y - ref[0]).abs().max()):.2e}, masks equal {bool((m == ref[1]).all())}, "
              f"max|dw - f32| {float((dw - ref[2]).abs().max()):.2e} of {float(ref[2].abs().max()):.2e}, "
              f"max|db - f32| {float((db - ref[3]).abs().max()):.2e} of {float(ref[3].abs().max()):.2e}", flush=True)
    L.bbbp_set_conv_winograd(old)


if __name__ == "__main__":
    main()
