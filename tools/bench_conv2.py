"""Direct vs Winograd form of the 32 -> 64 @ 64x64 conv stage: agreement and isolated kernel time (one GPU).

    python tools/bench_conv2.py [B]         # default B = 512
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bbbp_amd import _lib, ops


def timed(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    dev = torch.device("cuda")
    L = _lib.lib()
    g = torch.Generator().manual_seed(5)
    for b in (1, 3, 40):
        x = torch.relu(torch.randn(b, 32, 64, 64, generator=g)).to(dev)
        w = (0.2 * torch.randn(64, 32, 3, 3, generator=g)).to(dev)
        bias = (0.1 * torch.randn(64, generator=g)).to(dev)
        gy = torch.randn(b, 64, 32, 32, generator=g).to(dev)
        L.bbbp_set_conv_winograd(0)
        y0, m0 = ops.conv3x3_relu_pool_fwd(x, w, bias)
        dx0 = ops.conv3x3_relu_pool_bwd_data(gy, m0, w)
        for form, nm in ((3, "winograd"), (12, "split-bf16")):
            L.bbbp_set_conv_winograd(form)
            y1, m1 = ops.conv3x3_relu_pool_fwd(x, w, bias)
            dx1 = ops.conv3x3_relu_pool_bwd_data(gy, m0, w)
            torch.cuda.synchronize()
            print(f"B={b} {nm}: fwd max|diff| {float((y0 - y1).abs().max()):.3e} (max|y| {float(y0.abs().max()):.3f}), "
                  f"mask mismatch {float((m0 != m1).float().mean()):.2e}, dgrad max|diff| {float((dx0 - dx1).abs().max()):.3e} "
                  f"(max|dx| {float(dx0.abs().max()):.3f})", flush=True)
    x = torch.relu(torch.randn(B, 32, 64, 64, generator=g)).to(dev)
    w = (0.2 * torch.randn(64, 32, 3, 3, generator=g)).to(dev)
    bias = (0.1 * torch.randn(64, generator=g)).to(dev)
    gy = torch.randn(B, 64, 32, 32, generator=g).to(dev)
    flops = 2.0 * B * 64 * 64 * 64 * 288
    ref_dw = None
    for mask, name in ((0, "direct"), (3, "winograd"), (28, "split-bf16")):
        L.bbbp_set_conv_winograd(mask)
        y, m = ops.conv3x3_relu_pool_fwd(x, w, bias)
        tf = timed(lambda: ops.conv3x3_relu_pool_fwd(x, w, bias))
        td = timed(lambda: ops.conv3x3_relu_pool_bwd_data(gy, m, w))
        tw = timed(lambda: ops.conv3x3_relu_pool_bwd_weight(x, gy, m))
        dw, db = ops.conv3x3_relu_pool_bwd_weight(x, gy, m)
        if ref_dw is None:
            ref_dw, ref_m = dw, m
        print(f"{name:9s} B={B}: fwd {tf:.3f} ms ({flops / tf / 1e9:.1f} TFLOP/s direct-equivalent), dgrad {td:.3f} ms "
              f"({flops / td / 1e9:.1f}), wgrad {tw:.3f} ms ({flops / tw / 1e9:.1f}; max|dw - direct| "
              f"{float((dw - ref_dw).abs().max()):.2e} of {float(ref_dw.abs().max()):.2e}, masks equal {bool((m == ref_m).all())})", flush=True)
        if mask == 28 and os.environ.get("BBBP_B3_PROBE") == "1":
            import ctypes
            for nm, fn in (("fwd", lambda: ops.conv3x3_relu_pool_fwd(x, w, bias)), ("dgrad", lambda: ops.conv3x3_relu_pool_bwd_data(gy, m, w))):
                fn(); torch.cuda.synchronize()
                ph = (ctypes.c_uint64 * 4)()
                _lib.check(L.bbbp_conv_b3_phases(ph), "bbbp_conv_b3_phases")
                tot = sum(ph) or 1
                print(f"  split-bf16 {nm} phases (work-group 0, wave 0): load-issue {ph[0]} ({100 * ph[0] / tot:.1f} %), MFMA block {ph[1]} ({100 * ph[1] / tot:.1f} %), "
                      f"split+LDS+barrier {ph[2]} ({100 * ph[2] / tot:.1f} %), epilogue {ph[3]} ({100 * ph[3] / tot:.1f} %)", flush=True)
        if mask == 3 and os.environ.get("BBBP_WINO_PROBE") == "1":
            import ctypes
            for what, fn in (("fwd", lambda: ops.conv3x3_relu_pool_fwd(x, w, bias)), ("dgrad", lambda: ops.conv3x3_relu_pool_bwd_data(gy, m, w))):
                fn(); torch.cuda.synchronize()
                ph = (ctypes.c_uint64 * 4)()
                _lib.check(L.bbbp_conv_winograd_phases(ph), "phases")
                cyc, ticks = ctypes.c_uint64(0), ctypes.c_uint64(0)
                _lib.check(L.bbbp_conv_last_clock(ctypes.byref(cyc), ctypes.byref(ticks)), "clock")
                tot = sum(ph)
                print(f"   {what}: work-group 0 cycles {cyc.value}: init {ph[0]} ({100 * ph[0] / tot:.1f} %), k-steps {ph[1]} ({100 * ph[1] / tot:.1f} %), "
                      f"stage hand-over {ph[2]} ({100 * ph[2] / tot:.1f} %), output transform {ph[3]} ({100 * ph[3] / tot:.1f} %)", flush=True)
    L.bbbp_set_conv_winograd(0)


if __name__ == "__main__":
    main()
