"""f32-MFMA vs split-bf16 form of the large GEMMs (128 x 128 tile plans): isolated time per launch on one GPU.

    python tools/bench_gemm_forms.py
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bbbp_amd import _lib, ops
from tools.bench_conv2 import timed

SHAPES = [  # (what, layout, M, N, K)
    ("F=2048 in_proj fwd", "nt", 512, 6144, 2048), ("F=2048 linear1 fwd", "nt", 512, 2048, 2048),
    ("F=2048 linear1 dgrad", "nn", 512, 2048, 2048), ("F=2048 linear1 wgrad", "tn", 2048, 2048, 512),
    ("F=2048 in_proj wgrad", "tn", 6144, 2048, 512), ("F=2048 in_proj dgrad", "nn", 512, 2048, 6144),
    ("image FC fwd", "nt", 512, 128, 65536), ("image FC dgrad", "nn", 512, 65536, 128), ("image FC wgrad", "tn", 128, 65536, 512),
    ("B=4096 linear1 fwd F=2048", "nt", 4096, 2048, 2048),
]


def main():
    dev = torch.device("cuda")
    L = _lib.lib()
    g = torch.Generator().manual_seed(3)
    for what, layout, M, N, K in SHAPES:
        a = torch.randn(M, K, generator=g); b = torch.randn(K, N, generator=g)
        if layout == "nt":
            args = (a.to(dev), b.t().contiguous().to(dev)); kw = dict(trans_b=True)
        elif layout == "nn":
            args = (a.to(dev), b.to(dev)); kw = {}
        else:
            args = (a.t().contiguous().to(dev), b.to(dev)); kw = dict(trans_a=True)
        out = torch.empty(M, N, device=dev)
        res = {}
        for form in (0, 1):
            L.bbbp_set_gemm_split_bf16(form)
            res[form] = (timed(lambda: ops.gemm(*args, out=out, **kw)), out.clone())
        want = a.double() @ b.double() if M * N * K <= 2 ** 34 else None
        fl = 2.0 * M * N * K
        line = f"{what:28s} {layout} {M}x{N}x{K}: f32 {res[0][0] * 1e3:7.1f} us ({fl / res[0][0] / 1e9:6.1f} TFLOP/s), split-bf16 {res[1][0] * 1e3:7.1f} us ({fl / res[1][0] / 1e9:6.1f} TFLOP/s)"
        if want is not None:
            sc = float(want.abs().max())
            line += f"; max err / max|C|: f32 {float((res[0][1].cpu().double() - want).abs().max()) / sc:.2e}, split-bf16 {float((res[1][1].cpu().double() - want).abs().max()) / sc:.2e}"
        print(line, flush=True)
        if os.environ.get("BBBP_GEMM_B3_PROBE") == "1":
            import ctypes
            L.bbbp_set_gemm_split_bf16(1)
            ops.gemm(*args, out=out, **kw); torch.cuda.synchronize()
            ph = (ctypes.c_uint64 * 7)()
            _lib.check(L.bbbp_gemm_split_bf16_phases(ph), "bbbp_gemm_split_bf16_phases")
            tot = sum(ph[:5]) or 1
            print("    work-group 0 / wave 0 cycles: load-issue %d (%.1f %%), LDS reads + MFMA %d (%.1f %%), barrier %d (%.1f %%), split + LDS writes %d (%.1f %%), "
                  "barrier %d (%.1f %%); %d cycles in %.1f us = %.2f GHz" % (tuple(x for i in range(5) for x in (ph[i], 100.0 * ph[i] / tot)) + (ph[5], ph[6] / 100.0, ph[5] / (ph[6] * 10.0 + 1e-9))), flush=True)
    L.bbbp_set_gemm_split_bf16(1)


if __name__ == "__main__":
    main()
