"""Ablation of the 64x64 GEMM kernel (needs the debug build of csrc/gemm.hip made by hand: act bits 0x100 skip MFMA,
0x200 skip global loads after the first stage, 0x400 skip LDS store + barrier)."""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bbbp_amd import _lib
dev = torch.device("cuda:0"); L = _lib.lib(); st = torch.cuda.current_stream().cuda_stream
ws = torch.empty(64 << 20, dtype=torch.uint8, device=dev)
def run(name, M, N, K, act, reps=300):
    A = torch.randn(M * K + 64, device=dev); B = torch.randn(N * K + 64, device=dev); C = torch.empty(M * N + 64, device=dev)
    def go(n):
        for _ in range(n):
            L.bbbp_gemm_f32(st, 0, 1, M, N, K, 1.0, A.data_ptr(), K, B.data_ptr(), K, C.data_ptr(), N, None, None, 0, act, 1, 0, 0, 0, 0, ws.data_ptr(), ws.numel())
    go(10); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); go(reps); e1.record(); torch.cuda.synchronize()
    print(f"{name:50s} M={M} N={N} K={K}: {e0.elapsed_time(e1)/reps*1e3:7.2f} us/call", flush=True)
for K in (16, 64, 167, 334):
    run("full", 512, 167, K, 0)
    run("no MFMA (1 fma instead)", 512, 167, K, 0x100)
    run("no global loads after first stage", 512, 167, K, 0x200)
    run("no LDS store + barrier in the loop", 512, 167, K, 0x600)
    run("no loads, no store/barrier, no MFMA (reads only)", 512, 167, K, 0x700)
