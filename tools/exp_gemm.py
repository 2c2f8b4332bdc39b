"""Per-launch time of the encoder's GEMM shapes (back-to-back launches on one stream, ctypes direct)."""
import sys, os, time, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bbbp_amd import _lib
dev = torch.device("cuda:0")
L = _lib.lib()
st = torch.cuda.current_stream().cuda_stream
ws = torch.empty(64 << 20, dtype=torch.uint8, device=dev)
def run(name, ta, tb, M, N, K, lda, ldb, ldc, reps=200):
    A = torch.randn(max(M, K) * max(lda, 1) + 64, device=dev); B = torch.randn(max(N, K) * max(ldb, 1) + 64, device=dev)
    C = torch.empty(M * ldc + 64, device=dev)
    def go(n):
        for _ in range(n):
            L.bbbp_gemm_f32(st, ta, tb, M, N, K, 1.0, A.data_ptr(), lda, B.data_ptr(), ldb, C.data_ptr(), ldc, None, None, 0, 0, 1, 0, 0, 0, 0, ws.data_ptr(), ws.numel())
    go(10); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); go(reps); e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / reps * 1e3
    print(f"{name:34s} M={M:5d} N={N:5d} K={K:5d}: {us:7.2f} us/call  {2.0*M*N*K/us/1e6:7.2f} TFLOP/s", flush=True)
F = 167
run("QKV   x W_in^T   (NT)", 0, 1, 512, 3 * F, F, F, F, 3 * F)
run("scores Q K^T     (NT)", 0, 1, 512, 512, F, 3 * F, 3 * F, 512)
run("P V              (NN)", 0, 0, 512, F, 512, 512, 3 * F, F)
run("out_proj         (NT)", 0, 1, 512, F, F, F, F, F)
run("FFN1             (NT)", 0, 1, 512, 2048, F, F, F, 2048)
run("FFN2             (NT)", 0, 1, 512, F, 2048, 2048, 2048, F)
run("dW1 = dh^T y1    (TN)", 1, 0, 2048, F, 512, 2048, F, F)
run("dhff = dff W2    (NN)", 0, 0, 512, 2048, F, F, 2048, 2048)
run("img fc fwd       (NT)", 0, 1, 512, 128, 65536, 65536, 65536, 256)
run("img fc dX        (NN)", 0, 0, 512, 65536, 128, 256, 65536, 65536)
run("img fc dW        (TN)", 1, 0, 128, 65536, 512, 256, 65536, 65536)
run("square 4096      (NT)", 0, 1, 4096, 4096, 4096, 4096, 4096, 4096, reps=5)
