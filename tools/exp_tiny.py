"""Micro-experiment 2: per-kernel cost of a chain of trivial kernels on stream B, alone vs beside a persistent conv."""
import sys, os, time, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bbbp_amd import ops, _lib
dev = torch.device("cuda:0")
L = _lib.lib()
B = 512
x1 = torch.randn(B, 32, 64, 64, device=dev); w = torch.randn(64, 32, 3, 3, device=dev) * 0.1; b = torch.zeros(64, device=dev)
y = torch.empty(B, 64, 32, 32, device=dev); mask = torch.empty(B, 64, 32, 32, device=dev, dtype=torch.uint8)
wsb = L.bbbp_conv3x3_workspace_bytes(B, 32, 64, 64, 64); ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
tiny = torch.ones(64, device=dev)
big = torch.ones(512 * 2048, device=dev)
sA, sB = torch.cuda.Stream(), torch.cuda.Stream()
N = 300
def chain(t, n):
    st = sB.cuda_stream
    p, cnt = t.data_ptr(), t.numel()
    for _ in range(n):
        L.bbbp_scale(st, p, cnt, 1.0)
def conv(n):
    for _ in range(n):
        L.bbbp_conv3x3_relu_pool_fwd(sA.cuda_stream, x1.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(), mask.data_ptr(), B, 32, 64, 64, 64, ws.data_ptr(), wsb)
def run(label, t, with_conv):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    if with_conv:
        conv(8)
    time.sleep(0.001)
    h0 = time.perf_counter()
    e0.record(sB); chain(t, N); e1.record(sB)
    h1 = time.perf_counter()
    torch.cuda.synchronize()
    print(f"{label:44s} GPU {e0.elapsed_time(e1)/N*1e3:7.2f} us/kernel   host enqueue {(h1-h0)/N*1e6:6.2f} us/kernel", flush=True)
conv(2); chain(tiny, 10); torch.cuda.synchronize()
run("tiny (1 WG) alone", tiny, False)
run("tiny (1 WG) beside conv", tiny, True)
run("1M-element scale (4096 WG) alone", big, False)
run("1M-element scale (4096 WG) beside conv", big, True)
