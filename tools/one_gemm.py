"""One large GEMM shape repeated a few times (for rocprofv3 --pmc passes): python tools/one_gemm.py [layout M N K form]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bbbp_amd import _lib, ops

layout = sys.argv[1] if len(sys.argv) > 1 else "nt"
M, N, K = (int(v) for v in sys.argv[2:5]) if len(sys.argv) > 4 else (4096, 2048, 2048)
form = int(sys.argv[5]) if len(sys.argv) > 5 else 1
dev = torch.device("cuda")
_lib.lib().bbbp_set_gemm_split_bf16(form)
g = torch.Generator().manual_seed(3)
a = torch.randn(M, K, generator=g); b = torch.randn(K, N, generator=g)
if layout == "nt":
    args = (a.to(dev), b.t().contiguous().to(dev)); kw = dict(trans_b=True)
elif layout == "nn":
    args = (a.to(dev), b.to(dev)); kw = {}
else:
    args = (a.t().contiguous().to(dev), b.to(dev)); kw = dict(trans_a=True)
out = torch.empty(M, N, device=dev)
for _ in range(5):
    ops.gemm(*args, out=out, **kw)
torch.cuda.synchronize()
