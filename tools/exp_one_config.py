"""A few training steps of one configuration (for rocprofv3 --kernel-trace --stats): python tools/exp_one_config.py F B fused steps"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bbbp_amd
from bbbp_amd import _lib
from bbbp_amd.optim import AdamW
F, B, fused, steps = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
dev = torch.device("cuda:0")
_lib.lib().bbbp_set_fused_encoder(fused)
torch.manual_seed(0)
m = bbbp_amd.MixedInputModel(F, 128).to(dev).train()
opt = AdamW(m.parameters(), lr=1e-4, weight_decay=1e-5)
fp = torch.randn(B, F, device=dev); img = torch.randn(B, 49152, device=dev); y = torch.randn(B, device=dev)
crit = bbbp_amd.MSELoss()
for _ in range(steps):
    crit(m(fp, img).squeeze(), y).backward(); opt.step(); opt.zero_grad(set_to_none=True)
torch.cuda.synchronize()
