"""The wide/deep variant (Models/multi_input_data_regression_opt_transformer_cnn_opt_20250107_network.py:109-174: 12-layer encoder, 3-stage CNN
64/128/256, MultiModalAttentionFusion, 6-layer head) as one training step at B = 256: ms per step, molecules/s.  `rocprofv3 --kernel-trace
--stats -- python3 tools/bench_wide_deep.py` gives the kernel table (profiles/r04_kernel_stats_wide_deep.csv)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bbbp_amd import variants
from bbbp_amd.optim import AdamW

dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
torch.manual_seed(0)
m = variants.WideDeepMixedInputModel(167, 128).to(dev).train()
GRAPH = os.environ.get("BBBP_WIDE_GRAPH", "0") == "1"
opt = AdamW(m.parameters(), lr=1e-4, weight_decay=1e-5, capturable=GRAPH)
fp = torch.randn(B, 167, device=dev); img = torch.randn(B, 49152, device=dev); y = torch.randn(B, device=dev)


if GRAPH:
    from bbbp_amd.training import GraphedTrainStep
    graphed = GraphedTrainStep(m, opt)


def step():
    if GRAPH:
        graphed(fp, img, y)
    else:
        torch.nn.MSELoss()(m(fp, img).squeeze(), y).backward(); opt.step(); opt.zero_grad(set_to_none=True)


for _ in range(3): step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(steps): step()
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / steps
print("optimizer launches per step:", "1 (multi-tensor table)" if opt._tables.get(0) else "one per tensor", flush=True)
print(f"wide/deep B={B} train: {dt * 1e3:.2f} ms/step  {B / dt:.0f} molecules/s  (graph {int(GRAPH)}, BBBP_WIDE_OVERLAP={os.environ.get('BBBP_WIDE_OVERLAP', '1')}, conv mask {os.environ.get('BBBP_CONV_WINOGRAD', 'default')})", flush=True)
