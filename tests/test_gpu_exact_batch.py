"""GPU, 2 ranks on one device (gloo carries the CUDA tensors; the real runs use RCCL, one GPU per rank): the
exact-global-batch mode -- keys/values gathered across ranks in every encoder layer, BatchNorm on global statistics,
gradients averaged -- reproduces the single-process step at the full batch (mode 2)."""
import os
import socket
import sys

import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _zero_dropout(m):
    for mod in m.modules():
        if isinstance(mod, torch.nn.Dropout):
            mod.p = 0.0
        if isinstance(mod, torch.nn.MultiheadAttention):
            mod.dropout = 0.0


def _worker(rank, world, port):
    if True:
        sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
        os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        import torch.distributed as dist
        import bbbp_amd
        from bbbp_amd import distributed as D
        from bbbp_amd.variants import ExactBatchMixedInputModel
        from helpers import synth_inputs
        dist.init_process_group("gloo", rank=rank, world_size=world)
        dev = torch.device("cuda:0")
        torch.cuda.set_device(dev)
        F, B = 64, 12                                   # nhead 8, head_dim 8; shards of 6 rows
        fp, img, y = synth_inputs(99, B, F, 49152)
        torch.manual_seed(17)
        m = ExactBatchMixedInputModel(F, 128).to(dev)
        _zero_dropout(m); m.train()
        sl = D.shard_batch(B, rank, world)
        out = m(fp[sl].to(dev), img[sl].to(dev))
        loss = torch.nn.MSELoss()(out.squeeze(), y[sl].to(dev))
        loss.backward()
        D.allreduce_gradients(m, average=True)
        allout = D.gather_predictions(out.detach().reshape(-1))
        # reference: the fused single-GPU model at the full batch, same seed
        torch.manual_seed(17)
        ref = bbbp_amd.MixedInputModel(F, 128).to(dev)
        _zero_dropout(ref); ref.train()
        rout = ref(fp.to(dev), img.to(dev))
        torch.nn.MSELoss()(rout.squeeze(), y.to(dev)).backward()
        def close(a, b, what, rtol=2e-4, afrac=1e-4):
            a = a.double().cpu(); b = b.double().cpu()
            tol = rtol * b.abs() + afrac * b.abs().max() + 1e-30
            assert bool(((a - b).abs() <= tol).all()), f"rank {rank} {what}: max err {float((a - b).abs().max()):.3e} scale {float(b.abs().max()):.3e}"
        close(allout, rout.detach().reshape(-1), "outputs")
        for (k, p), (_, r) in zip(m.named_parameters(), ref.named_parameters()):
            if k.startswith("attention_fusion."):
                continue                                  # rounding-noise gradients (Parity)
            close(p.grad, r.grad, k)
        for k in ("fc.2.running_mean", "fc.2.running_var"):
            close(m.state_dict()[k], ref.state_dict()[k], k, rtol=1e-4)
        dist.barrier()
        dist.destroy_process_group()
        return "ok"


def test_two_ranks_equal_single_process_full_batch(dev):
    from helpers import run_ranks
    assert run_ranks(_worker, 2, timeout=300) == ["ok", "ok"]
