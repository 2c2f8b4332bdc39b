"""GPU, 2 ranks on one device (gloo; the real runs use RCCL): the all-reduce that starts under the backward pass
(distributed.OverlappedGradAllReduce) sums exactly what the plain single all-reduce sums."""
import os
import socket
import sys

import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port):
    if True:
        sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
        os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        import torch.distributed as dist
        import bbbp_amd
        from bbbp_amd import distributed as D
        from helpers import synth_inputs
        dist.init_process_group("gloo", rank=rank, world_size=world)
        dev = torch.device("cuda:0")
        torch.cuda.set_device(dev)
        F, B = 167, 16
        fp, img, y = synth_inputs(5 + rank, B, F, 49152)          # different data per rank
        results = []
        for overlapped in (False, True):
            torch.manual_seed(3)
            m = bbbp_amd.MixedInputModel(F, 128).to(dev).train()
            params = list(m.parameters())
            reducer = D.OverlappedGradAllReduce(m)
            torch.manual_seed(11)                                  # same dropout seeds in both runs
            for step in range(2):
                loss = bbbp_amd.MSELoss()(m(fp.to(dev), img.to(dev)).squeeze(), y.to(dev))
                loss.backward()
                n = reducer(params, average=True) if overlapped else D.allreduce_gradients(params, average=True)
                g = torch.cat([p.grad.flatten() for p in params]).clone()
                for p in params:
                    p.grad = None
            results.append((n, g.cpu()))
        (n0, g0), (n1, g1) = results
        # one collective vs image-FC weight + six encoder layers (last layer first) + the two remaining slices + conv tensors
        assert n0 == 1 and n1 == 10, (n0, n1)
        sched = reducer.schedule()
        assert [b for _, b, _, _ in sched] == [0, 7, 6, 5, 4, 3, 2, 1, 1, None]
        covered = sorted((lo, hi) for _, _, lo, hi in sched)
        assert covered[0][0] == 0 and covered[-1][1] == sum(p.numel() for p in params)
        assert all(a[1] == b[0] for a, b in zip(covered, covered[1:])), "the slices tile the flat gradient buffer exactly once"
        assert torch.equal(g0, g1), f"rank {rank}: max diff {float((g0 - g1).abs().max()):.3e}"
        # and it really is the mean over ranks: rank 1's view equals rank 0's
        ref = g1.to(dev).clone()
        dist.broadcast(ref, 0)
        assert torch.equal(ref.cpu(), g1)
        dist.destroy_process_group()
        return "ok"


def test_overlapped_allreduce_equals_plain_allreduce():
    from helpers import run_ranks
    assert run_ranks(_worker, 2, timeout=300) == ["ok", "ok"]
