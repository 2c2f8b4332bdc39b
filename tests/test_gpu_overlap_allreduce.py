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


def _train(dev, pipelined, world_reduce, steps=4, F=167, B=24, rank=0):
    """`steps` training steps; returns parameters, moments and the last loss.  pipelined: reducer.step(optimizer) -- the all-reduce
    and the AdamW of every early slice under the backward pass; else all-reduce (when world_reduce) followed by optimizer.step()."""
    import bbbp_amd
    from bbbp_amd import distributed as D
    from bbbp_amd.optim import AdamW
    from helpers import synth_inputs
    fp, img, y = (t.to(dev) for t in synth_inputs(40 + rank, 2 * B, F, 49152))
    torch.manual_seed(3)
    m = bbbp_amd.MixedInputModel(F, 128).to(dev).train()
    params = list(m.parameters())
    opt = AdamW(params, lr=1e-3, weight_decay=1e-5)
    reducer = D.OverlappedGradAllReduce(m, pipelined_step=pipelined) if (pipelined or world_reduce) else None
    world = D.world_size()
    torch.manual_seed(11)
    for i in range(steps):
        s = (i % 2) * B
        loss = bbbp_amd.MSELoss()(m(fp[s:s + B], img[s:s + B]).squeeze(), y[s:s + B])
        loss.backward()
        if pipelined:
            reducer.step(opt, params, grad_scale=1.0 / world)
        else:
            if world_reduce:
                reducer(params, average=False)
            opt.step(grad_scale=1.0 / world)
        opt.zero_grad(set_to_none=True)
    torch.cuda.synchronize()
    st = opt.state[params[0]]
    if reducer is not None:
        reducer.close()
    return (torch.cat([p.detach().flatten() for p in params]).cpu(), st["exp_avg"].flatten().cpu().clone(), int(st["step"]),
            float(loss.detach()))


def test_optimizer_step_pipelined_into_the_backward_pass_is_bit_identical(dev):
    """distributed.OverlappedGradAllReduce.step in a single process (no collectives, the pipelining stays): AdamW applied slice by
    slice on a side stream as soon as a slice's gradients are final and its parameters no longer read (the image-FC weight 1.2 ms
    before the pass ends) leaves exactly the parameters and moments of the one-launch step after four steps with dropout on -- a slice
    updated too early would change a later kernel of the same backward pass and every step after it."""
    p0, m0, s0, l0 = _train(dev, pipelined=False, world_reduce=False)
    p1, m1, s1, l1 = _train(dev, pipelined=True, world_reduce=False)
    assert s0 == s1 == 4 and l0 == l1
    assert torch.equal(p0, p1) and torch.equal(m0, m1)


def _pipelined_worker(rank, world, port):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    a = _train(dev, pipelined=False, world_reduce=True, steps=3, B=16, rank=rank)
    b = _train(dev, pipelined=True, world_reduce=True, steps=3, B=16, rank=rank)
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and a[2] == b[2] == 3
    ref = b[0].to(dev).clone()
    dist.broadcast(ref, 0)
    assert torch.equal(ref.cpu(), b[0]), "replicas stay identical"
    dist.destroy_process_group()
    return "ok"


def test_pipelined_allreduce_and_optimizer_two_ranks():
    from helpers import run_ranks
    assert run_ranks(_pipelined_worker, 2, timeout=300) == ["ok", "ok"]
