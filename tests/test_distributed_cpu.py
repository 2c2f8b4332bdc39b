"""CPU, gloo, world_size 2: the data-parallel helpers (flat single-collective gradient all-reduce, bucketed path,
parameter broadcast, batch sharding, prediction gather) give the single-process full-batch gradient."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _make_model(seed):
    torch.manual_seed(seed)
    return torch.nn.Sequential(torch.nn.Linear(12, 16), torch.nn.Tanh(), torch.nn.Linear(16, 1))


def _worker(rank, world, port, flat):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from bbbp_amd import distributed as D
    from bbbp_amd.models import flatten_parameters, flat_view_of
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    r, w, dev = D.init("gloo")
    assert (r, w, dev.type) == (rank, world, "cpu")
    model = _make_model(100 + rank)                      # different init per rank on purpose
    if flat:
        flatten_parameters(model)
    D.broadcast_parameters(model, src=0)
    g = torch.Generator().manual_seed(7)
    X, y = torch.randn(16, 12, generator=g), torch.randn(16, generator=g)
    sl = D.shard_batch(16, rank, world)
    out = model(X[sl]).squeeze(1)
    loss = ((out - y[sl]) ** 2).mean()
    if flat:                                             # gradients in one flat buffer, like MixedInputModel's backward
        grads = torch.autograd.grad(loss, list(model.parameters()))
        gflat = torch.cat([t.reshape(-1) for t in grads])
        off = 0
        for p in model.parameters():
            p.grad = gflat[off:off + p.numel()].view_as(p); off += p.numel()
        assert flat_view_of([p.grad for p in model.parameters()]) is not None
    else:
        loss.backward()
    ncoll = D.allreduce_gradients(model, average=True, bucket_bytes=256)
    preds = D.gather_predictions(out.detach())
    res = (rank, ncoll, [p.grad.clone() for p in model.parameters()], [p.detach().clone() for p in model.parameters()], preds)
    dist.barrier()
    dist.destroy_process_group()
    return res


@pytest.mark.parametrize("flat", [True, False])
def test_gloo_world2_matches_full_batch(flat):
    from helpers import run_ranks
    res = run_ranks(_worker, 2, args=(flat,), timeout=120)
    ref = _make_model(100)
    g = torch.Generator().manual_seed(7)
    X, y = torch.randn(16, 12, generator=g), torch.randn(16, generator=g)
    out = ref(X).squeeze(1)
    ((out - y) ** 2).mean().backward()
    for rank, ncoll, grads, params, preds in res:
        assert ncoll == (1 if flat else 2), ncoll              # one collective for flat gradients; 2 size-capped buckets otherwise
        for a, b in zip(params, ref.parameters()):
            assert torch.equal(a, b.detach())                    # broadcast from rank 0
        for a, b in zip(grads, ref.parameters()):
            torch.testing.assert_close(a, b.grad, rtol=1e-5, atol=1e-7)   # mean of shard gradients == full-batch gradient
        torch.testing.assert_close(preds, out.detach(), rtol=1e-6, atol=1e-7)


def test_shard_batch_edges():
    from bbbp_amd.distributed import shard_batch
    assert [shard_batch(10, r, 4) for r in range(4)] == [slice(0, 3), slice(3, 6), slice(6, 9), slice(9, 10)]
    assert shard_batch(2, 3, 4) == slice(2, 2)
    assert shard_batch(512, 7, 8) == slice(448, 512)
