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


def _exact_protocol_worker(rank, world, port):
    """The exact-global-batch engine's collective protocol on HOST buffers: the ctypes callback the engine would call
    (models._collective_fn: byte offsets into the call's workspace, floats per rank) moves K | V and dK | dV between two gloo ranks,
    the attention arithmetic in between is plain torch.  csrc/engine.hip does exactly these steps around its HIP launches."""
    import ctypes
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from bbbp_amd import distributed as D
    from bbbp_amd import models
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    D.init("gloo")
    Bg, F = 10, 6
    Bl = Bg // world
    g = torch.Generator().manual_seed(3)
    q, k, v, dctx = (torch.randn(Bg, F, generator=g) for _ in range(4))
    sl = D.shard_batch(Bg, rank, world)
    n = Bl * 2 * F                                        # floats per rank in the K | V exchange
    off_kvg, off_dkvg, off_dkvl = 64, 64 + 4 * n * world, 64 + 8 * n * world
    ws = torch.zeros(off_dkvl + 4 * n, dtype=torch.uint8)
    f32 = lambda off, cnt: ws[off:off + 4 * cnt].view(torch.float32)
    call = models._CollectiveCall(dist.group.WORLD, world, rank, False)
    call.ws = ws
    handle = models._collective_register(call)
    fn = models._collective_fn()
    # forward: pack this rank's K | V rows into its slot, all-gather in place
    f32(off_kvg + 4 * n * rank, n).view(Bl, 2 * F).copy_(torch.cat([k[sl], v[sl]], dim=1))
    assert fn(handle, 0, 0, 0, off_kvg + 4 * n * rank, off_kvg, n, None) == 0
    kvg = f32(off_kvg, n * world).view(Bg, 2 * F)
    assert torch.equal(kvg, torch.cat([k, v], dim=1))
    # this rank's queries against ALL keys; backward of ctx = softmax(q k^T) v for the local rows
    scale = F ** -0.5
    p = torch.softmax(scale * q[sl] @ kvg[:, :F].t(), dim=1)
    dv_all = p.t() @ dctx[sl]
    dp = dctx[sl] @ kvg[:, F:].t()
    ds = p * (dp - (dp * p).sum(dim=1, keepdim=True))
    dq = scale * ds @ kvg[:, :F]
    dk_all = scale * ds.t() @ q[sl]
    f32(off_dkvg, n * world).view(Bg, 2 * F).copy_(torch.cat([dk_all, dv_all], dim=1))
    assert fn(handle, 1, 1, 0, off_dkvg, off_dkvl, n, None) == 0
    dkv_local = f32(off_dkvl, n).view(Bl, 2 * F).clone()
    assert fn(handle + 1000, 0, 0, 0, 0, 0, 1, None) == 2      # unknown handle: refused, nothing touched
    models._COLLECTIVE_CALLS.pop(handle)
    dist.barrier()
    dist.destroy_process_group()
    return rank, dq, dkv_local


def test_exact_batch_collective_protocol_on_gloo_world2():
    from helpers import run_ranks
    res = run_ranks(_exact_protocol_worker, 2, timeout=120)
    Bg, F = 10, 6
    g = torch.Generator().manual_seed(3)
    q, k, v, dctx = (torch.randn(Bg, F, generator=g).requires_grad_(True) for _ in range(4))
    ctx = torch.softmax(F ** -0.5 * q @ k.t(), dim=1) @ v
    ctx.backward(dctx.detach())
    for rank, dq, dkv in res:
        rows = slice(rank * 5, rank * 5 + 5)
        torch.testing.assert_close(dq, q.grad[rows], rtol=1e-5, atol=1e-6)
        torch.testing.assert_close(dkv[:, :F], k.grad[rows], rtol=1e-5, atol=1e-6)
        torch.testing.assert_close(dkv[:, F:], v.grad[rows], rtol=1e-5, atol=1e-6)


def test_shard_batch_edges():
    from bbbp_amd.distributed import shard_batch
    assert [shard_batch(10, r, 4) for r in range(4)] == [slice(0, 3), slice(3, 6), slice(6, 9), slice(9, 10)]
    assert shard_batch(2, 3, 4) == slice(2, 2)
    assert shard_batch(512, 7, 8) == slice(448, 512)


def _exact_seed_worker(rank, world, port):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from bbbp_amd import distributed as D
    from bbbp_amd.variants import ExactBatchMixedInputModel
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    D.init("gloo")
    torch.manual_seed(20250113)                          # every rank seeds the host RNG alike, as bench.py and same-init setups do
    model = ExactBatchMixedInputModel(64, 128).train()
    seeds = [int(model._descriptor(8).seed) for _ in range(3)]
    model.eval()
    eval_seed = int(model._descriptor(8).seed)
    dist.barrier()
    dist.destroy_process_group()
    return rank, seeds, eval_seed


def test_exact_batch_ranks_draw_different_dropout_seeds_on_gloo_world2():
    """in exact-global-batch mode every rank took its dropout seed from an identically seeded host RNG and the kernels
    index their Philox streams by LOCAL row, so local row i drew the same masks on every rank.  The rank is now mixed into the call's
    seed: same host stream on both ranks, different seeds in their descriptors; rank 0 keeps the host stream's value (the single-process
    behaviour); eval-mode descriptors carry no seed."""
    from helpers import run_ranks
    res = sorted(run_ranks(_exact_seed_worker, 2, timeout=180))
    (r0, s0, e0), (r1, s1, e1) = res
    assert (r0, r1) == (0, 1) and e0 == 0 and e1 == 0
    assert all(a != b for a, b in zip(s0, s1)) and len(set(s0 + s1)) == 6
    torch.manual_seed(20250113)
    from bbbp_amd.variants import ExactBatchMixedInputModel
    ExactBatchMixedInputModel(64, 128)                   # consumes the same init stream
    want = [int(torch.randint(0, 2 ** 62, (1,)).item()) for _ in range(3)]
    assert s0 == want
    assert s1 == [((w + 0x9E3779B97F4A7C15) & (2 ** 62 - 1)) or 1 for w in want]


def _comm_diag_worker(rank, world, port):
    import sys
    import time
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import bench
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    state = {"cus": -1, "calls": []}
    grads = torch.ones(1 << 16)

    def set_cus(c):
        state["calls"].append(c); state["cus"] = c
        return 0

    def no_coll(i):
        time.sleep(0.004)

    def overlapped(i):                                   # rank 1 is the slow one at 0 reserved CUs; both are faster with 8
        time.sleep(0.004 + (0.030 if state["cus"] == 0 else 0.005) * (1 + rank))     # (no collective inside: gloo's jitter here is 10-30 ms)

    out = bench.comm_diagnostics(dist, rank, world, dist.barrier, overlapped, no_coll, lambda i: dist.all_reduce(grads.clone()), set_cus,
                                 grads.numel() * 4, candidates=(0, 8), steps=3, warmup=1)
    # the same pass with a plain-schedule candidate that beats both overlapped ones: the schedule switches, the CUs go back to 0
    state2 = {"cus": -1, "calls": [], "schedule": None}

    def set_cus2(c):
        state2["cus"] = c; state2["calls"].append(c)
        return 0

    out2 = bench.comm_diagnostics(dist, rank, world, dist.barrier, overlapped, no_coll, lambda i: dist.all_reduce(grads.clone()), set_cus2,
                                  grads.numel() * 4, candidates=(0, 8), steps=3, warmup=1,
                                  step_plain=lambda i: time.sleep(0.006), set_schedule=lambda name: state2.__setitem__("schedule", name))
    dist.barrier()
    dist.destroy_process_group()
    return rank, out, state, out2, state2


def test_bench_comm_diagnostics_fields_and_rank0_choice_on_gloo_world2():
    """bench.py's N-rank diagnostic pass on the CPU with stand-in steps: every field of the `comm` object is
    present, the times are the MAX over ranks (identical on both), the reserved-CU candidate with the smaller overlapped step is chosen
    by rank 0 and left set on EVERY rank, exposed = overlapped[chosen] - step without collectives."""
    from helpers import run_ranks
    res = sorted(run_ranks(_comm_diag_worker, 2, timeout=180), key=lambda r: r[0])
    (_, a, sa, a2, sa2), (_, b, sb, b2, sb2) = res
    for k in ("exposed_ms", "step_no_collective_ms", "plain_allreduce_ms", "comm_cus_chosen", "bytes", "algbw_GBps", "busbw_GBps", "overlapped_ms"):
        assert k in a and k in b, k
    assert a == b                                        # MAX over ranks + rank 0's broadcast choice: the same object everywhere
    assert a["comm_cus_chosen"] == 8 and sa["cus"] == 8 and sb["cus"] == 8 and sa["calls"] == [0, 8, 8]
    assert a["bytes"] == 4 << 16 and a["algbw_GBps"] > 0 and abs(a["busbw_GBps"] - a["algbw_GBps"]) < 1e-2 + 0.01 * a["algbw_GBps"]   # 2 (N-1) / N = 1
    assert a["overlapped_ms"]["0"] > a["overlapped_ms"]["8"] > a["step_no_collective_ms"] >= 3.9
    assert abs(a["exposed_ms"] - (a["overlapped_ms"]["8"] - a["step_no_collective_ms"])) < 1e-3
    assert 9.0 <= a["exposed_ms"] <= 20.0                # rank 1 sleeps 10 ms more than the collective-free step
    assert "schedule_chosen" not in a                    # no plain candidate offered: the overlapped schedule is not in question
    # with a plain-schedule candidate (6 ms) that beats the best overlapped one (14 ms): chosen everywhere, reserved CUs back to 0
    assert a2 == b2 and a2["schedule_chosen"] == "plain" and a2["comm_cus_chosen"] == 0 and 5.9 <= a2["plain_step_ms"] <= 12.0
    assert sa2["schedule"] == sb2["schedule"] == "plain" and sa2["cus"] == sb2["cus"] == 0
    assert abs(a2["exposed_ms"] - (a2["plain_step_ms"] - a2["step_no_collective_ms"])) < 1e-3
