"""One process per GPU on one 8 x MI355X node; gradients are summed with ONE RCCL all-reduce over xGMI per step.

The reference is single-process; this is the new data-parallel capability.  Mode implemented: replica data
parallel -- every rank runs the reference model on its own local batch (attention across the LOCAL batch, BatchNorm on
LOCAL statistics: exactly what torch DDP would do to the reference) and gradients are averaged.  Because the model keeps
all gradients in one flat buffer (models.py), the collective is a single 53.9 MB (F=167) all-reduce instead of 106 small
ones; xGMI is a point-to-point mesh, so one large message per peer is the shape RCCL handles best.
Backend "nccl" is RCCL on ROCm; "gloo" runs the same code on CPU tensors for tests.
"""
from __future__ import annotations

import os
from typing import Iterable, List, Optional

import torch
import torch.distributed as dist

from .models import flat_view_of


def init(backend: Optional[str] = None) -> tuple:
    """Initialise torch.distributed from the torchrun environment.  Returns (rank, world_size, device)."""
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29500")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC only on this driver
    use_gpu = torch.cuda.is_available()
    if backend is None:
        backend = "nccl" if use_gpu else "gloo"
    device = torch.device("cuda", local) if (use_gpu and backend == "nccl") else torch.device("cpu")
    if device.type == "cuda":
        torch.cuda.set_device(device)
    if world > 1 and not dist.is_initialized():
        kwargs = {"device_id": device} if device.type == "cuda" else {}
        dist.init_process_group(backend, rank=rank, world_size=world, **kwargs)
    return rank, world, device


def world_size(group=None) -> int:
    return dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1


def shard_batch(n: int, rank: int, world: int) -> slice:
    """Contiguous shard of a global batch of n samples for this rank (last ranks get the shorter shards)."""
    per = (n + world - 1) // world
    return slice(min(n, rank * per), min(n, (rank + 1) * per))


def _params_with_grad(params: Iterable[torch.Tensor]) -> List[torch.Tensor]:
    return [p for p in params if p.grad is not None]


def allreduce_gradients(module_or_params, average: bool = True, group=None, bucket_bytes: int = 64 << 20) -> int:
    """Sum (or average) ``.grad`` over all ranks.  Returns the number of collectives issued.
    Fast path: gradients that are consecutive views of one buffer -> ONE all-reduce in place.
    General path: size-capped buckets, flattened, reduced and scattered back."""
    params = list(module_or_params.parameters()) if isinstance(module_or_params, torch.nn.Module) else list(module_or_params)
    params = _params_with_grad(params)
    world = world_size(group)
    if world == 1 or not params:
        return 0
    flat = flat_view_of([p.grad for p in params])
    if flat is not None:
        dist.all_reduce(flat, group=group)
        if average:
            flat.div_(world)
        return 1
    n_coll, bucket, size = 0, [], 0

    def flush():
        nonlocal n_coll, bucket, size
        if not bucket:
            return
        buf = torch.cat([p.grad.reshape(-1) for p in bucket])
        dist.all_reduce(buf, group=group)
        if average:
            buf.div_(world)
        off = 0
        for p in bucket:
            n = p.grad.numel()
            p.grad.copy_(buf[off:off + n].view_as(p.grad))
            off += n
        n_coll += 1
        bucket, size = [], 0

    for p in params:
        b = p.grad.numel() * p.grad.element_size()
        if bucket and size + b > bucket_bytes:
            flush()
        bucket.append(p)
        size += b
    flush()
    return n_coll


class OverlappedGradAllReduce:
    """Gradient all-reduce for the fused ``MixedInputModel`` that starts before the backward pass has finished.

    The fused backward writes all gradients into one flat buffer.  Two slices are final early: the image-FC weight (62 % of
    the bytes at F = 167) after the first GEMM of the image branch, and everything except it and the four conv tensors
    when the fingerprint branch and its weight-gradient leaves are done (the image branch's last kernel is still
    running).  Call this right after ``loss.backward()`` returns (the GPU is then still ~2 ms from the end of the pass):
    those slices are reduced on a communication stream that waits only for the engine's bucket events, the conv tensors
    (76 KB) on the current stream, i.e. after the pass.  Every rank issues the same collectives in the same order.
    Falls back to ``allreduce_gradients`` whenever the layout or the events are not what it expects."""

    def __init__(self, model, group=None):
        self.model, self.group = model, group
        self.early = model.image_cnn[7].weight
        self.late = [model.image_cnn[0].weight, model.image_cnn[0].bias, model.image_cnn[3].weight, model.image_cnn[3].bias]
        self.comm = None

    def __call__(self, params, average: bool = False) -> int:
        from . import _lib
        world = world_size(self.group)
        if world == 1:
            return 0
        params = _params_with_grad(list(params))
        flat = flat_view_of([p.grad for p in params]) if params else None
        g = self.early.grad
        if flat is None or g is None or not flat.is_cuda or any(p.grad is None for p in self.late):
            return allreduce_gradients(params, average=average, group=self.group)
        es = flat.element_size()
        a0 = (g.data_ptr() - flat.data_ptr()) // es
        a1 = a0 + g.numel()
        c0 = (self.late[0].grad.data_ptr() - flat.data_ptr()) // es
        c1 = (self.late[-1].grad.data_ptr() - flat.data_ptr()) // es + self.late[-1].grad.numel()
        # expected order in the flat buffer: [ ... | conv tensors | image-FC weight | ... ]
        if not (0 <= c0 < c1 == a0 < a1 <= flat.numel()) or c1 - c0 != sum(p.grad.numel() for p in self.late):
            return allreduce_gradients(params, average=average, group=self.group)
        L = _lib.lib()
        if self.comm is None:
            self.comm = torch.cuda.Stream(device=flat.device)
        if L.bbbp_mixed_backward_wait_bucket(self.comm.cuda_stream, 0) != 0:
            return allreduce_gradients(params, average=average, group=self.group)     # no bucket event: plain path
        works, n = [], 0
        with torch.cuda.stream(self.comm):
            works.append(dist.all_reduce(flat[a0:a1], group=self.group, async_op=True)); n += 1
        second_early = L.bbbp_mixed_backward_wait_bucket(self.comm.cuda_stream, 1) == 0
        rest = [(0, c0), (a1, flat.numel())]
        if second_early:
            with torch.cuda.stream(self.comm):
                for lo, hi in rest:
                    if hi > lo:
                        works.append(dist.all_reduce(flat[lo:hi], group=self.group, async_op=True)); n += 1
            rest = []
        for lo, hi in rest + [(c0, c1)]:            # on the current stream: after the whole pass
            if hi > lo:
                dist.all_reduce(flat[lo:hi], group=self.group); n += 1
        for w in works:
            w.wait()                             # the current stream waits for the early buckets
        flat.record_stream(torch.cuda.current_stream())
        if average:
            flat.div_(world)
        return n


def broadcast_parameters(module: torch.nn.Module, src: int = 0, group=None) -> None:
    """Make every rank start from rank ``src``'s parameters and buffers (one collective when the parameters are flat)."""
    if world_size(group) == 1:
        return
    params = [p.data for p in module.parameters()]
    flat = flat_view_of(params)
    if flat is not None:
        dist.broadcast(flat, src, group=group)
    else:
        for p in params:
            dist.broadcast(p, src, group=group)
    for b in module.buffers():
        dist.broadcast(b, src, group=group)


def gather_predictions(pred: torch.Tensor, group=None) -> torch.Tensor:
    """Screening (BASELINE config 5): concatenate every rank's [n_local] predictions on all ranks (equal n_local)."""
    world = world_size(group)
    if world == 1:
        return pred
    out = [torch.empty_like(pred) for _ in range(world)]
    dist.all_gather(out, pred.contiguous(), group=group)
    return torch.cat(out)
