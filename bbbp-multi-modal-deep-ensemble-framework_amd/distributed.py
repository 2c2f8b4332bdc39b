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

import threading

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


_LIVE = {"count": 0, "graphs_before": 0}          # live OverlappedGradAllReduce objects and the graph mode the first one found
_LIVE_LOCK = threading.Lock()


class OverlappedGradAllReduce:
    """Gradient all-reduce for the fused engine models (``MixedInputModel`` and its concat variants) that starts before the
    backward pass has finished.

    The fused backward writes all gradients into one flat buffer, and slices of it are final early (engine.hip records an event
    for each): the image-FC weight (62 % of the bytes at F = 167) after the first GEMM of the image branch; the twelve tensors of
    encoder layer l when that layer's weight-gradient leaves are done -- layer L-1 first, 100 MB each at F = 2048, where the
    encoder is 94 % of the gradient bytes; and everything except the four conv tensors when the fingerprint
    branch and its leaves are done (the image branch's last kernel is still running).  Call this right after
    ``loss.backward()`` returns: every early slice is reduced on a communication stream that waits only for its event, the conv
    tensors (76 KB) on the current stream, i.e. after the pass.

    The SCHEDULE -- which slices, in which order -- is fixed at construction from the parameter layout, which is the same on
    every rank, so all ranks issue the same collectives in the same order by construction.  Nothing falls back silently: a
    missing event or an unexpected gradient layout raises.  HIP-graph replay of the engine calls is switched off while a reducer
    exists (a replayed backward records no bucket events, and ranks could capture on different steps)."""

    def __init__(self, model, group=None, min_world: int = 2, pipelined_step: bool = False):
        """``pipelined_step``: also allow ``step(optimizer, params)`` (the optimizer pipelined into the pass): the engine then records
        one more event per backward pass (the release point of the image-FC weight)."""
        import ctypes
        from . import _lib
        self.model, self.group, self.min_world = model, group, int(min_world)
        self.comm = None
        L = _lib.lib()
        # graph replay is off while ANY reducer exists: the first one remembers the process's mode, the last one to close restores it
        with _LIVE_LOCK:
            if _LIVE["count"] == 0:
                _LIVE["graphs_before"] = L.bbbp_set_graphs(0)
            else:
                L.bbbp_set_graphs(0)
            _LIVE["count"] += 1
        self.pipelined_step = bool(pipelined_step)
        self._release_before = L.bbbp_set_release_events(1) if self.pipelined_step else None
        self._closed = False
        params = list(model.parameters())
        offs, o = [], 0
        for p in params:
            offs.append(o); o += p.numel()
        self.total = o
        offs.append(o)
        desc = model._descriptor(2, draw_seed=False)      # layout query only: the global RNG stream is left untouched
        if L.bbbp_mixed_num_params(ctypes.byref(desc)) != len(params):
            raise RuntimeError("OverlappedGradAllReduce: the model's parameter list is not the fused engine's")
        first, count = ctypes.c_int(0), ctypes.c_int(0)
        self.early = []                                   # (bucket id, lo, hi) in reduction order
        order = [0] + [2 + l for l in range(desc.num_layers - 1, -1, -1)]
        for b in order:
            if L.bbbp_mixed_bucket_range(ctypes.byref(desc), b, ctypes.byref(first), ctypes.byref(count)) != 0:
                raise RuntimeError(f"OverlappedGradAllReduce: no range for gradient bucket {b}")
            self.early.append((b, offs[first.value], offs[first.value + count.value]))
        conv = [model.image_cnn[0].weight, model.image_cnn[0].bias, model.image_cnn[3].weight, model.image_cnn[3].bias]
        ids = [next(i for i, q in enumerate(params) if q is c) for c in conv]
        if ids != list(range(ids[0], ids[0] + 4)):
            raise RuntimeError("OverlappedGradAllReduce: the conv tensors are not consecutive parameters")
        self.late = (offs[ids[0]], offs[ids[0] + 4])
        # bucket 1: the complement of the early slices and the conv tensors
        taken = sorted([(lo, hi) for _, lo, hi in self.early] + [self.late])
        self.rest, cur = [], 0
        for lo, hi in taken:
            if lo > cur:
                self.rest.append((cur, lo))
            cur = max(cur, hi)
        if cur < self.total:
            self.rest.append((cur, self.total))

    def close(self) -> None:
        """Give graph replay back to the process (the mode it had before this reducer was built).  Idempotent; also runs when the
        reducer is garbage-collected."""
        if not getattr(self, "_closed", True):
            self._closed = True
            try:
                from . import _lib
                with _LIVE_LOCK:
                    _LIVE["count"] -= 1
                    if _LIVE["count"] == 0:
                        _lib.lib().bbbp_set_graphs(_LIVE["graphs_before"])
                if self._release_before is not None:
                    _lib.lib().bbbp_set_release_events(self._release_before)
            except Exception:      # noqa: BLE001  (interpreter shutdown)
                pass

    def __del__(self):
        self.close()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def schedule(self):
        """The collectives of one call, in order: (stream, bucket event waited for, lo, hi) -- identical on every rank."""
        return ([("comm", b, lo, hi) for b, lo, hi in self.early] + [("comm", 1, lo, hi) for lo, hi in self.rest]
                + [("current", None, self.late[0], self.late[1])])

    def __call__(self, params, average: bool = False) -> int:
        from . import _lib
        if self._closed:
            raise RuntimeError("OverlappedGradAllReduce: used after close()")
        world = world_size(self.group)
        if world < self.min_world:
            return 0
        params = list(params)
        if any(p.grad is None for p in params):
            raise RuntimeError("OverlappedGradAllReduce: every parameter must have a gradient (call it right after backward())")
        flat = flat_view_of([p.grad for p in params])
        if flat is None or not flat.is_cuda or flat.numel() != self.total:
            raise RuntimeError("OverlappedGradAllReduce: gradients are not the engine's one flat buffer; use allreduce_gradients")
        L = _lib.lib()
        if self.comm is None:
            self.comm = torch.cuda.Stream(device=flat.device)
        works, n = [], 0
        for where, bucket, lo, hi in self.schedule():
            if hi <= lo:
                continue
            if where == "comm":
                _lib.check(L.bbbp_mixed_backward_wait_bucket(self.comm.cuda_stream, bucket), "bbbp_mixed_backward_wait_bucket")
                with torch.cuda.stream(self.comm):
                    w = dist.all_reduce(flat[lo:hi], group=self.group, async_op=True)
                if w is not None:
                    works.append(w)
            else:
                dist.all_reduce(flat[lo:hi], group=self.group)        # after the whole pass, on the current stream
            n += 1
        for w in works:
            w.wait()                             # the current stream waits for the early slices
        torch.cuda.current_stream(flat.device).wait_stream(self.comm)
        flat.record_stream(self.comm)
        if average:
            flat.div_(world)
        return n


def _pipelined_step(self, optimizer, params, grad_scale: float = 1.0) -> int:
    """Gradient all-reduce AND the optimizer step, both pipelined into the backward pass (call right after ``loss.backward()``,
    instead of ``reducer(params); optimizer.step(grad_scale=...)``).  For every early slice, on the communication stream: wait for
    the slice's gradients (``bbbp_mixed_backward_wait_bucket``), all-reduce it (``world >= min_world`` only; a single process skips
    the collectives and keeps the pipelining), wait until the pass no longer READS the slice's parameters
    (``bbbp_mixed_backward_wait_released``) and apply fused AdamW to exactly that slice of the flat parameter / gradient / moment
    buffers.  The image-FC weight (62 % of the optimizer's bytes at F = 167) is updated ~1.2 ms before the pass ends, encoder layer
    l when layer l - 1's leaves are done; only the conv tensors (76 KB) are reduced and updated after the pass.  Element-wise
    arithmetic: the parameters afterwards are bit-identical to ``optimizer.step()``'s.  Returns the number of collectives issued."""
    from . import _lib, ops
    if self._closed:
        raise RuntimeError("OverlappedGradAllReduce: used after close()")
    if not self.pipelined_step:
        raise RuntimeError("OverlappedGradAllReduce.step needs OverlappedGradAllReduce(model, pipelined_step=True)")
    params = list(params)
    if any(p.grad is None for p in params):
        raise RuntimeError("OverlappedGradAllReduce.step: every parameter must have a gradient (call it right after backward())")
    # the layout is validated BEFORE the step is counted (begin_flat_step advances every parameter's step counter, and an
    # error after it left the bias correction one step ahead of the parameters)
    from .models import flat_view_of as _flat
    if sum(p.numel() for p in params) != self.total or not params[0].is_cuda:
        raise RuntimeError("OverlappedGradAllReduce.step: the optimizer's parameters are not this model's")
    g0 = _flat([p.grad for p in params])
    if g0 is None or g0.numel() != self.total or not g0.is_cuda:
        raise RuntimeError("OverlappedGradAllReduce.step: gradients are not the engine's one flat buffer; use allreduce_gradients + optimizer.step()")
    flat = optimizer.begin_flat_step()
    if flat is None:
        raise RuntimeError("OverlappedGradAllReduce.step needs bbbp_amd.optim.AdamW over the model's flat parameter buffer "
                           "(one parameter group, gradients from the fused backward)")
    pflat, gflat, m, v, step, hp = flat
    if gflat.numel() != self.total or pflat.numel() != self.total or not gflat.is_cuda:
        raise RuntimeError("OverlappedGradAllReduce.step: the optimizer's parameters are not this model's")
    L = _lib.lib()
    world = world_size(self.group)
    reduce_ = world >= self.min_world and world > 1
    if self.comm is None:
        self.comm = torch.cuda.Stream(device=gflat.device)
    n = 0

    def update(lo, hi):
        ops.adamw_step_(pflat[lo:hi], gflat[lo:hi], m[lo:hi], v[lo:hi], step, grad_scale=grad_scale, **hp)

    with torch.cuda.stream(self.comm):
        cs = self.comm.cuda_stream
        for bucket, lo, hi in self.early + [(1, lo, hi) for lo, hi in self.rest]:
            if hi <= lo:
                continue
            _lib.check(L.bbbp_mixed_backward_wait_bucket(cs, bucket), "bbbp_mixed_backward_wait_bucket")
            if reduce_:
                dist.all_reduce(gflat[lo:hi], group=self.group)          # on the communication stream, in schedule order
                n += 1
            _lib.check(L.bbbp_mixed_backward_wait_released(cs, bucket), "bbbp_mixed_backward_wait_released")
            update(lo, hi)
    lo, hi = self.late
    if hi > lo:
        if reduce_:
            dist.all_reduce(gflat[lo:hi], group=self.group)              # conv tensors: after the whole pass, on the current stream
            n += 1
        update(lo, hi)
    torch.cuda.current_stream(gflat.device).wait_stream(self.comm)
    for t in (pflat, gflat, m, v):
        t.record_stream(self.comm)
    return n


OverlappedGradAllReduce.step = _pipelined_step


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
