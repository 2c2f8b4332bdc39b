"""Fused AdamW for the hot path: same constructor arguments and arithmetic as ``torch.optim.AdamW``
as the reference uses it (Models/multi_input_data_regression_opt_transformer_cnn_20250113.py:172,
``optim.AdamW(model.parameters(), lr=1e-4, weight_decay=1e-5)``), run by ``bbbp_adamw_step``.

When the parameters (and their gradients) are consecutive views of one flat buffer -- which
``MixedInputModel`` arranges -- a whole step is ONE kernel launch over 13.5 M elements.  When only the parameters are flat and the gradients are
separate tensors (what autograd leaves behind for the per-op variants) it is still one launch, through a device table of gradient addresses
(``bbbp_adamw_step_multi``, round 4); otherwise one launch per tensor.  State tensors are exposed per parameter (``exp_avg``, ``exp_avg_sq``, ``step``) like
torch's, so ``state_dict()`` has the familiar shape.
"""
from __future__ import annotations

import torch

from . import ops
from .models import flat_view_of


class AdamW(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, defer=None, capturable=False):
        """``capturable`` (round 4): the step can be captured into a HIP graph (``training.GraphedTrainStep``) -- the step-dependent scalars
        (bias corrections, the scheduler's lr) are read from device memory, where ``step()`` stores them in stream order before its launch;
        a call made while the stream is capturing records the launch only, and ``advance()`` is what a replay loop calls before each replay.

        ``defer`` (round 4, opt-in): one parameter -- the image-FC weight of a ``MixedInputModel``, 62 % of the optimizer's bytes -- whose
        slice of the flat update runs on a library-owned side stream beside the start of the NEXT forward pass, which does not read it
        before its third kernel (``bbbp_adamw_step_deferred``).  The model's forward orders itself behind the slice; anything that reads
        that parameter outside a forward call (``state_dict()``, checkpoints, ``.cpu()``) must call ``synchronize()`` first.  The
        parameters are bit-identical to the undeferred step's."""
        if lr < 0 or eps < 0 or not (0 <= betas[0] < 1) or not (0 <= betas[1] < 1) or weight_decay < 0:
            raise ValueError("invalid AdamW hyper-parameter")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self._flat_state = {}
        self._flat_params = {}
        self._defer = defer
        self._capturable = bool(capturable)
        self._hyper = {}                # group index -> 8-float device tensor (capturable)
        self._tables = {}               # group index -> {gradient addresses: (device table, pinned host copy)}
        self._capture_staging = {}      # group index -> pinned buffer set aside by prepare_capture()
        self._graph_tables = []         # tables that belong to captured graphs

    def _hyper_of(self, gi, device):
        h = self._hyper.get(gi)
        if h is None or h.device != device:
            h = torch.zeros(8, dtype=torch.float32, device=device)
            self._hyper[gi] = h
        return h

    @torch.no_grad()
    def advance(self, grad_scale: float = 1.0) -> None:
        """capturable: count one step for every parameter and store its scalars to device memory (what ``step()`` does before its launch
        when it is not being captured).  Call before each replay of a captured step."""
        for gi, group in enumerate(self.param_groups):
            params = group["params"]
            if not params:
                continue
            if not self._state_is_flat(gi, params):
                self._init_group_state(gi, list(params))
            for p in params:
                self.state[p]["step"] += 1
            ops.adamw_hyper_store_(self._hyper_of(gi, params[0].device), self.state[params[0]]["step"], lr=group["lr"], betas=group["betas"],
                                   eps=group["eps"], weight_decay=group["weight_decay"], grad_scale=grad_scale)

    def _grad_table(self, gi, params):
        """Device table for ``bbbp_adamw_step_multi``: element offsets of the parameters in their flat buffer + the addresses of their
        gradients.  Cached by the addresses (the caching allocator hands a steady training loop the same blocks again and again); the
        pinned host copy stays alive with the entry because the upload is asynchronous (and, inside a capture, a node of the graph)."""
        key = tuple(p.grad.data_ptr() for p in params)
        cache = self._tables.setdefault(gi, {})
        hit = cache.get(key)
        if hit is None:
            capturing = torch.cuda.is_current_stream_capturing()
            offs = [0]
            for p in params:
                offs.append(offs[-1] + p.numel())
            values = torch.tensor(offs + list(key), dtype=torch.int64)
            if capturing:
                # pinned memory cannot be allocated while a stream is capturing: prepare_capture() set a staging buffer aside; the upload
                # becomes a node of the graph (replayed from that buffer, which therefore belongs to the graph from now on)
                host = self._capture_staging.pop(gi, None)
                if host is None or host.numel() != values.numel():
                    raise RuntimeError("AdamW: call prepare_capture() before capturing a step (GraphedTrainStep does)")
                host.copy_(values)
            else:
                if len(cache) >= 16:
                    cache.clear()
                host = values.pin_memory()
            dev = torch.empty(values.numel(), dtype=torch.int64, device=params[0].device)
            dev.copy_(host, non_blocking=True)
            hit = (dev, host)
            cache[key] = hit
            if capturing:
                self._graph_tables.append(hit)            # never evicted
        return hit[0]

    def prepare_capture(self) -> None:
        """Before a step is captured into a graph: set pinned staging memory aside for the gradient-address tables the captured ``step()``
        will upload (allocating pinned memory is not allowed during a capture)."""
        for gi, group in enumerate(self.param_groups):
            if group["params"]:
                self._capture_staging[gi] = torch.empty(2 * len(group["params"]) + 1, dtype=torch.int64).pin_memory()

    def synchronize(self) -> None:
        """Order the current stream behind a deferred slice (see ``defer``); a no-op when nothing is pending."""
        if self._defer is not None and self._defer.is_cuda:
            ops.param_sync()

    def state_dict(self):
        self.synchronize()
        return super().state_dict()

    def _init_group_state(self, gi, params):
        """Flat first/second-moment buffers for group ``gi``; ``state[p]`` holds views of them.  Moments and step counts a
        parameter already has (restored by ``load_state_dict``, or from steps taken before the group's layout changed)
        are carried over, never reset."""
        total = sum(p.numel() for p in params)
        dev = params[0].device
        m = torch.zeros(total, dtype=torch.float32, device=dev)
        v = torch.zeros(total, dtype=torch.float32, device=dev)
        off = 0
        for p in params:
            n = p.numel()
            mv, vv = m[off:off + n].view(p.shape), v[off:off + n].view(p.shape)
            old = self.state.get(p)
            step = 0
            if old:
                mv.copy_(old["exp_avg"]); vv.copy_(old["exp_avg_sq"])
                step = int(old["step"])
            self.state[p] = {"step": step, "exp_avg": mv, "exp_avg_sq": vv}
            off += n
        self._flat_state[gi] = (m, v)

    def _state_is_flat(self, gi, params) -> bool:
        """True while the parameters' moments are still views of this group's flat buffers (first and last are checked: the
        views are only ever replaced wholesale, by ``load_state_dict`` or ``_init_group_state``)."""
        flat = self._flat_state.get(gi)
        if flat is None:
            return False
        first, last = self.state.get(params[0]), self.state.get(params[-1])
        if not first or not last:
            return False
        end = flat[0].data_ptr() + 4 * (flat[0].numel() - params[-1].numel())
        return first["exp_avg"].data_ptr() == flat[0].data_ptr() and last["exp_avg"].data_ptr() == end

    def load_state_dict(self, state_dict):
        """torch's loader deep-copies the saved moments into fresh per-parameter tensors; the next ``step()`` notices that
        they are no longer views of the flat buffers (``_state_is_flat``) and re-homes them, values and step counts kept."""
        super().load_state_dict(state_dict)
        self._flat_state.clear()
        self._flat_params.clear()
        for st in self.state.values():
            if "step" in st and torch.is_tensor(st["step"]):
                st["step"] = int(st["step"])

    @torch.no_grad()
    def begin_flat_step(self):
        """For a step applied slice by slice (distributed.OverlappedGradAllReduce.step): one parameter group whose parameters, gradients
        and moments are flat buffers.  Counts the step and returns ``(params, grads, exp_avg, exp_avg_sq, step, hyper-parameters)``
        -- the caller then runs ``ops.adamw_step_`` on matching slices of the four buffers (element-wise arithmetic: any slicing gives
        the single launch's bits) -- or None when the layout is not flat (nothing is counted then: call ``step()``)."""
        if len(self.param_groups) != 1:
            return None
        group = self.param_groups[0]
        all_params = group["params"]
        if not all_params or any(p.grad is None for p in all_params):
            return None
        pflat = flat_view_of(all_params)
        gflat = flat_view_of([p.grad for p in all_params])
        if pflat is None or gflat is None:
            return None
        if not self._state_is_flat(0, all_params):
            self._init_group_state(0, list(all_params))
        steps = {self.state[p]["step"] for p in all_params}
        if len(steps) != 1:
            return None
        for p in all_params:
            self.state[p]["step"] += 1
        m, v = self._flat_state[0]
        hp = dict(lr=group["lr"], betas=group["betas"], eps=group["eps"], weight_decay=group["weight_decay"])
        return pflat, gflat, m, v, self.state[all_params[0]]["step"], hp

    @torch.no_grad()
    def step(self, closure=None, grad_scale: float = 1.0):
        """``grad_scale`` multiplies every gradient on the fly (1/world_size after a summing all-reduce)."""
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for gi, group in enumerate(self.param_groups):
            params = [p for p in group["params"] if p.grad is not None]
            if not params:
                continue
            if not self._state_is_flat(gi, group["params"]):
                self._init_group_state(gi, [p for p in group["params"]])
            hp = dict(lr=group["lr"], betas=group["betas"], eps=group["eps"], weight_decay=group["weight_decay"])
            capturing = self._capturable and torch.cuda.is_current_stream_capturing()
            if not capturing:                   # a captured call only records the launch; advance() counts the replays
                for p in params:
                    self.state[p]["step"] += 1
            step = self.state[params[0]]["step"]
            all_params = group["params"]
            pflat = gflat = None
            if len(params) == len(all_params):
                # the flat parameter view is cached while the first / last parameter stay where they were
                key = (all_params[0].data_ptr(), all_params[-1].data_ptr(), len(all_params))
                hit = self._flat_params.get(gi)
                if hit is None or hit[0] != key:
                    hit = (key, flat_view_of(all_params))
                    self._flat_params[gi] = hit
                pflat = hit[1]
                if pflat is not None:
                    # gradients written by the fused backward are consecutive views of one buffer: recognise that by the
                    # shared base and the running offset (cheaper than re-deriving the layout from data pointers)
                    g0 = all_params[0].grad
                    base = g0._base
                    if base is not None and g0.storage_offset() == 0 and base.numel() == pflat.numel() and base.is_contiguous():
                        off = 0
                        for q in all_params:
                            g = q.grad
                            if g._base is not base or g.storage_offset() != off:
                                base = None
                                break
                            off += g.numel()
                        gflat = base
                    if gflat is None:
                        gflat = flat_view_of([q.grad for q in all_params])
            same_step = all(self.state[p]["step"] == step for p in params)
            hyper = None
            if self._capturable and pflat is not None and same_step:
                hyper = self._hyper_of(gi, pflat.device)
                if not capturing:
                    ops.adamw_hyper_store_(hyper, step, grad_scale=grad_scale, **hp)
            elif self._capturable:
                raise RuntimeError("AdamW(capturable=True) needs every parameter of a group in one flat buffer (models.flatten_parameters), "
                                   "a gradient for each, and equal step counts")
            if pflat is not None and gflat is not None and same_step and hyper is None:
                m, v = self._flat_state[gi]
                d = self._defer
                lo = -1
                if d is not None and d.is_cuda and d.is_contiguous():
                    off = d.data_ptr() - pflat.data_ptr()
                    if off >= 0 and off % 4 == 0 and off // 4 + d.numel() <= pflat.numel():
                        lo = off // 4
                if lo >= 0:
                    ops.adamw_step_deferred_(pflat, gflat, m, v, step, lo, lo + d.numel(), grad_scale=grad_scale, **hp)
                else:
                    ops.adamw_step_(pflat, gflat, m, v, step, grad_scale=grad_scale, **hp)
            elif (pflat is not None and same_step and pflat.data_ptr() % 16 == 0
                  and all(q.grad.is_contiguous() and q.grad.dtype == torch.float32 and q.grad.is_cuda for q in all_params)):
                # flat parameters, separate gradient tensors: one launch through a table of gradient addresses
                m, v = self._flat_state[gi]
                ops.adamw_step_multi_(pflat, m, v, self._grad_table(gi, all_params), len(all_params), max(step, 1), grad_scale=grad_scale,
                                      hyper=hyper, **hp)
            elif hyper is not None:
                raise RuntimeError("AdamW(capturable=True): gradients must be contiguous float32 device tensors")
            else:
                for p in params:
                    st = self.state[p]
                    g = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
                    ops.adamw_step_(p.data, g, st["exp_avg"], st["exp_avg_sq"], st["step"], grad_scale=grad_scale, **hp)
        return loss
