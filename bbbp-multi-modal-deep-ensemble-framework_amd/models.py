"""Host-side mirror of the reference's nn.Module interface for the hot path.

``MixedInputModel(fingerprint_size, image_feature_size)`` keeps the constructor signature, the
``forward(fingerprint[B,F], image[B,49152]) -> [B,1]`` contract and every ``state_dict`` key of
the reference class (Models/multi_input_data_regression_opt_transformer_cnn_20250113.py:68-119,
identical in Models/multi_input_data_regression_opt_transformer_cnn.py:71-135), so the
reference's training loop (``model.to(device)``, ``optim.AdamW(model.parameters())``,
``model(fp, img).squeeze()``, ``loss.backward()``, ``torch.save(model.state_dict())``,
``pickle.dump(model)``) runs unchanged.  The parameter containers are the same stock torch modules
created in the same order, so ``torch.manual_seed(s)`` gives bit-identical initial weights; their
``forward`` methods are never called -- all arithmetic runs in libbbbp_hip.so through ONE
autograd node per model call (``bbbp_mixed_forward`` / ``bbbp_mixed_backward``).
"""
from __future__ import annotations

import contextlib
import ctypes
import weakref
from typing import List

import torch
import torch.nn as nn
from torch.utils.data import Dataset

from . import _lib, ops


def reference_nhead(fingerprint_size: int) -> int:
    """Head count rule of the reference (…20250113.py:71-73): F // 8, lowered until it divides F."""
    nhead = max(1, fingerprint_size // 8)
    while fingerprint_size % nhead:
        nhead -= 1
    return nhead


class MixedDataset(Dataset):
    """(fingerprint, image, label) triples as float32 tensors (…20250113.py:31-45)."""

    def __init__(self, fingerprints, images, labels):
        self.fingerprints, self.images, self.labels = fingerprints, images, labels

    def __len__(self):
        return len(self.labels)

    def __getitem__(self, idx):
        as_f32 = lambda a: torch.as_tensor(a[idx]).to(torch.float32).clone()
        return as_f32(self.fingerprints), as_f32(self.images), as_f32(self.labels)


def flatten_parameters(module: nn.Module) -> torch.Tensor:
    """Re-home all parameters of ``module`` as views of ONE contiguous fp32 buffer (named_parameters
    order) and return it.  Values are preserved, Parameter objects keep their identity (optimizers
    stay valid), state_dict keys do not change.  Lets AdamW and the RCCL all-reduce run as one
    flat launch each."""
    params = [p for p in module.parameters()]
    total = sum(p.numel() for p in params)
    if total == 0:
        return torch.empty(0)
    flat = torch.empty(total, dtype=params[0].dtype, device=params[0].device)
    off = 0
    with torch.no_grad():
        for p in params:
            n = p.numel()
            view = flat[off:off + n].view(p.shape)
            view.copy_(p.data)
            p.data = view
            off += n
    return flat


def flat_view_of(params: List[torch.Tensor]):
    """The flat buffer if ``params`` are consecutive views of one storage, else None."""
    if not params:
        return None
    base = params[0]
    ptr = base.data_ptr()
    for p in params:
        if p.data_ptr() != ptr or not p.is_contiguous() or p.dtype != base.dtype:
            return None
        ptr += p.numel() * p.element_size()
    total = sum(p.numel() for p in params)
    return torch.as_strided(base, (total,), (1,), base.storage_offset()) if total else None


class MultiHeadAttentionFusion(nn.Module):
    """Parameter container + standalone forward of the reference fusion block
    (…20250113.py:48-65): cat -> per head Linear/Tanh/Linear -> softmax over heads -> weighted sum."""

    def __init__(self, input_dim, num_heads=4, hidden_dim=128):
        super().__init__()
        self.attention_heads = nn.ModuleList([
            nn.Sequential(nn.Linear(input_dim, hidden_dim), nn.Tanh(), nn.Linear(hidden_dim, 1))
            for _ in range(num_heads)])
        self.softmax = nn.Softmax(dim=1)

    def forward(self, x1, x2):
        from .functional import attention_fusion
        return attention_fusion(self, x1, x2)


def _cached_ptrs(model, attr, tensors):
    """ctypes array of the tensors' device pointers, rebuilt only when the storage moved (e.g. after .to())."""
    # every pointer takes part in the key (first / last only missed a middle parameter whose storage was replaced, e.g.
    # `p.data = p.data.clone()` on one layer -- the engine then read a stale pointer); ~150 data_ptr() calls, 20 us
    key = tuple(t.data_ptr() for t in tensors)
    cache = _PTR_CACHE.setdefault(model, {})          # kept off the module so that pickle.dump(model) still works
    hit = cache.get(attr)
    if hit is None or hit[0] != key:
        hit = (key, _lib.ptr_array(list(key)))
        cache[attr] = hit
    return hit[1]


_PTR_CACHE = weakref.WeakKeyDictionary()
_PARAM_CACHE = weakref.WeakKeyDictionary()


def _param_list(model):
    """``list(model.parameters())`` without walking the module tree on every call (0.3 ms of a 1.6 ms step at the reference's batch size
    32): the list is kept with the (``_modules`` dict, name, child) and (``_parameters`` dict, name, parameter) slots it was read from,
    and every call re-checks those ~150 slots by identity -- a replaced parameter or sub-module rebuilds it."""
    hit = _PARAM_CACHE.get(model)
    if hit is not None:
        mods, slots, params = hit
        for d, n, m in mods:
            if d.get(n) is not m:
                break
        else:
            for d, n, q in slots:
                if d.get(n) is not q:
                    break
            else:
                return params
    mods, slots, params, seen = [], [], [], set()
    for _, mod in model.named_modules():
        for n, child in mod._modules.items():
            mods.append((mod._modules, n, child))
        for n, q in mod._parameters.items():
            slots.append((mod._parameters, n, q))
            if q is not None and id(q) not in seen:
                seen.add(id(q))
                params.append(q)
    if [id(q) for q in params] != [id(q) for q in model.parameters()]:          # a layout this walk does not reproduce: no cache
        return list(model.parameters())
    _PARAM_CACHE[model] = (mods, slots, params)
    return params
_LAST_WS = weakref.WeakKeyDictionary()


def _exact_batch_group(model):
    """(group, world, rank) of a model in exact-global-batch mode (``model.exact_batch``): its own group when given, else the default
    group of an initialised ``torch.distributed``; a single process is world 1 -- the same engine path with no-op collectives."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return None, 1, 0
    group = getattr(model, "group", None)
    group = group if group is not None else dist.group.WORLD
    return group, dist.get_world_size(group), dist.get_rank(group)


class _CollectiveCall:
    """What one forward / backward pair of the exact-global-batch engine needs in its collective callback."""
    __slots__ = ("ws", "group", "world", "rank", "errors", "force")

    def __init__(self, group, world, rank, force):
        self.ws, self.group, self.world, self.rank, self.errors, self.force = None, group, world, rank, [], force


_COLLECTIVE_CALLS = {}          # handle (bbbp_mixed_desc.collective_ctx) -> _CollectiveCall; dropped after the backward call
_COLLECTIVE_NEXT = [1]
_COLLECTIVE_MAX_PENDING = 64    # exact-batch forward calls that may wait for their backward call at once (gradient accumulation over micro-batches)
_COLLECTIVE_ERRORS = []         # errors raised inside the callback for handles that have no call object any more
_COLLECTIVE_DROPPED = set()     # handles evicted above: their backward call gets a clear error instead of 'callback returned 2'
_COLLECTIVE_FN = []             # ONE ctypes thunk for the process: a thunk per call is a reference cycle that keeps its workspace alive until a GC


def _collective_register(call):
    handle = _COLLECTIVE_NEXT[0]
    _COLLECTIVE_NEXT[0] += 1
    _COLLECTIVE_CALLS[handle] = call
    while len(_COLLECTIVE_CALLS) > _COLLECTIVE_MAX_PENDING:      # forward calls whose backward never came (evaluation under autograd): oldest first
        dropped = next(iter(_COLLECTIVE_CALLS))
        _COLLECTIVE_CALLS.pop(dropped)
        _COLLECTIVE_DROPPED.add(dropped)
    return handle


def _collective_fn():
    """bbbp_collective_fn (include/bbbp_hip.h) over torch.distributed: the engine names its buffers as byte offsets into the call's
    workspace; the collective is issued under the engine's stream, so it is ordered after the launches that produced its input and
    before the ones that read its output.  RCCL ("nccl") takes the tensor forms; gloo (rehearsal: ranks sharing one GPU, CPU-side
    transport) has neither an in-place all-gather nor a reduce-scatter, so it gathers into chunks / all-reduces and slices.
    ``force`` (tests): issue the library calls at world size 1 too (the RCCL path on a one-GPU box)."""
    if _COLLECTIVE_FN:
        return _COLLECTIVE_FN[0]
    import torch.distributed as dist

    def cb(handle, op, what, layer, send_off, recv_off, count, stream):
        call = _COLLECTIVE_CALLS.get(handle)
        if call is None:
            if handle in _COLLECTIVE_DROPPED:
                _COLLECTIVE_ERRORS.append(RuntimeError(
                    f"exact-batch mode: more than {_COLLECTIVE_MAX_PENDING} forward calls were waiting for their backward call; this one's "
                    "collective state was dropped -- call backward() (or run under torch.no_grad()) before issuing that many forwards"))
            return 2
        try:
            ws, group, world, rank = call.ws, call.group, call.world, call.rank
            real = world > 1 or (call.force and group is not None)

            def f32(off, n):
                return ws[off:off + 4 * n].view(torch.float32)

            if ws.is_cuda:
                scope = torch.cuda.stream(torch.cuda.ExternalStream(int(stream), device=ws.device) if stream
                                          else torch.cuda.default_stream(ws.device))
            else:                                             # host buffers: the CPU rehearsal of the protocol (tests/test_distributed_cpu.py)
                scope = contextlib.nullcontext()
            with scope:
                if op == 0:                                   # BBBP_COLL_ALLGATHER, in place
                    if real:
                        buf = f32(recv_off, count * world)
                        mine = buf[rank * count:(rank + 1) * count]
                        if dist.get_backend(group) == "gloo":
                            dist.all_gather(list(buf.chunk(world)), mine.clone(), group=group)
                        else:
                            dist.all_gather_into_tensor(buf, mine, group=group)
                elif op == 1:                                 # BBBP_COLL_REDUCE_SCATTER
                    send, recv = f32(send_off, count * world), f32(recv_off, count)
                    if not real:
                        recv.copy_(send)
                    elif dist.get_backend(group) == "gloo":
                        dist.all_reduce(send, group=group)
                        recv.copy_(send[rank * count:(rank + 1) * count])
                    else:
                        dist.reduce_scatter_tensor(recv, send, group=group)
                else:
                    raise RuntimeError(f"unknown collective op {op}")
            return 0
        except BaseException as e:        # noqa: BLE001 -- an exception must not unwind through the C frames
            call.errors.append(e)
            return 1

    _COLLECTIVE_FN.append(_lib.CollectiveFn(cb))
    return _COLLECTIVE_FN[0]


class _MixedFn(torch.autograd.Function):
    """One autograd node for the whole model: forward and backward are single C-ABI calls."""

    @staticmethod
    def forward(ctx, model, inference, fingerprint, image, *params):
        L = _lib.lib()
        B = fingerprint.shape[0]
        desc = model._descriptor(B, inference)
        ctx.call, ctx.handle = None, 0
        if getattr(model, "exact_batch", False):
            group, world, rank = _exact_batch_group(model)
            ctx.call = _CollectiveCall(group, world, rank, getattr(model, "exact_force_collectives", False))
            ctx.handle = _collective_register(ctx.call)
            desc.collective = ctypes.cast(_collective_fn(), ctypes.c_void_p).value
            desc.collective_ctx = ctx.handle
        ws_bytes = L.bbbp_mixed_workspace_bytes(ctypes.byref(desc))
        if ws_bytes == 0:
            _lib.check(1, "bbbp_mixed_workspace_bytes")
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=fingerprint.device)
        if ctx.call is not None:
            ctx.call.ws = ws
        out = torch.empty((B, 1), dtype=torch.float32, device=fingerprint.device)
        if not params:
            raise RuntimeError("the model has no parameters")
        pp = _cached_ptrs(model, "_pp", params)
        # dtype / device of every parameter: checked once per pointer array (a .to() / .half() moves the storage and rebuilds it)
        chk = _PTR_CACHE[model].get("_pp_checked")
        if chk is None or chk[0] is not pp or chk[1] != fingerprint.device:
            if any(p.dtype != torch.float32 or p.device != fingerprint.device for p in params):
                raise RuntimeError("the HIP kernels are float32-only: every parameter must be float32 on the inputs' device "
                                   f"({fingerprint.device}); undo .half()/.double() or move the model")
            _PTR_CACHE[model]["_pp_checked"] = (pp, fingerprint.device)
        bn = model.fc[2]
        if any(b.dtype != torch.float32 or b.device != fingerprint.device for b in (bn.running_mean, bn.running_var)):
            raise RuntimeError("BatchNorm running statistics must be float32 on the inputs' device")
        bnp = _cached_ptrs(model, "_bnp", (bn.running_mean, bn.running_var))
        rc = L.bbbp_mixed_forward(ops._stream(), ctypes.byref(desc), pp, bnp, fingerprint.data_ptr(), image.data_ptr(),
                                  out.data_ptr(), ws.data_ptr(), ws_bytes)
        if ctx.call is not None and (inference or rc):
            _COLLECTIVE_CALLS.pop(ctx.handle, None)                  # no backward call will follow
        if rc and ctx.call is not None and ctx.call.errors:
            raise RuntimeError("bbbp_mixed_forward: collective failed") from ctx.call.errors[0]
        _lib.check(rc, "bbbp_mixed_forward")
        ctx.desc, ctx.ws, ctx.ws_bytes, ctx.pp = desc, ws, ws_bytes, pp
        lay = _PTR_CACHE[model].get("_grad_layout")
        if lay is None or lay[4] is not pp:
            sizes = [p.numel() for p in params]
            offs, off = [], 0
            for n in sizes:
                offs.append(4 * off)
                off += n
            lay = (off, sizes, [None if p.dim() == 1 else p.shape for p in params], offs, pp)
            _PTR_CACHE[model]["_grad_layout"] = lay
        ctx.layout = lay
        if getattr(model, "keep_workspace", False):
            _LAST_WS[model] = (desc, ws)
        ctx.save_for_backward(fingerprint, image, *params)
        return out

    @staticmethod
    def backward(ctx, dout):
        fingerprint, image, *params = ctx.saved_tensors
        L = _lib.lib()
        dout = dout.contiguous()
        # the gradient views of one flat buffer: one split call, a reshape only where a parameter is not 1-D, byte offsets kept per
        # pointer array (the views themselves must be new objects every step: autograd adopts them as .grad only while nobody else holds them)
        lay = ctx.layout
        gflat = torch.empty(lay[0], dtype=torch.float32, device=dout.device)
        grads = [g if s is None else g.view(s) for g, s in zip(gflat.split(lay[1]), lay[2])]
        base = gflat.data_ptr()
        gp = (ctypes.c_void_p * len(params))(*[base + o for o in lay[3]])
        rc = L.bbbp_mixed_backward(ops._stream(), ctypes.byref(ctx.desc), ctx.pp, gp, fingerprint.data_ptr(),
                                   image.data_ptr(), dout.data_ptr(), ctx.ws.data_ptr(), ctx.ws_bytes)
        if ctx.call is not None:
            _COLLECTIVE_CALLS.pop(ctx.handle, None)
        if rc and ctx.call is not None and ctx.call.errors:
            raise RuntimeError("bbbp_mixed_backward: collective failed") from ctx.call.errors[0]
        if rc and _COLLECTIVE_ERRORS:
            raise _COLLECTIVE_ERRORS.pop()
        _lib.check(rc, "bbbp_mixed_backward")
        return (None, None, None, None, *grads)


class MixedInputModel(nn.Module):
    """Drop-in for the reference ``MixedInputModel`` (…20250113.py:68-119): 6-layer post-norm
    Transformer encoder over the fingerprint (attending across the mini-batch), 2-stage conv/ReLU/pool
    CNN over the 3x128x128 image, attention fusion and a BatchNorm regression head."""

    NUM_LAYERS = 6            # encoder depth; 0 = no encoder (fingerprint_fc reads the fingerprint itself)
    FUSION = "attention"      # "attention": MultiHeadAttentionFusion(256, 4); "concat": torch.cat of the branch outputs

    def __init__(self, fingerprint_size, image_feature_size):
        super().__init__()
        self.fingerprint_size = int(fingerprint_size)
        self.image_feature_size = int(image_feature_size)
        self.nhead = reference_nhead(self.fingerprint_size)
        F, S = self.fingerprint_size, self.image_feature_size
        # same modules, same creation order as the reference => same state_dict keys and RNG stream
        if self.NUM_LAYERS > 0:
            self.fingerprint_transformer = nn.TransformerEncoder(
                nn.TransformerEncoderLayer(d_model=F, nhead=self.nhead), num_layers=self.NUM_LAYERS, enable_nested_tensor=False)
        self.fingerprint_fc = nn.Sequential(nn.Linear(F, 128), nn.ReLU())
        self.image_cnn = nn.Sequential(
            nn.Conv2d(3, 32, kernel_size=3, stride=1, padding=1), nn.ReLU(), nn.MaxPool2d(kernel_size=2, stride=2),
            nn.Conv2d(32, 64, kernel_size=3, stride=1, padding=1), nn.ReLU(), nn.MaxPool2d(kernel_size=2, stride=2),
            nn.Flatten(), nn.Linear(64 * (S // 4) * (S // 4), 128), nn.ReLU())
        if self.FUSION == "attention":
            self.attention_fusion = MultiHeadAttentionFusion(256, num_heads=4)
        self.fc = nn.Sequential(nn.Linear(256, 256), nn.ReLU(), nn.BatchNorm1d(256), nn.Linear(256, 128), nn.ReLU(),
                                nn.Linear(128, 64), nn.ReLU(), nn.Linear(64, 1))
        if S != 128:
            # the reference's forward hard-codes view(-1, 3, 128, 128) (…20250113.py:114)
            raise ValueError("image_feature_size must be 128: forward reshapes images to 3x128x128")
        flatten_parameters(self)

    # nn.Module.to()/cuda()/float() re-create every parameter; put them back into one flat buffer
    def _apply(self, fn, *args, **kwargs):
        out = super()._apply(fn, *args, **kwargs)
        flatten_parameters(self)
        return out

    def flat_parameters(self) -> torch.Tensor:
        params = list(self.parameters())
        flat = flat_view_of(params)
        return flat if flat is not None else flatten_parameters(self)

    def _descriptor(self, batch: int, inference: bool = False, draw_seed: bool = True) -> _lib.MixedDesc:
        """``draw_seed=False``: a descriptor for layout queries only (parameter count, gradient buckets) -- it must not consume the
        global RNG stream, or constructing a reducer would shift every later dropout seed."""
        p, dff, layers = 0.0, 2048, 0
        if self.NUM_LAYERS > 0:
            layers = len(self.fingerprint_transformer.layers)
            layer0 = self.fingerprint_transformer.layers[0]
            p, dff = float(layer0.dropout.p), layer0.linear1.out_features
            for m in (layer0.dropout1, layer0.dropout2):
                if float(m.p) != p:
                    raise RuntimeError("encoder dropout probabilities must agree")
            if float(layer0.self_attn.dropout) != p:
                raise RuntimeError("attention dropout must equal the layer dropout")
        training = bool(self.training)
        seed = int(torch.randint(0, 2 ** 62, (1,)).item()) if (training and p > 0 and draw_seed) else 0
        world, rank = 0, 0
        if getattr(self, "exact_batch", False):
            _, world, rank = _exact_batch_group(self)
            # the kernels index their Philox streams by LOCAL row; ranks that seed the host RNG alike (the usual same-init setup) would
            # otherwise draw the SAME masks for local row i on every rank, which no single-GPU step at the global batch does: the rank
            # is mixed into the call's seed
            if seed and rank:
                seed = (seed + rank * 0x9E3779B97F4A7C15) & (2 ** 62 - 1) or 1
        return _lib.MixedDesc(batch=batch, fingerprint_size=self.fingerprint_size, nhead=self.nhead, num_layers=layers,
                              dim_feedforward=dff, training=int(training), dropout_p=p, seed=seed, need_input_grad=0,
                              fusion=0 if self.FUSION == "attention" else 1, inference=int(inference and not training),
                              world=world, rank=rank)

    def debug_ffn_gates(self):
        """Test hook: per encoder layer, the [B, dim_feedforward] uint8 ReLU decisions of ``linear1`` in the most recent
        forward call made with ``self.keep_workspace = True`` (``bbbp_mixed_debug_ffn_gate``)."""
        if self not in _LAST_WS:
            raise RuntimeError("set model.keep_workspace = True before the forward call")
        desc, ws = _LAST_WS[self]
        L, gates = _lib.lib(), []
        for l in range(desc.num_layers):
            g = torch.empty((desc.batch, desc.dim_feedforward), dtype=torch.uint8, device=ws.device)
            _lib.check(L.bbbp_mixed_debug_ffn_gate(ops._stream(), ctypes.byref(desc), ws.data_ptr(), l, g.data_ptr()),
                       "bbbp_mixed_debug_ffn_gate")
            gates.append(g)
        return gates

    def debug_pool_masks(self):
        """Test hook: the u8 pooling / ReLU decisions ([B,32,64,64], [B,64,32,32]) of the two conv stages in the most recent forward
        call made with ``self.keep_workspace = True`` (``bbbp_mixed_debug_pool_mask``: 0..3 = first maximum of the 2x2 window in
        scan order, 4 = ReLU inactive)."""
        if self not in _LAST_WS:
            raise RuntimeError("set model.keep_workspace = True before the forward call")
        desc, ws = _LAST_WS[self]
        L, masks = _lib.lib(), []
        for stage, shape in ((1, (desc.batch, 32, 64, 64)), (2, (desc.batch, 64, 32, 32))):
            m = torch.empty(shape, dtype=torch.uint8, device=ws.device)
            _lib.check(L.bbbp_mixed_debug_pool_mask(ops._stream(), ctypes.byref(desc), ws.data_ptr(), stage, m.data_ptr()),
                       "bbbp_mixed_debug_pool_mask")
            masks.append(m)
        return masks

    def forward(self, fingerprint, image):
        if not fingerprint.is_cuda:
            raise RuntimeError("MixedInputModel runs on MI355X only (HIP kernels); move the model and inputs to 'cuda'. "
                               "There is no CPU fallback.")
        if fingerprint.dim() != 2 or fingerprint.shape[1] != self.fingerprint_size:
            raise RuntimeError(f"fingerprint must be [B, {self.fingerprint_size}], got {tuple(fingerprint.shape)}")
        B = fingerprint.shape[0]
        if image.numel() != B * 3 * 128 * 128:
            raise RuntimeError(f"shape '[-1, 3, 128, 128]' is invalid for input of size {image.numel()} with batch {B}")
        fingerprint = fingerprint.to(torch.float32).contiguous()
        image = image.to(torch.float32).contiguous()
        params = _param_list(self)
        # forward-only calls (torch.no_grad(), eval loops, screening) take the small inference workspace
        inference = not (torch.is_grad_enabled() and any(p.requires_grad for p in params))
        out = _MixedFn.apply(self, inference, fingerprint, image, *params)
        if self.training:
            self.fc[2].num_batches_tracked += 1
        return out


class ConcatMixedInputModel(MixedInputModel):
    """Drop-in for the earliest Transformer+CNN ``MixedInputModel``
    (Descriptors/multi_input_data_regression_opt_round_2_transformer_cnn.py:45-102): same encoder, ``fingerprint_fc`` and CNN as
    the flagship, but the branch outputs are fused by a plain ``torch.cat`` (:99) -- there is no ``attention_fusion`` block and
    no such ``state_dict`` keys.  Runs on the same fused engine (``bbbp_mixed_desc.fusion = 1``)."""
    FUSION = "concat"


class TwoBranchConcatModel(MixedInputModel):
    """BASELINE config 2: MACCS ``Linear(F,128)+ReLU`` + the 2-stage image CNN + ``torch.cat`` + BatchNorm regression head.
    NO exact reference script exists for it: it is the class above
    (Descriptors/multi_input_data_regression_opt_round_2_transformer_cnn.py:45-102) with the encoder omitted, i.e. its
    ``forward`` lines :95-101 without :91-92; ``state_dict`` keys are that class's minus ``fingerprint_transformer.*``, so
    ``two.load_state_dict({k: v for k, v in full.state_dict().items() if not k.startswith("fingerprint_transformer.")})``
    carries weights over.  Engine: ``num_layers = 0``, ``fusion = 1``."""
    NUM_LAYERS = 0
    FUSION = "concat"


class _MSEFn(torch.autograd.Function):
    """loss and d loss / d pred from ONE kernel (bbbp_mse); backward scales the stored gradient."""

    @staticmethod
    def forward(ctx, pred, target):
        loss, dpred = ops.mse(pred, target)
        ctx.save_for_backward(dpred)
        ctx.shape = pred.shape
        return loss.view(())

    @staticmethod
    def backward(ctx, grad_out):
        (dpred,) = ctx.saved_tensors
        return (dpred * grad_out).view(ctx.shape), None


class MSELoss(nn.Module):
    """Drop-in for ``nn.MSELoss()`` (reduction 'mean'; ...20250113.py:173,189) for predictions / targets on the GPU: value
    and gradient come from one fused kernel instead of seven elementwise/reduction launches.  The target gets no gradient."""

    def forward(self, pred, target):
        if pred.shape != target.shape:
            raise RuntimeError(f"MSELoss: shapes differ: {tuple(pred.shape)} vs {tuple(target.shape)}")
        return _MSEFn.apply(pred, target)
