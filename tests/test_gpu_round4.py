"""Round 4 engine features against the schedules they replace (GPU): the optimizer slice deferred to a side stream, the LayerNorm-absorbing
GEMMs inside the whole-model forward, the software-pipelined conv1 forward inside training plans."""
import os

import numpy as np
import pytest
import torch

import bbbp_amd
from bbbp_amd import _lib, ops
from bbbp_amd.optim import AdamW
from helpers import synth_inputs

pytestmark = pytest.mark.gpu


@pytest.fixture
def dev():
    return torch.device("cuda:0")


def _steps(dev, defer, steps=6, B=48, F=167, seed=3, read_between=False):
    torch.manual_seed(seed)
    model = bbbp_amd.MixedInputModel(F, 128).to(dev).train()
    params = list(model.parameters())
    opt = AdamW(params, lr=1e-3, weight_decay=1e-5, defer=model.image_cnn[7].weight if defer else None)
    fp, img, y = (t.to(dev) for t in synth_inputs(11, 2 * B, F, 49152))
    crit = bbbp_amd.MSELoss()
    torch.manual_seed(99)                                   # the dropout seeds of both runs
    losses = []
    for i in range(steps):
        s = (i % 2) * B
        loss = crit(model(fp[s:s + B], img[s:s + B]).squeeze(), y[s:s + B])
        loss.backward()
        opt.step()
        opt.zero_grad(set_to_none=True)
        if read_between:
            # a reader outside the library between two steps: allocations on the current stream that could take the freed gradient block,
            # and a read of the deferred tensor after synchronize()
            junk = torch.randn(B, 49152, device=dev)
            opt.synchronize()
            losses.append(float(model.image_cnn[7].weight.abs().sum()) + 0.0 * float(junk[0, 0]))
        else:
            losses.append(float(loss.detach()))
    opt.synchronize()
    torch.cuda.synchronize()
    st = opt.state[model.image_cnn[7].weight]
    return (torch.cat([p.detach().flatten() for p in params]).cpu(), st["exp_avg"].flatten().cpu().clone(), st["exp_avg_sq"].flatten().cpu().clone(),
            int(st["step"]), losses)


@pytest.mark.parametrize("read_between", [False, True])
def test_deferred_image_fc_slice_is_bit_identical_to_the_one_launch_step(dev, read_between):
    """optim.AdamW(defer=image-FC weight) / bbbp_adamw_step_deferred: 62 % of the optimizer's bytes updated on a side stream beside the next
    forward pass, which waits for the slice right before its image FC.  Parameters, both moments and every loss after six steps (dropout on)
    are bit-identical to the one-launch step's -- a forward pass that read the weight too early, or a gradient buffer recycled under the
    side stream, would change them; with `read_between` the test also reads the tensor between steps through synchronize()."""
    a = _steps(dev, defer=False, read_between=read_between)
    b = _steps(dev, defer=True, read_between=read_between)
    assert a[3] == b[3] == 6
    assert a[4] == b[4]
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and torch.equal(a[2], b[2])


def test_deferred_step_then_eval_forward_and_state_dict_see_the_update(dev):
    """After a deferred step, an eval-mode forward of the SAME model and of ANOTHER model, state_dict() of the optimizer and a plain
    adamw_step_ are all ordered behind the slice (bbbp_param_wait at the library's entry points)."""
    torch.manual_seed(1)
    model = bbbp_amd.MixedInputModel(64, 128).to(dev).train()
    other = bbbp_amd.MixedInputModel(64, 128).to(dev).eval()
    w = model.image_cnn[7].weight
    opt = AdamW(model.parameters(), lr=1e-2, defer=w)
    ref = AdamW([torch.nn.Parameter(p.detach().clone()) for p in model.parameters()], lr=1e-2)
    fp, img, y = (t.to(dev) for t in synth_inputs(5, 16, 64, 49152))
    bbbp_amd.MSELoss()(model(fp, img).squeeze(), y).backward()
    for q, p in zip(ref.param_groups[0]["params"], model.parameters()):
        q.grad = p.grad.detach().clone()
    ref.step()
    opt.step()
    with torch.no_grad():
        other(fp, img)                                      # another model's forward: waits at its start
        model.eval()
        out = model(fp, img)                                # this model's forward: waits before the image FC
    sd = opt.state_dict()
    assert sd["state"]
    torch.cuda.synchronize()
    want = [q.detach() for q in ref.param_groups[0]["params"]]
    for p, q in zip(model.parameters(), want):
        assert torch.equal(p.detach(), q)
    assert torch.isfinite(out).all()


def test_layernorm_absorbing_engine_matches_the_standalone_layernorm_schedule(dev):
    """bbbp_set_ln_absorb(1): norm1 -> linear1 and norm2 -> the next in_proj / fingerprint_fc absorbed by the consuming GEMM, the dropout +
    residual in the producing GEMM's epilogue (same Philox elements).  Outputs, loss and every gradient of one training step with dropout ON
    agree with the default schedule to rounding, at the folded one-head width (167) and a multi-head width (64: the out_proj epilogue)."""
    L = _lib.lib()
    for F, B in ((167, 96), (64, 40)):
        fp, img, y = (t.to(dev) for t in synth_inputs(21, B, F, 49152))
        res = []
        for mode in (0, 1):
            old = L.bbbp_set_ln_absorb(mode)
            try:
                torch.manual_seed(7)
                model = bbbp_amd.MixedInputModel(F, 128).to(dev).train()
                torch.manual_seed(123)
                out = model(fp, img)
                loss = bbbp_amd.MSELoss()(out.squeeze(), y)
                loss.backward()
                res.append((out.detach().clone(), float(loss), {n: p.grad.detach().clone() for n, p in model.named_parameters()}))
            finally:
                L.bbbp_set_ln_absorb(old)
        (o0, l0, g0), (o1, l1, g1) = res
        assert float((o0 - o1).abs().max()) <= 2e-5 * float(o0.abs().max()) + 1e-6
        assert abs(l0 - l1) <= 1e-5 * abs(l0)
        for n in g0:
            if n.startswith("attention_fusion."):
                continue                                    # exact gradient 0: rounding noise in any implementation
            scale = float(g0[n].abs().max())
            assert float((g0[n] - g1[n]).abs().max()) <= 2e-4 * scale + 1e-9, (F, n, float((g0[n] - g1[n]).abs().max()), scale)


def test_pipelined_conv1_forward_in_training_plans_matches_f32_kernel(dev):
    """BBBP_C1_TRAIN (default on since round 4): the software-pipelined split-bf16 conv1 forward inside a training step, one work-group per
    CU.  Against the same step with conv mask bit 6 cleared (the f32 kernel of rounds 1-3): pooled activations agree to 2 ulp-ish, so output,
    loss and every gradient agree to rounding; the pooling decisions differ only at near-ties."""
    L = _lib.lib()
    F, B = 167, 64
    fp, img, y = (t.to(dev) for t in synth_inputs(33, B, F, 49152))
    res = []
    old = L.bbbp_get_conv_winograd()
    for mask in (old & ~64, old | 64):
        L.bbbp_set_conv_winograd(mask)
        try:
            torch.manual_seed(7)
            model = bbbp_amd.MixedInputModel(F, 128).to(dev).train()
            model.keep_workspace = True
            torch.manual_seed(5)
            out = model(fp, img)
            loss = bbbp_amd.MSELoss()(out.squeeze(), y)
            loss.backward()
            res.append((out.detach().clone(), float(loss), {n: p.grad.detach().clone() for n, p in model.named_parameters()}, model.debug_pool_masks()[0].clone()))
        finally:
            L.bbbp_set_conv_winograd(old)
    (o0, l0, g0, m0), (o1, l1, g1, m1) = res
    assert float((m0 != m1).float().mean()) < 1e-4
    assert float((o0 - o1).abs().max()) <= 2e-5 * float(o0.abs().max()) + 1e-6
    for n in g0:
        if n.startswith("attention_fusion."):
            continue
        scale = float(g0[n].abs().max())
        tol = 1e-2 if n.startswith("image_cnn.") else 5e-4           # the conv gradients: 10^6-term sums behind pooling decisions that flip at near-ties
        assert float((g0[n] - g1[n]).abs().max()) <= tol * scale + 1e-9, (n, float((g0[n] - g1[n]).abs().max()), scale)


# ---- multi-tensor / capturable AdamW and the graph-captured step of the per-op variants -------------------------------------------------
def _flat_params(dev, sizes, seed):
    g = torch.Generator().manual_seed(seed)
    flat = torch.randn(sum(int(np.prod(s)) for s in sizes), generator=g).to(dev)
    params, off = [], 0
    for s in sizes:
        n = int(np.prod(s))
        params.append(torch.nn.Parameter(flat[off:off + n].view(s)))
        off += n
    return flat, params


def test_multi_tensor_adamw_is_bit_identical_to_one_launch_per_tensor(dev):
    # odd sizes: tensors that start at offsets 1, 2, 3 mod 4 of the flat buffer, a tensor shorter than a float4, one longer than a launch wave
    sizes = [(167,), (3, 5), (1,), (2,), (513, 167), (4,), (167, 501), (7,), (1024, 256), (3,)]
    flat_a, pa = _flat_params(dev, sizes, 5)
    flat_b, pb = _flat_params(dev, sizes, 5)
    oa = AdamW(pa, lr=3e-3, weight_decay=1e-2)
    ob = [torch.zeros_like(flat_b), torch.zeros_like(flat_b)]
    g = torch.Generator().manual_seed(6)
    for step in range(1, 5):
        grads = [torch.randn(s, generator=g).to(dev) * 10.0 ** float(torch.randint(-6, 2, (1,), generator=g)) for s in sizes]
        for p, gr in zip(pa, grads):
            p.grad = gr.clone()                         # separate tensors: the multi-tensor launch
        assert ops is not None
        oa.step()
        off = 0
        for p, gr in zip(pb, grads):                     # the yardstick: one launch per tensor on views of flat buffers
            n = p.numel()
            ops.adamw_step_(p.data.view(-1), gr.contiguous().view(-1), ob[0][off:off + n], ob[1][off:off + n], step, lr=3e-3, weight_decay=1e-2)
            off += n
        assert torch.equal(flat_a, flat_b), f"step {step}"
        assert torch.equal(oa._flat_state[0][0], ob[0]) and torch.equal(oa._flat_state[0][1], ob[1])
    assert len(oa._tables[0]) >= 1                       # it did go through the table


def test_capturable_adamw_equals_the_plain_step_and_follows_the_learning_rate(dev):
    sizes = [(167,), (64, 167), (5,)]
    flat_a, pa = _flat_params(dev, sizes, 8)
    flat_b, pb = _flat_params(dev, sizes, 8)
    oa, ob = AdamW(pa, lr=1e-3, weight_decay=1e-2, capturable=True), AdamW(pb, lr=1e-3, weight_decay=1e-2)
    g = torch.Generator().manual_seed(9)
    for step in range(4):
        for o in (oa, ob):
            o.param_groups[0]["lr"] = 1e-3 / (1 + step)           # what a scheduler does
        grads = [torch.randn(s, generator=g).to(dev) for s in sizes]
        for p, q, gr in zip(pa, pb, grads):
            p.grad, q.grad = gr.clone(), gr.clone()
        oa.step(); ob.step()
        assert torch.equal(flat_a, flat_b), f"step {step}"
    assert oa.state[pa[0]]["step"] == ob.state[pb[0]]["step"] == 4


def _no_dropout(model):
    for m in model.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0


def _wide(dev, seed, dropout):
    from bbbp_amd import variants
    torch.manual_seed(seed)
    m = variants.WideDeepMixedInputModel(64, 128).to(dev).train()
    if not dropout:
        _no_dropout(m)
    return m


def test_graph_captured_step_equals_the_eager_step(dev):
    """Dropout off: five steps through GraphedTrainStep (two eager, capture, three replays) against five eager steps with the plain
    optimizer, on alternating batches -- same kernels in the same order, so the same bits (compared with a float32-rounding margin)."""
    from bbbp_amd.training import GraphedTrainStep
    B, F = 8, 64
    fp, img, y = (t.to(dev) for t in synth_inputs(21, 2 * B, F, 49152))
    ma, mb = _wide(dev, 4, False), _wide(dev, 4, False)
    oa = AdamW(ma.parameters(), lr=1e-3, weight_decay=1e-5, capturable=True)
    ob = AdamW(mb.parameters(), lr=1e-3, weight_decay=1e-5)
    graphed = GraphedTrainStep(ma, oa, eager_steps=2)
    la, lb = [], []
    for i in range(5):
        s = (i % 2) * B
        la.append(float(graphed(fp[s:s + B], img[s:s + B], y[s:s + B])))
        loss = torch.nn.MSELoss()(mb(fp[s:s + B], img[s:s + B]).squeeze(), y[s:s + B])
        loss.backward(); ob.step(); ob.zero_grad(set_to_none=True)
        lb.append(float(loss))
    assert graphed.graph is not None and graphed.calls == 5
    np.testing.assert_allclose(la, lb, rtol=1e-5)
    pa = torch.cat([p.detach().reshape(-1) for p in ma.parameters()])
    pb = torch.cat([p.detach().reshape(-1) for p in mb.parameters()])
    assert float((pa - pb).abs().max()) <= 1e-6 + 1e-5 * float(pb.abs().max())
    assert oa.state[next(iter(ma.parameters()))]["step"] == 5


def test_graph_captured_step_draws_new_dropout_masks_on_every_replay(dev):
    """lr = 0 and one fixed batch: the parameters never move, so the loss of a replay changes only through its dropout masks -- every
    replay must give another loss (the recorded seeds are frozen; the device-side step counter is what moves the streams)."""
    from bbbp_amd.training import GraphedTrainStep
    B, F = 8, 64
    fp, img, y = (t.to(dev) for t in synth_inputs(22, B, F, 49152))
    m = _wide(dev, 5, True)
    opt = AdamW(m.parameters(), lr=0.0, weight_decay=0.0, capturable=True)
    graphed = GraphedTrainStep(m, opt, eager_steps=1)
    losses = [float(graphed(fp, img, y)) for _ in range(6)]
    assert graphed.graph is not None
    assert len(set(losses[1:])) == 5, losses
    # ... and the eager ops that run afterwards are back on seed-only streams (the base is not left set)
    x = torch.randn(64, 64, device=dev)
    assert torch.equal(ops.dropout(x, 0.5, 123), ops.dropout(x, 0.5, 123))
    assert ops._SEED_BASE == 0


def test_train_fold_on_graph_steps_follows_the_eager_loop(dev):
    """The published loop's shape -- train mode in epoch 1 only, a ragged last batch -- through ``train_fold(graph_steps=True)``: one graph
    per (batch shape, mode), eager steps before each capture; dropout off, so the loss history must be the eager loop's."""
    from bbbp_amd.training import train_fold
    N, F, bs = 20, 64, 8
    fp, img, y = (t.to(dev) for t in synth_inputs(31, N + 6, F, 49152))
    train, test = (fp[:N], img[:N], y[:N]), (fp[N:], img[N:], y[N:])
    orders = [np.random.RandomState(e).permutation(N) for e in range(5)]
    hist = []
    for graph in (False, True):
        m = _wide(dev, 6, False)
        hist.append(train_fold(m, train, test, epochs=5, batch_size=bs, lr=1e-3, batch_orders=orders, graph_steps=graph))
    np.testing.assert_allclose(hist[1]["train_loss"], hist[0]["train_loss"], rtol=2e-4)
    np.testing.assert_allclose(hist[1]["val_loss"], hist[0]["val_loss"], rtol=2e-4)


@pytest.mark.parametrize("name,F,I", [("DenseMLPModel", 167, 768), ("PCAFusionModel", 64, 128)])
def test_graph_captured_step_of_the_mlp_variants(dev, name, F, I):
    """The dense and the PCA-fusion variants (BatchNorm running statistics, several dropout sites) through GraphedTrainStep against the
    eager loop, dropout off: same losses, same parameters, same running statistics after six steps."""
    from bbbp_amd import variants
    from bbbp_amd.training import GraphedTrainStep
    B = 16
    fp, img, y = (t.to(dev) for t in synth_inputs(41, 2 * B, F, I))
    models, opts = [], []
    for capt in (True, False):
        torch.manual_seed(7)
        m = getattr(variants, name)(F, I).to(dev).train()
        _no_dropout(m)
        models.append(m); opts.append(AdamW(m.parameters(), lr=1e-3, weight_decay=1e-5, capturable=capt))
    graphed = GraphedTrainStep(models[0], opts[0], eager_steps=2)
    for i in range(6):
        s = (i % 2) * B
        la = float(graphed(fp[s:s + B], img[s:s + B], y[s:s + B]))
        loss = torch.nn.MSELoss()(models[1](fp[s:s + B], img[s:s + B]).squeeze(), y[s:s + B])
        loss.backward(); opts[1].step(); opts[1].zero_grad(set_to_none=True)
        assert abs(la - float(loss)) <= 1e-5 * max(1.0, abs(float(loss))), (i, la, float(loss))
    assert graphed.graph is not None
    for (k, a), (_, b) in zip(models[0].state_dict().items(), models[1].state_dict().items()):
        if a.dtype.is_floating_point:
            assert float((a - b).abs().max()) <= 1e-6 + 1e-5 * float(b.abs().max()), k
        else:
            assert torch.equal(a, b), k


def test_adamw_kernel_follows_torch_optim_adamw_to_the_ulp(dev):
    """The reference's optimizer IS torch.optim.AdamW (R:172).  Same parameters, same gradients (magnitudes over eight decades), three
    steps with a changing learning rate: torch's float32 CPU step against the HIP kernel -- the kernel spells out torch's own sequence
    of roundings, so the moments are torch's bits and the parameters are torch's bits on > 99 % of the elements, a few units in the last place of
    the update's operands elsewhere."""
    n = 200_000
    g = torch.Generator().manual_seed(12)
    p0 = torch.randn(n, generator=g)
    ref = torch.nn.Parameter(p0.clone())
    ropt = torch.optim.AdamW([ref], lr=3e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, foreach=False)
    mine = torch.nn.Parameter(p0.clone().to(dev))
    opt = AdamW([mine], lr=3e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2)
    prev = p0.clone()
    for step in range(3):
        grad = torch.randn(n, generator=g) * 10.0 ** torch.randint(-6, 2, (n,), generator=g).float()
        for o in (ropt, opt):
            o.param_groups[0]["lr"] = 3e-3 / (1 + step)
        ref.grad, mine.grad = grad.clone(), grad.clone().to(dev)
        ropt.step(); opt.step()
        a, b = mine.detach().cpu(), ref.detach()
        # one unit in the last place AT THE SCALE OF THE OPERANDS of the final addition (p * decay + update: where the two cancel, one
        # ulp of an operand is many ulps of the small result)
        scale = torch.maximum(prev.abs(), b.abs())
        err = (a - b).abs() / (scale * 2.0 ** -23 + 1e-45)
        # torch's own bits depend on the host's vector ISA (tools/adamw_ulp_probe.py: its AVX2 addcdiv kernel agrees with this sequence on
        # all but 1e-4 of the elements, its AVX-512 kernel on all but 2.4e-3, each time by one unit in the last place of an operand)
        assert float(err.max()) <= 4.0 * (1 + step), (step, float(err.max()))
        assert float((a != b).float().mean()) < 1e-2 * (1 + step), (step, float((a != b).float().mean()))
        prev = b.clone()
        st = ropt.state[ref]
        assert torch.equal(opt.state[mine]["exp_avg"].cpu(), st["exp_avg"]), step         # the moments depend on the gradients only
        assert torch.equal(opt.state[mine]["exp_avg_sq"].cpu(), st["exp_avg_sq"]), step


@pytest.mark.parametrize("cin,cout", [(32, 64), (64, 128)])
@pytest.mark.parametrize("keep_mask", [True, False])
def test_pipelined_conv2_forward_is_bit_identical_to_the_two_work_group_kernel(dev, cin, cout, keep_mask):
    """conv_b3p_kernel<FWD> (one work-group per CU, software-pipelined: what training plans run beside the encoder chain) performs the same
    products in the same order as conv_b3_kernel: outputs and pooling decisions must be the same bits -- incl. a batch that leaves some
    work-groups without a strip and one that gives every work-group several."""
    L = _lib.lib()
    for B in (1, 3, 40):
        g = torch.Generator().manual_seed(100 + B)
        x = torch.randn(B, cin, 64, 64, generator=g).to(dev)
        w = (torch.randn(cout, cin, 3, 3, generator=g) * 0.1).to(dev)
        b = torch.randn(cout, generator=g).to(dev)
        outs = []
        for pipe in (0, 1):
            prev = L.bbbp_set_conv2_fwd_pipe(pipe)
            try:
                outs.append(ops.conv3x3_relu_pool_fwd(x, w, b, keep_mask=keep_mask))
            finally:
                L.bbbp_set_conv2_fwd_pipe(prev)
        (y0, m0), (y1, m1) = outs
        assert torch.equal(y0, y1), B
        if keep_mask:
            assert torch.equal(m0, m1), B
