"""GPU: every op-level C-ABI entry point against the CPU oracle / float64 torch on the same seeded inputs."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from bbbp_amd import _lib, ops
from oracle import reference_cpu as oracle
from helpers import assert_close, golden

pytestmark = pytest.mark.gpu


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


GEMM_SHAPES = [(512, 501, 167), (512, 167, 167), (512, 2048, 167), (512, 167, 2048), (7, 128, 167), (1, 1, 64),
               (33, 65, 17), (512, 128, 4096), (256, 256, 256), (130, 70, 9), (64, 64, 1),
               # the latency path: K slices inside a work-group, 2x2 sub-tiles, single / exact / ragged last chunk
               (512, 167, 512), (512, 167, 501), (2048, 167, 512), (40, 24, 16), (40, 24, 32), (40, 24, 33), (3, 5, 700),
               (1024, 1024, 64),
               # short K over > 4 tiles per CU: the 128 x 128 tile with 16-deep stages (the image FC's input gradient)
               (384, 49152, 128), (512, 33000, 100),
               # 128 x 128 tiles on the bf16 pipe with split operands: full tiles, ragged M / N, K not a multiple of the 32-deep stage,
               # split-K over a deep K; (516, 1028, 514): K % 4 != 0 keeps the f32 kernel for the k-contiguous layouts
               (512, 2048, 2048), (500, 2044, 516), (260, 6144, 2048), (2048, 2048, 512), (516, 1028, 514),
               # 64 x 64 tiles over the whole K (gemm_b3s_kernel: fewer than 256 tiles of 128 x 128 but >= 2.5 tiles of 64 x 64 per CU, K >= 512):
               # the F = 2048 encoder's in_proj at B = 512, ragged M / N with a 6-element K tail, and one that stays on the 128-tile plan
               (512, 6144, 2048), (200, 10240, 1030), (512, 2048, 6144),
               # a K that ends in a partial 32-deep stage on the bf16 pipe (B3Loader::load_tail): linear1 / Q K^T of a 4096-row screening
               # batch at the MACCS width (K = 167 = 5 stages + 7), and a tail of exactly one element
               (4096, 2048, 167), (4096, 4096, 167), (1024, 2048, 161),
               # tall, deep-K product with one or two column tiles (linear2 of a 4096-row screening batch): 128 x 128 plan, ragged N, split-K
               (4096, 167, 2048), (2048, 130, 1024)]


@pytest.fixture(params=[1, 0], ids=["split-bf16", "f32"])
def gemm_form(request):
    """Large products (128 x 128 tile plans) run on the bf16 matrix pipe with operands split into three bf16 pieces
    (bbbp_set_gemm_split_bf16, default on) or on the f32 MFMA."""
    L = _lib.lib()
    old = L.bbbp_set_gemm_split_bf16(request.param)
    yield request.param
    L.bbbp_set_gemm_split_bf16(old)


@pytest.mark.parametrize("M,N,K", GEMM_SHAPES)
@pytest.mark.parametrize("layout", ["nt", "nn", "tn"])
def test_gemm_layouts(dev, M, N, K, layout, gemm_form):
    a = rnd(M, K, seed=M + K); b = rnd(K, N, seed=N + 7)
    want = (a.double() @ b.double())
    scale = (a.abs().double() @ b.abs().double())
    if layout == "nt":
        got = ops.gemm(a.to(dev), b.t().contiguous().to(dev), trans_b=True)
    elif layout == "nn":
        got = ops.gemm(a.to(dev), b.to(dev))
    else:
        got = ops.gemm(a.t().contiguous().to(dev), b.to(dev), trans_a=True)
    err = (got.cpu().double() - want).abs()
    assert (err <= 2e-6 * scale + 1e-30).all(), f"max err ratio {(err / (scale + 1e-30)).max():.3e}"


@pytest.mark.parametrize("M,N,K,layout", [(512, 2048, 2048, "nt"), (512, 2048, 2048, "nn"), (512, 167, 65536, "nt"), (130, 300, 4100, "nt"),
                                          (2048, 2048, 512, "tn")])
def test_split_k_reduction_inside_the_gemm_launch_equals_the_reduce_kernel(dev, M, N, K, layout):
    """csrc/gemm.hip (GemmParams.arrivals): the K range that arrives last at an output tile sums the tile's slabs in split order and
    applies the epilogue (bias, ReLU, residual) itself.  Same summation order as gemm_splitk_reduce_kernel => the same bits, whoever
    arrives last; three launches in a row reuse the self-resetting counters.  Shapes: the F = 2048 encoder's linear1 and its input
    gradient at B = 512 (8 K ranges), the image FC forward (K = 65536), ragged tiles with a K tail, and an unsplit weight gradient."""
    L = _lib.lib()
    a = rnd(M, K, seed=M + K); b = rnd(K, N, seed=N + 7)
    bias = rnd(N, seed=3).to(dev); res = rnd(M, N, seed=4).to(dev)
    def run():
        kw = dict(bias=bias, residual=res, act="relu")
        if layout == "nt":
            return ops.gemm(a.to(dev), b.t().contiguous().to(dev), trans_b=True, **kw)
        if layout == "nn":
            return ops.gemm(a.to(dev), b.to(dev), **kw)
        return ops.gemm(a.t().contiguous().to(dev), b.to(dev), trans_a=True, **kw)
    old = L.bbbp_set_gemm_fold_reduce(0)
    try:
        want = run()
        L.bbbp_set_gemm_fold_reduce(1)
        for _ in range(3):
            got = run()
            assert torch.equal(got, want)
    finally:
        L.bbbp_set_gemm_fold_reduce(old)
    ref = torch.relu(a.double() @ b.double() + bias.cpu().double()) + res.cpu().double()
    scale = a.abs().double() @ b.abs().double() + 1.0
    assert ((got.cpu().double() - ref).abs() <= 2e-6 * scale).all()


def test_gemm_epilogues_and_slices(dev):
    M, N, K = 100, 70, 50
    a, w, bias, res = rnd(M, K, seed=1), rnd(N, K, seed=2), rnd(N, seed=3), rnd(M, N, seed=4)
    for act, fn in (("relu", torch.relu), ("tanh", torch.tanh), (None, lambda x: x)):
        want = fn(0.5 * (a.double() @ w.double().t()) + bias.double()) + res.double()
        got = ops.gemm(a.to(dev), w.to(dev), trans_b=True, alpha=0.5, bias=bias.to(dev), residual=res.to(dev), act=act)
        assert_close(got.cpu().numpy(), want.numpy(), rtol=1e-5, what=f"epilogue {act}")
    # write into a column slice of a wider buffer (how the branches fill `combined`)
    buf = torch.zeros(M, 2 * N, device=dev)
    ops.gemm(a.to(dev), w.to(dev), trans_b=True, out=buf[:, N:])
    assert_close(buf[:, N:].cpu().numpy(), (a.double() @ w.double().t()).numpy(), rtol=1e-5, what="slice out")
    assert float(buf[:, :N].abs().max()) == 0.0
    # batched (attention heads)
    qa, kb = rnd(5, 40, 8, seed=5), rnd(5, 40, 8, seed=6)
    got = ops.gemm(qa.to(dev), kb.to(dev), trans_b=True, alpha=0.25)
    assert_close(got.cpu().numpy(), (0.25 * qa.double() @ kb.double().transpose(1, 2)).numpy(), rtol=1e-5, what="batched")


@pytest.mark.parametrize("B,D,NH", [(512, 167, 1), (96, 8, 8), (40, 700, 1), (512, 2048, 1)])
def test_gemm_grouped_pairs_and_gate(dev, B, D, NH):
    """The attention-backward pairs (TN | NT and NN | TN) through one grouped call == separate float64 products."""
    F = D * NH
    pd, dctx, v = rnd(NH, B, B, seed=1), rnd(NH, B, D, seed=2), rnd(NH, B, D, seed=3)
    got = ops.gemm_grouped([dict(a=pd.to(dev), b=dctx.to(dev), trans_a=True),
                            dict(a=dctx.to(dev), b=v.to(dev), trans_b=True)])
    want = [pd.double().transpose(1, 2) @ dctx.double(), dctx.double() @ v.double().transpose(1, 2)]
    ds, k, q = rnd(NH, B, B, seed=4), rnd(NH, B, D, seed=5), rnd(NH, B, D, seed=6)
    got += ops.gemm_grouped([dict(a=ds.to(dev), b=k.to(dev), alpha=0.3), dict(a=ds.to(dev), b=q.to(dev), trans_a=True, alpha=0.3)])
    want += [0.3 * ds.double() @ k.double(), 0.3 * ds.double().transpose(1, 2) @ q.double()]
    for i, (g_, w_) in enumerate(zip(got, want)):
        assert_close(g_.cpu().numpy(), w_.numpy(), rtol=2e-5, atol_frac=2e-6, what=f"grouped product {i}")
    # gate: ReLU/dropout backward folded into the epilogue (both kernel families: small -> latency path, large -> LDS tiles)
    for (M, N, K) in ((B, 300, 167), (1024, 2048, 512)):
        a, w, gate, res = rnd(M, K, seed=7), rnd(K, N, seed=8), rnd(M, N, seed=9), rnd(M, N, seed=10)
        want = (a.double() @ w.double()) * (gate.double() > 0) * 1.25 + res.double()
        got1 = ops.gemm(a.to(dev), w.to(dev), gate=gate.to(dev), gate_scale=1.25, residual=res.to(dev))
        assert_close(got1.cpu().numpy(), want.numpy(), rtol=2e-5, atol_frac=2e-6, what=f"gated {M}x{N}x{K}")
        want2 = ((a.double() @ w.double()) + res.double()) * (gate.double() > 0) * 1.25
        got2 = ops.gemm(a.to(dev), w.to(dev), gate=gate.to(dev), gate_scale=1.25, residual=res.to(dev), gate_after_residual=True)
        assert_close(got2.cpu().numpy(), want2.numpy(), rtol=2e-5, atol_frac=2e-6, what=f"gate after residual {M}x{N}x{K}")
    assert ops.gemm_grouped([]) == []


@pytest.mark.parametrize("M,N,K", [(512, 2048, 167), (37, 300, 64), (16, 16, 16)])
def test_gemm_output_dropout_draws_the_dropout_kernels_stream(dev, M, N, K):
    """bbbp_gemm_desc.drop_p: linear1 + ReLU + dropout as ONE launch == the GEMM followed by bbbp_dropout with the same seed, bit for bit
    (element (m, n) takes the keep-scale of element m * N + n of the stream); the large-product path refuses it."""
    a, w, bias = rnd(M, K, seed=21).to(dev), rnd(N, K, seed=22).to(dev), rnd(N, seed=23).to(dev)
    two = ops.dropout(ops.gemm(a, w, trans_b=True, bias=bias, act="relu"), 0.1, 987654321)
    one = ops.gemm_grouped([dict(a=a, b=w, trans_b=True, bias=bias, act="relu", dropout_p=0.1, dropout_seed=987654321)])[0]
    assert torch.equal(one, two)
    kept = float((one != 0).float().mean()) / max(float((ops.gemm(a, w, trans_b=True, bias=bias, act="relu") != 0).float().mean()), 1e-9)
    assert 0.8 < kept < 0.97 or M * N < 1000
    big_a, big_w = torch.zeros(4096, 512, device=dev), torch.zeros(2048, 512, device=dev)
    with pytest.raises(RuntimeError):
        ops.gemm_grouped([dict(a=big_a, b=big_w, trans_b=True, dropout_p=0.1, dropout_seed=1)])


@pytest.mark.parametrize("M,N,K,p", [(512, 167, 167, 0.1), (512, 167, 2048, 0.1), (37, 64, 300, 0.0), (5, 256, 16, 0.25), (16, 1, 7, 0.0)])
def test_linear_layernorm_fused_matches_gemm_then_layernorm(dev, M, N, K, p):
    """out_proj -> dropout + residual -> LayerNorm as one launch == the GEMM followed by bbbp_layernorm_fwd (same Philox elements,
    so the dropped positions are identical; sums are ordered differently, hence rounding-level tolerances)."""
    x, w, b = rnd(M, K, seed=31).to(dev), rnd(N, K, seed=32, scale=0.2).to(dev), rnd(N, seed=33).to(dev)
    res, gam, bet = rnd(M, N, seed=34).to(dev), (1 + 0.1 * rnd(N, seed=35)).to(dev), rnd(N, seed=36).to(dev)
    y1, z1, m1, r1 = ops.linear_layernorm_fwd(x, w, b, res, gam, bet, dropout_p=p, seed=4242)
    y0, z0, m0, r0 = ops.layernorm_fwd(ops.gemm(x, w, trans_b=True, bias=b), res, gam, bet, dropout_p=p, seed=4242)
    if p > 0:
        lin = ops.gemm(x, w, trans_b=True, bias=b)
        assert torch.equal((z1 - res).abs() < 1e-12 * 0, (z0 - res).abs() < 1e-12 * 0)          # shapes only; the real check follows
        dropped0 = ((z0 - res).abs() <= 1e-6 * lin.abs().clamp_min(1e-3)) & (lin.abs() > 1e-2)
        dropped1 = ((z1 - res).abs() <= 1e-6 * lin.abs().clamp_min(1e-3)) & (lin.abs() > 1e-2)
        assert torch.equal(dropped0, dropped1)
    assert_close(z1.cpu().numpy(), z0.cpu().numpy(), rtol=2e-5, atol_frac=2e-6, what="z")
    assert_close(y1.cpu().numpy(), y0.cpu().numpy(), rtol=1e-4, atol_frac=1e-5, what="y")
    assert_close(m1.cpu().numpy(), m0.cpu().numpy(), rtol=1e-4, atol_frac=1e-5, what="mean")
    assert_close(r1.cpu().numpy(), r0.cpu().numpy(), rtol=1e-4, atol_frac=1e-6, what="rstd")
    # against float64
    if p == 0:
        zz = x.double().cpu() @ w.double().cpu().t() + b.double().cpu() + res.double().cpu()
        want = torch.nn.functional.layer_norm(zz, (N,), gam.double().cpu(), bet.double().cpu(), 1e-5)
        assert_close(y1.cpu().numpy(), want.numpy(), rtol=1e-4, atol_frac=1e-5, what="y vs float64")
    with pytest.raises(RuntimeError):
        ops.linear_layernorm_fwd(torch.zeros(4, 8, device=dev), torch.zeros(300, 8, device=dev), None, None, torch.ones(300, device=dev), torch.zeros(300, device=dev))


@pytest.mark.parametrize("M,N,K,act,p", [(512, 2048, 167, "relu", 0.1), (512, 501, 167, None, 0.0), (37, 128, 167, "relu", 0.0), (5, 33, 64, None, 0.0),
                                         (100, 70, 176, "tanh", 0.25), (33, 17, 192, None, 0.0), (7, 16, 16, None, 0.0), (3, 5, 7, "relu", 0.0)])
def test_layernorm_absorbed_by_the_consuming_linear(dev, M, N, K, act, p):
    """bbbp_layernorm_linear_fwd (round 4): out = dropout(act(LayerNorm(z) W^T + b)) with the LayerNorm taken from the product's own operand
    stream, and y / mean / rstd written by the same launch.  Against (i) the float64 composition, (ii) the two-launch schedule it
    replaces -- bbbp_layernorm_fwd followed by the GEMM with the same dropout stream: y, mean, rstd and out to rounding, the same dropped
    positions.  Rows with a LARGE common offset (mean >> spread: the pivot
    keeps the row sums from cancelling) are part of the case."""
    z = rnd(M, K, seed=41)
    z[: max(1, M // 3)] += 30.0                                # mean / std ~ 30 (the stand-alone kernel's own (z - mean) loses ~2e-6 there)
    z = z.to(dev)
    gam, bet = (1 + 0.2 * rnd(K, seed=42)).to(dev), (0.3 * rnd(K, seed=43)).to(dev)
    w, b = rnd(N, K, seed=44, scale=0.2).to(dev), rnd(N, seed=45).to(dev)
    acts = {None: 0, "relu": 1, "tanh": 2}
    out, y, mean, rstd = ops.layernorm_linear_fwd(z, gam, bet, w, b, act=acts[act], dropout_p=p, seed=777)
    y0, _, m0, r0 = ops.layernorm_fwd(z.clone(), None, gam, bet)
    # (the writer work-groups run the row kernel's arithmetic: equal to the last bits or so -- two translation units, two FMA contractions)
    assert_close(mean.cpu().numpy(), m0.cpu().numpy(), rtol=1e-6, atol_frac=1e-7, what="mean")
    assert_close(rstd.cpu().numpy(), r0.cpu().numpy(), rtol=2e-6, atol_frac=1e-7, what="rstd")
    assert_close(y.cpu().numpy(), y0.cpu().numpy(), rtol=1e-5, atol_frac=2e-6, what="y")
    if p > 0 and ops._lib.lib().bbbp_gemm_folds_asum(M, N, K, 1):
        two = ops.gemm_grouped([dict(a=y0, b=w, trans_b=True, bias=b, act=act, dropout_p=p, dropout_seed=777)])[0]
    else:
        two = ops.gemm(y0, w, trans_b=True, bias=b, act=act)
        if p > 0:
            two = ops.dropout(two, p, 777)
    assert torch.equal(out == 0, two == 0) or act == "relu" and float(((out == 0) != (two == 0)).float().mean()) < 1e-4   # ReLU near-zeros may differ
    assert_close(out.cpu().numpy(), two.cpu().numpy(), rtol=2e-4, atol_frac=5e-5, what="out vs layernorm + gemm")
    zz = z.double().cpu()
    want_y = torch.nn.functional.layer_norm(zz, (K,), gam.double().cpu(), bet.double().cpu(), 1e-5)
    lin = want_y @ w.double().cpu().t() + b.double().cpu()
    lin = torch.relu(lin) if act == "relu" else torch.tanh(lin) if act == "tanh" else lin
    keep = (two != 0).cpu() | (lin == 0) if p > 0 else torch.ones_like(lin, dtype=torch.bool)
    scale = 1.0 / (1.0 - p) if p > 0 else 1.0
    assert_close(torch.where(keep, out.cpu().double(), torch.zeros_like(lin)).numpy(), torch.where(keep, lin * scale, torch.zeros_like(lin)).numpy(),
                 rtol=2e-4, atol_frac=2e-5, what="out vs float64")
    with pytest.raises(RuntimeError):
        ops.layernorm_linear_fwd(torch.zeros(4, 200, device=dev), torch.ones(200, device=dev), torch.zeros(200, device=dev), torch.zeros(8, 200, device=dev), None)


def test_gemm_errors(dev):
    with pytest.raises(RuntimeError):
        ops.gemm(torch.zeros(4, 5), torch.zeros(5, 6))                    # CPU tensors: no fallback
    with pytest.raises(RuntimeError):
        ops.gemm(torch.zeros(4, 5, device=dev), torch.zeros(6, 6, device=dev))
    with pytest.raises(RuntimeError):
        ops.gemm(torch.zeros(4, 5, device=dev, dtype=torch.float64), torch.zeros(5, 6, device=dev, dtype=torch.float64))
    out = ops.gemm(torch.zeros(0, 5, device=dev), torch.zeros(5, 6, device=dev))   # empty batch
    assert out.shape == (0, 6)


def _conv_case(dev, B, cin, cout, hw, seed, uniform_patches=False):
    x = rnd(B, cin, hw, hw, seed=seed)
    if uniform_patches:        # flat regions => exact ties inside pooling windows (white image background)
        x[:, :, : hw // 2, :] = 1.0
    w = rnd(cout, cin, 3, 3, seed=seed + 1, scale=0.2)
    b = rnd(cout, seed=seed + 2, scale=0.1)
    # float64 oracle: torch's own fp32 CPU weight-gradient carries ~2e-3 (of max) summation error at these sizes
    xr, wr, br = (t_.double().clone().requires_grad_(True) for t_ in (x, w, b))
    y, mask = ops.conv3x3_relu_pool_fwd(x.to(dev), w.to(dev), b.to(dev))
    assert int(mask.max()) <= 4
    # The gradients are sums over POOLING DECISIONS (which of four pre-activations is the maximum; whether it is positive).  A window whose
    # two largest values agree to float32 rounding may legitimately route its gradient to either -- ONE such window moves 9 * cin entries of
    # dw by |gy * x| (round 4: the 128 -> 256 stage at B = 20 on the split-bf16 forward has one in 1.3 M windows).  So, as
    # tests/test_gpu_parity_sizes.py does at B = 512: (i) the kernel's decisions may differ from float64's own only at near-ties -- the value
    # it picked within 1e-5 (of the mean |pre-activation|) of float64's maximum, at < 1e-4 of the windows; (ii) the oracle then takes the
    # kernel's decisions, and every gradient element is compared.
    pre = []
    with torch.no_grad():
        own = oracle.conv3x3_relu_pool(xr, wr, br, pre=pre)
    dec64 = oracle.pool_decisions(pre[0])
    m_cpu = mask.cpu()
    differ = m_cpu != dec64
    if differ.any():
        win = oracle.pool_windows(pre[0])
        picked = torch.where(m_cpu < 4, win.gather(-1, m_cpu.clamp(max=3).long().unsqueeze(-1)).squeeze(-1), torch.zeros_like(own))
        gap = (own - picked).abs()[differ]
        assert float(gap.max()) <= 1e-5 * float(pre[0].abs().mean()) + 1e-12, float(gap.max())
        assert float(differ.float().mean()) < 1e-4
    yr = oracle.conv3x3_relu_pool(xr, wr, br, pool_mask=m_cpu)
    gy = rnd(*yr.shape, seed=seed + 3)
    yr.backward(gy.double())
    assert_close(y.cpu().numpy(), yr.detach().numpy(), rtol=1e-5, atol_frac=1e-6, what="conv fwd")
    dw, db = ops.conv3x3_relu_pool_bwd_weight(x.to(dev), gy.to(dev), mask)
    assert_close(dw.cpu().numpy(), wr.grad.numpy(), rtol=1e-4, atol_frac=2e-5, what="conv dw")
    assert_close(db.cpu().numpy(), br.grad.numpy(), rtol=1e-4, atol_frac=2e-5, what="conv db")
    if cin != 3:
        dx = ops.conv3x3_relu_pool_bwd_data(gy.to(dev), mask, w.to(dev))
        assert_close(dx.cpu().numpy(), xr.grad.numpy(), rtol=1e-4, atol_frac=2e-5, what="conv dx")


@pytest.fixture(params=[0, 32, 96], ids=["f32", "split-bf16-wgrad", "split-bf16-fwd+wgrad"])
def conv1_algo(request):
    """The forms of the 3 -> 32 @ 128x128 stage: f32 MFMA kernels; bit 5 of bbbp_set_conv_winograd: the weight gradient on the bf16
    matrix pipe with split operands (conv_b3.hip); bit 6 (round 3): the forward too (conv_b3c1.hip: channel-innermost LDS strip, a
    lane's operand is two taps = two aligned 8-byte reads, no operand assembly -- round 2's first split-bf16 forward assembled operands
    per lane with 72 alignbits per 36 MFMAs and lost)."""
    L = _lib.lib()
    old = L.bbbp_get_conv_winograd()
    _lib.check(L.bbbp_set_conv_winograd(request.param), "bbbp_set_conv_winograd")
    yield request.param
    L.bbbp_set_conv_winograd(old)


@pytest.mark.parametrize("B", [1, 3, 9])
def test_conv1_3to32(dev, B, conv1_algo):
    _conv_case(dev, B, 3, 32, 128, seed=10 + B)


def test_conv1_split_bf16_forward_keeps_ties_and_matches_f32(dev):
    """conv_b3c1.hip against the f32 form on an image with a flat band (the white background of the depictions: the four pre-activations
    of a pooling window are computed from identical inputs, hence bit-equal in either form, and PyTorch's first-maximum rule must pick
    position 0 -- or 4 when the value is not positive): masks identical on the band and on >= 99.99 % of the random part, values to
    float32 rounding; a ragged batch (B = 5 < one strip set per work-group) and the image borders are in the comparison."""
    L = _lib.lib()
    old = L.bbbp_get_conv_winograd()
    x = rnd(5, 3, 128, 128, seed=91)
    x[:, :, 20:90, :] = 1.0                                  # flat band incl. the left / right borders
    x[1] = 1.0                                               # a completely flat image: every window a tie, every border case
    w, b = rnd(32, 3, 3, 3, seed=92, scale=0.2), rnd(32, seed=93, scale=0.1)
    try:
        L.bbbp_set_conv_winograd(0)
        y0, m0 = ops.conv3x3_relu_pool_fwd(x.to(dev), w.to(dev), b.to(dev))
        L.bbbp_set_conv_winograd(64)
        y1, m1 = ops.conv3x3_relu_pool_fwd(x.to(dev), w.to(dev), b.to(dev))
    finally:
        L.bbbp_set_conv_winograd(old)
    assert float((y1 - y0).abs().max()) <= 2e-6 * float(y0.abs().max())
    band = slice(11, 44)                                     # pooled rows whose 3x3 neighbourhoods lie inside the flat band
    assert torch.equal(m1[:, :, band, 1:63], m0[:, :, band, 1:63])
    assert set(torch.unique(m1[:, :, band, 1:63]).tolist()) <= {0, 4}
    assert torch.equal(m1[1], m0[1])                         # the flat image, borders included
    assert float((m1 != m0).float().mean()) < 1e-4           # near-ties on the random part only


@pytest.fixture(params=[0, 3, 28, 28 | 128, 28 | 128 | 256], ids=["direct", "winograd", "split-bf16", "split-bf16-sparse-wgrad-8w", "split-bf16-sparse-wgrad-4w"])
def conv2_algo(request):
    """The forms of the 32 -> 64 @ 64x64 stage (bbbp_set_conv_winograd): direct implicit GEMM on the f32 MFMA, Winograd F(2x2,3x3),
    the direct form on the bf16 matrix pipe with every float32 operand split into three bf16 pieces (conv_b3.hip: forward, data
    gradient and weight gradient), and that form with the weight gradient on the 2:4 structured-sparse MFMA (the pooled gradient is the
    compressed operand; 8-wave and 4-wave work-groups)."""
    L = _lib.lib()
    old = L.bbbp_get_conv_winograd()
    _lib.check(L.bbbp_set_conv_winograd(request.param), "bbbp_set_conv_winograd")
    yield request.param
    L.bbbp_set_conv_winograd(old)


@pytest.mark.parametrize("B", [1, 2, 5])
def test_conv2_32to64(dev, B, conv2_algo):
    _conv_case(dev, B, 32, 64, 64, seed=20 + B)


def test_conv2_winograd_keeps_ties_and_matches_direct(dev):
    """Flat regions (the white background of the depictions) give bit-equal outputs inside a pooling window in both forms,
    so the arg-max mask -- which routes the gradient -- is the same; values agree to float32 rounding."""
    L = _lib.lib()
    old = L.bbbp_get_conv_winograd()
    x = torch.relu(rnd(4, 32, 64, 64, seed=77))
    x[:, :, 8:40, :] = 1.0                                   # flat band
    w, b = rnd(64, 32, 3, 3, seed=78, scale=0.2), rnd(64, seed=79, scale=0.1)
    gy = rnd(4, 64, 32, 32, seed=80)
    try:
        L.bbbp_set_conv_winograd(0)
        y0, m0 = ops.conv3x3_relu_pool_fwd(x.to(dev), w.to(dev), b.to(dev))
        dx0 = ops.conv3x3_relu_pool_bwd_data(gy.to(dev), m0, w.to(dev))
        L.bbbp_set_conv_winograd(3)
        y1, m1 = ops.conv3x3_relu_pool_fwd(x.to(dev), w.to(dev), b.to(dev))
        dx1 = ops.conv3x3_relu_pool_bwd_data(gy.to(dev), m0, w.to(dev))
    finally:
        L.bbbp_set_conv_winograd(old)
    flat = slice(5, 19)                                      # pooled rows whose 4x4 patches lie inside the flat band
    assert torch.equal(m0[:, :, flat, 1:-1], m1[:, :, flat, 1:-1])
    assert float((m0 != m1).float().mean()) < 1e-4
    assert_close(y1.cpu().numpy(), y0.cpu().numpy(), rtol=1e-5, atol_frac=2e-6, what="winograd vs direct fwd")
    assert_close(dx1.cpu().numpy(), dx0.cpu().numpy(), rtol=1e-5, atol_frac=2e-6, what="winograd vs direct dgrad")


@pytest.mark.parametrize("cin,cout,hw,B", [(3, 64, 128, 2), (64, 128, 64, 2), (128, 256, 32, 3), (128, 256, 32, 20)])
def test_conv_wide_deep_shapes(dev, cin, cout, hw, B):
    """The three conv stages of the wide/deep variant (Models/..._opt_20250107_network.py:129-138): output-channel
    blocks in forward / data-gradient, (input block, output block) pairs in the weight gradient."""
    _conv_case(dev, B, cin, cout, hw, seed=100 + cin + B)


def test_conv_pool_ties_follow_first_max(dev, conv2_algo):
    _conv_case(dev, 2, 3, 32, 128, seed=31, uniform_patches=True)
    _conv_case(dev, 2, 32, 64, 64, seed=32, uniform_patches=True)


def test_conv_many_strips_persistent_loop(dev, conv2_algo):
    """More strips than work-groups: exercises the grid-stride loop and the double buffer across strips."""
    _conv_case(dev, 40, 32, 64, 64, seed=41)
    _conv_case(dev, 24, 3, 32, 128, seed=42)


@pytest.mark.parametrize("form", [0, 3, 28, 64], ids=["f32", "winograd", "split-bf16", "split-bf16-conv1"])
def test_conv_forward_without_mask_equals_forward_with_mask(dev, form):
    """Forward-only plans (eval loop, screening) pass a NULL mask: the pooled activations are bit-identical to the training call's and
    nothing is written where the decisions would go (every forward form: conv.hip, conv_wino.hip, conv_b3.hip, conv_b3c1.hip)."""
    L = _lib.lib()
    old = L.bbbp_get_conv_winograd()
    try:
        L.bbbp_set_conv_winograd(form)
        for cin, cout, hw, B in ((3, 32, 128, 5), (32, 64, 64, 3)):
            x, w, b = rnd(B, cin, hw, hw, seed=7).to(dev), rnd(cout, cin, 3, 3, seed=8, scale=0.2).to(dev), rnd(cout, seed=9, scale=0.1).to(dev)
            y0, m0 = ops.conv3x3_relu_pool_fwd(x, w, b)
            y1, m1 = ops.conv3x3_relu_pool_fwd(x, w, b, keep_mask=False)
            assert m1 is None and m0 is not None and torch.equal(y0, y1)
    finally:
        L.bbbp_set_conv_winograd(old)


def test_conv_rejects_unsupported(dev):
    with pytest.raises(RuntimeError):
        ops.conv3x3_relu_pool_fwd(torch.zeros(1, 5, 64, 64, device=dev), torch.zeros(8, 5, 3, 3, device=dev), torch.zeros(8, device=dev))


@pytest.mark.parametrize("rows,cols", [(5, 7), (512, 167), (33, 2048), (4, 64), (512, 2048), (3, 1024), (7, 1500), (2, 4096), (5, 1022), (3, 4100)])
def test_layernorm(dev, rows, cols):
    x, r = rnd(rows, cols, seed=1), rnd(rows, cols, seed=2)
    gam, bet, dy = rnd(cols, seed=3), rnd(cols, seed=4), rnd(rows, cols, seed=5)
    xr, rr, gr, br = (t.clone().double().requires_grad_(True) for t in (x, r, gam, bet))
    yr = F.layer_norm(xr + rr, (cols,), gr, br, 1e-5)
    yr.backward(dy.double())
    y, z, mean, rstd = ops.layernorm_fwd(x.to(dev), r.to(dev), gam.to(dev), bet.to(dev))
    assert_close(y.cpu().numpy(), yr.detach().numpy(), rtol=1e-5, what="ln fwd")
    dz, dx, dg, db = ops.layernorm_bwd(dy.to(dev), z, gam.to(dev), mean, rstd)
    assert_close(dz.cpu().numpy(), xr.grad.numpy(), rtol=1e-4, atol_frac=2e-5, what="ln dx")
    assert_close(dg.cpu().numpy(), gr.grad.numpy(), rtol=1e-4, atol_frac=2e-5, what="ln dgamma")
    assert_close(db.cpu().numpy(), br.grad.numpy(), rtol=1e-4, atol_frac=2e-5, what="ln dbeta")


@pytest.mark.parametrize("rows,cols", [(6, 2048), (3, 1500), (5, 167)])
def test_layernorm_dropout_draws_the_dropout_kernels_stream(dev, rows, cols):
    """z = dropout(x) + residual inside the LayerNorm kernels (the work-group-per-row form for 1024..4096 columns draws one Philox block
    per float4, the wave-per-row form one per element): both must give the elements bbbp_dropout draws for the contiguous [rows, cols]
    tensor -- bit for bit -- and the backward's dx must apply the same mask."""
    x, r, gam, bet, dy = rnd(rows, cols, seed=1).to(dev), rnd(rows, cols, seed=2).to(dev), rnd(cols, seed=3).to(dev), rnd(cols, seed=4).to(dev), rnd(rows, cols, seed=5).to(dev)
    p, seed = 0.3, 777
    y, z, mean, rstd = ops.layernorm_fwd(x, r, gam, bet, dropout_p=p, seed=seed)
    xd = ops.dropout(x, p, seed)
    assert torch.equal(z, xd + r)
    kept = (xd != 0).float().mean().item()
    assert abs(kept - (1 - p)) < 0.06
    want = F.layer_norm(z.double().cpu(), (cols,), gam.double().cpu(), bet.double().cpu(), 1e-5)
    assert_close(y.cpu().numpy(), want.numpy(), rtol=1e-5, what="ln(dropout) fwd")
    dz, dx, _, _ = ops.layernorm_bwd(dy, z, gam, mean, rstd, dropout_p=p, seed=seed)
    assert torch.equal(dx, ops.dropout(dz, p, seed))


@pytest.mark.parametrize("rows,cols", [(9, 9), (512, 512), (3, 4096), (70, 33)])
def test_softmax(dev, rows, cols):
    x, dp = rnd(rows, cols, seed=1, scale=3.0), rnd(rows, cols, seed=2)
    xr = x.clone().double().requires_grad_(True)
    pr = torch.softmax(xr, dim=-1)
    pr.backward(dp.double())
    p, pd = ops.softmax_fwd(x.to(dev))
    assert pd is p
    assert_close(p.cpu().numpy(), pr.detach().numpy(), rtol=1e-5, what="softmax")
    ds = ops.softmax_bwd(dp.to(dev), p)
    assert_close(ds.cpu().numpy(), xr.grad.numpy(), rtol=1e-4, atol_frac=2e-5, what="softmax bwd")


def test_batchnorm_golden_and_random(dev):
    g = golden("ops")
    x, w, b = (torch.from_numpy(g[k]).to(dev) for k in ("bn/x", "bn/w", "bn/b"))
    rm, rv = torch.zeros(6, device=dev), torch.ones(6, device=dev)
    y, sm, sr = ops.batchnorm1d_fwd(x, w, b, rm, rv, training=True)
    assert_close(y.cpu().numpy(), g["bn/y_train"], rtol=1e-5, what="bn train")
    assert_close(rm.cpu().numpy(), g["bn/running_mean"], rtol=1e-5, what="bn running_mean")
    assert_close(rv.cpu().numpy(), g["bn/running_var"], rtol=1e-5, what="bn running_var")
    dx, dg, db = ops.batchnorm1d_bwd(torch.from_numpy(g["bn/gy"]).to(dev), x, w, sm, sr, training=True)
    assert_close(dx.cpu().numpy(), g["bn/gx"], rtol=1e-4, atol_frac=2e-5, what="bn dx")
    assert_close(dg.cpu().numpy(), g["bn/gw"], rtol=1e-4, atol_frac=2e-5, what="bn dgamma")
    assert_close(db.cpu().numpy(), g["bn/gb"], rtol=1e-4, atol_frac=2e-5, what="bn dbeta")
    y2, _, _ = ops.batchnorm1d_fwd(x, w, b, rm, rv, training=False)
    assert_close(y2.cpu().numpy(), g["bn/y_eval"], rtol=1e-5, what="bn eval")
    with pytest.raises(RuntimeError, match="more than 1 value per channel"):
        ops.batchnorm1d_fwd(x[:1].contiguous(), w, b, rm, rv, training=True)     # same failure as nn.BatchNorm1d
    # larger random case, eval-mode backward
    X = rnd(300, 256, seed=9).to(dev); W = rnd(256, seed=10).to(dev); Bb = rnd(256, seed=11).to(dev)
    rm, rv = rnd(256, seed=12).to(dev), (rnd(256, seed=13).abs() + 0.5).to(dev)
    dy = rnd(300, 256, seed=14)
    Xr = X.cpu().double().requires_grad_(True)
    yr = F.batch_norm(Xr, rm.cpu().double(), rv.cpu().double(), W.cpu().double(), Bb.cpu().double(), False, 0.1, 1e-5)
    yr.backward(dy.double())
    y, sm, sr = ops.batchnorm1d_fwd(X, W, Bb, rm, rv, training=False)
    dx, _, _ = ops.batchnorm1d_bwd(dy.to(dev), X, W, sm, sr, training=False)
    assert_close(y.cpu().numpy(), yr.detach().numpy(), rtol=1e-5, what="bn eval big")
    assert_close(dx.cpu().numpy(), Xr.grad.numpy(), rtol=1e-4, what="bn eval dx")


def test_bias_act_bwd_and_mse(dev):
    y = torch.relu(rnd(50, 70, seed=1)); dy = rnd(50, 70, seed=2)
    want = dy * (y > 0)
    d = dy.clone().to(dev)
    db = ops.bias_act_bwd(d, y.to(dev), act="relu")
    assert_close(d.cpu().numpy(), want.numpy(), rtol=1e-6, what="relu bwd")
    assert_close(db.cpu().numpy(), want.double().sum(0).numpy(), rtol=1e-5, what="db")
    pred, tgt = rnd(37, seed=3), rnd(37, seed=4)
    loss, dpred = ops.mse(pred.to(dev), tgt.to(dev))
    assert_close(loss.cpu().numpy(), [float(((pred - tgt) ** 2).mean())], rtol=1e-5, what="mse")
    assert_close(dpred.cpu().numpy(), (2 * (pred - tgt) / 37).numpy(), rtol=1e-5, what="dmse")


def test_dropout_statistics_and_determinism(dev):
    x = torch.ones(1 << 20, device=dev)
    y1, y2, y3 = ops.dropout(x, 0.1, seed=5), ops.dropout(x, 0.1, seed=5), ops.dropout(x, 0.1, seed=6)
    assert torch.equal(y1, y2) and not torch.equal(y1, y3)
    keep = float((y1 > 0).float().mean())
    assert abs(keep - 0.9) < 2e-3
    assert abs(float(y1.mean()) - 1.0) < 3e-3 and abs(float(y1.max()) - 1 / 0.9) < 1e-6


def test_adamw_matches_oracle(dev):
    p, g_ = rnd(1000, seed=1), rnd(1000, seed=2)
    m, v = torch.zeros(1000), torch.zeros(1000)
    pd, md, vd = p.clone().to(dev), m.clone().to(dev), v.clone().to(dev)
    for step in range(1, 4):
        gs = g_ * step
        oracle.adamw_step(p, gs, m, v, step)
        ops.adamw_step_(pd, gs.to(dev), md, vd, step)
    assert_close(pd.cpu().numpy(), p.numpy(), rtol=1e-5, what="adamw p")
    assert_close(vd.cpu().numpy(), v.numpy(), rtol=1e-5, what="adamw v")


@pytest.mark.parametrize("M,N,K", [(512, 167, 2048), (512, 2048, 167), (512, 501, 167), (37, 1, 64), (512, 128, 167), (7, 256, 256),
                                   (512, 16, 16), (3, 5, 3)])
def test_weight_gradient_gemm_folds_the_bias_gradient(dev, M, N, K):
    """dW = dy^T x with db = column sums of dy out of the same launch (bbbp_gemm_desc.asum: a virtual all-ones column of x):
    the Linear shapes of the encoder and the head, N a multiple of 16 (the ones column opens a new tile) or not."""
    from bbbp_amd import ops
    g = torch.Generator().manual_seed(M * 7 + N)
    dy = torch.randn(M, N, generator=g).to(dev)
    x = torch.randn(M, K, generator=g).to(dev)
    dw, db = ops.linear_weight_bias_grad(dy, x)
    ref_w = dy.double().t() @ x.double()
    ref_b = dy.double().sum(dim=0)
    assert_close(dw.cpu().numpy(), ref_w.cpu().numpy(), rtol=1e-5, atol_frac=2e-6, what="dW")
    assert_close(db.cpu().numpy(), ref_b.cpu().numpy(), rtol=1e-5, atol_frac=2e-6, what="db")
    # a strided dy (the first 128 of 256 columns, as fingerprint_fc's gradient inside `combined`)
    if N >= 4:
        wide = torch.randn(M, 2 * N, generator=g).to(dev)
        dw2, db2 = ops.linear_weight_bias_grad(wide[:, :N], x)
        assert_close(db2.cpu().numpy(), wide[:, :N].double().sum(dim=0).cpu().numpy(), rtol=1e-5, atol_frac=2e-6, what="db strided")
        assert_close(dw2.cpu().numpy(), (wide[:, :N].double().t() @ x.double()).cpu().numpy(), rtol=1e-5, atol_frac=2e-6, what="dW strided")
