"""No GPU: the algebra csrc/fold.hip relies on, in float64 against torch autograd.  For one head,
    out_proj(Pd V) = Pd (x W'^T + 1 b'^T) + bo        with  W' = Wo Wv,  b' = Wo bv                       (forward)
    dWo = dW' Wv^T + db' bv^T,   d[Wv | bv] = Wo^T [dW' | db'],   dbo = column sums of dz                  (gradients, unfolded)
where dW' | db' is the gradient of the folded block, whatever the (dropped, not row-normalised) attention weights Pd are
(nn.TransformerEncoderLayer's self-attention block as the reference builds it for nhead = 1, ...20250113.py:75-78)."""
import torch


def test_folded_out_proj_is_the_same_function_and_its_gradients_unfold():
    g = torch.Generator().manual_seed(11)
    B, F = 23, 17
    rnd = lambda *s: torch.randn(*s, generator=g, dtype=torch.float64)
    x = rnd(B, F)
    Wv, bv, Wo, bo = (rnd(F, F).requires_grad_(), rnd(F).requires_grad_(), rnd(F, F).requires_grad_(), rnd(F).requires_grad_())
    keep = (torch.rand(B, B, generator=g) > 0.1).double() / 0.9
    Pd = torch.softmax(rnd(B, B), dim=1) * keep            # dropped attention weights: rows do NOT sum to one
    dz = rnd(B, F)
    # the reference's order
    z = (Pd @ (x @ Wv.t() + bv)) @ Wo.t() + bo
    z.backward(dz)
    want = [t.grad.clone() for t in (Wv, bv, Wo, bo)]
    # folded: one block of the in_proj weight, no out_proj GEMM
    with torch.no_grad():
        Wf, bf = Wo @ Wv, Wo @ bv
    Wf.requires_grad_(); bf.requires_grad_()
    zf = Pd @ (x @ Wf.t() + bf) + bo.detach()
    assert torch.allclose(zf, z, rtol=1e-12, atol=1e-12)
    zf.backward(dz)
    dWf, dbf = Wf.grad, bf.grad
    with torch.no_grad():
        dWo = dWf @ Wv.t() + torch.outer(dbf, bv)
        dWv, dbv = Wo.t() @ dWf, Wo.t() @ dbf
        dbo = dz.sum(0)
    for got, ref in zip((dWv, dbv, dWo, dbo), want):
        assert torch.allclose(got, ref, rtol=1e-10, atol=1e-10)
    # what the backward chain uses instead of out_proj's input gradient: dVW = Pd^T dz, dPd = dz VW^T
    VW = (x @ Wf.t() + bf).detach()
    V = (x @ Wv.t() + bv).detach()
    dctx = dz @ Wo.detach()
    assert torch.allclose(dz @ VW.t(), dctx @ V.t(), rtol=1e-10, atol=1e-10)            # dPd either way
    assert torch.allclose((Pd.t() @ dz) @ Wo.detach(), Pd.t() @ dctx, rtol=1e-10, atol=1e-10)      # dV = dVW Wo


def test_bias_rides_in_the_folded_bias_when_rows_sum_to_one():
    """Forward-only plans under the fused attention kernel (no dropout): softmax rows sum to one, so bo is carried by b' = Wo bv + bo."""
    g = torch.Generator().manual_seed(12)
    B, F = 19, 13
    rnd = lambda *s: torch.randn(*s, generator=g, dtype=torch.float64)
    x, Wv, bv, Wo, bo = rnd(B, F), rnd(F, F), rnd(F), rnd(F, F), rnd(F)
    P = torch.softmax(rnd(B, B), dim=1)
    z = (P @ (x @ Wv.t() + bv)) @ Wo.t() + bo
    zf = P @ (x @ (Wo @ Wv).t() + (Wo @ bv + bo))
    assert torch.allclose(zf, z, rtol=1e-12, atol=1e-12)
