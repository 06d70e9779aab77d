"""Host-side pieces added around the kernels: `_LevelPos` (level position embedding with a hand-written backward) and
`linear.Linear` (nn.Linear routed through token_linear) against the plain framework expressions on the CPU."""
import pytest
import torch
import torch.nn.functional as F

from monodetr_amd.monodetr.depthaware_transformer import _LevelPos, fused_first_layers, mlp_rest, MLP
from monodetr_amd.monodetr.linear import Linear


def test_level_pos_matches_the_broadcast_adds():
    g = torch.Generator().manual_seed(0)
    shapes = [(6, 10), (3, 5), (2, 3), (1, 2)]
    pos = [torch.randn(2, 16, h, w, generator=g) for h, w in shapes]
    emb = torch.randn(4, 16, generator=g, requires_grad=True)
    out = _LevelPos.apply(emb, torch.float32, *pos)
    ref_emb = emb.detach().clone().requires_grad_(True)
    ref = torch.cat([p.flatten(2).transpose(1, 2) + ref_emb[l].view(1, 1, -1) for l, p in enumerate(pos)], 1)
    assert torch.equal(out, ref)
    dy = torch.randn(ref.shape, generator=g)
    out.backward(dy)
    ref.backward(dy)
    assert (emb.grad - ref_emb.grad).abs().max() <= 1e-5 * ref_emb.grad.abs().max()
    # bf16 output, fp32 parameter: the gradient comes back in the parameter's dtype
    emb2 = emb.detach().clone().requires_grad_(True)
    _LevelPos.apply(emb2, torch.bfloat16, *pos).backward(dy.to(torch.bfloat16))
    assert emb2.grad.dtype == torch.float32 and (emb2.grad - ref_emb.grad).abs().max() <= 2e-2 * ref_emb.grad.abs().max()


def test_level_pos_back_propagates_into_a_learned_position_embedding():
    """position_embedding: learned (position_encoding.py:58-84): row_embed / col_embed receive their gradient through
    ``pos_l + level_embed[l]`` as in the reference (depthaware_transformer.py:215-218); float64 stays float64."""
    from monodetr_amd.monodetr.position_encoding import PositionEmbeddingLearned
    from monodetr_amd.utils.misc import NestedTensor
    torch.manual_seed(4)
    shapes = [(6, 10), (3, 5)]

    def run(custom):
        torch.manual_seed(5)
        pe = PositionEmbeddingLearned(8).double()
        emb = torch.randn(2, 16, dtype=torch.float64, requires_grad=True, generator=torch.Generator().manual_seed(6))
        pos = [pe(NestedTensor(torch.zeros(2, 3, h, w, dtype=torch.float64), torch.zeros(2, h, w, dtype=torch.bool))) for h, w in shapes]
        assert all(p.requires_grad for p in pos)
        if custom:
            out = _LevelPos.apply(emb, torch.float64, *pos)
        else:
            out = torch.cat([p.flatten(2).transpose(1, 2) + emb[l].view(1, 1, -1) for l, p in enumerate(pos)], 1)
        dy = torch.randn(out.shape, dtype=torch.float64, generator=torch.Generator().manual_seed(7))
        out.backward(dy)
        return out.detach(), emb.grad, pe.row_embed.weight.grad, pe.col_embed.weight.grad

    got, want = run(True), run(False)
    assert got[2] is not None and got[3] is not None
    for a, b in zip(got, want):
        assert a.dtype == torch.float64 and (a - b).abs().max() <= 1e-13 * max(1.0, b.abs().max().item())


def test_linear_keeps_nn_linear_semantics_and_keys():
    torch.manual_seed(1)
    a, b = Linear(8, 5), torch.nn.Linear(8, 5)
    b.load_state_dict(a.state_dict())                                  # same parameter names
    x = torch.randn(3, 7, 8, requires_grad=True)
    ya, yb = a(x), b(x)
    assert torch.equal(ya, yb)
    ya.sum().backward()
    assert a.weight.grad is not None and isinstance(a, torch.nn.Linear)


def test_fused_first_layers_equal_the_separate_heads():
    torch.manual_seed(2)
    heads = [MLP(16, 16, 6, 3), MLP(16, 16, 3, 2), MLP(16, 16, 2, 2), torch.nn.Linear(16, 3)]
    x = torch.randn(2, 5, 16)
    parts = fused_first_layers(x, heads)
    assert len(parts) == 4 and [p.shape[-1] for p in parts] == [16, 16, 16, 3]
    for h, part in zip(heads, parts):
        want = h(x)
        got = mlp_rest(h, part) if isinstance(h, MLP) else part
        assert (got - want).abs().max() < 1e-5
    # a single-layer MLP: its first layer is its output
    one = MLP(16, 16, 4, 1)
    assert (mlp_rest(one, fused_first_layers(x, [one])[0]) - one(x)).abs().max() < 1e-5


def test_split_rows_gradient_is_the_concatenation_and_mha_matches_torch():
    """linear.split_rows == Tensor.split with one gradient assembly (absent block -> zeros); attention.MultiheadAttention through it
    reproduces nn.MultiheadAttention's outputs and packed-parameter gradients in its three projection forms."""
    from monodetr_amd.monodetr.attention import MultiheadAttention
    from monodetr_amd.monodetr.linear import split_rows
    torch.manual_seed(3)
    w = torch.randn(12, 5, requires_grad=True)
    a, b, c = split_rows(w, 4, 4, 4)
    (a.sum() * 2 + c.square().sum()).backward()                                   # b unused
    want = torch.cat((torch.full((4, 5), 2.0), torch.zeros(4, 5), 2 * w.detach()[8:]))
    assert torch.equal(w.grad, want)

    mine, ref = MultiheadAttention(64, 2), torch.nn.MultiheadAttention(64, 2, batch_first=True)
    ref.load_state_dict(mine.state_dict())
    x, y, z = (torch.randn(2, 7, 64) for _ in range(3))
    for q, k, v in ((x, x, x), (x, x, y), (x, y, y), (x, y, z)):
        mine.zero_grad(set_to_none=True); ref.zero_grad(set_to_none=True)
        o1 = mine.forward_batch_first(q, k, v)
        o2 = ref(q, k, v, need_weights=False)[0]
        assert (o1 - o2).abs().max() < 1e-5
        o1.square().sum().backward(); o2.square().sum().backward()
        for n, p in mine.named_parameters():
            g2 = dict(ref.named_parameters())[n].grad
            assert (p.grad - g2).abs().max() <= 1e-4 * max(1.0, g2.abs().max().item()), n


def test_token_linear_skip_chain_matches_plain_autograd():
    """linear._TokenLinearSkip: (y, x') = ((x + pos) W^T + b, x); the gradient arriving over x' and the layer's own input gradient
    are summed inside the input-gradient product.  A chain of two (the encoder layer's value / query projections) followed by a
    residual use reproduces plain autograd's gradients for x, both weights and both biases."""
    from monodetr_amd.monodetr.linear import _TokenLinearSkip
    torch.manual_seed(5)
    x = torch.randn(2, 9, 16, requires_grad=True)
    pos = torch.randn(2, 9, 16)
    w1, b1 = torch.randn(16, 16, requires_grad=True), torch.randn(16, requires_grad=True)
    w2, b2 = torch.randn(24, 16, requires_grad=True), torch.randn(24, requires_grad=True)

    def plain():
        v = torch.nn.functional.linear(x, w1, b1)
        p = torch.nn.functional.linear(x + pos, w2, b2)
        return (v.sin().sum() + p.cos().sum() + (x * x).sum())

    def chained():
        v, x1 = _TokenLinearSkip.apply(x, w1, b1, None)
        p, x2 = _TokenLinearSkip.apply(x1, w2, b2, pos)
        return (v.sin().sum() + p.cos().sum() + (x2 * x2).sum())

    want = torch.autograd.grad(plain(), [x, w1, b1, w2, b2])
    got = torch.autograd.grad(chained(), [x, w1, b1, w2, b2])
    for a, b in zip(want, got):
        assert (a - b).abs().max() <= 1e-5 * max(1.0, a.abs().max().item())
    # the residual branch unused: the skip output receives no gradient
    v, x1 = _TokenLinearSkip.apply(x, w1, b1, None)
    g = torch.autograd.grad(v.sum(), [x, w1])
    assert torch.allclose(g[0], w1.sum(0).expand_as(x), atol=1e-5)


@pytest.mark.parametrize("shared", [False, True])
def test_token_linear_skip_never_writes_into_the_arriving_gradient(shared):
    """The residual-path gradient is summed inside the input-gradient product into a NEW tensor: whoever else holds the arriving
    gradient (a second consumer of the same tensor, retain_grad(), a hook that kept it) sees it unchanged.  (Round 4 wrote into it
    when a reference-count census said nobody else could reach it; round 5 replaced the census by not writing.)"""
    from monodetr_amd.monodetr.linear import _TokenLinearSkip
    torch.manual_seed(0)
    x = torch.randn(6, 5, 8, requires_grad=True)
    w, b = torch.randn(4, 8, requires_grad=True), torch.randn(4, requires_grad=True)
    other = torch.randn(6, 5, 8, requires_grad=True)
    kept = []
    m = other * 1.0
    y, xs = _TokenLinearSkip.apply(x, w, b, None)
    xs.register_hook(lambda g: kept.append((g, g.clone())))
    z = (xs + m) if shared else (xs * 2.0 + m)
    ((y * y).sum() + (z * 3.0).sum()).backward()
    assert torch.equal(kept[0][0], kept[0][1])                         # the arriving gradient is as it arrived
    assert x.grad.data_ptr() != kept[0][0].data_ptr()
    x2, o2 = x.detach().clone().requires_grad_(True), other.detach().clone().requires_grad_(True)
    w2, b2 = w.detach().clone().requires_grad_(True), b.detach().clone().requires_grad_(True)
    y2 = torch.nn.functional.linear(x2, w2, b2)
    z2 = (x2 + o2) if shared else (x2 * 2.0 + o2)
    ((y2 * y2).sum() + (z2 * 3.0).sum()).backward()
    for got, want in ((x.grad, x2.grad), (other.grad, o2.grad), (w.grad, w2.grad), (b.grad, b2.grad)):
        assert (got - want).abs().max() <= 1e-5 * max(1.0, want.abs().max().item())
    # retain_grad(): x' keeps the gradient it received, x gets the sum
    x3 = torch.randn(6, 5, 8, requires_grad=True)
    y3, xs3 = _TokenLinearSkip.apply(x3, w, b, None)
    xs3.retain_grad()
    ((y3 * y3).sum() + (xs3.view(30, 8) * 3.0).sum()).backward()
    assert torch.equal(xs3.grad, torch.full_like(xs3, 3.0)) and not torch.equal(x3.grad, xs3.grad)
