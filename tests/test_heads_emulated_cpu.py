"""monodetr/heads.py on the HIP-on-CPU shim: a decoder level's five prediction heads through the grouped fp32 kernels
(csrc/sgemm.hip) against the same modules evaluated by the framework in float64 -- every output, every parameter gradient, the
input gradient with the next layer's gradient summed in (`x'`), for an fp32 and for a bf16 decoder output."""
import copy

import pytest
import torch
from torch import nn

import native_emul


@pytest.fixture()
def heads():
    from monodetr_amd import sgemm_ext
    from monodetr_amd.monodetr import heads as H
    sgemm_ext._backend = native_emul.lib()
    was, H.ENABLED = H.ENABLED, True
    yield H
    H.ENABLED = was
    sgemm_ext._backend = None


def _modules(seed, hidden=256):
    from monodetr_amd.monodetr.depthaware_transformer import MLP
    torch.manual_seed(seed)
    return (MLP(hidden, hidden, 6, 3), MLP(hidden, hidden, 3, 2), MLP(hidden, hidden, 2, 2), MLP(hidden, hidden, 24, 2), nn.Linear(hidden, 3))


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_heads_level_matches_the_modules_in_float64(heads, dtype):
    mods = _modules(5)
    ref = [copy.deepcopy(m).double() for m in mods]
    B, Q = 2, 75                                                      # 150 rows: ragged against every tile size
    x = torch.randn(B, Q, 256).to(dtype).requires_grad_(True)
    out = heads.heads_level(x, *mods)
    assert out is not None
    delta, size, depth, angle, logits, xs = out
    assert all(t.dtype == torch.float32 for t in (delta, size, depth, angle, logits)) and xs.dtype == dtype
    gs = [torch.randn_like(t) for t in (delta, size, depth, angle, logits)]
    skip_w = torch.randn(B, Q, 256)
    total = sum((t * g).sum() for t, g in zip((delta, size, depth, angle, logits), gs)) + (xs.float() * skip_w).sum()
    total.backward()
    xd = x.detach().double().requires_grad_(True)
    outs = [m(xd) for m in ref]
    (sum((t * g.double()).sum() for t, g in zip(outs, gs)) + (xd * skip_w.double()).sum()).backward()
    for got, want in zip((delta, size, depth, angle, logits), outs):
        assert (got.double() - want).abs().max() <= 2e-5 * max(1.0, float(want.abs().max()))
    gx_tol = 2.0 ** -7 if dtype == torch.bfloat16 else 2e-5           # bf16: the gradient itself is rounded to bf16 once
    assert (x.grad.double() - xd.grad).abs().max() <= gx_tol * float(xd.grad.abs().max())
    for m, r in zip(mods, ref):
        for (n, p), (_, q) in zip(m.named_parameters(), r.named_parameters()):
            assert p.grad is not None and (p.grad.double() - q.grad).abs().max() <= 2e-5 * max(1.0, float(q.grad.abs().max())), n


def test_heads_level_declines_what_it_was_not_built_for(heads):
    mods = _modules(6)
    x = torch.randn(1, 8, 256)
    assert heads.heads_level(x, *mods) is not None
    assert heads.heads_level(x, mods[0], mods[1], mods[2], mods[3], mods[1]) is None       # class head is not a Linear
    half = [copy.deepcopy(m).to(torch.bfloat16) for m in mods]
    assert heads.heads_level(x, *half) is None                                              # parameters must be fp32
    heads.ENABLED = False
    assert heads.heads_level(x, *mods) is None
