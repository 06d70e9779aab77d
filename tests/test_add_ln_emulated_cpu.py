"""csrc/add_ln.hip -- y = LayerNorm(a + dropout(b)) -- with its launcher and C-ABI entries on the HIP-on-CPU shim,
through the product's autograd wrapper (monodetr_amd/add_ln_ext.py): values and all four gradients against
nn.LayerNorm / F.dropout semantics, for every width and both I/O types; the dropout mask is checked against the hash
of csrc/add_ln_math.h evaluated independently in numpy."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

import native_emul


@pytest.fixture()
def ext():
    from monodetr_amd import add_ln_ext
    add_ln_ext._backend = native_emul.lib()
    yield add_ln_ext
    add_ln_ext._backend = None


def keep_mask(seed, n, p):
    """add_ln_math.h: ln_hash / ln_threshold, vectorised."""
    M32, M64 = np.uint64(0xFFFFFFFF), np.uint64(0xFFFFFFFFFFFFFFFF)
    with np.errstate(over='ignore'):
        z = (np.arange(n, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15) + np.uint64(seed)) & M64
        x = ((z & M32) ^ (z >> np.uint64(32))).astype(np.uint64)
        x ^= x >> np.uint64(16)
        x = (x * np.uint64(0x7FEB352D)) & M32
        x ^= x >> np.uint64(15)
        x = (x * np.uint64(0x846CA68B)) & M32
        x ^= x >> np.uint64(16)
    t = p * 4294967296.0
    thresh = 0xFFFFFFFF if t >= 4294967295.0 else int(t)
    return torch.from_numpy((x >= np.uint64(thresh)).astype(np.float32))


@pytest.mark.parametrize("rows,C,dtype,p", [(37, 256, torch.float32, 0.0), (130, 256, torch.float32, 0.1), (5000, 256, torch.bfloat16, 0.1),
                                            (9, 128, torch.float32, 0.3), (6, 512, torch.bfloat16, 0.0), (1, 512, torch.float32, 0.5)])
def test_fused_add_layernorm_matches_the_framework_operators(ext, rows, C, dtype, p):
    g = torch.Generator().manual_seed(rows + C)
    a = torch.randn(3, rows, C, generator=g)[1].to(dtype).requires_grad_(True)
    b = (torch.randn(rows, C, generator=g) * 0.7 + 0.2).to(dtype).requires_grad_(True)
    gamma = (1 + 0.1 * torch.randn(C, generator=g)).requires_grad_(True)
    beta = (0.1 * torch.randn(C, generator=g)).requires_grad_(True)
    dy = torch.randn(rows, C, generator=g).to(dtype)
    seed = 123456789
    y = ext.fused_add_layernorm(a, b, gamma, beta, 1e-5, p, seed=seed)
    assert y.dtype == dtype and y.shape == (rows, C)
    y.backward(dy)
    got = (y.detach(), a.grad.clone(), b.grad.clone(), gamma.grad.clone(), beta.grad.clone())
    for t in (a, b, gamma, beta):
        t.grad = None
    keep = keep_mask(seed, rows * C, p).view(rows, C) if p > 0 else torch.ones(rows, C)
    if p > 0:
        assert abs(keep.mean().item() - (1 - p)) < 4 * (p * (1 - p) / (rows * C)) ** 0.5 + 1e-3
    s = (a.float() + b.float() * keep / (1 - p)).to(dtype)               # the framework's rounding points
    ref = F.layer_norm(s.float(), (C,), gamma, beta, 1e-5).to(dtype)
    ref.backward(dy)
    want = (ref.detach(), a.grad, b.grad, gamma.grad, beta.grad)
    tol = 2e-2 if dtype == torch.bfloat16 else 2e-5
    for name, x, w in zip(("y", "da", "db", "dgamma", "dbeta"), got, want):
        err = (x.float() - w.float()).abs().max().item()
        assert err <= tol * max(1.0, w.float().abs().max().item()), (name, err)
    assert torch.equal(got[2] == 0, (keep == 0) | (want[2] == 0))        # dropped elements get exactly zero gradient


def test_residual_layernorm_helper_and_module_integration(ext, monkeypatch):
    """The helper the model calls: off -> the three framework operators; on -> the kernel (eval mode: no dropout;
    train mode: a fresh device-resident seed per call)."""
    torch.manual_seed(0)
    norm, drop = torch.nn.LayerNorm(256), torch.nn.Dropout(0.1)
    a, b = torch.randn(2, 50, 256), torch.randn(2, 50, 256)
    monkeypatch.setattr(ext, "ENABLED", False)
    drop.eval()
    ref = ext.residual_layernorm(a, b, norm, drop)
    assert torch.equal(ref, norm(a + b))
    monkeypatch.setattr(ext, "ENABLED", True)
    assert (ext.residual_layernorm(a, b, norm, drop) - ref).abs().max() < 2e-5
    drop.train()
    y1, y2 = ext.residual_layernorm(a, b, norm, drop), ext.residual_layernorm(a, b, norm, drop)
    assert not torch.equal(y1, y2) and (y1 - ref).abs().max() > 1e-3     # masks differ between calls
    odd = torch.randn(2, 50, 96)
    assert torch.equal(ext.residual_layernorm(odd, odd, torch.nn.LayerNorm(96), None), torch.nn.LayerNorm(96)(odd + odd))   # unsupported width: fallback


def test_bf16_affine_parameters_are_read_without_casts(ext):
    """A bf16 model body keeps LayerNorm's weight and bias in bf16: the kernel reads them as they are and the gradients
    come back in bf16."""
    g = torch.Generator().manual_seed(5)
    rows, C = 300, 256
    a = torch.randn(rows, C, generator=g).to(torch.bfloat16).requires_grad_(True)
    b = torch.randn(rows, C, generator=g).to(torch.bfloat16).requires_grad_(True)
    gamma = (1 + 0.1 * torch.randn(C, generator=g)).to(torch.bfloat16).requires_grad_(True)
    beta = (0.1 * torch.randn(C, generator=g)).to(torch.bfloat16).requires_grad_(True)
    dy = torch.randn(rows, C, generator=g).to(torch.bfloat16)
    y = ext.fused_add_layernorm(a, b, gamma, beta, 1e-5, 0.0)
    y.backward(dy)
    assert gamma.grad.dtype == beta.grad.dtype == torch.bfloat16
    a2, b2, g2, be2 = (t.detach().float().requires_grad_(True) for t in (a, b, gamma, beta))
    ref = F.layer_norm((a2 + b2).to(torch.bfloat16).float(), (C,), g2, be2, 1e-5)
    ref.backward(dy.float())
    for got, want in ((y, ref), (a.grad, a2.grad), (gamma.grad, g2.grad), (beta.grad, be2.grad)):
        assert (got.float() - want.detach().float()).abs().max() <= 3e-2 * max(1.0, want.abs().max().item())
