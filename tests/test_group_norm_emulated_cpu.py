"""csrc/group_norm.hip -- GroupNorm (+ ReLU) of channels-last activations, 8 channels per group -- with its launcher and C-ABI
entries on the HIP-on-CPU shim, through the product's autograd wrapper (monodetr_amd/group_norm_ext.py): values and all
three gradients against torch.nn.functional.group_norm (+ relu) in float64, both I/O types, parameter types, pixel counts
that do and do not fill the row chunks, a large common offset (the moments must not cancel)."""
import pytest
import torch
import torch.nn.functional as F

import native_emul


@pytest.fixture()
def ext():
    from monodetr_amd import group_norm_ext
    group_norm_ext._backend = native_emul.lib()
    yield group_norm_ext
    group_norm_ext._backend = None


@pytest.mark.parametrize("shape,dtype,pdtype,relu,offset", [
    ((2, 256, 24, 80), torch.bfloat16, torch.bfloat16, True, 0.0),     # the depth head's stage, 1 920 pixels: 30 chunks of 64 rows
    ((2, 256, 6, 20), torch.bfloat16, torch.float32, False, 0.0),      # input projection of the coarsest level: 2 chunks, the last ragged
    ((1, 64, 7, 9), torch.float32, torch.float32, True, 0.0),          # 8 vectors per row, 32 row lanes, fewer rows than one chunk
    ((3, 128, 50, 70), torch.float32, torch.float32, False, 300.0),    # 3 500 pixels: chunk rows above the minimum; mean >> std
    ((1, 2048, 3, 5), torch.bfloat16, torch.bfloat16, True, 0.0),      # 256 vectors per row: one row lane
])
def test_group_norm_kernel_matches_the_framework_operator(ext, shape, dtype, pdtype, relu, offset):
    g = torch.Generator().manual_seed(sum(shape))
    N, C, H, W = shape
    G = C // 8
    x = (torch.randn(shape, generator=g) * 1.7 + offset).to(dtype).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    w = (torch.randn(C, generator=g) * 0.5 + 1.0).to(pdtype).requires_grad_(True)
    b = (torch.randn(C, generator=g) * 0.5).to(pdtype).requires_grad_(True)
    dy = torch.randn(shape, generator=g).to(dtype).contiguous(memory_format=torch.channels_last)
    assert ext.supported(x, w, b, G)
    y = ext.group_norm(x, w, b, G, 1e-5, relu)
    assert y.dtype == dtype and y.shape == x.shape and y.is_contiguous(memory_format=torch.channels_last)
    y.backward(dy)
    got = (y.detach().double(), x.grad.double(), w.grad.double(), b.grad.double())
    xr, wr, br = (t.detach().double().requires_grad_(True) for t in (x, w, b))
    ref = F.group_norm(xr, G, wr, br, 1e-5)
    if relu:
        ref = F.relu(ref)
    ref.backward(dy.double())
    want = (ref.detach(), xr.grad, wr.grad, br.grad)
    # one bf16 rounding of outputs of magnitude ~|w| * 4 + |b|; fp32 arithmetic otherwise (the offset case loses log2(300 / 1.7) bits of x)
    out_tol = 2 ** -8 if dtype == torch.bfloat16 else (3e-4 if offset else 2e-5)
    for name, a, r in zip(("y", "dx", "dw", "db"), got, want):
        tol = out_tol * (4 if (name in ("dw", "db") and pdtype == torch.bfloat16) else 1)
        scale = max(1.0, r.abs().max().item())
        if relu and name != "y" and dtype == torch.bfloat16:
            # a pre-activation within rounding of 0 may fall on the other side of the mask: compare away from those elements
            pre = F.group_norm(x.detach().double(), G, w.detach().double(), b.detach().double(), 1e-5)
            near = (pre.abs() < 1e-3)
            if name == "dx":
                assert near.float().mean() < 0.01
                a, r = a.masked_fill(near, 0.0), r.masked_fill(near, 0.0)
                tol *= 4                                              # the group sums a, b still see the flipped elements
            else:
                tol *= 8
        assert (a - r).abs().max().item() <= tol * scale, (name, (a - r).abs().max().item(), scale)


def test_group_norm_module_keeps_the_reference_parameters_and_falls_back(ext, monkeypatch):
    """group_norm_ext.GroupNorm: nn.GroupNorm's parameter names; kernel when enabled and eligible, framework operators
    otherwise (NCHW-contiguous input, other group sizes), with the folded ReLU either way."""
    torch.manual_seed(0)
    m = ext.GroupNorm(32, 256, relu=True)
    assert sorted(k for k, _ in m.named_parameters()) == ["bias", "weight"]
    x = torch.randn(2, 256, 5, 6).contiguous(memory_format=torch.channels_last)
    monkeypatch.setattr(ext, "ENABLED", True)
    y1 = m(x)
    assert type(y1.grad_fn).__name__ == "_GroupNormBackward"
    y2 = m(x.contiguous())                                            # NCHW memory: not eligible
    assert type(y2.grad_fn).__name__ != "_GroupNormBackward"
    assert (y1 - y2).abs().max() < 1e-5 and (y1 >= 0).all()
    monkeypatch.setattr(ext, "ENABLED", False)
    assert type(m(x).grad_fn).__name__ != "_GroupNormBackward"
    assert not ext.supported(x, torch.ones(256), torch.zeros(256), 16)     # 16 channels per group


def test_group_norm_c_abi_rejects_what_the_kernel_cannot_do():
    from monodetr_amd import _capi
    L = native_emul.lib()
    assert L.mdetr_group_norm_workspace_bytes(2, 100, 256, 16) == 0 and L.mdetr_group_norm_workspace_bytes(2, 100, 256, 32) > 0
    x = torch.zeros(1, 4, 256)
    st = torch.zeros(1, 32, 2)
    ws = torch.zeros(16)
    rc = L.mdetr_group_norm_forward(0, 0, x.data_ptr(), x.data_ptr(), x.data_ptr(), x.data_ptr(), st.data_ptr(), ws.data_ptr(), ws.numel() * 4,
                                    1, 4, 256, 32, 1e-5, 0, -1, None)
    assert rc == -1                                                   # MDETR_E_ARG: workspace too small
    rc = L.mdetr_group_norm_forward(0, 2, x.data_ptr(), x.data_ptr(), x.data_ptr(), x.data_ptr(), st.data_ptr(), ws.data_ptr(), 1 << 30,
                                    1, 4, 256, 32, 1e-5, 0, -1, None)
    assert rc == -1                                                   # MDETR_E_ARG: bf16 parameters with an fp32 activation
