"""csrc/small_wgrad.hip -- dW = dY^T X and db = column sums of dY for a few thousand token rows -- with its launcher (chunk
partials + colsum) and C-ABI entry on the HIP-on-CPU shim, through monodetr_amd/small_wgrad_ext.py and through
token_linear's backward: against float64 products, both I/O types, ragged last chunks, strided operands, one chunk only."""
import pytest
import torch

import native_emul


@pytest.fixture()
def ext():
    from monodetr_amd import small_wgrad_ext
    small_wgrad_ext._backend = native_emul.lib()
    yield small_wgrad_ext
    small_wgrad_ext._backend = None


@pytest.mark.parametrize("T,N,K,dtype,out_dtype,strided", [
    (4400, 256, 256, torch.bfloat16, torch.bfloat16, False),     # the decoder's layers: 32 chunks of 160 rows, the last one short
    (550, 128, 256, torch.bfloat16, torch.float32, True),        # attention weights of one image; operands are column slices
    (37, 64, 64, torch.float32, torch.float32, False),           # a single chunk, fewer rows than one staging step + a ragged one
    (1000, 512, 64, torch.float32, torch.float32, False),        # packed q / k projection width
    (8192, 64, 128, torch.bfloat16, torch.bfloat16, False),      # the largest row count taken
    (4400, 6, 256, torch.float32, torch.float32, False),         # a prediction head's last layer: the guarded form, 24-byte dY rows
    (1100, 3, 64, torch.bfloat16, torch.bfloat16, False),        # 6-byte dY rows
    (700, 24, 128, torch.float32, torch.bfloat16, False),
    (900, 136, 64, torch.float32, torch.float32, False),         # (the packed first layers are 1 032 wide: whole tiles + a ragged one)
    (15360, 61, 64, torch.float32, torch.float32, False),        # the depth-embedding table's gradient: 61 rows over B H W positions
    (300, 64, 64, torch.bfloat16, torch.float32, "odd"),         # whole tiles, but dY rows that 16-byte loads cannot take
])
def test_small_wgrad_kernel_matches_float64_products(ext, T, N, K, dtype, out_dtype, strided):
    g = torch.Generator().manual_seed(T + N + K)
    if strided == "odd":
        big_y = torch.randn(T, N + 3, generator=g).to(dtype)
        dy, x = big_y[:, 3:], torch.randn(T, K, generator=g).to(dtype)
    elif strided:
        big_y, big_x = torch.randn(T, N + 64, generator=g).to(dtype), torch.randn(T, K + 128, generator=g).to(dtype)
        dy, x = big_y[:, 64:], big_x[:, :K]
    else:
        dy, x = torch.randn(T, N, generator=g).to(dtype), torch.randn(T, K, generator=g).to(dtype)
    assert ext.supported(dy, x)
    dw, db = ext.small_wgrad(dy, x, out_dtype)
    assert dw.shape == (N, K) and db.shape == (N,) and dw.dtype == db.dtype == out_dtype
    rw, rb = dy.double().t() @ x.double(), dy.double().sum(0)
    tol = 2 ** -8 if out_dtype == torch.bfloat16 else 1e-5      # one rounding of an fp32 sum
    assert (dw.double() - rw).abs().max() <= tol * rw.abs().max()
    assert (db.double() - rb).abs().max() <= tol * max(1.0, rb.abs().max().item())


def test_small_wgrad_rejects_other_shapes(ext):
    a = torch.zeros(100, 256)
    assert not ext.supported(a, torch.zeros(100, 100))                   # K not a multiple of 64
    assert not ext.supported(torch.zeros(9000, 128), torch.zeros(9000, 64))    # too many rows (the split-K library path)
    assert not ext.supported(torch.zeros(100, 4096), torch.zeros(100, 256))    # too wide: the chunk partials would outweigh the operands
    assert not ext.supported(torch.zeros(70000, 8), torch.zeros(70000, 64))    # too many rows even for the narrow form
    assert not ext.supported(a, torch.zeros(100, 256, dtype=torch.bfloat16))
    assert not ext.supported(a[:, :64], a[:, 1:65])                      # misaligned rows of x (dY may be anything: guarded loads)
    with pytest.raises(RuntimeError):
        ext.small_wgrad(a, torch.zeros(100, 100))
    assert native_emul.lib().mdetr_small_wgrad_workspace_bytes(100, 256, 100) == 0


def test_token_linear_backward_through_small_wgrad(ext, monkeypatch):
    """token_linear's autograd function with the kernel on == F.linear's gradients (rows below the big-T split-K path)."""
    import torch.nn.functional as F
    from monodetr_amd.monodetr import linear
    monkeypatch.setattr(ext, "ENABLED", True)
    g = torch.Generator().manual_seed(0)
    x = torch.randn(2, 2200, 64, generator=g, requires_grad=True)
    w = (torch.randn(128, 64, generator=g) * 0.1).requires_grad_(True)
    b = torch.zeros(128, requires_grad=True)
    dy = torch.randn(2, 2200, 128, generator=g)
    y = linear._TokenLinear.apply(x, w, b, False)
    y.backward(dy)
    got = (x.grad.clone(), w.grad.clone(), b.grad.clone())
    x.grad = w.grad = b.grad = None
    F.linear(x, w, b).backward(dy)
    for a, r in zip(got, (x.grad, w.grad, b.grad)):
        assert (a - r).abs().max() <= 1e-4 * max(1.0, r.abs().max().item())


def test_depth_embedding_table_gradient_through_small_wgrad(ext, monkeypatch):
    """depth_predictor._HatTimesTable: hat @ table with the table gradient from the kernel (hat as dY, dOut as X) == autograd's."""
    from monodetr_amd.monodetr.depth_predictor.depth_predictor import _HatTimesTable
    monkeypatch.setattr(ext, "ENABLED", True)
    g = torch.Generator().manual_seed(5)
    coord = torch.rand(2, 24, 80, generator=g) * 60
    hat = (1 - (coord.unsqueeze(-1) - torch.arange(61.0)).abs()).clamp(min=0).requires_grad_(True)
    table = torch.randn(61, 256, generator=g).requires_grad_(True)
    gout = torch.randn(2, 24, 80, 256, generator=g)
    want = torch.autograd.grad(hat @ table, [hat, table], gout)
    got = torch.autograd.grad(_HatTimesTable.apply(hat, table), [hat, table], gout)
    for a, b in zip(want, got):
        assert (a - b).abs().max() <= 1e-5 * max(1.0, a.abs().max().item())
