"""csrc/sgemm.hip (grouped fp32 products on the f32-input matrix instruction) on the HIP-on-CPU shim, through the product's wrapper
(monodetr_amd/sgemm_ext.py) and the C ABI: tile lookup in a group, ragged edges in every dimension, contraction lengths that are
not multiples of the slab, operands that are column slices of wider tensors, bf16 operands, several terms per product, the whole
epilogue (bias, partial ReLU, res, mask, bf16 result) and the TN form's eight-way split with its column sums.  Against float64;
the bound is fp32 accumulation's (sgemm.hip computes the k-ordered fmaf chain, nothing narrower)."""
import pytest
import torch

import native_emul


@pytest.fixture()
def ext():
    from monodetr_amd import sgemm_ext
    sgemm_ext._backend = native_emul.lib()
    yield sgemm_ext
    sgemm_ext._backend = None


def _close(got, ref, absref, K):
    """|got - ref| <= 2 K 2^-24 sum|a||b| + one output rounding (bf16 results: 2^-8 of the value)."""
    bound = 2.0 * K * 2.0 ** -24 * absref + (2.0 ** -8 * ref.abs() if got.dtype == torch.bfloat16 else 2.0 ** -23 * ref.abs()) + 1e-30
    bad = (got.double() - ref).abs() > bound
    assert not bad.any(), (int(bad.sum()), float(((got.double() - ref).abs() / bound).max()))


def test_nt_group_of_ragged_products_with_every_tail(ext):
    torch.manual_seed(0)
    T = 150
    x = torch.randn(T, 256).to(torch.bfloat16)                        # the decoder output of the bf16 body
    wide = torch.randn(T, 300)                                        # operands as column slices of a wider tensor
    w1, b1 = torch.randn(70, 256), torch.randn(70)
    w2, b2 = torch.randn(6, 40), torch.randn(6)
    w3 = torch.randn(33, 37)                                          # contraction 37: not a multiple of four either
    res = torch.randn(T, 33).to(torch.bfloat16)
    mask = torch.randn(T, 33)
    o1 = torch.full((T, 80), 7.0)                                     # the result is a column slice too: columns 5 .. 74
    o2 = torch.empty(T, 6)
    o3 = torch.empty(T, 33, dtype=torch.bfloat16)
    ext.grouped(ext.NT, [
        ext.Problem([(x, w1)], o1[:, 5:75], bias=b1, relu_cols=64),
        ext.Problem([(wide[:, 100:140], w2)], o2, bias=b2),
        ext.Problem([(wide[:, 3:40], w3)], o3, res=res, mask=mask, relu_cols=True),
    ])
    xd = x.double()
    r1 = xd @ w1.double().t() + b1.double()
    r1[:, :64] = r1[:, :64].clamp(min=0)
    _close(o1[:, 5:75], r1, xd.abs() @ w1.double().abs().t() + b1.abs().double(), 256)
    assert (o1[:, :5] == 7.0).all() and (o1[:, 75:] == 7.0).all()     # nothing outside the slice is touched
    a2 = wide[:, 100:140].double()
    _close(o2, a2 @ w2.double().t() + b2.double(), a2.abs() @ w2.double().abs().t() + b2.abs().double(), 40)
    a3 = wide[:, 3:40].double()
    r3 = (a3 @ w3.double().t() + res.double()).clamp(min=0) * (mask > 0)
    _close(o3, r3, a3.abs() @ w3.double().abs().t() + res.double().abs(), 37)


def test_nn_input_gradient_with_terms_split_over_several_tensors(ext):
    """dX = [dH_a | dH_b | d_cls] [W_a; W_b; W_c] without either concatenation: three terms, contraction 256 + 96 + 3."""
    torch.manual_seed(1)
    T = 131
    dh = torch.randn(T, 352)
    dcls = torch.randn(T, 3)
    wa, wb, wc = torch.randn(256, 256), torch.randn(96, 256), torch.randn(3, 256)
    skip = torch.randn(T, 256).to(torch.bfloat16)
    saved = torch.randn(T, 256)
    out = torch.empty(T, 256, dtype=torch.bfloat16)
    plain = torch.empty(T, 256)
    ext.grouped(ext.NN, [
        ext.Problem([(dh[:, :256], wa), (dh[:, 256:], wb), (dcls, wc)], out, res=skip),
        ext.Problem([(dcls, wc)], plain, mask=saved),
    ])
    ref = dh[:, :256].double() @ wa.double() + dh[:, 256:].double() @ wb.double() + dcls.double() @ wc.double() + skip.double()
    absref = dh[:, :256].double().abs() @ wa.double().abs() + dh[:, 256:].double().abs() @ wb.double().abs() + dcls.double().abs() @ wc.double().abs() + skip.double().abs()
    _close(out, ref, absref, 355)
    _close(plain, (dcls.double() @ wc.double()) * (saved > 0), dcls.double().abs() @ wc.double().abs(), 3)


@pytest.mark.parametrize("T", [4400 // 8, 37, 1])
def test_tn_weight_gradients_and_column_sums(ext, T):
    torch.manual_seed(2 + T)
    dy = torch.randn(T, 300)
    x = torch.randn(T, 256).to(torch.bfloat16)
    h = torch.randn(T, 70)
    dw1, db1 = torch.empty(45, 256), torch.empty(45)
    dw2, db2 = torch.empty(6, 70), torch.empty(6)
    dw3 = torch.empty(33, 33)
    ext.grouped(ext.TN, [
        ext.Problem([(dy[:, 10:55], x)], dw1, colsum=db1),
        ext.Problem([(dy[:, 100:106], h)], dw2, colsum=db2),
        ext.Problem([(dy[:, 200:233], dy[:, 250:283])], dw3),
    ])
    for got, a, b in ((dw1, dy[:, 10:55], x), (dw2, dy[:, 100:106], h), (dw3, dy[:, 200:233], dy[:, 250:283])):
        _close(got, a.double().t() @ b.double(), a.double().abs().t() @ b.double().abs(), T)
    _close(db1, dy[:, 10:55].double().sum(0), dy[:, 10:55].double().abs().sum(0), T)
    _close(db2, dy[:, 100:106].double().sum(0), dy[:, 100:106].double().abs().sum(0), T)


def test_products_are_the_fp32_fma_chain_not_something_narrower(ext):
    """Operands with 24 significant bits whose products need all of them: a bf16-split or tf32-like path would lose 1e-3 here."""
    torch.manual_seed(3)
    a = (1.0 + torch.rand(64, 32) * 2.0 ** -12).float()              # every element differs from 1 only below bit 12
    b = torch.ones(64, 32)
    b[:, 1::2] = -1.0                                                 # row sums cancel the leading ones exactly
    out = torch.empty(64, 64)
    ext.grouped(ext.NT, [ext.Problem([(a, b)], out)])
    ref = a.double() @ b.double().t()
    assert (out.double() - ref).abs().max() <= 32 * 2.0 ** -24 * 2.0   # a narrower product would be off by ~2^-9 relative to |ref| ~ 2^-12
    assert ref.abs().max() > 1e-5


def test_argument_errors_are_reported_not_launched(ext):
    a, b, out = torch.randn(8, 16), torch.randn(4, 16), torch.empty(8, 4)
    with pytest.raises(ValueError):
        ext.grouped(ext.NN, [ext.Problem([(a, b)], out)])            # NN wants B [K, N]
    with pytest.raises(RuntimeError, match="TN: one term"):
        ext.grouped(ext.TN, [ext.Problem([(torch.randn(5, 8), torch.randn(5, 4))], out, bias=torch.zeros(4))])
    with pytest.raises(RuntimeError, match="colsum belongs to TN"):
        ext.grouped(ext.NT, [ext.Problem([(a, b)], out, colsum=torch.zeros(8))])
