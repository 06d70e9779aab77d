"""csrc/tgemm.hip -- the REAL kernel source, launcher and C-ABI entry -- on the HIP-on-CPU shim (tests/native_emul.py): slab
pipeline (one and two register sets), the four tile shapes, ragged T / N / K, strided operands, the NN form's in-register
transposition, every part of the epilogue (bias bf16 / fp32, residual, in-place accumulation, ReLU, dropout, fp32 output),
held element by element to the fp64 product of the same bf16 operands (tests/gemm_bounds.py)."""
import ctypes

import pytest
import torch

import native_emul
from gemm_bounds import assert_product_close
from conftest import tune


def run(a, w, bias=None, res=None, relu=False, nn=False, out=None, out_dtype=torch.bfloat16, p=0.0, seed=0):
    from monodetr_amd import tgemm_ext
    old = tgemm_ext._backend
    tgemm_ext._backend = native_emul.lib()
    try:
        assert tgemm_ext.supported(a, w, nn=nn, res=res, bias=bias, out=out)
        return tgemm_ext.tgemm(a, w, bias, res, relu=relu, nn=nn, out=out, out_dtype=out_dtype, dropout_p=p, seed=seed)
    finally:
        tgemm_ext._backend = old


def problem(T, K, N, nn, seed):
    g = torch.Generator().manual_seed(seed)
    a = (torch.randn(T, K, generator=g) * 0.5).to(torch.bfloat16)
    w = (torch.randn(K, N, generator=g) * 0.1).to(torch.bfloat16) if nn else (torch.randn(N, K, generator=g) * 0.1).to(torch.bfloat16)
    b = torch.randn(N, generator=g).to(torch.bfloat16)
    r = torch.randn(T, N, generator=g).to(torch.bfloat16)
    return a, w, b, r


def reference(a, w, nn, bias=None, res=None, relu=False):
    wd = w.double() if nn else w.double().t()
    ref = a.double() @ wd
    mag = a.double().abs() @ wd.abs()
    if bias is not None:
        ref = ref + bias.double()
        mag = mag + bias.double().abs()
    if res is not None:
        ref = ref + res.double()
        mag = mag + res.double().abs()
    if relu:
        ref = ref.clamp(min=0)
    return ref, mag


SHAPES = [
    # T, K, N
    (300, 256, 256),        # ragged last token tile
    (64, 64, 64),           # one slab
    (1, 8, 8),              # one token, one piece, K below a slab
    (97, 128, 264),         # N = 264: a ragged feature tile with 8 live features
    (130, 1032, 72),        # K = 1032: 17 slabs, the last with one live piece
    (260, 192, 136),        # three slabs (odd count: the two-set ring's tail)
    (2100, 320, 128),
]


@pytest.mark.parametrize("tile", ["128x128", "128x64", "64x128", "64x64"])
@pytest.mark.parametrize("pf", ["1", "2"])
@pytest.mark.parametrize("nn", [False, True])
def test_tgemm_plain_products_every_tile_and_pipeline(monkeypatch, tile, pf, nn):
    tune(monkeypatch, tgemm_tile=tile)
    tune(monkeypatch, tgemm_pf=pf)
    for T, K, N in SHAPES[:6]:
        a, w, _, _ = problem(T, K, N, nn, T + K + N)
        y = run(a, w, nn=nn)
        ref, mag = reference(a, w, nn)
        assert_product_close(y, ref, mag, K, "T=%d K=%d N=%d" % (T, K, N))


@pytest.mark.parametrize("nn", [False, True])
@pytest.mark.parametrize("T,K,N", SHAPES)
def test_tgemm_launcher_default_tiles_with_the_whole_tail(T, K, N, nn):
    a, w, b, r = problem(T, K, N, nn, 7 * T + K + N)
    y = run(a, w, bias=b, res=r, relu=True, nn=nn)
    ref, mag = reference(a, w, nn, b, r, True)
    assert_product_close(y, ref, mag, K, "bias + residual + relu")
    y32 = run(a, w, bias=b.float(), nn=nn, out_dtype=torch.float32)          # fp32 bias, fp32 output, no tail
    ref, mag = reference(a, w, nn, b)
    assert y32.dtype == torch.float32
    assert_product_close(y32, ref, mag, K, "fp32 output")


def test_tgemm_accumulates_into_its_output_and_respects_row_strides():
    T, K, N = 200, 128, 136
    a, w, b, r = problem(T, K, N, True, 5)
    big_a = torch.zeros(T, K + 24, dtype=torch.bfloat16)
    big_a[:, :K] = a
    big_w = torch.zeros(K, N + 8, dtype=torch.bfloat16)
    big_w[:, :N] = w
    out = torch.full((T, N + 16), 3.0, dtype=torch.bfloat16)
    out[:, :N] = r
    view = out[:, :N]
    y = run(big_a[:, :K], big_w[:, :N], res=view, nn=True, out=view)          # y += a w (the residual-path gradient's accumulation)
    assert y.data_ptr() == out.data_ptr()
    ref, mag = reference(a, w, True, None, r)
    assert_product_close(out[:, :N], ref, mag, K, "in-place accumulation")
    assert bool((out[:, N:] == 3.0).all())                                     # nothing written beyond column N


def test_tgemm_dropout_makes_the_decisions_of_bias_act():
    """relu + dropout in the epilogue == mdetr_bias_act_forward(relu, dropout) on the product's fp32 values: same hash, same
    element index, same scale -- so bias_act's backward (dy where y > 0, scaled) serves both."""
    from monodetr_amd import bias_act_ext
    T, K, N, p, seed = 150, 64, 72, 0.25, 1234
    a, w, b, _ = problem(T, K, N, False, 11)
    y = run(a, w, bias=b, relu=True, p=p, seed=seed)
    pre = run(a, w, bias=b, out_dtype=torch.float32)                           # fp32 pre-activations of the same products
    L = native_emul.lib()
    want = torch.empty(T, N, dtype=torch.float32)
    rc = L.mdetr_bias_act_forward(0, 0, pre.data_ptr(), None, None, want.data_ptr(), T, N, 1, p, seed, None, -1, None)
    assert rc == 0, ctypes.string_at(L.mdetr_last_error())
    assert torch.equal(y, want.to(torch.bfloat16))
    kept = (y != 0).float().mean().item()
    assert 0.2 < kept < 0.55                                                   # ~ half positive, three quarters of those kept


def test_tgemm_rejects_what_it_cannot_run():
    L = native_emul.lib()
    a = torch.zeros(16, 64, dtype=torch.bfloat16)
    w = torch.zeros(8, 64, dtype=torch.bfloat16)
    y = torch.zeros(16, 8, dtype=torch.bfloat16)
    args = lambda **kw: [kw.get("a", a.data_ptr()), w.data_ptr(), None, None, y.data_ptr(), kw.get("T", 16), kw.get("N", 8), kw.get("K", 64),
                         kw.get("lda", 64), 64, 0, 8, kw.get("flags", 0), kw.get("p", 0.0), 0, None, -1, None]
    assert L.mdetr_tgemm(*args()) == 0
    assert L.mdetr_tgemm(*args(T=0)) == 0
    for bad in (dict(K=60), dict(N=12), dict(lda=60), dict(flags=64), dict(p=0.5), dict(a=a.data_ptr() + 2), dict(T=-1)):
        assert L.mdetr_tgemm(*args(**bad)) < 0, bad
        assert b"mdetr_tgemm" in ctypes.string_at(L.mdetr_last_error())


@pytest.mark.parametrize("grid", ["8", "16"])
@pytest.mark.parametrize("pf", ["1", "2"])
@pytest.mark.parametrize("nn", [False, True])
def test_tgemm_persistent_workgroups_walk_several_tiles(monkeypatch, grid, pf, nn):
    """Few workgroups, many tiles each: the slab sequence runs across tile boundaries (the next tile's first slabs are fetched during
    this tile's last products and parked tail), dead row tiles of the rounded-up grid are skipped, column tiles alternate."""
    tune(monkeypatch, tgemm_grid=grid)
    tune(monkeypatch, tgemm_pf=pf)
    tune(monkeypatch, tgemm_tile="64x64")
    for T, K, N in ((700, 192, 136), (1500, 64, 72), (330, 328, 200)):        # 11 / 24 / 6 row tiles x 3 / 2 / 4 column tiles
        a, w, b, r = problem(T, K, N, nn, T + K + N + 1)
        y = run(a, w, bias=b, res=r, relu=True, nn=nn)
        ref, mag = reference(a, w, nn, b, r, True)
        assert_product_close(y, ref, mag, K, "grid=%s T=%d K=%d N=%d" % (grid, T, K, N))
    tune(monkeypatch, tgemm_tile="128x128")
    a, w, b, r = problem(2100, 320, 264, nn, 99)
    y = run(a, w, bias=b, nn=nn)
    ref, mag = reference(a, w, nn, b)
    assert_product_close(y, ref, mag, 320, "128x128 persistent")


def test_tgemm_eight_wave_form_of_the_big_tile(monkeypatch):
    """The plain forward product on the 128 x 128 tile runs with 512 threads (waves 2 x 4) and one register set; MDETR_TGEMM_WAVES=4
    restores four waves -- the same values bit for bit (same products in the same order per element)."""
    tune(monkeypatch, tgemm_tile="128x128")
    tune(monkeypatch, tgemm_grid="8")
    for T, K, N in ((700, 192, 264), (130, 1032, 72)):
        a, w, b, r = problem(T, K, N, False, T + K)
        tune(monkeypatch, tgemm_waves=None)
        y8 = run(a, w, bias=b, relu=True)
        ref, mag = reference(a, w, False, b, None, True)
        assert_product_close(y8, ref, mag, K, "8 waves T=%d" % T)
        tune(monkeypatch, tgemm_waves="4")
        assert torch.equal(run(a, w, bias=b, relu=True), y8)


@pytest.mark.parametrize("T,K,N", [(300, 256, 256), (97, 128, 264), (1, 8, 8), (2100, 64, 128), (130, 72, 1032)])
@pytest.mark.parametrize("with_res", [False, True])
def test_masked_input_gradient(T, K, N, with_res, monkeypatch):
    """mdetr_tgemm_masked: y = mask > 0 ? a w + res : 0 -- the ReLU backward of the layer's input where the input gradient leaves."""
    from monodetr_amd import tgemm_ext
    monkeypatch.setattr(tgemm_ext, "_backend", native_emul.lib())
    a, w, _, r = problem(T, K, N, True, T + K + N)
    g = torch.Generator().manual_seed(7)
    mask = torch.randn(T, N, generator=g).clamp(min=0).to(torch.bfloat16)          # a ReLU output: zeros and positives
    mask[0, 0] = float("nan") if T * N > 1 else mask[0, 0]                         # (NaN <= 0 is false: the gradient passes, as in threshold_backward)
    res = r if with_res else None
    assert tgemm_ext.masked_supported(a, w, mask, res)
    y = tgemm_ext.tgemm_masked(a, w, mask, res)
    ref, mag = reference(a, w, True, None, res)
    keep = ~(mask.double() <= 0)
    assert bool((y[~keep] == 0).all())
    assert_product_close(torch.where(keep, y.double(), torch.zeros_like(ref)).to(y.dtype), torch.where(keep, ref, torch.zeros_like(ref)), mag, K)
    for tile in ("64x64", "128x64", "64x128", "128x128"):
        tune(monkeypatch, tgemm_tile=tile)
        assert torch.equal(tgemm_ext.tgemm_masked(a, w, mask, res), y)
