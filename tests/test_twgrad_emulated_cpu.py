"""csrc/twgrad.hip -- the REAL kernel source, launcher and C-ABI entry (mdetr_token_wgrad) -- on the HIP-on-CPU shim: transposing-read
fragments (the shim restates ds_read_b64_tr_b16 as measured on the chip), the four tile shapes, ragged T / N / C, several chunks
per tile, the bias gradient riding on the dY fragments; against fp64 products of the same bf16 operands."""
import ctypes

import pytest
import torch

import native_emul
from conftest import tune


def run(x, dy, with_bias, form="1"):
    L = native_emul.lib()
    T, C = x.shape
    N = dy.shape[1]
    chunks = L.mdetr_token_wgrad_chunks(T, C, N)
    assert chunks >= 1
    cols = N * C + (N if with_bias else 0)
    part = torch.full((chunks, cols), float("nan"))
    rc = L.mdetr_token_wgrad(x.data_ptr(), dy.data_ptr(), part.data_ptr(), part.numel(), T, C, N, 1 if with_bias else 0, -1, None)
    assert rc == 0, ctypes.string_at(L.mdetr_last_error())
    assert not torch.isnan(part).any()                                 # every chunk wrote every element
    tot = part.double().sum(0)
    return tot[:N * C].view(N, C), (tot[N * C:] if with_bias else None), chunks


@pytest.mark.parametrize("T,C,N", [
    (136, 64, 32),          # 64 x 64 tiles, 5 slabs (ragged last), one chunk
    (1000, 128, 160),       # 128 x 128 tiles, N = 160: a ragged second row tile
    (264, 64, 256),         # 128 (n) x 64 (c)
    (300, 264, 72),         # C = 264: three column tiles, the last with 8 live columns; N = 72
    (4101, 256, 256),       # many slabs: several chunks per tile; T % 8 != 0
    (33, 8, 8),             # two slabs, everything ragged
])
def test_twgrad_source_on_the_cpu_shim(monkeypatch, T, C, N):
    tune(monkeypatch, twgrad=None)
    g = torch.Generator().manual_seed(T + C + N)
    x = (torch.randn(T, C, generator=g) * 0.5).to(torch.bfloat16)
    dy = (torch.randn(T, N, generator=g) * 0.2).to(torch.bfloat16)
    dw, db, chunks = run(x, dy, True)
    rw, rb = dy.double().t() @ x.double(), dy.double().sum(0)
    mag = dy.double().abs().t() @ x.double().abs()
    assert bool(((dw - rw).abs() <= 4 * (T ** 0.5) * 2.0 ** -23 * mag + 1e-30).all())      # fp32 accumulation of exact products
    assert bool(((db - rb).abs() <= 4 * (T ** 0.5) * 2.0 ** -23 * dy.double().abs().sum(0) + 1e-30).all())
    dw2, none, _ = run(x, dy, False)
    assert none is None and torch.equal(dw2, dw)


def test_twgrad_chunking_is_a_partition(monkeypatch):
    """More workgroups asked for than slabs allow: every chunk keeps at least one slab, the sum over chunks is the whole product."""
    tune(monkeypatch, twgrad_wgs="8192")
    g = torch.Generator().manual_seed(1)
    x = (torch.randn(700, 64, generator=g)).to(torch.bfloat16)
    dy = (torch.randn(700, 64, generator=g)).to(torch.bfloat16)
    dw, db, chunks = run(x, dy, True)
    assert chunks == 5                                                 # 22 slabs, at least four per chunk
    assert (dw - dy.double().t() @ x.double()).abs().max() <= 1e-3
    tune(monkeypatch, twgrad="0")                            # the 1x1 case of csrc/conv_wgrad.hip answers the same entry point
    x8, dy8 = x[:696].contiguous(), dy[:696].contiguous()
    dw0, db0, _ = run(x8, dy8, True)
    assert (dw0 - dy8.double().t() @ x8.double()).abs().max() <= 1e-3


@pytest.mark.parametrize("T,C,N", [
    (4101, 256, 256),       # four tiles; an odd number of slabs in the last chunk: one group's last slab is beyond it
    (1000, 160, 136),       # ragged tiles in both directions
    (40, 128, 128),         # two slabs in all: one per group
])
def test_two_wave_groups_per_workgroup_give_the_same_sums(monkeypatch, T, C, N):
    """twgrad_kg=2: eight waves, the even / odd slabs of a chunk on two groups with their own LDS buffers, accumulators (and the bias
    sums) added through LDS -- half the partials; against the one-group kernel on the same operands."""
    g = torch.Generator().manual_seed(T * 3 + C + N)
    x = (torch.randn(T, C, generator=g) * 0.5).to(torch.bfloat16)
    dy = (torch.randn(T, N, generator=g) * 0.2).to(torch.bfloat16)
    tune(monkeypatch, twgrad_kg="1", twgrad_wgs="64")
    dw1, db1, chunks1 = run(x, dy, True)
    tune(monkeypatch, twgrad_kg="2", twgrad_wgs="64")
    dw2, db2, chunks2 = run(x, dy, True)
    assert chunks2 <= chunks1
    rw, rb = dy.double().t() @ x.double(), dy.double().sum(0)
    mag = dy.double().abs().t() @ x.double().abs()
    for dw, db in ((dw1, db1), (dw2, db2)):
        assert bool(((dw - rw).abs() <= 4 * (T ** 0.5) * 2.0 ** -23 * mag + 1e-30).all())
        assert bool(((db - rb).abs() <= 4 * (T ** 0.5) * 2.0 ** -23 * dy.double().abs().sum(0) + 1e-30).all())
