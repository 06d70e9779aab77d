"""csrc/colsum.hip on the HIP-on-CPU shim through the C ABI (mdetr_column_sum_to): tall matrices (many row blocks, two
launches), the single-row-block form with few rows and very many columns (the chunk sums of the split-K weight gradients:
row lanes traded for workgroups), strided rows, both input types and both output types."""
import pytest
import torch

import native_emul


@pytest.mark.parametrize("rows,cols,dtype,out_dtype,ld_extra", [
    (1000, 256, torch.bfloat16, torch.float32, 0),       # 4 row blocks: partial rows + final pass
    (64, 65536, torch.bfloat16, torch.bfloat16, 0),      # split-K chunk sum: one row block, 8 192 column vectors -> 16-vector workgroups
    (16, 65792, torch.float32, torch.bfloat16, 0),       # small_wgrad's partials (n k + n columns), fp32 in, bf16 out
    (3, 1024, torch.float32, torch.float32, 0),          # fewer rows than row lanes could use
    (300, 64, torch.float32, torch.float32, 64),         # a column slice of a wider matrix
    (257, 8, torch.bfloat16, torch.float32, 0),          # one column vector
])
def test_column_sum_matches_float64(rows, cols, dtype, out_dtype, ld_extra):
    L = native_emul.lib()
    g = torch.Generator().manual_seed(rows + cols)
    big = torch.randn(rows, cols + ld_extra, generator=g).to(dtype)
    x = big[:, :cols]
    need = L.mdetr_column_sum_workspace_bytes(rows, cols)
    ws = torch.empty(max(need, 16), dtype=torch.uint8)
    out = torch.full((cols,), 7.0).to(out_dtype)
    code = lambda dt: 2 if dt == torch.bfloat16 else 0      # noqa: E731  (MDETR_BF16 / MDETR_F32)
    rc = L.mdetr_column_sum_to(code(dtype), x.data_ptr(), out.data_ptr(), code(out_dtype), ws.data_ptr(), ws.numel(), rows, cols, x.stride(0), -1, None)
    assert rc == 0
    ref = x.double().sum(0)
    tol = 2 ** -8 if out_dtype == torch.bfloat16 else 1e-5
    assert (out.double() - ref).abs().max() <= tol * max(1.0, ref.abs().max().item())


def test_grouped_chunk_sums_match_the_single_sums_bit_for_bit():
    """mdetr_chunk_sums through monodetr_amd/chunk_sums.py: several partial matrices of different shapes in one launch -- more jobs
    than one argument block holds, ragged last column tiles, both result types -- equal to the fixed-order sum computed per matrix;
    registered inside ``deferred()`` the results arrive with the flush, and a flush in the middle is harmless."""
    from monodetr_amd import chunk_sums
    chunk_sums._backend = native_emul.lib()
    was, chunk_sums.ENABLED = chunk_sums.ENABLED, True
    try:
        g = torch.Generator().manual_seed(11)
        shapes = [(128, 65792), (33, 36864), (1, 1028), (7, 4)] + [(3 + i % 5, 256 + 4 * i) for i in range(50)]
        parts = [torch.randn(c, n, generator=g) for c, n in shapes]
        dts = [torch.bfloat16 if i % 2 else torch.float32 for i in range(len(parts))]

        def ordered(p, dt):
            s = torch.zeros(p.shape[1])
            for k in range(p.shape[0]):
                s = s + p[k]
            return s.to(dt)
        with chunk_sums.deferred():
            outs = [chunk_sums.chunk_sum(p, dt) for p, dt in zip(parts[:30], dts[:30])]
            chunk_sums.flush()
            assert all(torch.equal(o, ordered(p, dt)) for o, p, dt in zip(outs, parts[:30], dts[:30]))
            outs += [chunk_sums.chunk_sum(p, dt) for p, dt in zip(parts[30:], dts[30:])]
        assert len(chunk_sums._pending) == 0
        for o, p, dt in zip(outs, parts, dts):
            assert o.dtype == dt and torch.equal(o, ordered(p, dt))
        now = chunk_sums.chunk_sum(parts[0], torch.float32)                   # outside the context: computed at once
        assert torch.equal(now, ordered(parts[0], torch.float32))
        with pytest.raises(RuntimeError):
            chunk_sums.chunk_sum(torch.randn(4, 6), torch.float32)            # 6 columns: not a multiple of 4
    finally:
        chunk_sums.ENABLED = was
        chunk_sums._backend = None
