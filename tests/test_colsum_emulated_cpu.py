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
