"""csrc/colsum.hip (bias gradient of the token-wise linear layers) vs torch's fp64 column sum."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("shape", [(81600, 256), (81600, 1024), (4400, 64), (257, 8), (1, 48), (5000, 2048 + 64)])
def test_column_sum_matches_fp64(dtype, shape):
    from monodetr_amd.colsum_ext import column_sum, supported
    torch.manual_seed(shape[0] + shape[1])
    x = (torch.randn(shape, device="cuda") + 0.25).to(dtype)
    assert supported(x)
    got = column_sum(x)
    ref = x.double().sum(0)
    assert got.dtype == torch.float32 and got.shape == (shape[1],)
    # fp32 accumulation of T terms of magnitude ~1: error ~ sqrt(T) * 6e-8 * |x|, far below this bound
    assert (got.double() - ref).abs().max() < 2e-5 * max(shape[0], 1) ** 0.5 + 1e-6 * ref.abs().max()
    assert torch.equal(got, column_sum(x))                                  # deterministic


def test_column_sum_strided_rows_and_token_linear_bias_grad():
    from monodetr_amd.colsum_ext import column_sum, supported
    from monodetr_amd.monodetr.linear import token_linear
    big = torch.randn(6000, 512, device="cuda")
    x = big[:, 128:384]                                                      # ld = 512, offset keeps 16-byte alignment
    assert supported(x) and not x.is_contiguous()
    assert (column_sum(x).double() - x.double().sum(0)).abs().max() < 1e-3
    assert not supported(big[:, 1:257])                                      # misaligned rows are refused, not mis-summed
    # through the layer: db of token_linear == db of F.linear
    inp = torch.randn(8192, 64, device="cuda", requires_grad=True)
    w = torch.randn(96, 64, device="cuda", requires_grad=True)
    b = torch.randn(96, device="cuda", requires_grad=True)
    g = torch.randn(8192, 96, device="cuda")
    ref = torch.autograd.grad(torch.nn.functional.linear(inp, w, b), (inp, w, b), g)
    got = torch.autograd.grad(token_linear(inp, w, b), (inp, w, b), g)
    for a, c in zip(ref, got):
        assert (a - c).abs().max() <= 2e-3 * a.abs().max()


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("shape", [(81600, 256), (64, 65536), (200, 48), (4400, 128)])
def test_column_sum_written_as_bf16_is_the_rounded_fp32_sum(dtype, shape):
    """mdetr_column_sum_to(..., MDETR_BF16): the gradient of a bf16 parameter without a cast launch -- exactly the fp32
    result rounded once, for the one-block case (rows <= 256: the split-K weight gradients) and the two-pass case."""
    from monodetr_amd.colsum_ext import column_sum
    torch.manual_seed(shape[0] * 3 + shape[1])
    x = (torch.randn(shape, device="cuda") + 0.1).to(dtype)
    f32 = column_sum(x)
    b16 = column_sum(x, torch.bfloat16)
    assert b16.dtype == torch.bfloat16 and torch.equal(b16, f32.to(torch.bfloat16))
