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


def test_chunk_sums_grouped_launch_equals_the_single_launches():
    """mdetr_chunk_sums vs mdetr_column_sum_to on the partials' shapes of the step (chunks x (N C + N) fp32): same order, same bits."""
    from monodetr_amd import chunk_sums
    from monodetr_amd.colsum_ext import column_sum
    torch.manual_seed(3)
    shapes = [(128, 65792), (64, 147456), (32, 1024 * 256 + 1024), (8, 2048 * 512), (1, 1028)] + [(16 + i, 256 * 64 + 64) for i in range(60)]
    parts = [torch.randn(c, n, device="cuda") for c, n in shapes]
    was, chunk_sums.ENABLED = chunk_sums.ENABLED, True
    try:
        with chunk_sums.deferred():
            outs = [chunk_sums.chunk_sum(p, torch.bfloat16 if i % 2 else torch.float32) for i, p in enumerate(parts)]
        for i, (o, p) in enumerate(zip(outs, parts)):
            ref = p[0].clone()
            for k in range(1, p.shape[0]):
                ref += p[k]
            assert torch.equal(o, ref.to(o.dtype)), i
            assert (o.double() - p.double().sum(0)).abs().max() <= (2.0 ** -7 if o.dtype == torch.bfloat16 else 1e-4) * p.double().sum(0).abs().max()
    finally:
        chunk_sums.ENABLED = was


def test_deferred_chunk_sums_give_the_same_gradients_as_immediate_ones():
    """One backward pass of the whole model (bf16 committed kernels, dropout off, the training shape, launched the way the product
    launches its eager iterations: on the side stream) with the chunk sums batched, against every sum computed on the spot by the
    same kernel.  In the batched run every registered result is filled with NaN until its flush (chunk_sums.POISON): a consumer
    inside the backward pass that read a registered sum too early -- AccumulateGrad cloning a gradient that two parameters share,
    a concatenation, a cast -- leaves NaN in that parameter.  (Not bit-for-bit: two runs of the bf16 step differ by a bf16 rounding
    here and there, test_trainer_gpu.py measures that spread.)"""
    import bench
    from model_init import disable_dropout_
    from monodetr_amd import chunk_sums
    dev = torch.device("cuda", 0)
    names = tuple(sorted(bench.COMMITTED_SWITCHES["bf16"]))
    assert "MDETR_CHUNK_SUMS" in names
    grads = {}
    try:
        for mode in ("immediate", "deferred"):
            chunk_sums.IMMEDIATE, chunk_sums.POISON = mode == "immediate", mode == "deferred"
            step = bench.TrainStep(dev, 8, "bf16", size=(384, 1280), switches=names, graph=True)
            disable_dropout_(step.raw_model)
            step.optimizer.step = lambda *a, **k: None                   # gradients only
            step._eager(step.inputs)
            torch.cuda.synchronize()
            grads[mode] = {n: p.grad.detach().float().clone() for n, p in step.raw_model.named_parameters() if p.grad is not None}
            del step
    finally:
        chunk_sums.IMMEDIATE = chunk_sums.POISON = False
        bench.apply_switches(set())
    a, b = grads["immediate"], grads["deferred"]
    assert set(a) == set(b) and len(a) > 300
    assert not [n for n in b if not torch.isfinite(b[n]).all()], [n for n in b if not torch.isfinite(b[n]).all()][:8]
    # (against the tensor's own size, but not below 1e-3 of the largest gradient: the key-projection biases of the decoder's self-
    #  attention have gradients that cancel to ~0 -- softmax ignores a constant added to every key's logit)
    floor = 1e-3 * max(float(t.abs().max()) for t in a.values())
    worst = max((float((a[n] - b[n]).abs().max() / a[n].abs().max().clamp_min(floor)), n) for n in a)
    assert worst[0] <= 2.0 ** -5, worst
