"""csrc/bias_act.hip -- y = dropout(relu(x + bias + skip)) -- with its launcher and C-ABI entries on the HIP-on-CPU shim,
through the product's autograd wrapper (monodetr_amd/bias_act_ext.py): values and gradients against the framework
operators it replaces, channels_last 4-D and token-matrix layouts, both I/O types, row widths that do and do not divide
the grid stride, sizes above and below one grid pass; the dropout mask against the hash evaluated in numpy."""
import pytest
import torch
import torch.nn.functional as F

import native_emul
from test_add_ln_emulated_cpu import keep_mask


@pytest.fixture()
def ext():
    from monodetr_amd import bias_act_ext
    bias_act_ext._backend = native_emul.lib()
    yield bias_act_ext
    bias_act_ext._backend = None


def rounded(t, dtype):
    return t.to(dtype).float()


@pytest.mark.parametrize("shape,dtype,bias_dtype,use_skip,relu", [
    ((2, 64, 5, 7), torch.bfloat16, torch.bfloat16, False, True),      # 3x3 convolution tail: shift + ReLU
    ((2, 64, 5, 7), torch.bfloat16, torch.float32, False, True),
    ((1, 256, 9, 11), torch.bfloat16, None, True, True),               # bottleneck tail: + identity, ReLU
    ((3, 24, 4, 5), torch.float32, torch.float32, True, True),         # 6 vectors per row: not a divisor of the grid stride
    ((2, 40, 33, 65), torch.bfloat16, torch.bfloat16, True, True),     # 5 vectors per row, several grid passes
    ((1, 2048, 3, 4), torch.bfloat16, torch.bfloat16, False, False),   # shift only
    ((700, 256), torch.float32, None, False, True),                    # token matrix
    ((2, 8200, 128), torch.bfloat16, torch.float32, True, True),       # > 2048 blocks x 256 lanes x 1 vector: the 4-deep loop
])
def test_bias_act_matches_the_framework_operators(ext, shape, dtype, bias_dtype, use_skip, relu):
    g = torch.Generator().manual_seed(sum(shape))
    cl = len(shape) == 4
    C = shape[1] if cl else shape[-1]

    def make():
        t = torch.randn(shape, generator=g).to(dtype)
        return (t.contiguous(memory_format=torch.channels_last) if cl else t)

    x, skip = make().requires_grad_(True), (make().requires_grad_(True) if use_skip else None)
    bias = (torch.randn(C, generator=g) * 0.5).to(bias_dtype) if bias_dtype is not None else None
    dy = make()
    assert ext.supported(x, bias, skip)
    y = ext.bias_act(x, bias, skip, relu=relu)
    assert y.dtype == dtype and y.shape == x.shape and y.stride() == x.stride()
    y.backward(dy)
    got = [y.detach(), x.grad.clone()] + ([skip.grad.clone()] if use_skip else [])
    x.grad = None
    if use_skip:
        skip.grad = None
    bshape = (1, C, 1, 1) if cl else (C,)
    pre = x.float() + (bias.float().view(bshape) if bias is not None else 0.0) + (skip.float() if use_skip else 0.0)
    ref = (F.relu(pre) if relu else pre).to(dtype)                    # one rounding, as the kernel
    ref.backward(dy)
    want = [ref.detach(), x.grad] + ([skip.grad] if use_skip else [])
    assert torch.equal(got[0], want[0])                               # fp32 arithmetic + round-to-nearest-even: exact
    for a, b in zip(got[1:], want[1:]):
        assert torch.equal(a, b)                                      # the gradient is a select: exact as well


@pytest.mark.parametrize("rows,C,dtype,p", [(130, 256, torch.float32, 0.1), (5000, 256, torch.bfloat16, 0.1), (9, 1024, torch.bfloat16, 0.4)])
def test_bias_act_relu_dropout_of_the_ffn(ext, rows, C, dtype, p):
    g = torch.Generator().manual_seed(rows)
    x = torch.randn(rows, C, generator=g).to(dtype).requires_grad_(True)
    dy = torch.randn(rows, C, generator=g).to(dtype)
    seed = 987654321
    y = ext.bias_act(x, None, None, relu=True, dropout_p=p, seed=seed)
    y.backward(dy)
    keep = keep_mask(seed, rows * C, p).view(rows, C)
    assert abs(keep.mean().item() - (1 - p)) < 4 * (p * (1 - p) / (rows * C)) ** 0.5 + 1e-3
    scale = torch.tensor(1.0 / (1.0 - p), dtype=torch.float32)
    ref = (F.relu(x.detach().float()) * scale * keep).to(dtype)
    assert torch.equal(y.detach(), ref)
    want = torch.where((x.detach().float() > 0) & (keep > 0), dy.float() * scale, torch.zeros(())).to(dtype)
    assert torch.equal(x.grad, want)
    # a fresh seed per call when none is given: two calls draw different masks
    a = ext.bias_act(x.detach(), None, None, relu=True, dropout_p=0.5)
    b = ext.bias_act(x.detach(), None, None, relu=True, dropout_p=0.5)
    assert not torch.equal(a == 0, b == 0)


def test_bias_act_special_values_and_refusals(ext):
    x = torch.tensor([[float("nan"), -0.0, float("inf"), -float("inf"), 1.0, -1.0, 0.0, 2.0]], dtype=torch.bfloat16)
    y = ext.bias_act(x, None, None, relu=True)
    ref = F.relu(x)
    assert torch.equal(torch.isnan(y), torch.isnan(ref)) and torch.equal(torch.nan_to_num(y.float(), 7.0), torch.nan_to_num(ref.float(), 7.0))
    z = torch.randn(4, 12)                                            # 3 vectors per row (fp32)
    assert ext.supported(z) and not ext.supported(z.to(torch.bfloat16))      # 12 % 8 != 0
    assert not ext.supported(torch.randn(2, 8, 3, 3))                 # NCHW-contiguous: channel is not the fastest index
    assert not ext.supported(z, torch.randn(12, requires_grad=True))  # a bias that wants a gradient is not this kernel's business
    assert not ext.supported(z, None, torch.randn(4, 12).t().contiguous().t())
    with pytest.raises(RuntimeError):
        ext.bias_act(torch.randn(2, 8, 3, 3))
    with pytest.raises(RuntimeError):
        ext.bias_act(z, relu=False, dropout_p=0.1)
    lib = native_emul.lib()
    assert lib.mdetr_bias_act_forward(0, 0, z.data_ptr(), None, None, z.data_ptr(), 4, 10, 1, 0.0, 0, None, -1, None) != 0     # cols % 4
    assert lib.mdetr_bias_act_forward(0, 2, z.data_ptr(), z.data_ptr(), None, z.data_ptr(), 4, 12, 1, 0.0, 0, None, -1, None) != 0   # bf16 bias, f32 io
    assert lib.mdetr_bias_act_forward(0, 0, z.data_ptr(), None, None, z.data_ptr(), 4, 12, 1, 1.0, 0, None, -1, None) != 0     # p = 1
    assert lib.mdetr_bias_act_forward(0, 0, None, None, None, None, 0, 12, 1, 0.0, 0, None, -1, None) == 0                     # empty
    # in place (y = x) is allowed by the C ABI
    w = torch.randn(6, 16)
    want = F.relu(w + 1.0)
    one = torch.ones(16)
    assert lib.mdetr_bias_act_forward(0, 0, w.data_ptr(), one.data_ptr(), None, w.data_ptr(), 6, 16, 1, 0.0, 0, None, -1, None) == 0
    assert torch.equal(w, want)


def test_bottleneck_and_ffn_sites_take_the_kernel_and_agree_with_the_default_path(ext):
    """The call sites: a channels_last ResNet bottleneck (3x3 tail, residual tail) and the FFN helper, kernel on vs off."""
    from monodetr_amd.monodetr import linear
    from monodetr_amd.monodetr.backbone import Bottleneck, FrozenBatchNorm2d
    from torch import nn
    torch.manual_seed(3)
    down = nn.Sequential(nn.Conv2d(16, 32, 1, 2, bias=False), FrozenBatchNorm2d(32))
    block = Bottleneck(16, 8, stride=2, downsample=down).to(memory_format=torch.channels_last)
    for m in block.modules():
        if isinstance(m, FrozenBatchNorm2d):
            m.weight.uniform_(0.5, 1.5); m.bias.normal_(); m.running_mean.normal_(); m.running_var.uniform_(0.5, 2.0)
    x = torch.randn(2, 16, 10, 14).contiguous(memory_format=torch.channels_last)
    dy = torch.randn(2, 32, 5, 7).contiguous(memory_format=torch.channels_last)
    calls = []
    real = ext.bias_act
    results = {}
    for on in (False, True):
        ext.ENABLED = on
        ext.bias_act = lambda *a, **k: (calls.append(on), real(*a, **k))[1]
        try:
            xi = x.clone().requires_grad_(True)
            block.zero_grad()
            y = block(xi)
            y.backward(dy)
            results[on] = [y.detach(), xi.grad] + [p.grad.clone() for p in block.parameters()]
        finally:
            ext.ENABLED, ext.bias_act = False, real
    # conv1's and conv2's tails and the residual tail (on the GPU conv1 is a GEMM with the shift -- and with
    # MDETR_GEMM_RELU the ReLU -- in its epilogue, as conv3 always is)
    assert calls == [True, True, True]
    for a, b in zip(results[False], results[True]):
        assert (a - b).abs().max().item() <= 1e-5 * max(1.0, a.abs().max().item())

    lin, drop = nn.Linear(256, 256), nn.Dropout(0.1)
    t = torch.randn(3, 50, 256, requires_grad=True)
    ref = linear.ffn_hidden(t, lin, None, F.relu, tokenwise=False)                   # no dropout module: the plain path
    assert torch.equal(ref, F.relu(lin(t)))
    ext.ENABLED = True
    try:
        h = linear.ffn_hidden(t, lin, drop, F.relu, tokenwise=False)
        kept = h != 0
        assert 0.3 < kept.float().mean().item() < 0.6                               # ~ half positive, 90 % of those kept
        assert torch.allclose(h[kept], (ref / 0.9)[kept], rtol=1e-6, atol=1e-6)
        h.sum().backward()
        assert t.grad is not None and lin.weight.grad is not None and torch.isfinite(lin.weight.grad).all()
        drop.eval()
        assert torch.equal(linear.ffn_hidden(t, lin, drop, F.relu, tokenwise=False), ref)   # eval: no dropout, framework path
        assert torch.equal(linear.ffn_hidden(t, lin, drop, F.gelu, tokenwise=False), F.gelu(lin(t)))
    finally:
        ext.ENABLED = False


def test_library_gemm_with_relu_epilogue_matches_linear_then_relu():
    """MDETR_GEMM_RELU: _TokenLinear with the ReLU in the library GEMM's epilogue (torch._addmm_activation), values and all
    three gradients against relu(F.linear)."""
    from monodetr_amd.monodetr import linear
    torch.manual_seed(1)
    x = torch.randn(4, 2048, 64, requires_grad=True)
    w, b = torch.randn(96, 64, requires_grad=True), torch.randn(96, requires_grad=True)
    dy = torch.randn(4, 2048, 96)
    assert not linear._kernel_relu(x, w, b)
    linear._GEMM_RELU = True
    try:
        assert linear._kernel_relu(x, w, b) and not linear._kernel_relu(x, w, None)
        y = linear._TokenLinear.apply(x, w, b, True)
    finally:
        linear._GEMM_RELU = False
    y.backward(dy)
    got = [y.detach(), x.grad.clone(), w.grad.clone(), b.grad.clone()]
    x.grad = w.grad = b.grad = None
    ref = F.relu(F.linear(x, w, b))
    ref.backward(dy)
    for a, r in zip(got, [ref.detach(), x.grad, w.grad, b.grad]):
        assert (a - r).abs().max().item() <= 2e-4 * max(1.0, r.abs().max().item())
