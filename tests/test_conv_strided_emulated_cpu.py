"""csrc/conv_taps.hip (stride-2 convolutions and their input gradients as pixel-parity classes) and csrc/conv_wgrad.hip (the
weight gradient: split-K over pixel tiles, operands transposed in registers) with their launchers and C-ABI entries on the
HIP-on-CPU shim, through the product's autograd wrappers, against F.conv2d evaluated in fp32 on the same bf16 inputs: ragged
tiles, odd and even map sizes, several channel slabs / blocks, borders, the 1x1 projection shortcut."""
import pytest
import torch
import torch.nn.functional as F

import native_emul
from conftest import tune


@pytest.fixture()
def ext():
    from monodetr_amd import conv3x3_ext, conv_taps_ext, conv_wgrad_ext
    lib = native_emul.lib()
    conv_taps_ext._backend = conv_wgrad_ext._backend = conv3x3_ext._backend = lib
    yield conv_taps_ext
    conv_taps_ext._backend = conv_wgrad_ext._backend = conv3x3_ext._backend = None


def close(got, want, what, tol=1.2e-2):
    err = (got.float() - want.float()).abs().max().item()
    assert err <= tol * max(1.0, want.float().abs().max().item()), (what, err)


@pytest.mark.parametrize("B,H,W,C,N,k,relu,use_shift", [
    (2, 10, 70, 64, 64, 3, True, True),       # two column tiles of the output (35 wide), ragged rows
    (1, 8, 64, 128, 128, 3, False, True),     # exactly one output tile, two channel slabs, two output groups of 64
    (1, 9, 13, 64, 192, 3, True, False),      # odd map: the last input row / column is read by tap 0 / 1 only
    (2, 6, 6, 64, 64, 3, False, False),       # a map smaller than the tile
    (1, 12, 40, 256, 64, 1, False, True),     # the projection shortcut: 1x1 / stride 2 (only even pixels are read)
    (1, 7, 9, 64, 128, 1, False, False),      # ... on an odd map
    (1, 12, 40, 512, 64, 3, False, True),     # few output tiles, 16 channel slabs: the contraction is split 4 ways (fp32 partials + sum)
])
def test_strided_convolution_matches_conv2d(ext, B, H, W, C, N, k, relu, use_shift, monkeypatch):
    # (the launcher narrows the workgroup's output-channel block for problems with few tiles -- every test here: pin it to the
    # widest block the layer allows for half of the cases so that each instantiation runs)
    if (H + W) % 2 == 0:
        tune(monkeypatch, conv_taps_nb="4")
    g = torch.Generator().manual_seed(B * 1000 + H * W + C + N + k)
    x = torch.randn(B, C, H, W, generator=g).to(torch.bfloat16).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    w = (torch.randn(N, C, k, k, generator=g) / (k * C ** 0.5)).to(torch.bfloat16).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    shift = torch.randn(N, generator=g) * 0.5 if use_shift else None
    pad = 1 if k == 3 else 0
    OH, OW = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    dy = torch.randn(B, N, OH, OW, generator=g).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    assert ext.supported(x, w, padding=(pad, pad))
    assert (ext._split_count(B, OH, OW, N, C, k, relu) == 4) == (C == 512)
    y = ext.conv_strided(x, w, shift, relu=relu)
    assert y.shape == (B, N, OH, OW) and y.dtype == torch.bfloat16 and y.is_contiguous(memory_format=torch.channels_last)
    y.backward(dy)
    got = (y.detach(), x.grad.clone(), w.grad.clone())
    assert got[2].shape == w.shape
    x.grad = w.grad = None
    ref = F.conv2d(x.float(), w.float(), shift, stride=2, padding=pad)
    ref = F.relu(ref) if relu else ref
    mask = (y.detach() > 0) if relu else torch.ones_like(ref, dtype=torch.bool)
    gx, gw = torch.autograd.grad(F.conv2d(x.float(), w.float(), None, stride=2, padding=pad), (x, w), dy.float() * mask)
    close(got[0], ref, "y")
    close(got[1], gx, "dx")
    close(got[2], gw, "dw", tol=1.5e-2)


@pytest.mark.parametrize("B,H,W,C,N,k,stride", [
    (2, 9, 37, 64, 64, 3, 1),                 # two bands (the second 1 row), two column tiles (the second 5 wide), N < 128
    (1, 16, 32, 128, 160, 3, 1),              # two input-channel blocks, two output blocks (the second 32 live channels)
    (3, 5, 8, 64, 128, 3, 1),                 # a map smaller than a tile; three images = three pixel tiles
    (2, 18, 70, 64, 128, 3, 2),               # stride 2: de-interleaved columns, 9 x 35 output
    (1, 7, 9, 128, 64, 3, 2),                 # stride 2 on an odd map
    (2, 12, 40, 128, 256, 1, 2),              # the projection shortcut's weight gradient
])
def test_weight_gradient_matches_autograd(ext, B, H, W, C, N, k, stride):
    from monodetr_amd import conv_wgrad_ext
    g = torch.Generator().manual_seed(B + H * W + C + N + k + stride)
    x = torch.randn(B, C, H, W, generator=g).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    pad = 1 if k == 3 else 0
    OH, OW = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    dy = torch.randn(B, N, OH, OW, generator=g).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    assert conv_wgrad_ext.supported(x, dy, k, stride)
    for dtype in (torch.bfloat16, torch.float32):
        dw = conv_wgrad_ext.weight_gradient(x, dy, k, stride, dtype)
        assert dw.shape == (N, C, k, k) and dw.dtype == dtype
        w = torch.zeros(N, C, k, k, requires_grad=True)
        ref, = torch.autograd.grad(F.conv2d(x.float(), w, None, stride=stride, padding=pad), w, dy.float())
        close(dw, ref, "dw", tol=1e-2 if dtype == torch.bfloat16 else 1e-5)


@pytest.mark.parametrize("T,K,N", [(136, 64, 32), (1000, 128, 160), (264, 64, 256)])
def test_token_weight_and_bias_gradient_from_one_kernel(ext, T, K, N):
    """mdetr_token_wgrad: dW = dy^T x and db = column sums of dy for a token-wise linear layer, the bias gradient riding on the dy
    operand of the 1x1 weight-gradient kernel (ragged last tile, several chunks, N beyond one 128-channel block)."""
    from monodetr_amd import conv_wgrad_ext
    g = torch.Generator().manual_seed(T + K + N)
    x = torch.randn(T, K, generator=g).to(torch.bfloat16)
    dy = torch.randn(T, N, generator=g).to(torch.bfloat16)
    for dtype, tol in ((torch.float32, 1e-5), (torch.bfloat16, 1e-2)):
        dw, db = conv_wgrad_ext.token_weight_gradient(x, dy, dtype, bias=True)
        assert dw.shape == (N, K) and db.shape == (N,) and dw.dtype == db.dtype == dtype
        close(dw, dy.float().t() @ x.float(), "dw", tol=tol)
        close(db, dy.float().sum(0), "db", tol=tol)
        dw2, none = conv_wgrad_ext.token_weight_gradient(x, dy, dtype, bias=False)
        assert none is None and torch.equal(dw2, dw)


def test_stride1_convolution_takes_the_weight_gradient_kernel(ext):
    """conv3x3_ext (stride 1): forward + input gradient on conv3x3.hip, weight gradient now on conv_wgrad.hip."""
    from monodetr_amd import conv3x3_ext
    g = torch.Generator().manual_seed(9)
    x = torch.randn(2, 64, 6, 33, generator=g).to(torch.bfloat16).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    w = (torch.randn(64, 64, 3, 3, generator=g) / 24).to(torch.bfloat16).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    dy = torch.randn(2, 64, 6, 33, generator=g).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    y = conv3x3_ext.conv3x3(x, w, None, relu=False)
    y.backward(dy)
    gw, = torch.autograd.grad(F.conv2d(x.float(), w.float(), None, padding=1), w, dy.float())
    close(w.grad, gw, "dw")


def test_refusals(ext):
    lib = native_emul.lib()
    x = torch.randn(1, 64, 8, 8).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    w = torch.randn(64, 64, 3, 3).to(torch.bfloat16)
    assert ext.supported(x, w) and not ext.supported(x, w, stride=(1, 1)) and not ext.supported(x, w, padding=(0, 0))
    assert not ext.supported(x.float(), w.float()) and not ext.supported(x.contiguous(), w)
    with pytest.raises(RuntimeError):
        ext.conv_strided(x.float(), w.float())
    assert lib.mdetr_conv_wgrad_chunks(1, 8, 8, 64, 8, 8, 64, 3, 1) >= 1
    buf = torch.empty(8, dtype=torch.float32)
    assert lib.mdetr_conv_wgrad(x.data_ptr(), x.data_ptr(), buf.data_ptr(), 8, 1, 8, 8, 64, 8, 8, 64, 3, 1, -1, None) != 0     # partial buffer too small
    assert lib.mdetr_conv_wgrad(x.data_ptr(), x.data_ptr(), buf.data_ptr(), 1 << 30, 1, 8, 8, 64, 5, 8, 64, 3, 1, -1, None) != 0   # wrong output map
    assert lib.mdetr_conv_wgrad(x.data_ptr(), x.data_ptr(), buf.data_ptr(), 1 << 30, 1, 8, 8, 48, 8, 8, 64, 3, 1, -1, None) != 0   # C % 64
    d = torch.tensor([1, 8, 8, 64, 4, 4, 64, 3, 3, 3, 1, 1, 0, 1, 0, 1, 0, 1024, 256, 64, 576, 192, 64], dtype=torch.int64)
    assert lib.mdetr_conv_taps(x.data_ptr(), w.data_ptr(), None, x.data_ptr(), d.data_ptr(), 0, -1, None) != 0                   # 3x3 taps need stride 2


@pytest.mark.parametrize("B,H,W", [(1, 8, 64), (2, 13, 75), (1, 30, 200), (1, 5, 6)])
def test_stem_matches_conv2d(B, H, W):
    """csrc/conv_stem.hip: 7x7 / stride 2 / pad 3 on the 3-channel image, shift + ReLU: borders on every side, ragged tiles,
    several column tiles per workgroup (W = 200: 100 output columns = 4 tiles)."""
    from monodetr_amd import conv_stem_ext
    conv_stem_ext._backend = native_emul.lib()
    try:
        g = torch.Generator().manual_seed(H * W + B)
        x = torch.randn(B, 3, H, W, generator=g).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        w = (torch.randn(64, 3, 7, 7, generator=g) / 12).to(torch.bfloat16)
        shift = torch.randn(64, generator=g) * 0.3
        assert conv_stem_ext.supported(x, w)
        y = conv_stem_ext.conv_stem(x, w, shift)
        ref = F.relu(F.conv2d(x.float(), w.float(), shift, stride=2, padding=3))
        assert y.shape == ref.shape and y.is_contiguous(memory_format=torch.channels_last)
        close(y, ref, "stem")
        # transposition-detecting: the packed weight is not symmetric in (t, e, ch)
        p = conv_stem_ext.pack_weight(w)
        assert p.shape == (64, 176) and float(p[5, 2 * 24 + 4 * 3 + 1]) == float(w[5, 1, 2, 4]) and float(p[:, 21:24].abs().max()) == 0 and float(p[:, 168:].abs().max()) == 0
    finally:
        conv_stem_ext._backend = None
