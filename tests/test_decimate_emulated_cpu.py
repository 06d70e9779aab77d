"""csrc/decimate.hip (every second pixel of a channels-last activation, and the adjoint) on the HIP-on-CPU shim, through
monodetr_amd/decimate_ext.py: against tensor slicing and its autograd, even / odd extents, both element sizes; and the 1x1 /
stride-2 convolution assembled from it against F.conv2d."""
import pytest
import torch
import torch.nn.functional as F

import native_emul


@pytest.fixture()
def ext():
    from monodetr_amd import decimate_ext
    decimate_ext._backend = native_emul.lib()
    yield decimate_ext
    decimate_ext._backend = None


@pytest.mark.parametrize("B,C,H,W,dtype", [(2, 64, 12, 40, torch.bfloat16), (1, 8, 7, 9, torch.bfloat16), (3, 4, 5, 6, torch.float32),
                                           (1, 256, 1, 1, torch.bfloat16), (2, 16, 2, 3, torch.float32)])
def test_decimate_matches_slicing_and_its_adjoint(ext, B, C, H, W, dtype):
    g = torch.Generator().manual_seed(B * 1000 + H * 10 + W)
    x = torch.randn(B, C, H, W, generator=g).to(dtype).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    assert ext.supported(x)
    y = ext.decimate2(x)
    want = x.detach()[:, :, ::2, ::2]
    assert y.shape == want.shape and y.is_contiguous(memory_format=torch.channels_last) and torch.equal(y, want)
    gy = torch.randn(want.shape, generator=g).to(dtype)
    (gx,) = torch.autograd.grad(y, x, gy)
    ref = torch.zeros_like(x.detach())
    ref[:, :, ::2, ::2] = gy
    assert gx.is_contiguous(memory_format=torch.channels_last) and torch.equal(gx, ref)


def test_rejects_what_it_cannot_take(ext):
    assert not ext.supported(torch.zeros(1, 4, 4, 4, dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last))   # 8-byte pixels
    assert not ext.supported(torch.zeros(1, 8, 4, 4, dtype=torch.bfloat16))                                                  # NCHW
    with pytest.raises(RuntimeError):
        ext.decimate2(torch.zeros(1, 8, 4, 4, dtype=torch.bfloat16))


def test_projection_shortcut_as_gather_plus_token_gemm(ext):
    """conv(1x1, stride 2)(x) == linear over the tokens of decimate2(x), values and all gradients (fp32)."""
    g = torch.Generator().manual_seed(4)
    x = torch.randn(2, 16, 6, 10, generator=g).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    w = (torch.randn(24, 16, 1, 1, generator=g) * 0.2).requires_grad_(True)
    b = torch.randn(24, generator=g).requires_grad_(True)
    want = F.conv2d(x, w, b, stride=2)
    xd = ext.decimate2(x)
    got = F.linear(xd.permute(0, 2, 3, 1), w.view(24, 16), b).permute(0, 3, 1, 2)
    assert (want - got).abs().max() < 1e-5
    gy = torch.randn(want.shape, generator=g)
    for a, c in zip(torch.autograd.grad(want, [x, w, b], gy), torch.autograd.grad(got, [x, w, b], gy)):
        assert (a - c).abs().max() < 1e-5


@pytest.mark.parametrize("B,C,H,W", [(2, 64, 12, 40), (1, 8, 7, 9), (1, 16, 1, 1), (2, 8, 2, 5)])
def test_maxpool_matches_the_framework(ext, B, C, H, W):
    g = torch.Generator().manual_seed(H * 100 + W)
    x = torch.randn(B, C, H, W, generator=g).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    assert ext.maxpool_supported(x)
    y = ext.maxpool3x3s2(x)
    want = F.max_pool2d(x.float(), 3, 2, 1).to(torch.bfloat16)
    assert y.shape == want.shape and y.is_contiguous(memory_format=torch.channels_last) and torch.equal(y, want)
    assert not ext.maxpool_supported(x.clone().requires_grad_(True))          # forward only: anything that needs a gradient stays with torch


def test_flat_gather_of_many_tensors(ext):
    """mdetr_gather_flat (the optimizer's flat gradient buffer): tensors of odd sizes and alignments, more than one launch's worth of
    them (256 pointers per launch), several chunks per tensor -- every byte lands at its offset, the padding between stays."""
    import ctypes
    lib = native_emul.lib()
    g = torch.Generator().manual_seed(3)
    sizes = [1, 7, 64, 4097, 30000, 3] + [5 + (i * 37) % 211 for i in range(300)]
    esz, chunk, pad = 2, 4096, 64
    srcs, offs, total = [], [], 0
    for i, n in enumerate(sizes):
        base = torch.randn(n + 3, generator=g).to(torch.bfloat16)
        srcs.append(base[(i % 3):(i % 3) + n])                                   # bases at 0, 2 and 4 bytes past an allocation
        offs.append(total)
        total += -(-n // pad) * pad
    flat = torch.full((total,), 7.0, dtype=torch.bfloat16)
    bt, bs, begin = [], [], []
    for i, n in enumerate(sizes):
        begin.append(len(bt))
        for s0 in range(0, n * esz, chunk):
            bt.append(i); bs.append(s0)
    begin.append(len(bt))
    dst_off = torch.tensor([o * esz for o in offs], dtype=torch.int64)
    nbytes = torch.tensor([n * esz for n in sizes], dtype=torch.int64)
    blk_t, blk_s = torch.tensor(bt, dtype=torch.int32), torch.tensor(bs, dtype=torch.int64)
    ptrs = (ctypes.c_void_p * len(sizes))(*[t.data_ptr() for t in srcs])
    rc = lib.mdetr_gather_flat(ptrs, len(sizes), (ctypes.c_int * len(begin))(*begin), flat.data_ptr(), dst_off.data_ptr(), nbytes.data_ptr(),
                               blk_t.data_ptr(), blk_s.data_ptr(), chunk, -1, None)
    assert rc == 0
    for t, o, n in zip(srcs, offs, sizes):
        assert torch.equal(flat[o:o + n], t)
        assert bool((flat[o + n:o + -(-n // pad) * pad] == 7.0).all())
