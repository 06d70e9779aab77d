"""csrc/token_gemm.hip -- the REAL kernel source, launcher and C-ABI entry -- run on the HIP-on-CPU shim
(tests/native_emul.py): LDS weight staging, the slab hand-over, tile scheduling across waves and workgroups, ragged
tails in T and N, strided inputs, bias / ReLU epilogue addressing, for every (K, NB) instantiation.  The matrix
instruction is emulated with the operand layout validated on the GPU through attn.hip."""
import ctypes

import pytest
import torch

import native_emul


def run(x, w, bias, relu, ldy=None):
    L = native_emul.lib()
    T, K = x.shape
    N = w.shape[0]
    ldy = N if ldy is None else ldy
    y = torch.full((T, ldy), 7.0, dtype=torch.bfloat16)
    rc = L.mdetr_token_linear(x.data_ptr(), w.data_ptr(), bias.data_ptr() if bias is not None else None, y.data_ptr(),
                              T, N, K, x.stride(0), ldy, 1 if relu else 0, 0, None)
    assert rc == 0, ctypes.string_at(L.mdetr_last_error())
    return y


@pytest.mark.parametrize("T,K,N,relu,use_bias", [
    (300, 256, 256, False, True),       # NB = 8, ragged last tile (300 = 9 * 32 + 12), 10 tiles over 3 workgroups
    (64, 256, 128, True, True),         # NB = 4
    (97, 128, 264, False, False),       # K = 128; N = 264: two column blocks, the second with 8 live features
    (33, 512, 136, True, True),         # K = 512, NB = 4, two column blocks
    (1, 256, 8, False, True),           # one token, one quad pair
    (170, 64, 256, True, True),         # K = 64: a single slab per tile (the backbone's 64 -> 256 expansions)
    (70, 64, 72, False, False),
    (2100, 256, 256, True, True),       # 66 tiles over 17 workgroups
    (33000, 128, 16, True, True),       # 1032 tiles > 4 waves x 256 workgroups (the launch cap): waves wrap to a second tile,
                                        # with the next tile's loads requested across the wrap
])
def test_token_gemm_source_on_the_cpu_shim_matches_linear(T, K, N, relu, use_bias):
    g = torch.Generator().manual_seed(T + K + N)
    x = (torch.randn(T, K, generator=g) * 0.5).to(torch.bfloat16)
    w = (torch.randn(N, K, generator=g) * 0.1).to(torch.bfloat16)
    b = torch.randn(N, generator=g).to(torch.bfloat16) if use_bias else None
    y = run(x, w, b, relu)
    ref = x.double() @ w.double().t() + (b.double() if use_bias else 0)
    if relu:
        ref = ref.clamp(min=0)
    err = (y.double() - ref).abs().max().item()
    assert err <= 2e-2 * max(1.0, ref.abs().max().item()), err          # bf16 output rounding


@pytest.mark.parametrize("T,N,relu,use_bias", [
    (300, 256, False, True),        # two column blocks of 128 features, ragged last tile
    (64, 128, True, True),
    (1, 8, False, True),            # one token, one quad pair: the other rows of the tile re-read row 0
    (2100, 256, True, True),        # 66 tiles
    (97, 264, False, False),        # three column blocks, the last with 8 live features
    (33000, 16, True, True),        # NB = 2 form (N <= 64); more tiles than waves in flight: the refill of a slab's registers across tiles
    (130, 64, False, True),
])
def test_token_gemm_direct_form_on_the_cpu_shim_matches_linear(monkeypatch, T, N, relu, use_bias):
    """MDETR_TOKEN_GEMM_DIRECT=1 (K = 256): the lane loads its own MFMA operand from global memory, weight fragments are read one
    k-step ahead, rows beyond T re-read row T - 1 and are not stored."""
    monkeypatch.setenv("MDETR_TOKEN_GEMM_DIRECT", "1")
    K = 256
    g = torch.Generator().manual_seed(T + N)
    x = (torch.randn(T, K, generator=g) * 0.5).to(torch.bfloat16)
    w = (torch.randn(N, K, generator=g) * 0.1).to(torch.bfloat16)
    b = torch.randn(N, generator=g).to(torch.bfloat16) if use_bias else None
    y = run(x, w, b, relu)
    monkeypatch.setenv("MDETR_TOKEN_GEMM_DIRECT", "0")
    y0 = run(x, w, b, relu)
    ref = x.double() @ w.double().t() + (b.double() if use_bias else 0)
    if relu:
        ref = ref.clamp(min=0)
    assert (y.double() - ref).abs().max().item() <= 2e-2 * max(1.0, ref.abs().max().item())
    assert torch.equal(y, y0)                                # same products in the same order as the LDS-staged form


@pytest.mark.parametrize("T,K,N,relu", [(33, 512, 136, True), (97, 128, 264, False), (170, 64, 256, True), (70, 64, 72, False), (33000, 64, 64, True)])
def test_token_gemm_direct_form_other_contraction_lengths(monkeypatch, T, K, N, relu):
    g = torch.Generator().manual_seed(T + K + N)
    x = (torch.randn(T, K, generator=g) * 0.5).to(torch.bfloat16)
    w = (torch.randn(N, K, generator=g) * 0.1).to(torch.bfloat16)
    b = torch.randn(N, generator=g).to(torch.bfloat16)
    monkeypatch.setenv("MDETR_TOKEN_GEMM_DIRECT", "1")
    y = run(x, w, b, relu)
    monkeypatch.setenv("MDETR_TOKEN_GEMM_DIRECT", "0")
    assert torch.equal(y, run(x, w, b, relu))


def test_token_gemm_direct_form_respects_row_strides(monkeypatch):
    monkeypatch.setenv("MDETR_TOKEN_GEMM_DIRECT", "1")
    g = torch.Generator().manual_seed(1)
    T, K, N = 70, 256, 40
    big = (torch.randn(T, K + 64, generator=g)).to(torch.bfloat16)
    x = big[:, 32:32 + K]
    w = (torch.randn(N, K, generator=g) * 0.1).to(torch.bfloat16)
    y = run(x, w, None, False, ldy=N + 12)
    ref = (x.double() @ w.double().t())
    assert (y[:, :N].double() - ref).abs().max() <= 2e-2 * ref.abs().max()
    assert torch.all(y[:, N:] == 7.0)


def test_token_gemm_respects_row_strides_and_leaves_padding_untouched():
    g = torch.Generator().manual_seed(0)
    T, K, N = 70, 128, 40
    big = (torch.randn(T, K + 64, generator=g)).to(torch.bfloat16)
    x = big[:, 32:32 + K]                                   # ldx = K + 64, base offset 64 bytes (16-byte aligned)
    w = (torch.randn(N, K, generator=g) * 0.1).to(torch.bfloat16)
    y = run(x, w, None, False, ldy=N + 12)
    ref = (x.double() @ w.double().t())
    assert (y[:, :N].double() - ref).abs().max() <= 2e-2 * ref.abs().max()
    assert torch.all(y[:, N:] == 7.0)                        # columns beyond N belong to the caller


def test_token_gemm_rejects_unsupported_shapes_through_the_c_abi():
    L = native_emul.lib()
    x = torch.zeros(8, 96, dtype=torch.bfloat16)
    w = torch.zeros(8, 96, dtype=torch.bfloat16)
    y = torch.zeros(8, 8, dtype=torch.bfloat16)
    rc = L.mdetr_token_linear(x.data_ptr(), w.data_ptr(), None, y.data_ptr(), 8, 8, 96, 96, 8, 0, 0, None)
    assert rc != 0 and b"K" in ctypes.string_at(L.mdetr_last_error())


@pytest.mark.parametrize("relu", [False, True])
def test_token_linear_autograd_function_with_the_emulated_kernel(relu, monkeypatch):
    """The autograd integration the model uses (monodetr/linear.py _TokenLinear): forward through the kernel (ReLU in
    its epilogue), input gradient through the kernel with the transposed weight, weight / bias gradients in torch."""
    from monodetr_amd import token_gemm_ext
    from monodetr_amd.monodetr import linear
    monkeypatch.setattr(token_gemm_ext, "_backend", native_emul.lib())
    monkeypatch.setattr(linear, "_TOKEN_GEMM", True)
    g = torch.Generator().manual_seed(4)
    x = (torch.randn(2, 150, 256, generator=g) * 0.5).to(torch.bfloat16).requires_grad_(True)
    w = (torch.randn(128, 256, generator=g) * 0.1).to(torch.bfloat16).requires_grad_(True)
    b = torch.randn(128, generator=g).to(torch.bfloat16).requires_grad_(True)
    dy = torch.randn(2, 150, 128, generator=g).to(torch.bfloat16)
    y = linear._TokenLinear.apply(x, w, b, relu)
    y.backward(dy)
    got = (y.detach(), x.grad.clone(), w.grad.clone(), b.grad.clone())
    x.grad = w.grad = b.grad = None
    ref = torch.nn.functional.linear(x.float(), w.float(), b.float())
    ref = ref.clamp(min=0) if relu else ref
    ref.backward(dy.float())
    for name, a, r in zip(("y", "dx", "dw", "db"), got, (ref.detach(), x.grad, w.grad, b.grad)):
        assert a.dtype == torch.bfloat16
        assert (a.float() - r.float()).abs().max() <= 3e-2 * max(1.0, r.float().abs().max().item()), name


@pytest.mark.parametrize("T,K,N,relu,use_bias", [
    (300, 256, 256, False, True),       # five 64-token tiles over five workgroups, ragged last tile (300 = 4 * 64 + 44)
    (64, 256, 128, True, True),         # half of the waves own no live feature
    (1, 256, 8, False, True),           # one token, one quad pair: 32-token tiles, the other rows re-read row 0
    (97, 128, 264, False, False),       # two column blocks (the second with 8 live features) sharing the token ranges
    (170, 64, 256, True, True),         # K = 64: a tile's pieces do not fill the workgroup's threads
    (70, 64, 72, False, False),
    (2100, 256, 256, True, True),       # 66 tiles of 32 tokens over 66 workgroups
    (20000, 256, 24, True, True),       # 313 tiles of 64 over 256 workgroups: a second tile per workgroup through the other buffer
    (70000, 128, 16, False, True),      # 1 094 tiles over 256 workgroups: four to five tiles each -- both stages refilled, both breaks
    (33, 512, 136, True, True),         # K = 512: 32 KB of weight per wave, 32-token tiles only
    (9000, 512, 16, False, True),       # ... 282 tiles over 256 workgroups
])
@pytest.mark.parametrize("ystage", ["1", "0"])
def test_token_gemm_weight_in_registers_form_on_the_cpu_shim(monkeypatch, T, K, N, relu, use_bias, ystage):
    """MDETR_TOKEN_GEMM_DIRECT=2: the weight slice of a wave in registers, the tokens of a tile shared by the workgroup's eight
    waves through LDS, two register stages ahead; the output tile leaves through LDS in whole rows (ystage 1, the default) or in
    8-byte pieces straight from the accumulators (0).  Same products in the same order as the LDS-weight form: identical bits."""
    monkeypatch.setenv("MDETR_TOKEN_GEMM_YSTAGE", ystage)
    g = torch.Generator().manual_seed(T + K + N)
    x = (torch.randn(T, K, generator=g) * 0.5).to(torch.bfloat16)
    w = (torch.randn(N, K, generator=g) * 0.1).to(torch.bfloat16)
    b = torch.randn(N, generator=g).to(torch.bfloat16) if use_bias else None
    monkeypatch.setenv("MDETR_TOKEN_GEMM_DIRECT", "2")
    y = run(x, w, b, relu)
    monkeypatch.setenv("MDETR_TOKEN_GEMM_DIRECT", "0")
    y0 = run(x, w, b, relu)
    ref = x.double() @ w.double().t() + (b.double() if use_bias else 0)
    if relu:
        ref = ref.clamp(min=0)
    assert (y.double() - ref).abs().max().item() <= 2e-2 * max(1.0, ref.abs().max().item())
    assert torch.equal(y, y0)


def test_token_gemm_weight_in_registers_form_with_strided_rows(monkeypatch):
    """Row strides larger than the row (a column slice of a wider matrix in, a wider matrix out): untouched columns keep their values."""
    monkeypatch.setenv("MDETR_TOKEN_GEMM_DIRECT", "2")
    g = torch.Generator().manual_seed(9)
    wide = (torch.randn(150, 320, generator=g) * 0.5).to(torch.bfloat16)
    x = wide[:, 64:320]                                                  # [150, 256], ldx = 320
    w = (torch.randn(40, 256, generator=g) * 0.1).to(torch.bfloat16)
    ref = x.double() @ w.double().t()
    for ldy in (48, 44):                                                 # 44: rows only 8-byte aligned -> the two-piece row stores
        y = run(x, w, None, False, ldy=ldy)
        assert (y[:, :40].double() - ref).abs().max().item() <= 2e-2 * max(1.0, ref.abs().max().item())
        assert torch.all(y[:, 40:] == 7.0)


@pytest.mark.parametrize("tt,per_cu", [("32", "2"), ("64", "1"), ("32", "1")])
def test_token_gemm_weight_in_registers_form_launch_knobs(monkeypatch, tt, per_cu):
    """MDETR_TOKEN_GEMM_WS_TT / _PER_CU (A/B runs): the forced token tile and the two-workgroups-per-CU launch cap change the
    schedule -- 40 000 rows = 1 250 tiles of 32 over 512 workgroups, or 625 of 64 over 256 -- not a bit of the result."""
    g = torch.Generator().manual_seed(77)
    x = (torch.randn(40000, 128, generator=g) * 0.5).to(torch.bfloat16)
    w = (torch.randn(24, 128, generator=g) * 0.1).to(torch.bfloat16)
    b = torch.randn(24, generator=g).to(torch.bfloat16)
    monkeypatch.setenv("MDETR_TOKEN_GEMM_DIRECT", "0")
    y0 = run(x, w, b, True)
    monkeypatch.setenv("MDETR_TOKEN_GEMM_DIRECT", "2")
    monkeypatch.setenv("MDETR_TOKEN_GEMM_WS_TT", tt)
    monkeypatch.setenv("MDETR_TOKEN_GEMM_WS_PER_CU", per_cu)
    assert torch.equal(run(x, w, b, True), y0)
