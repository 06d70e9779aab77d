"""csrc/attn.hip (GPU-validated) on the HIP-on-CPU shim, through the product's autograd wrapper.  Purpose: close the
loop on the shim's matrix-instruction emulation -- its operand / accumulator layout is the one attn.hip relies on, and
attn.hip is validated on the GPU against fp64 (tests/test_attn_gpu.py); reproducing those results here means the
emulated layout is the hardware's, which is what the token-GEMM emulation (tests/test_token_gemm_emulated_cpu.py)
assumes.  Also runs the dropout path against the hash mask restated in tests/test_attn_gpu.py."""
import pytest
import torch

import native_emul
from test_attn_gpu import keep_mask, reference
from conftest import tune


# default build, and the build with the conflict-free staging map (-DMDETR_ATTN_STAGE_REMAP=1)
@pytest.fixture(params=[(), ("MDETR_ATTN_STAGE_REMAP=1",)], ids=["default", "stage_remap"])
def ext(request):
    from monodetr_amd import attn_ext
    attn_ext._backend = native_emul.lib(request.param)
    yield attn_ext
    attn_ext._backend = None


@pytest.mark.parametrize("B,H,Lq,Lk,dtype,masked", [(1, 2, 70, 130, torch.float32, True), (2, 1, 33, 65, torch.bfloat16, False),
                                                    (1, 1, 1, 1, torch.float32, False), (2, 2, 50, 50, torch.float32, False)])
def test_emulated_attention_matches_fp64(ext, B, H, Lq, Lk, dtype, masked):
    torch.manual_seed(B * 100 + Lq)
    E = H * 32
    q, k, v = (torch.randn(B, L, E).to(dtype).requires_grad_(True) for L in (Lq, Lk, Lk))
    go = torch.randn(B, Lq, E).to(dtype)
    kpm = (torch.rand(B, Lk) < 0.3) if masked else None
    out = ext.fused_attention(q, k, v, H, key_padding_mask=kpm)
    out.backward(go)
    qd, kd, vd = (t.detach().double().requires_grad_(True) for t in (q, k, v))
    ref = reference(qd, kd, vd, H, kpm)
    ref.backward(go.double())
    tol, gtol = (2e-5, 2e-5) if dtype == torch.float32 else (2e-2, 3e-2)      # fp32: three-part split operands, fp32 products
    assert out.dtype == dtype and (out.double() - ref).abs().max() < tol * max(1.0, ref.abs().max().item())
    for g, r, name in ((q.grad, qd.grad, "dq"), (k.grad, kd.grad, "dk"), (v.grad, vd.grad, "dv")):
        assert (g.double() - r).abs().max() < gtol * max(1.0, r.abs().max().item()), name


def test_emulated_attention_dropout_uses_the_documented_hash(ext):
    torch.manual_seed(2)
    B, H, Lq, Lk, p, seed = 1, 2, 40, 70, 0.1, 0x1234567890ABCDEF
    E = H * 32
    q, k, v = (torch.randn(B, L, E, requires_grad=True) for L in (Lq, Lk, Lk))
    go = torch.randn(B, Lq, E)
    out = ext.fused_attention(q, k, v, H, dropout_p=p, seed=seed)
    out.backward(go)
    keep = keep_mask(seed, B, H, Lq, Lk, p)
    qd, kd, vd = (t.detach().double().requires_grad_(True) for t in (q, k, v))
    ref = reference(qd, kd, vd, H, None, keep, p)
    ref.backward(go.double())
    assert (out.double() - ref).abs().max() < 2e-5 * max(1.0, ref.abs().max().item())
    for g, r in ((q.grad, qd.grad), (k.grad, kd.grad), (v.grad, vd.grad)):
        assert (g.double() - r).abs().max() < 2e-5 * max(1.0, r.abs().max().item())


@pytest.mark.parametrize("masked,p", [(False, 0.0), (True, 0.1)])
def test_emulated_attention_with_the_key_range_split_over_two_wave_groups(ext, masked, p, monkeypatch):
    """Few query tiles and several key tiles (the decoder's 550 x 1920 in small): the bf16 forward and dQ kernels put two 4-wave
    groups into a workgroup, each walking half of the key tiles, and merge the partial softmax states through LDS -- against the
    fp64 reference with the documented dropout mask, a ragged last tile (330 = 5 tiles + 10 keys: the second group's last tile is
    all padding) and a key-padding mask."""
    tune(monkeypatch, attn_ksplit="1")                 # (the launcher reads it per call; off by default: measured slower)
    torch.manual_seed(11)
    B, H, Lq, Lk, seed = 2, 2, 70, 330, 0x0FEDCBA987654321
    E = H * 32
    q, k, v = (torch.randn(B, L, E).to(torch.bfloat16).requires_grad_(True) for L in (Lq, Lk, Lk))
    go = torch.randn(B, Lq, E).to(torch.bfloat16)
    kpm = (torch.rand(B, Lk) < 0.3) if masked else None
    out = ext.fused_attention(q, k, v, H, dropout_p=p, seed=seed if p > 0 else None, key_padding_mask=kpm)
    out.backward(go)
    keep = keep_mask(seed, B, H, Lq, Lk, p) if p > 0 else None
    qd, kd, vd = (t.detach().double().requires_grad_(True) for t in (q, k, v))
    ref = reference(qd, kd, vd, H, kpm, keep, p)
    ref.backward(go.double())
    assert (out.double() - ref).abs().max() < 2e-2 * max(1.0, ref.abs().max().item())
    for g, r, name in ((q.grad, qd.grad, "dq"), (k.grad, kd.grad, "dk"), (v.grad, vd.grad, "dv")):
        assert (g.double() - r).abs().max() < 3e-2 * max(1.0, r.abs().max().item()), name
