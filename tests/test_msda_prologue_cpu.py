"""The fused MSDA prologue (csrc/msda_prologue_math.h: softmax over L*P + sampling-location arithmetic) against
the module's PyTorch formulation (ms_deform_attn.py:139-160) and its autograd, on the CPU through the host build:
2- and 6-component reference points, reference points as an expanded (broadcast) view, fp32 and bf16 I/O."""
import pytest
import torch
import torch.nn.functional as F

import backends
import native_host


def torch_prologue(offsets, logits, ref, shapes, P):
    B, Lq, M, L = offsets.shape[:4]
    weights = F.softmax(logits.float(), -1).view(B, Lq, M, L, P)
    r = ref.float()[:, :, None, :, None, :]
    off = offsets.float()
    if ref.shape[-1] == 2:
        wh = shapes.flip(-1).float()
        loc = r + off / wh[None, None, None, :, None, :]
    else:
        extent = r[..., 2::2] + r[..., 3::2]
        loc = r[..., :2] + off / P * extent * 0.5
    return loc, weights


@pytest.fixture(params=backends.BACKENDS)
def backend(request):
    from monodetr_amd import msda_prologue_ext
    msda_prologue_ext._backend = backends.get(request.param)
    yield msda_prologue_ext
    msda_prologue_ext._backend = None


@pytest.mark.parametrize("R,expanded,dtype,LP", [(2, False, torch.float32, (4, 4)), (6, True, torch.float32, (4, 4)),
                                                 (6, False, torch.float32, (3, 2)), (2, False, torch.bfloat16, (4, 4)),
                                                 (6, True, torch.bfloat16, (4, 4))])
def test_prologue_matches_module_formulas_and_autograd(backend, R, expanded, dtype, LP):
    L, P = LP
    B, Lq, M = 2, 37, 8
    g = torch.Generator().manual_seed(R + L)
    shapes = torch.tensor([(48, 160), (24, 80), (12, 40), (6, 20)][:L], dtype=torch.int64)
    offsets = (torch.randn(B, Lq, M, L, P, 2, generator=g) * 3).to(dtype).requires_grad_(True)
    logits = torch.randn(B, Lq, M, L * P, generator=g).to(dtype).requires_grad_(True)
    if expanded:
        base = torch.rand(B, Lq, R, generator=g).to(dtype).requires_grad_(True)
        ref = base[:, :, None].expand(-1, -1, L, -1)
    else:
        base = torch.rand(B, Lq, L, R, generator=g).to(dtype).requires_grad_(True)
        ref = base
    g_loc = torch.randn(B, Lq, M, L, P, 2, generator=g)
    g_att = torch.randn(B, Lq, M, L, P, generator=g)

    loc_r, att_r = torch_prologue(offsets, logits, ref, shapes, P)
    ref_grads = torch.autograd.grad([loc_r, att_r], [offsets, logits, base], [g_loc, g_att])
    loc, att = backend.msda_prologue(offsets, logits, ref, shapes)
    got_grads = torch.autograd.grad([loc, att], [offsets, logits, base], [g_loc, g_att])

    assert loc.dtype == att.dtype == torch.float32
    assert (loc - loc_r).abs().max() < 1e-6 and (att - att_r).abs().max() < 1e-6       # both evaluate in fp32
    tol = 1e-5 if dtype == torch.float32 else 2e-2                                       # bf16 gradients are rounded once
    for a, b in zip(ref_grads, got_grads):
        assert a.shape == b.shape and a.dtype == b.dtype
        assert (a.float() - b.float()).abs().max() <= tol * max(1.0, a.float().abs().max().item())


def test_module_with_the_fused_prologue_matches_default(backend, oracle):
    """MSDeformAttn.forward with MDETR_MSDA_PROLOGUE on == off (fp32), outputs and all parameter gradients."""
    from monodetr_amd.monodetr.ops.functions import ms_deform_attn_func as F_
    from monodetr_amd.monodetr.ops.modules import ms_deform_attn as mod
    saved = F_.MSDA
    F_.MSDA = oracle.OracleMSDA
    packed_was = mod._PACKED_PROJECTION
    try:
        torch.manual_seed(0)
        m = mod.MSDeformAttn(256, 4, 8, 4)
        with torch.no_grad():
            m.sampling_offsets.weight.normal_(0, 0.02); m.attention_weights.weight.normal_(0, 0.1)
        shapes = torch.tensor([(12, 20), (6, 10), (3, 5), (2, 3)], dtype=torch.int64)
        start = torch.cat((shapes.new_zeros(1), shapes.prod(1).cumsum(0)[:-1]))
        S = int(shapes.prod(1).sum())
        src = torch.randn(2, S, 256)
        query = torch.randn(2, 9, 256)
        refp = torch.rand(2, 9, 6)[:, :, None].expand(-1, -1, 4, -1) * 0.5 + 0.1
        res = {}
        for flag in (False, True, "packed"):
            mod._FUSED_PROLOGUE = bool(flag)
            mod._PACKED_PROJECTION = flag == "packed"          # (the host stand-in has no packed entry point: the module splits the GEMM output)
            m.zero_grad(set_to_none=True)
            out = m(query, refp, src, shapes, start)
            out.square().sum().backward()
            res[flag] = (out.detach(), {n: p.grad.clone() for n, p in m.named_parameters()})
        mod._FUSED_PROLOGUE = False
        for flag in (True, "packed"):
            assert (res[False][0] - res[flag][0]).abs().max() < 1e-5
            for n, gr in res[False][1].items():
                assert (gr - res[flag][1][n]).abs().max() <= 1e-4 * max(1.0, gr.abs().max().item()), (flag, n)
    finally:
        F_.MSDA = saved
        mod._FUSED_PROLOGUE = False
        mod._PACKED_PROJECTION = packed_was


def test_fp32_reference_points_stay_fp32_with_bf16_projections(backend):
    """A bf16 model body keeps its reference points in fp32 (coordinates in bf16 would be quantised to ~1/256 of the
    image): the kernel reads them as they are, so the locations differ from the all-fp32 formula only through the
    bf16 offsets."""
    g = torch.Generator().manual_seed(9)
    B, Lq, M, L, P = 2, 9, 8, 4, 4
    shapes = torch.tensor([[48, 160], [24, 80], [12, 40], [6, 20]])
    off = torch.randn(B, Lq, M, L, P, 2, generator=g).to(torch.bfloat16)
    lg = torch.randn(B, Lq, M, L * P, generator=g).to(torch.bfloat16)
    ref = (torch.rand(B, Lq, L, 2, generator=g) * 0.9 + 0.05).requires_grad_(True)            # fp32, values that bf16 cannot hold
    loc, w = backend.msda_prologue(off, lg, ref, shapes)
    want = ref.detach()[:, :, None, :, None, :] + off.float() / shapes.flip(-1)[None, None, None, :, None, :]
    assert (loc - want).abs().max() < 1e-6                                                    # bf16-rounded refs would be off by ~2e-3
    assert (ref.detach().to(torch.bfloat16).float() - ref.detach()).abs().max() > 1e-3
    loc.sum().backward()
    assert ref.grad.dtype == torch.float32 and torch.allclose(ref.grad, torch.full_like(ref.grad, M * P * 2.0 / 2), atol=1e-4)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("R", [2, 6])
def test_vector_and_scalar_kernels_agree_bit_for_bit(dtype, R):
    """L = P = 4 with 16-byte aligned tensors takes the 16-byte-access kernels (prologue_*_vec44); the same call on
    pointers that are off by one element takes the element-wise ones.  Same arithmetic, so the same bits -- on the
    HIP-on-CPU shim, through the C ABI."""
    import native_emul
    lib = native_emul.lib()
    g = torch.Generator().manual_seed(R)
    B, Lq, M, L, P = 2, 21, 8, 4, 4
    shapes = torch.tensor([[48, 160], [24, 80], [12, 40], [6, 20]], dtype=torch.int64)
    code = 2 if dtype == torch.bfloat16 else 0
    n = B * Lq * M * L * P

    def padded(t):                                   # (aligned copy, copy displaced by one element)
        buf = torch.zeros(t.numel() + 16, dtype=t.dtype)
        a, b = buf[:t.numel()], torch.zeros(t.numel() + 16, dtype=t.dtype)[1:t.numel() + 1]
        a.copy_(t.reshape(-1)); b.copy_(t.reshape(-1))
        assert a.data_ptr() % 16 == 0 and b.data_ptr() % 16 != 0
        return a, b

    off = padded((torch.randn(2 * n, generator=g) * 3).to(dtype))
    lg = padded(torch.randn(n, generator=g).to(dtype))
    ref = (torch.rand(B, Lq, L, R, generator=g) * 0.8 + 0.1).contiguous()
    g_loc, g_att = padded(torch.randn(2 * n, generator=g)), padded(torch.randn(n, generator=g))
    geom = (B, Lq, M, L, P, R, ref.stride(0), ref.stride(1), ref.stride(2))
    out = []
    for k in (0, 1):
        loc, att = padded(torch.zeros(2 * n))[k], padded(torch.zeros(n))[k]
        assert lib.mdetr_msda_prologue_forward(code, 0, off[k].data_ptr(), lg[k].data_ptr(), ref.data_ptr(), shapes.data_ptr(),
                                               loc.data_ptr(), att.data_ptr(), *geom, -1, None) == 0
        g_off, g_lg = padded(torch.zeros(2 * n, dtype=dtype))[k], padded(torch.zeros(n, dtype=dtype))[k]
        g_ref = torch.empty(B, Lq, L, R)
        assert lib.mdetr_msda_prologue_backward(code, 0, off[k].data_ptr(), ref.data_ptr(), shapes.data_ptr(), att.data_ptr(),
                                                g_loc[k].data_ptr(), g_att[k].data_ptr(), g_off.data_ptr(), g_lg.data_ptr(),
                                                g_ref.data_ptr(), *geom, -1, None) == 0
        out.append((loc.clone(), att.clone(), g_off.clone(), g_lg.clone(), g_ref))
    for a, b in zip(out[0][:4], out[1][:4]):
        assert torch.equal(a, b)
    # grad_ref is a float atomic sum over the M heads of a query: the same terms, in lane order
    assert (out[0][4] - out[1][4]).abs().max().item() <= 1e-5 * max(1.0, out[0][4].abs().max().item())


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("R", [2, 6])
def test_packed_projection_output_gives_the_same_prologue(R, dtype):
    """mdetr_msda_prologue_*_packed reads offsets | logits out of ONE [B, Lq, 384] projection output and writes the two
    gradients back into one tensor of that layout: bit-identical to the two-tensor form (same kernels, pitched rows)."""
    from monodetr_amd import msda_prologue_ext as ext
    ext._backend = backends.get("emul")
    try:
        g = torch.Generator().manual_seed(R)
        B, Lq, M, L, P = 2, 37, 8, 4, 4
        shapes = torch.tensor([[48, 160], [24, 80], [12, 40], [6, 20]])
        packed = (torch.randn(B, Lq, M * L * P * 3, generator=g) * 2).to(dtype).requires_grad_(True)
        ref = (torch.rand(B, Lq, L, R, generator=g) * 0.8 + 0.1).requires_grad_(True)
        g_loc = torch.randn(B, Lq, M, L, P, 2, generator=g)
        g_att = torch.randn(B, Lq, M, L, P, generator=g)
        assert ext.packed_supported(packed, ref, L, P)
        loc, att = ext.msda_prologue_packed(packed, ref, shapes, M, L, P)
        gp, gr = torch.autograd.grad([loc, att], [packed, ref], [g_loc, g_att])
        off = packed[..., :M * L * P * 2].reshape(B, Lq, M, L, P, 2)
        lg = packed[..., M * L * P * 2:].reshape(B, Lq, M, L * P)
        loc2, att2 = ext.msda_prologue(off, lg, ref, shapes)
        gp2, gr2 = torch.autograd.grad([loc2, att2], [packed, ref], [g_loc, g_att])
        assert torch.equal(loc, loc2) and torch.equal(att, att2)
        assert torch.equal(gp, gp2)
        assert (gr - gr2).abs().max() <= 1e-5 * max(1.0, gr2.abs().max().item())            # atomics over the heads: order differs
    finally:
        ext._backend = None
