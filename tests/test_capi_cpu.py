"""CPU-side checks of the drop-in boundary: the C-ABI library loads, exports every symbol that
include/monodetr_amd.h declares, and the host shim mirrors the reference's error behaviour.
No kernel is launched here (no GPU in the build container)."""
import ctypes
import os
import re

import pytest
import torch

from conftest import ROOT, make_problem


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "monodetr_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mdetr_[a-z0-9_]+)\s*\(", text)))


@pytest.fixture(scope="module")
def built_lib():
    from monodetr_amd import build
    return build.build()


def test_library_exports_every_declared_symbol(built_lib):
    from monodetr_amd import _capi
    names = _declared_symbols()
    assert "mdetr_msda_forward" in names and "mdetr_msda_backward" in names
    handle = ctypes.CDLL(built_lib)
    for n in names:
        assert hasattr(handle, n), "libmonodetr_amd.so does not export %s" % n
        assert n in _capi.SIGNATURES, "monodetr_amd/_capi.py has no signature for %s" % n
    assert sorted(_capi.SIGNATURES) == names


def test_abi_version_and_variant_dispatch(built_lib):
    from monodetr_amd import _capi
    lib = _capi.lib()
    assert lib.mdetr_abi_version() == _capi.ABI_VERSION
    # fast path = f32, 32 channels per head, L*P a multiple of 4 (default config M=8,D=32,L=4,P=4)
    assert lib.mdetr_msda_variant(_capi.MDETR_F32, 8, 32, 4, 4) == 1
    assert lib.mdetr_msda_variant(_capi.MDETR_F32, 3, 32, 2, 2) == 1
    assert lib.mdetr_msda_variant(_capi.MDETR_F64, 8, 32, 4, 4) == 0
    assert lib.mdetr_msda_variant(_capi.MDETR_F32, 8, 30, 4, 4) == 0
    assert lib.mdetr_msda_variant(_capi.MDETR_F32, 8, 32, 3, 3) == 0


def test_argument_validation_without_gpu(built_lib):
    from monodetr_amd import _capi
    lib = _capi.lib()
    rc = lib.mdetr_msda_forward(7, None, None, None, None, None, None, 1, 1, 1, 1, 1, 1, 1, 0, None)
    assert rc == -1 and b"dtype" in lib.mdetr_last_error()
    rc = lib.mdetr_msda_forward(0, None, None, None, None, None, None, 1, 4, 1, 1, 1, 1, 1, 0, None)
    assert rc == -1 and b"null" in lib.mdetr_last_error()
    rc = lib.mdetr_msda_forward(0, 8, 16, 16, 16, 16, 16, 1, 4, 1, 1, 1, 1, 1, 0, None)
    assert rc == -3 and b"aligned" in lib.mdetr_last_error()
    # empty problems succeed without touching the device
    assert lib.mdetr_msda_forward(0, None, None, None, None, None, None, 0, 4, 1, 1, 1, 1, 1, 0, None) == 0
    assert lib.mdetr_msda_forward(0, None, None, None, None, None, None, 2, 4, 1, 1, 1, 0, 1, 0, None) == 0
    with pytest.raises(RuntimeError, match="code -1"):
        _capi.check(lib.mdetr_msda_backward(0, *([None] * 9), 1, 1, 0, 1, 1, 1, 1, 0, None), "bwd")


def test_extension_module_rejects_cpu_tensors_like_the_reference():
    """ops/src/ms_deform_attn.h:38 and cpu/ms_deform_attn_cpu.cpp:26,39: no CPU implementation."""
    from monodetr_amd import msda_ext
    p = make_problem(1, 2, 4, 3, [(3, 3)], 2, torch.float32)
    with pytest.raises(RuntimeError, match="Not implemented on the CPU"):
        msda_ext.ms_deform_attn_forward(p["value"], p["shapes"], p["level_start"], p["loc"], p["attn"], 64)
    with pytest.raises(RuntimeError, match="Not implemented on the CPU"):
        msda_ext.ms_deform_attn_backward(p["value"], p["shapes"], p["level_start"], p["loc"], p["attn"], p["grad_out"], 64)


def test_function_has_no_silent_fallback():
    from monodetr_amd.monodetr.ops.functions import MSDeformAttnFunction
    p = make_problem(1, 2, 4, 3, [(3, 3)], 2, torch.float32)
    with pytest.raises(RuntimeError, match="Not implemented on the CPU"):
        MSDeformAttnFunction.apply(p["value"], p["shapes"], p["level_start"], p["loc"], p["attn"], 64)


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from monodetr_amd import _capi
    monkeypatch.setattr(_capi, "_lib", None)
    monkeypatch.setattr(_capi, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(RuntimeError, match="not built"):
        _capi.lib()


def test_module_surface_and_init():
    """Parameter names / shapes / init of MSDeformAttn (ops/modules/ms_deform_attn.py:94-120)."""
    from monodetr_amd.monodetr.ops.modules import MSDeformAttn, MSDeformAttn_cross, MultiheadAttention  # noqa
    m = MSDeformAttn(256, 4, 8, 4)
    sd = m.state_dict()
    assert {k: tuple(v.shape) for k, v in sd.items()} == {
        "sampling_offsets.weight": (256, 256), "sampling_offsets.bias": (256,),
        "attention_weights.weight": (128, 256), "attention_weights.bias": (128,),
        "value_proj.weight": (256, 256), "value_proj.bias": (256,),
        "output_proj.weight": (256, 256), "output_proj.bias": (256,)}
    assert m.im2col_step == 64
    assert sd["sampling_offsets.weight"].abs().max() == 0 and sd["attention_weights.weight"].abs().max() == 0
    b = sd["sampling_offsets.bias"].view(8, 4, 4, 2)
    assert torch.allclose(b[0, :, :, 0], torch.tensor([1., 2., 3., 4.]).expand(4, 4))      # head 0 points +x
    assert torch.allclose(b[2, 0, :, 1], torch.tensor([1., 2., 3., 4.]))                   # head 2 points +y
    assert torch.allclose(b[1, 0, 3], torch.tensor([4., 4.]))                              # 45 deg, max-norm
    with pytest.raises(ValueError):
        MSDeformAttn(250, 4, 8, 4)


def test_module_forward_on_cpu_with_oracle_backend(monkeypatch, oracle):
    """Host-side arithmetic of MSDeformAttn.forward (:138-155): sampling locations for 2-d and 6-d
    reference points, softmax over L*P.  The operator itself is swapped for the CPU oracle *by the
    test*; the product package has no CPU path."""
    from monodetr_amd.monodetr.ops.functions import ms_deform_attn_func as F_
    from monodetr_amd.monodetr.ops.modules import MSDeformAttn
    from oracle.msda_torch_ref import msda_grid_sample
    monkeypatch.setattr(F_, "MSDA", oracle.OracleMSDA)
    torch.manual_seed(0)
    shapes = torch.tensor([(6, 8), (3, 4)])
    start = torch.tensor([0, 48])
    m = MSDeformAttn(32, 2, 4, 2).double()
    with torch.no_grad():                                  # make offsets/weights input dependent
        m.sampling_offsets.weight.normal_(0, 0.05)
        m.attention_weights.weight.normal_(0, 0.5)
    q = torch.randn(2, 5, 32, dtype=torch.float64)
    src = torch.randn(2, 60, 32, dtype=torch.float64, requires_grad=True)
    mask = torch.zeros(2, 60, dtype=torch.bool)
    mask[1, -7:] = True
    for ref in (torch.rand(2, 5, 2, 2, dtype=torch.float64),
                torch.cat([torch.rand(2, 5, 2, 2), torch.rand(2, 5, 2, 4) * 0.2], -1).double()):
        out = m(q, ref, src, shapes, start, mask)
        # independent evaluation with plain torch ops + grid_sample
        v = m.value_proj(src).masked_fill(mask[..., None], 0).view(2, 60, 4, 8)
        off = m.sampling_offsets(q).view(2, 5, 4, 2, 2, 2)
        w = m.attention_weights(q).view(2, 5, 4, 4).softmax(-1).view(2, 5, 4, 2, 2)
        if ref.shape[-1] == 2:
            loc = ref[:, :, None, :, None, :] + off / torch.tensor([[8., 6.], [4., 3.]], dtype=torch.float64)[None, None, None, :, None, :]
        else:
            r = ref[:, :, None, :, None, :]
            loc = r[..., :2] + off / 2 * torch.stack([r[..., 2] + r[..., 3], r[..., 4] + r[..., 5]], -1) * 0.5
        want = m.output_proj(msda_grid_sample(v, shapes, loc, w))
        assert (out - want).abs().max() < 1e-12
        g1, = torch.autograd.grad(out.sum(), src, retain_graph=True)
        g2, = torch.autograd.grad(want.sum(), src)
        assert (g1 - g2).abs().max() < 1e-10


# ---- host (CPU) twins of the operator: mdetr_msda_forward_cpu / _backward_cpu (reference stubs cpu/ms_deform_attn_cpu.cpp:17-40) ----
@pytest.mark.parametrize("B,M,D,Lq,shapes,P,dtype,lo,hi", [
    (1, 2, 2, 2, [(6, 4), (3, 2)], 2, torch.float64, 0.0, 1.0),                      # the reference's own test problem (ops/test.py:21-28)
    (2, 8, 32, 37, [(12, 40), (6, 20), (3, 10), (2, 5)], 4, torch.float32, -0.2, 1.2),
    (1, 3, 30, 9, [(5, 7), (1, 1)], 3, torch.float64, -0.1, 1.1),
])
def test_cpu_entry_points_match_the_oracle(built_lib, oracle, B, M, D, Lq, shapes, P, dtype, lo, hi):
    from monodetr_amd import msda_ext
    p = make_problem(B, M, D, Lq, shapes, P, dtype, seed=3, lo=lo, hi=hi)
    with pytest.raises(RuntimeError, match="Not implemented on the CPU"):           # default: the reference's behaviour
        msda_ext.ms_deform_attn_forward(p["value"], p["shapes"], p["level_start"], p["loc"], p["attn"], 64)
    msda_ext.allow_cpu(True)
    try:
        out = msda_ext.ms_deform_attn_forward(p["value"], p["shapes"], p["level_start"], p["loc"], p["attn"], 64)
        gv, gl, ga = msda_ext.ms_deform_attn_backward(p["value"], p["shapes"], p["level_start"], p["loc"], p["attn"], p["grad_out"], 64)
    finally:
        msda_ext.allow_cpu(False)
    ref = oracle.forward(p["value"], p["shapes"], p["level_start"], p["loc"], p["attn"])
    rv, rl, ra = oracle.backward(p["value"], p["shapes"], p["level_start"], p["loc"], p["attn"], p["grad_out"])
    tol = 1e-12 if dtype == torch.float64 else 1e-5
    for got, want in ((out, ref), (gv, rv), (gl, rl), (ga, ra)):
        assert got.shape == want.shape and got.dtype == want.dtype
        assert (got - want).abs().max() <= tol * max(1.0, want.abs().max().item())


def test_baseline_config_1_runs_on_the_cpu_through_the_c_abi(built_lib):
    """BASELINE.json configs[0]: configs/monodetr.yaml on a CPU, batch_size 1, two KITTI-shaped synthetic images, one
    training iteration each -- plumbing.  The reference cannot run it (its operator raises on the CPU); here the C ABI's
    host entry points carry the operator (MDETR_MSDA_CPU / allow_cpu) and everything else is the same code as on the GPU.
    Pass = finite loss, and every trainable parameter outside the known-unused set receives a gradient."""
    import bench
    from monodetr_amd import msda_ext
    msda_ext.allow_cpu(True)
    try:
        step = bench.TrainStep(torch.device("cpu"), 1, "fp32", switches=())
        losses = []
        for sample in range(2):
            step.inputs = bench.synthetic_batch(1, 384, 1280, 1000 + sample, torch.device("cpu"))
            losses.append(float(step().detach()))
    finally:
        msda_ext.allow_cpu(False)
    assert all(l == l and abs(l) < 1e6 for l in losses), losses
    unused = ("label_enc", "sa_v_proj", "decoder.query_scale", "decoder.ref_point_head")
    for n, p in step.raw_model.named_parameters():
        if p.requires_grad and not any(u in n for u in unused):
            assert p.grad is not None and torch.isfinite(p.grad).all(), n


def test_conv3x3_launch_geometry_for_the_resnet_stages(built_lib):
    """mdetr_conv3x3_plan (host only): the tile shape / channel width csrc/conv3x3.hip takes at B = 8, 384 x 1280 -- chosen for ONE round of
    workgroups where the stage allows it (layer3: 8 x 16 tiles of 2 x 16 blocks at 64 channels = 480 workgroups on 512 places, no empty
    block at W = 80) and for no matrix instructions on empty columns (layer4, W = 40: 8-wide blocks).  100 WC + 10 GC + NB."""
    from monodetr_amd import _capi
    lib = _capi.lib()
    plan = lambda H, W, N, B=8: lib.mdetr_conv3x3_plan(B, H, W, N)          # noqa: E731
    assert plan(96, 320, 64) == 3212 and plan(48, 160, 128) == 3212          # exact fits: 1 x 32 blocks, 4 x 32 tiles, 64 channels
    assert plan(24, 80, 256) == 1612                                         # layer3 and the depth head
    assert plan(12, 40, 512) // 100 == 8 and plan(12, 40, 512) % 10 == 1     # layer4: 8-wide blocks, 32 channels (768 workgroups, one round)
    assert plan(0, 1, 32) < 0
