"""Pin the CPU oracle (oracle/msda_oracle.c) to the reference's own implementation.

Golden vectors were produced by tests/golden/make_golden.py from the reference's
ms_deform_attn_core_pytorch (ops/functions/ms_deform_attn_func.py:41-61) + autograd.
Tolerances: fp64 uses the reference's own allclose defaults (rtol 1e-5, atol 1e-8, ops/test.py:40)
and a much tighter 1e-12 absolute bound; fp32 uses rtol 1e-2 / atol 1e-3 (ops/test.py:56) and a
tighter 1e-6 absolute bound (values are O(1e-2)).
"""
import pytest
import torch

from conftest import load_golden, make_problem


def _fwd(oracle, g, dtype=None):
    cast = (lambda t: t.to(dtype)) if dtype is not None else (lambda t: t)
    return oracle.forward(cast(g["value"]), g["shapes"], g["level_start"], cast(g["loc"]), cast(g["attn"]))


def test_reference_test_problem_f64(oracle):
    g = load_golden("msda_ref_test_f64")
    out = _fwd(oracle, g)
    assert torch.allclose(out, g["out"])                         # the reference's own criterion
    assert (out - g["out"]).abs().max() < 1e-12


def test_reference_test_problem_f32(oracle):
    g = load_golden("msda_ref_test_f32")
    out = _fwd(oracle, g)
    assert torch.allclose(out, g["out"], rtol=1e-2, atol=1e-3)   # the reference's own criterion
    assert (out - g["out"]).abs().max() < 1e-8


@pytest.mark.parametrize("D", [30, 32, 64, 71])
def test_gradcheck_problems_match_reference_autograd(oracle, D):
    g = load_golden("msda_grad_d%d" % D)
    out = _fwd(oracle, g)
    assert (out - g["out"]).abs().max() < 1e-12
    gv, gl, ga = oracle.backward(g["value"], g["shapes"], g["level_start"], g["loc"], g["attn"], g["grad_out"])
    assert (gv - g["grad_value"]).abs().max() < 1e-12
    assert (gl - g["grad_loc"]).abs().max() < 1e-10
    assert (ga - g["grad_attn"]).abs().max() < 1e-12


def test_kitti_small_f64_and_f32(oracle):
    g = load_golden("msda_kitti_small")
    out64 = _fwd(oracle, g, torch.float64)
    assert (out64 - g["out_f64"]).abs().max() < 1e-12
    gv, gl, ga = oracle.backward(g["value"].double(), g["shapes"], g["level_start"], g["loc"].double(),
                                 g["attn"].double(), g["grad_out"].double())
    assert (gv - g["grad_value_f64"]).abs().max() < 1e-10
    assert (gl - g["grad_loc_f64"]).abs().max() < 1e-9
    assert (ga - g["grad_attn_f64"]).abs().max() < 1e-10

    out32 = _fwd(oracle, g)
    assert out32.dtype == torch.float32
    assert torch.allclose(out32, g["out_f32"], rtol=1e-2, atol=1e-3)
    assert (out32.double() - g["out_f64"]).abs().max() < 1e-6
    gv, gl, ga = oracle.backward(g["value"], g["shapes"], g["level_start"], g["loc"], g["attn"], g["grad_out"])
    assert (gv.double() - g["grad_value_f64"]).abs().max() < 1e-5
    assert (gl.double() - g["grad_loc_f64"]).abs().max() < 1e-4
    assert (ga.double() - g["grad_attn_f64"]).abs().max() < 1e-5


def test_border_forward_matches_reference(oracle):
    """Locations on/around every window edge (.cuh:288 window test, :56-78 corner bounds)."""
    g = load_golden("msda_border")
    out = _fwd(oracle, g)
    assert (out - g["out"]).abs().max() < 1e-12


def test_border_backward_matches_reference_off_the_edges(oracle):
    """grid_sample and the CUDA kernel differ in d/dloc only AT pixel coordinate -1 exactly
    (the kernel's strict `> -1` window, .cuh:288, zeroes the sample; grid_sample still
    differentiates the weight of the in-range corner).  Everywhere else they agree."""
    g = load_golden("msda_border")
    gv, gl, ga = oracle.backward(g["value"], g["shapes"], g["level_start"], g["loc"], g["attn"], g["grad_out"])
    assert (gv - g["grad_value"]).abs().max() < 1e-12
    assert (ga - g["grad_attn"]).abs().max() < 1e-12
    idx = oracle.indices(g["shapes"], g["loc"])
    inwin = idx[..., 0].bool()
    diff = (gl - g["grad_loc"]).abs().amax(-1)
    assert diff[inwin].max() < 1e-10
    # out-of-window samples: the kernel semantics give exact zeros (.cuh:365-374)
    assert gl[~inwin].abs().max() == 0 and ga[~inwin].abs().max() == 0


def test_indices_follow_kernel_formula(oracle):
    """floor(loc*size - 0.5) with the product rounded first (double literal, .cuh:285-286)."""
    p = make_problem(2, 4, 8, 33, [(7, 9), (3, 5)], 3, torch.float32, seed=11, lo=-0.3, hi=1.3)
    idx = oracle.indices(p["shapes"], p["loc"])
    H = p["shapes"][:, 0].float().view(1, 1, 1, -1, 1)
    W = p["shapes"][:, 1].float().view(1, 1, 1, -1, 1)
    h_im = p["loc"][..., 1] * H - 0.5        # torch fp32 ops round each step, like the kernel
    w_im = p["loc"][..., 0] * W - 0.5
    inwin = (h_im > -1) & (w_im > -1) & (h_im < H) & (w_im < W)
    assert torch.equal(idx[..., 0].bool(), inwin)
    assert torch.equal(idx[..., 1][inwin], torch.floor(h_im)[inwin].int())
    assert torch.equal(idx[..., 2][inwin], torch.floor(w_im)[inwin].int())


def test_oracle_matches_port_of_reference_cpu_path(oracle):
    """oracle/msda_torch_ref.py (the timed cpu_baseline 'port') agrees with the C oracle."""
    from oracle.msda_torch_ref import msda_grid_sample
    p = make_problem(2, 8, 32, 50, [(12, 40), (6, 20), (3, 10), (2, 5)], 4, torch.float64, seed=2, lo=-0.1, hi=1.1)
    a = oracle.forward(p["value"], p["shapes"], p["level_start"], p["loc"], p["attn"])
    b = msda_grid_sample(p["value"], p["shapes"], p["loc"], p["attn"])
    assert (a - b).abs().max() < 1e-12


def test_empty_query_set(oracle):
    p = make_problem(1, 2, 4, 0, [(3, 3)], 2, torch.float32)
    out = oracle.forward(p["value"], p["shapes"], p["level_start"], p["loc"], p["attn"])
    assert out.shape == (1, 0, 8)
    gv, gl, ga = oracle.backward(p["value"], p["shapes"], p["level_start"], p["loc"], p["attn"], p["grad_out"])
    assert gv.abs().max() == 0 and gl.numel() == 0 and ga.numel() == 0


# ---- the product's own ms_deform_attn_core_pytorch against the vectors recorded from the reference's ----
def test_product_core_pytorch_function_matches_the_recorded_reference_outputs():
    """monodetr_amd...ops.functions.ms_deform_attn_core_pytorch (mirror of ops/functions/ms_deform_attn_func.py:41-61, what
    the reference's ops/test.py:19 imports) reproduces the outputs -- and through autograd the gradients -- the reference's
    own function produced (tests/golden/make_golden.py), in fp64 and fp32."""
    from monodetr_amd.monodetr.ops.functions import MSDeformAttnFunction, ms_deform_attn_core_pytorch   # both names, as ops/test.py:19
    assert MSDeformAttnFunction is not None
    g = load_golden("msda_ref_test_f64")
    out = ms_deform_attn_core_pytorch(g["value"], g["shapes"], g["loc"], g["attn"])
    assert out.shape == g["out"].shape and (out - g["out"]).abs().max() < 1e-14
    for D in (30, 32, 64, 71):
        g = load_golden("msda_grad_d%d" % D)
        v, l, a = (g[k].clone().requires_grad_(True) for k in ("value", "loc", "attn"))
        out = ms_deform_attn_core_pytorch(v, g["shapes"], l, a)
        assert (out - g["out"]).abs().max() < 1e-14
        out.backward(g["grad_out"])
        assert (v.grad - g["grad_value"]).abs().max() < 1e-13
        assert (l.grad - g["grad_loc"]).abs().max() < 1e-12
        assert (a.grad - g["grad_attn"]).abs().max() < 1e-13
    g = load_golden("msda_kitti_small")
    out32 = ms_deform_attn_core_pytorch(g["value"], g["shapes"], g["loc"], g["attn"])
    assert out32.dtype == torch.float32 and (out32 - g["out_f32"]).abs().max() < 1e-7
