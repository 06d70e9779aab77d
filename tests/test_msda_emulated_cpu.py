"""The MSDA kernel sources (csrc/msda.hip, csrc/msda_tiled.hip) -- kernels, launchers and C-ABI entries, unmodified --
run on the HIP-on-CPU shim (tests/native_emul.py).

Two purposes.  (1) The fp32 / fp64 paths are GPU-validated: running them here against the oracle validates the
EMULATOR (fibers, wave barriers, DPP controls, LDS, the fixed-point tile scatter).  (2) With the emulator trusted,
the mixed-precision bf16 instantiations (mdetr_msda_forward_bf16 / backward_bf16), which have not met a GPU yet, are
checked against the fp32 kernels on the widened tensors -- the same comparison the pending GPU test makes."""
import ctypes

import pytest
import torch

import native_emul
from conftest import make_problem
from oracle import msda_oracle as oracle

SMALL = [(12, 40), (6, 20), (3, 10), (2, 5)]          # the KITTI pyramid / 4, S = 635
TINY = [(8, 24), (4, 12), (2, 6), (1, 3)]             # S = 255: keeps the tile-scatter emulation (1024-thread blocks) to seconds
ODD = [(1, 1), (2, 3), (1, 7), (5, 1)]


def _check(rc):
    assert rc == 0, ctypes.string_at(native_emul.lib().mdetr_last_error())


def fwd(p, code=0):
    L = native_emul.lib()
    B, S, M, D = p["value"].shape
    Lq, Lv, P = p["loc"].shape[1], p["loc"].shape[3], p["loc"].shape[4]
    out = torch.empty(B, Lq, M * D, dtype=p["value"].dtype)
    _check(L.mdetr_msda_forward(code, p["value"].data_ptr(), p["shapes"].data_ptr(), p["level_start"].data_ptr(),
                                p["loc"].data_ptr(), p["attn"].data_ptr(), out.data_ptr(), B, S, M, D, Lv, Lq, P, 0, None))
    return out


def bwd(p, code=0, tiled=False):
    L = native_emul.lib()
    B, S, M, D = p["value"].shape
    Lq, Lv, P = p["loc"].shape[1], p["loc"].shape[3], p["loc"].shape[4]
    gv, gl, ga = torch.full_like(p["value"], 9.0), torch.empty_like(p["loc"]), torch.empty_like(p["attn"])
    common = (code, p["value"].data_ptr(), p["shapes"].data_ptr(), p["level_start"].data_ptr(), p["loc"].data_ptr(),
              p["attn"].data_ptr(), p["grad_out"].data_ptr(), gv.data_ptr(), gl.data_ptr(), ga.data_ptr(), B, S, M, D, Lv, Lq, P)
    if tiled:
        n = L.mdetr_msda_backward_workspace_bytes(code, p["shapes"].data_ptr(), p["level_start"].data_ptr(), B, S, M, D, Lv, Lq, P)
        assert n > 0
        ws = torch.empty(n, dtype=torch.uint8)
        _check(L.mdetr_msda_backward_ex(*common, p["shapes"].data_ptr(), p["level_start"].data_ptr(), ws.data_ptr(), n, 0, None))
    else:
        _check(L.mdetr_msda_backward(*common, 0, None))
    return gv, gl, ga


def close(a, b, tol):
    return (a.double() - b.double()).abs().max().item() <= tol * max(1.0, b.double().abs().max().item())


# ---- (1) the emulator reproduces the GPU-validated paths -----------------------------------------------------------
@pytest.mark.parametrize("B,M,Lq,shapes,P,lo,hi", [
    (1, 8, None, TINY, 4, 0.0, 1.0),           # fast path (record kernel), self-attention: atomics and tile scatter
    (1, 8, 50, SMALL, 4, -0.2, 1.2),           # decoder-like, samples outside the maps
    (2, 3, 37, ODD, 4, -0.1, 1.1),             # degenerate maps, ragged pair count (111 pairs)
    (1, 2, 9, [(6, 4), (3, 2)], 2, 0.0, 1.0),  # L = P = 2: the run-time-shape fast path
])
def test_emulated_fp32_kernels_match_the_oracle(B, M, Lq, shapes, P, lo, hi):
    S = sum(h * w for h, w in shapes)
    p = make_problem(B, M, 32, S if Lq is None else Lq, shapes, P, torch.float32, seed=5, lo=lo, hi=hi)
    ref = oracle.forward(p["value"].double(), p["shapes"], p["level_start"], p["loc"].double(), p["attn"].double())
    assert close(fwd(p), ref, 1e-6)
    rv, rl, ra = oracle.backward(p["value"].double(), p["shapes"], p["level_start"], p["loc"].double(), p["attn"].double(),
                                 p["grad_out"].double())
    for tiled in ([False, True] if Lq is None else [False]):
        gv, gl, ga = bwd(p, tiled=tiled)
        assert close(gv, rv, 1e-5) and close(gl, rl, 1e-4) and close(ga, ra, 1e-4), tiled


def test_emulated_generic_kernel_fp64_and_odd_channel_count():
    p = make_problem(1, 2, 30, 7, [(6, 4), (3, 2)], 3, torch.float64, seed=2)
    ref = oracle.forward(p["value"], p["shapes"], p["level_start"], p["loc"], p["attn"])
    assert close(fwd(p, code=1), ref, 1e-12)
    rv, rl, ra = oracle.backward(p["value"], p["shapes"], p["level_start"], p["loc"], p["attn"], p["grad_out"])
    gv, gl, ga = bwd(p, code=1)
    assert close(gv, rv, 1e-12) and close(gl, rl, 1e-12) and close(ga, ra, 1e-12)


def test_emulated_gather_indices_are_bit_exact():
    L = native_emul.lib()
    p = make_problem(2, 4, 32, 21, SMALL, 4, torch.float32, seed=8, lo=-0.3, hi=1.3)
    idx = torch.empty(2, 21, 4, 4, 4, 4, dtype=torch.int32)
    _check(L.mdetr_msda_indices(0, p["shapes"].data_ptr(), p["loc"].data_ptr(), idx.data_ptr(), 2, 4, 4, 21, 4, 0, None))
    assert torch.equal(idx, oracle.indices(p["shapes"], p["loc"]))


# ---- (2) the mixed-precision instantiations ------------------------------------------------------------------------
@pytest.mark.parametrize("B,M,Lq,shapes,lo,hi", [
    (1, 8, None, TINY, -0.15, 1.15),           # self-attention: bf16 gather kernel + bf16 tile scatter
    (2, 8, 50, SMALL, -0.15, 1.15),            # cross attention: atomics variant
    (1, 1, 3, ODD, 0.0, 1.0),                  # ragged tail, degenerate maps
])
def test_emulated_bf16_kernels_match_the_fp32_kernels_on_widened_tensors(B, M, Lq, shapes, lo, hi):
    L = native_emul.lib()
    S = sum(h * w for h, w in shapes)
    Lq_ = S if Lq is None else Lq
    p = make_problem(B, M, 32, Lq_, shapes, 4, torch.float32, seed=3, lo=lo, hi=hi)
    vb, gb = (p["value"] * 100).to(torch.bfloat16), p["grad_out"].to(torch.bfloat16)
    wide = dict(p, value=vb.float(), grad_out=gb.float())
    out = torch.empty(B, Lq_, M * 32, dtype=torch.bfloat16)
    _check(L.mdetr_msda_forward_bf16(vb.data_ptr(), p["shapes"].data_ptr(), p["level_start"].data_ptr(), p["loc"].data_ptr(),
                                     p["attn"].data_ptr(), out.data_ptr(), B, S, M, 32, 4, Lq_, 4, 0, None))
    ref = fwd(wide)
    assert torch.equal(out, ref.to(torch.bfloat16))                    # same fp32 arithmetic, then one rounding
    gv, gl, ga = torch.full((B, S, M, 32), 9.0), torch.empty_like(p["loc"]), torch.empty_like(p["attn"])
    n = L.mdetr_msda_backward_workspace_bytes(0, p["shapes"].data_ptr(), p["level_start"].data_ptr(), B, S, M, 32, 4, Lq_, 4) if Lq is None else 0
    ws = torch.empty(max(n, 1), dtype=torch.uint8)
    _check(L.mdetr_msda_backward_bf16(vb.data_ptr(), p["shapes"].data_ptr(), p["level_start"].data_ptr(), p["loc"].data_ptr(),
                                      p["attn"].data_ptr(), gb.data_ptr(), gv.data_ptr(), gl.data_ptr(), ga.data_ptr(),
                                      B, S, M, 32, 4, Lq_, 4, p["shapes"].data_ptr() if n else None,
                                      p["level_start"].data_ptr() if n else None, ws.data_ptr() if n else None, n, 0, None))
    rv, rl, ra = bwd(wide, tiled=bool(n))
    assert close(gv, rv, 1e-6) and close(gl, rl, 1e-6) and close(ga, ra, 1e-6)


class _EmulModule:
    """The extension-module object as MSDeformAttnFunction sees it, backed by the emulated C ABI (CPU tensors)."""

    @staticmethod
    def bf16_supported(value, loc):
        return value.dtype == torch.bfloat16 and value.shape[3] == 32 and loc.shape[3] == 4 and loc.shape[4] == 4

    @staticmethod
    def ms_deform_attn_forward(value, shapes, start, loc, attn, im2col_step):
        return fwd(dict(value=value, shapes=shapes, level_start=start, loc=loc, attn=attn))

    @staticmethod
    def ms_deform_attn_backward(value, shapes, start, loc, attn, grad_out, im2col_step):
        return list(bwd(dict(value=value, shapes=shapes, level_start=start, loc=loc, attn=attn, grad_out=grad_out)))

    @staticmethod
    def ms_deform_attn_forward_bf16(value, shapes, start, loc, attn):
        L = native_emul.lib()
        B, S, M, D = value.shape
        Lq = loc.shape[1]
        out = torch.empty(B, Lq, M * D, dtype=torch.bfloat16)
        _check(L.mdetr_msda_forward_bf16(value.data_ptr(), shapes.data_ptr(), start.data_ptr(), loc.data_ptr(), attn.data_ptr(),
                                         out.data_ptr(), B, S, M, D, 4, Lq, 4, 0, None))
        return out

    @staticmethod
    def ms_deform_attn_backward_bf16(value, shapes, start, loc, attn, grad_out):
        L = native_emul.lib()
        B, S, M, D = value.shape
        Lq = loc.shape[1]
        gv, gl, ga = torch.empty(B, S, M, D), torch.empty_like(loc), torch.empty_like(attn)
        _check(L.mdetr_msda_backward_bf16(value.data_ptr(), shapes.data_ptr(), start.data_ptr(), loc.data_ptr(), attn.data_ptr(),
                                          grad_out.data_ptr(), gv.data_ptr(), gl.data_ptr(), ga.data_ptr(), B, S, M, D, 4, Lq, 4,
                                          None, None, None, 0, 0, None))
        return [gv, gl, ga]


def test_autograd_function_with_native_bf16_matches_the_widening_path(monkeypatch):
    """MSDeformAttnFunction as a bf16 model calls it (bf16 value, locations and weights): MDETR_MSDA_BF16 on vs off."""
    from monodetr_amd.monodetr.ops.functions import ms_deform_attn_func as F
    monkeypatch.setattr(F, "MSDA", _EmulModule)
    p = make_problem(2, 8, 32, 50, SMALL, 4, torch.float32, seed=6, lo=-0.1, hi=1.1)
    res = {}
    for native in (False, True):
        monkeypatch.setattr(F, "_NATIVE_BF16", native)
        v = (p["value"] * 100).to(torch.bfloat16).requires_grad_(True)
        l = p["loc"].to(torch.bfloat16).requires_grad_(True)
        a = p["attn"].to(torch.bfloat16).requires_grad_(True)
        out = F.MSDeformAttnFunction.apply(v, p["shapes"], p["level_start"], l, a, 64)
        out.backward(p["grad_out"].to(torch.bfloat16))
        res[native] = (out.detach(), v.grad, l.grad, a.grad)
    for x, y in zip(res[False], res[True]):
        assert x.dtype == y.dtype == torch.bfloat16
        assert (x.float() - y.float()).abs().max() <= 2 ** -7 * max(1.0, x.float().abs().max().item())
