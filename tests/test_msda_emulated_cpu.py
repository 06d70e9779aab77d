"""The MSDA kernel sources (csrc/msda.hip, csrc/msda_tiled.hip, csrc/msda_fused.hip) -- kernels, launchers and C-ABI
entries, unmodified -- run on the HIP-on-CPU shim (tests/native_emul.py).

Two purposes.  (1) The fp32 / fp64 paths are GPU-validated: running them here against the oracle validates the
EMULATOR (fibers, wave barriers, DPP controls, LDS, the fixed-point tile scatter).  (2) With the emulator trusted,
the mixed-precision bf16 instantiations (mdetr_msda_forward_bf16 / backward_bf16), which have not met a GPU yet, are
checked against the fp32 kernels on the widened tensors -- the same comparison the pending GPU test makes."""
import ctypes

import pytest
import torch

import native_emul
from conftest import make_problem, tune
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


def bwd(p, code=0, tiled=False, path=None, ws=None):
    """path: None = C ABI without workspace (global atomics); "tiled" / "fused" / "atomic" = mdetr_msda_backward_ex with the
    grad_value strategy selected through MDETR_TUNE="msda_bwd=..." (`tiled=True` is the round-1 spelling of path="tiled")."""
    import os
    L = native_emul.lib()
    path = "tiled" if tiled and path is None else path
    B, S, M, D = p["value"].shape
    Lq, Lv, P = p["loc"].shape[1], p["loc"].shape[3], p["loc"].shape[4]
    gv, gl, ga = torch.full_like(p["value"], 9.0), torch.full_like(p["loc"], 7.0), torch.full_like(p["attn"], 5.0)
    common = (code, p["value"].data_ptr(), p["shapes"].data_ptr(), p["level_start"].data_ptr(), p["loc"].data_ptr(),
              p["attn"].data_ptr(), p["grad_out"].data_ptr(), gv.data_ptr(), gl.data_ptr(), ga.data_ptr(), B, S, M, D, Lv, Lq, P)
    if path is not None:
        saved = os.environ.get("MDETR_TUNE")
        os.environ["MDETR_TUNE"] = ",".join(x for x in ("msda_bwd=" + path, saved) if x)
        try:
            n = L.mdetr_msda_backward_workspace_bytes(code, p["shapes"].data_ptr(), p["level_start"].data_ptr(), B, S, M, D, Lv, Lq, P)
            assert n > 0
            if ws is None or ws.numel() < n:
                ws = torch.randint(0, 255, (n,), dtype=torch.uint8)         # a fresh workspace holds anything
            _check(L.mdetr_msda_backward_ex(*common, p["shapes"].data_ptr(), p["level_start"].data_ptr(), ws.data_ptr(), ws.numel(), 0, None))
        finally:
            if saved is None:
                del os.environ["MDETR_TUNE"]
            else:
                os.environ["MDETR_TUNE"] = saved
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
    paths = [None, "fused"] + (["tiled"] if Lq is None else [])
    if 32 % (len(shapes) * P):
        paths = [None]                                      # the one-pass kernel's pre-pass reads attn in 16-byte pieces
    for path in paths:
        gv, gl, ga = bwd(p, path=path)
        assert close(gv, rv, 1e-5) and close(gl, rl, 1e-4) and close(ga, ra, 1e-4), path


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
    n = L.mdetr_msda_backward_workspace_bytes(0, p["shapes"].data_ptr(), p["level_start"].data_ptr(), B, S, M, 32, 4, Lq_, 4)
    ws = torch.randint(0, 255, (max(n, 1),), dtype=torch.uint8)
    _check(L.mdetr_msda_backward_bf16(vb.data_ptr(), p["shapes"].data_ptr(), p["level_start"].data_ptr(), p["loc"].data_ptr(),
                                      p["attn"].data_ptr(), gb.data_ptr(), gv.data_ptr(), gl.data_ptr(), ga.data_ptr(),
                                      B, S, M, 32, 4, Lq_, 4, p["shapes"].data_ptr() if n else None,
                                      p["level_start"].data_ptr() if n else None, ws.data_ptr() if n else None, n, 0, None))
    rv, rl, ra = bwd(wide, path="fused" if n else None)
    assert close(gv, rv, 1e-6) and close(gl, rl, 1e-6) and close(ga, ra, 1e-6)
    if n:                                                  # integer accumulation: the two element types agree bit for bit
        assert torch.equal(gv, rv)


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


# ---- (3) the one-pass backward (csrc/msda_fused.hip) ------------------------------------------------------------------
def _fused_env(monkeypatch, th, tw, reach, whole, chunks):
    tune(monkeypatch, msda_tile_h=th, msda_tile_w=tw, msda_reach=reach, msda_whole_level_cells=whole, msda_chunks=chunks)


def _oracle_bwd(p):
    return oracle.backward(p["value"].double(), p["shapes"], p["level_start"], p["loc"].double(), p["attn"].double(), p["grad_out"].double())


@pytest.mark.parametrize("th,tw,reach,whole,chunks", [(4, 8, 2, 30, 3), (3, 5, 1, 0, 1), (16, 32, 4, 512, 8)])
@pytest.mark.parametrize("B,M,Lq,shapes,lo,hi", [
    (1, 2, None, TINY, 0.0, 1.0),              # self-attention, uniform locations: most corners leave every block's reach
    (2, 2, None, SMALL, -0.3, 1.3),            # + samples outside the maps
    (2, 2, 50, SMALL, -0.2, 1.2),              # cross-attention: every block scans every query
    (1, 1, 3, ODD, 0.0, 1.0),                  # degenerate maps
])
def test_fused_backward_matches_the_oracle_for_every_plan(monkeypatch, th, tw, reach, whole, chunks, B, M, Lq, shapes, lo, hi):
    """Core tiles + candidate queries (mode 0), whole-level chunks (mode 1), scan-all tiles (mode 2), the `far` buffer, with
    tile geometries that split the small test pyramids the way the default plan splits the 48x160 one."""
    _fused_env(monkeypatch, th, tw, reach, whole, chunks)
    S = sum(h * w for h, w in shapes)
    p = make_problem(B, M, 32, S if Lq is None else Lq, shapes, 4, torch.float32, seed=3, lo=lo, hi=hi)
    gv, gl, ga = bwd(p, path="fused")
    rv, rl, ra = _oracle_bwd(p)
    assert close(gv, rv, 1e-5) and close(gl, rl, 1e-6) and close(ga, ra, 1e-6)


def test_fused_backward_near_samples_stay_in_lds_and_are_deterministic(monkeypatch):
    """Offsets of at most `reach` cells (the model's initial star pattern, ops/modules/ms_deform_attn.py:107-114): nothing
    takes the global-atomic route (the `far` flag in the workspace header stays 0), and the result is independent of the
    tile geometry bit for bit (integer accumulation), which the fp32-atomic paths are not."""
    S = sum(h * w for h, w in SMALL)
    p = make_problem(2, 2, 32, S, SMALL, 4, torch.float32, seed=4)
    ref = torch.cat([torch.stack(torch.meshgrid((torch.arange(h) + 0.5) / h, (torch.arange(w) + 0.5) / w, indexing="ij"), -1).reshape(-1, 2)
                     for h, w in SMALL])[:, [1, 0]]                         # (x, y) of every query
    g = torch.Generator().manual_seed(1)
    off = (torch.rand(2, S, 2, 4, 4, 2, generator=g) * 2 - 1) * 1.4        # |offset| + 1.5 <= reach (3 cells) in each level's own units
    off[:, :, :, :, 0] = off[:, :, :, :, 0].round()                        # some exactly integer: corners of weight 0 are skipped
    sizes = torch.tensor([[w, h] for h, w in SMALL], dtype=torch.float32).view(1, 1, 1, 4, 1, 2)
    p["loc"] = (ref.view(1, S, 1, 1, 1, 2) + off / sizes).contiguous()
    # (fp32 oracle: with pixel coordinates ON integers the floor is decided by the fp32 rounding of loc * size, .cuh:285-286)
    rv, rl, ra = oracle.backward(p["value"], p["shapes"], p["level_start"], p["loc"], p["attn"], p["grad_out"])
    outs = []
    for th, tw in ((4, 8), (3, 7)):
        _fused_env(monkeypatch, th, tw, 3, 30, 3)
        ws = torch.randint(0, 255, (1 << 22,), dtype=torch.uint8)
        gv, gl, ga = bwd(p, path="fused", ws=ws)
        assert ws[8:12].view(torch.int32).item() == 0                       # Header.far
        assert close(gv, rv, 1e-5) and close(gl, rl, 1e-5) and close(ga, ra, 1e-5)
        outs.append(gv)
    assert torch.equal(outs[0], outs[1])


def test_fused_backward_repeats_a_pass_when_a_cell_draws_too_many_contributions(monkeypatch):
    """All 255 x 4 samples of a level on ONE footprint: > 511 contributions per cell, the 32-bit fields would overflow;
    the block notices from its counts and repeats the pass at half the resolution."""
    _fused_env(monkeypatch, 4, 8, 2, 30, 1)
    S = sum(h * w for h, w in TINY)
    p = make_problem(1, 2, 32, S, TINY, 4, torch.float32, seed=9)
    p["loc"] = torch.full_like(p["loc"], 0.5) + 0.01 * torch.rand(p["loc"].shape, generator=torch.Generator().manual_seed(2))
    gv, gl, ga = bwd(p, path="fused")
    rv, rl, ra = _oracle_bwd(p)
    assert close(gv, rv, 2e-5) and close(gl, rl, 1e-6) and close(ga, ra, 1e-6)


def test_fused_backward_non_finite_gradients_and_workspace_reuse(monkeypatch):
    """inf / NaN in grad_out: no fixed-point scale exists, every corner takes the global-atomic route and propagates like the
    reference's atomics; the same workspace then serves a finite call of another size and the first size again (the `far`
    buffer's all-zero invariant is re-established whenever the layout changes)."""
    _fused_env(monkeypatch, 4, 8, 2, 30, 3)
    S = sum(h * w for h, w in TINY)
    ws = torch.randint(0, 255, (1 << 22,), dtype=torch.uint8)
    p = make_problem(1, 2, 32, S, TINY, 4, torch.float32, seed=5, lo=0.1, hi=0.9)
    bad = dict(p, grad_out=p["grad_out"].clone())
    bad["grad_out"][0, 7, 3] = float("inf")
    bad["grad_out"][0, 100, 40] = float("nan")
    gv, gl, ga = bwd(bad, path="fused", ws=ws)
    rv, rl, ra = oracle.backward(bad["value"], bad["shapes"], bad["level_start"], bad["loc"], bad["attn"], bad["grad_out"])
    fin = torch.isfinite(rv)
    assert torch.equal(torch.isfinite(gv), fin) and (~fin).any()
    assert (gv[fin] - rv[fin]).abs().max() < 1e-5 * max(1.0, rv[fin].abs().max().item())
    fl = torch.isfinite(rl)
    assert torch.equal(torch.isfinite(gl), fl) and (gl[fl] - rl[fl]).abs().max() < 1e-5 * max(1.0, rl[fl].abs().max().item())
    for B, Lq in ((2, 17), (1, S), (2, 17)):
        q = make_problem(B, 2, 32, Lq, TINY, 4, torch.float32, seed=B + Lq, lo=-0.2, hi=1.2)
        gv, gl, ga = bwd(q, path="fused", ws=ws)
        rv, rl, ra = _oracle_bwd(q)
        assert close(gv, rv, 1e-5) and close(gl, rl, 1e-6) and close(ga, ra, 1e-6), (B, Lq)


def _bwd_bf16(p, vb, gb):
    L = native_emul.lib()
    B, S, M, _ = p["value"].shape
    Lq = p["loc"].shape[1]
    gv, gl, ga = torch.full((B, S, M, 32), 9.0), torch.full_like(p["loc"], 7.0), torch.full_like(p["attn"], 5.0)
    n = L.mdetr_msda_backward_workspace_bytes(0, p["shapes"].data_ptr(), p["level_start"].data_ptr(), B, S, M, 32, 4, Lq, 4)
    assert n > 0
    ws = torch.randint(0, 255, (n,), dtype=torch.uint8)
    _check(L.mdetr_msda_backward_bf16(vb.data_ptr(), p["shapes"].data_ptr(), p["level_start"].data_ptr(), p["loc"].data_ptr(),
                                      p["attn"].data_ptr(), gb.data_ptr(), gv.data_ptr(), gl.data_ptr(), ga.data_ptr(),
                                      B, S, M, 32, 4, Lq, 4, p["shapes"].data_ptr(), p["level_start"].data_ptr(), ws.data_ptr(), n, 0, None))
    return gv, gl, ga


@pytest.mark.parametrize("threads,lps,groups", [(512, 8, 2), (1024, 4, 2), (768, 4, 4), (1024, 2, 2)])      # (each instantiation family once)
@pytest.mark.parametrize("th,tw,reach,whole,chunks", [(4, 8, 2, 30, 3)])
def test_fused_backward_kernel_variants(monkeypatch, threads, lps, groups, th, tw, reach, whole, chunks):
    """The launch variants behind MDETR_TUNE msda_threads / msda_lps / msda_groups (8-, 12- or 16-wave workgroups; 8 lanes x 4 channels or 4 lanes
    x 8 channels per sample, the latter with the packed-bf16 dot products): the bf16 form against the fp32 form on the same
    (widened) tensors and against the oracle, on plans that use tiles with candidates, whole-level chunks, scan-all tiles and
    the `far` buffer (whose fp32 atomics make grad_value depend on the order of the waves: not bit-equal across variants here;
    the integer-accumulated part is, see test_emulated_bf16_kernels_match_the_fp32_kernels_on_widened_tensors)."""
    _fused_env(monkeypatch, th, tw, reach, whole, chunks)
    tune(monkeypatch, msda_threads=str(threads))
    tune(monkeypatch, msda_lps=str(lps))
    tune(monkeypatch, msda_groups=str(groups))
    S = sum(h * w for h, w in SMALL)
    for Lq in (S, 50):
        p = make_problem(1, 2, 32, Lq, SMALL, 4, torch.float32, seed=3, lo=-0.2, hi=1.2)
        vb, gb = (p["value"] * 100).to(torch.bfloat16), p["grad_out"].to(torch.bfloat16)
        wide = dict(p, value=vb.float(), grad_out=gb.float())
        gv, gl, ga = _bwd_bf16(p, vb, gb)
        fv, fl, fa = bwd(wide, path="fused")
        assert close(gv, fv, 1e-6) and close(gl, fl, 1e-6) and close(ga, fa, 1e-6)
        rv, rl, ra = _oracle_bwd(wide)
        assert close(gv, rv, 1e-5) and close(gl, rl, 1e-6) and close(ga, ra, 1e-6)


@pytest.mark.parametrize("lps", [4, 2])
def test_fused_backward_bf16_lane_layouts_agree_bit_for_bit_on_near_samples(monkeypatch, lps):
    """Near samples only (no fp32 atomics anywhere): tiles with candidate queries, bf16 operands, either lane layout and
    16-wave workgroups -- grad_value equals the fp32 form's on the widened tensors bit for bit."""
    S = sum(h * w for h, w in SMALL)
    p = make_problem(2, 2, 32, S, SMALL, 4, torch.float32, seed=11)
    ref = torch.cat([torch.stack(torch.meshgrid((torch.arange(h) + 0.5) / h, (torch.arange(w) + 0.5) / w, indexing="ij"), -1).reshape(-1, 2)
                     for h, w in SMALL])[:, [1, 0]]
    off = (torch.rand(2, S, 2, 4, 4, 2, generator=torch.Generator().manual_seed(2)) * 2 - 1) * 1.4
    sizes = torch.tensor([[w, h] for h, w in SMALL], dtype=torch.float32).view(1, 1, 1, 4, 1, 2)
    p["loc"] = (ref.view(1, S, 1, 1, 1, 2) + off / sizes).contiguous()
    vb, gb = (p["value"] * 100).to(torch.bfloat16), p["grad_out"].to(torch.bfloat16)
    wide = dict(p, value=vb.float(), grad_out=gb.float())
    _fused_env(monkeypatch, 4, 8, 3, 30, 3)
    tune(monkeypatch, msda_lps=str(lps))
    gv, gl, ga = _bwd_bf16(p, vb, gb)
    fv, fl, fa = bwd(wide, path="fused")
    assert torch.equal(gv, fv) and close(gl, fl, 1e-6) and close(ga, fa, 1e-6)


def test_fused_backward_fixed_point_error_bound_with_a_wide_dynamic_range(monkeypatch):
    """grad_value is accumulated in 32-bit fixed point at ONE power-of-two scale per call (csrc/msda_fused.hip): a contribution
    w * attn * g is rounded to a multiple of q = 2^(e - 22), 2^e >= max|grad_out| * max|attn| (RNE, unbiased).  The absolute error
    of a cell is therefore at most n q / 2 for its n contributions -- independent of the cell's own magnitude -- where the
    reference's fp32 atomics lose ~2^-24 of the running sum per term.  With gradients spanning six decades across the queries
    the small tokens' entries keep that ABSOLUTE bound (the documented trade: ADVICE round 2); grad_loc / grad_attn are computed
    in floating point and keep their relative accuracy."""
    _fused_env(monkeypatch, 4, 8, 3, 30, 3)
    S = sum(h * w for h, w in SMALL)
    p = make_problem(2, 2, 32, S, SMALL, 4, torch.float32, seed=21)
    g = torch.Generator().manual_seed(5)
    decades = torch.pow(10.0, -6.0 * torch.rand(2, S, 1, generator=g))        # per query: 1 ... 1e-6
    p["grad_out"] = p["grad_out"] * decades
    gv, gl, ga = bwd(p, path="fused")
    rv, rl, ra = _oracle_bwd(p)
    mx = float(p["grad_out"].abs().max() * p["attn"].abs().max())
    import math
    q = 2.0 ** (math.frexp(mx)[1] - 22)
    # contributions per cell: at most every sample of the level that can reach it; here bounded by the level's sample count
    n_max = 2 * S * 4
    err = (gv.double() - rv).abs().max().item()
    assert err <= 0.5 * q * math.sqrt(n_max) * 8, (err, q)               # random-walk accumulation of +-q/2 roundings, 8 sigma
    assert err <= 1e-5 * rv.abs().max().item()                             # and tiny against the large entries
    assert close(gl, rl, 1e-6) and close(ga, ra, 1e-6)
