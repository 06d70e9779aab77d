"""csrc/sgemm.hip on the GPU: the grouped fp32 products at the shapes of the prediction heads (B x 550 = 4 400 rows), against
float64; a decoder level's heads (monodetr/heads.py) against the modules; the training step with and without the family."""
import copy

import pytest
import torch
from torch import nn

pytestmark = pytest.mark.gpu


def _close(got, ref, absref, K):
    """|got - ref| <= 2 K 2^-24 sum|a||b| + one output rounding (fp32 accumulation, nothing narrower)."""
    bound = 2.0 * K * 2.0 ** -24 * absref + (2.0 ** -8 * ref.abs() if got.dtype == torch.bfloat16 else 2.0 ** -23 * ref.abs()) + 1e-30
    bad = (got.double() - ref).abs() > bound
    assert not bad.any(), (int(bad.sum()), float(((got.double() - ref).abs() / bound).max()))


def test_sgemm_nt_nn_tn_at_the_heads_shapes_vs_float64():
    from monodetr_amd import sgemm_ext as ext
    torch.manual_seed(0)
    dev = "cuda"
    T = 4400
    x = torch.randn(T, 256, device=dev).to(torch.bfloat16)
    w1 = [torch.randn(256, 256, device=dev) * 0.06 for _ in range(4)]
    b1 = [torch.randn(256, device=dev) for _ in range(4)]
    wc, bc = torch.randn(3, 256, device=dev), torch.randn(3, device=dev)
    h1 = torch.empty(T, 1024, device=dev)
    logits = torch.empty(T, 3, device=dev)
    ext.grouped(ext.NT, [ext.Problem([(x, w)], h1[:, 256 * i:256 * (i + 1)], bias=b, relu_cols=True) for i, (w, b) in enumerate(zip(w1, b1))]
                + [ext.Problem([(x, wc)], logits, bias=bc)])
    xd = x.double()
    for i in range(4):
        ref = (xd @ w1[i].double().t() + b1[i].double()).clamp(min=0)
        _close(h1[:, 256 * i:256 * (i + 1)], ref, xd.abs() @ w1[i].double().abs().t() + b1[i].double().abs(), 256)
    _close(logits, xd @ wc.double().t() + bc.double(), xd.abs() @ wc.double().abs().t() + bc.double().abs(), 256)
    # input gradient over a contraction split across five tensors, the residual-path gradient and the bf16 rounding inside
    dh1 = torch.randn(T, 1024, device=dev)
    dcls = torch.randn(T, 3, device=dev)
    skip = torch.randn(T, 256, device=dev).to(torch.bfloat16)
    dx = torch.empty(T, 256, device=dev, dtype=torch.bfloat16)
    ext.grouped(ext.NN, [ext.Problem([(dh1[:, 256 * i:256 * (i + 1)], w1[i]) for i in range(4)] + [(dcls, wc)], dx, res=skip)])
    ref = sum(dh1[:, 256 * i:256 * (i + 1)].double() @ w1[i].double() for i in range(4)) + dcls.double() @ wc.double() + skip.double()
    absref = sum(dh1[:, 256 * i:256 * (i + 1)].double().abs() @ w1[i].double().abs() for i in range(4)) + dcls.double().abs() @ wc.double().abs() + skip.double().abs()
    _close(dx, ref, absref, 1027)
    # masked input gradient with a 6-long contraction
    g6, w6, saved = torch.randn(T, 6, device=dev), torch.randn(6, 256, device=dev), torch.randn(T, 256, device=dev)
    dh2 = torch.empty(T, 256, device=dev)
    ext.grouped(ext.NN, [ext.Problem([(g6, w6)], dh2, mask=saved)])
    _close(dh2, (g6.double() @ w6.double()) * (saved > 0), g6.double().abs() @ w6.double().abs(), 6)
    # weight gradients over 4 400 rows with their column sums
    dws = [torch.empty(256, 256, device=dev) for _ in range(4)] + [torch.empty(3, 256, device=dev), torch.empty(6, 256, device=dev)]
    dbs = [torch.empty(256, device=dev) for _ in range(4)] + [torch.empty(3, device=dev), torch.empty(6, device=dev)]
    ops = [(dh1[:, 256 * i:256 * (i + 1)], x) for i in range(4)] + [(dcls, x), (g6, saved)]
    ext.grouped(ext.TN, [ext.Problem([op], dw, colsum=db) for op, dw, db in zip(ops, dws, dbs)])
    for (a, b), dw, db in zip(ops, dws, dbs):
        _close(dw, a.double().t() @ b.double(), a.double().abs().t() @ b.double().abs(), T)
        _close(db, a.double().sum(0), a.double().abs().sum(0), T)
    # twice the same launch: bit-identical (fixed summation order, no atomics)
    again = [torch.empty_like(d) for d in dws]
    ext.grouped(ext.TN, [ext.Problem([op], dw) for op, dw in zip(ops, again)])
    assert all(torch.equal(a, b) for a, b in zip(dws, again))


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_heads_level_matches_the_modules_on_the_gpu(dtype):
    from monodetr_amd.monodetr import heads as H
    from monodetr_amd.monodetr.depthaware_transformer import MLP
    torch.manual_seed(5)
    mods = [MLP(256, 256, 6, 3), MLP(256, 256, 3, 2), MLP(256, 256, 2, 2), MLP(256, 256, 24, 2), nn.Linear(256, 3)]
    mods = [m.cuda() for m in mods]
    ref = [copy.deepcopy(m).double() for m in mods]
    B, Q = 8, 550
    x = torch.randn(B, Q, 256, device="cuda").to(dtype).requires_grad_(True)
    # Rows with a hidden pre-activation within 2e-5 of zero get NO upstream gradient: the ReLU's derivative jumps there, an fp32
    # evaluation may sit on the other side of the jump than float64 (4.5 M decisions per level: a handful do), and the comparison
    # would measure that coin toss instead of the kernels (~2 % of the rows; the forward outputs are compared on every row)
    with torch.no_grad():
        xd0 = x.detach().double()
        pre = [m.layers[0](xd0) for m in ref[:4]]
        near = torch.stack([p.abs().amin(-1) for p in pre] + [ref[0].layers[1](pre[0].clamp(min=0)).abs().amin(-1)]).amin(0) < 2e-5
        keep = (~near).float().unsqueeze(-1)
    assert 0.0 < float(near.float().mean()) < 0.2
    was, H.ENABLED = H.ENABLED, True
    try:
        out = H.heads_level(x, *mods)
        assert out is not None
        delta, size, depth, angle, logits, xs = out
        gs = [torch.randn_like(t) * keep for t in (delta, size, depth, angle, logits)]
        skip_w = torch.randn(B, Q, 256, device="cuda")
        (sum((t * g).sum() for t, g in zip((delta, size, depth, angle, logits), gs)) + (xs.float() * skip_w).sum()).backward()
    finally:
        H.ENABLED = was
    xd = x.detach().double().requires_grad_(True)
    outs = [m(xd) for m in ref]
    (sum((t * g.double()).sum() for t, g in zip(outs, gs)) + (xd * skip_w.double()).sum()).backward()
    for got, want in zip((delta, size, depth, angle, logits), outs):
        assert (got.detach().double() - want.detach()).abs().max() <= 1e-5 * max(1.0, float(want.abs().max()))
    gx_tol = 2.0 ** -7 if dtype == torch.bfloat16 else 1e-5
    assert (x.grad.double() - xd.grad).abs().max() <= gx_tol * float(xd.grad.abs().max())
    for m, r in zip(mods, ref):
        for (n, p), (_, q) in zip(m.named_parameters(), r.named_parameters()):
            assert (p.grad.double() - q.grad).abs().max() <= 1e-5 * max(1.0, float(q.grad.abs().max())), n


@pytest.mark.parametrize("precision", ["bf16", "fp32"])
def test_training_step_with_the_grouped_heads_matches_the_modules(precision):
    """The committed list with and without MDETR_HEADS: the heads computed by grouped fp32 launches or by the modules (library
    GEMMs) give the same loss trajectory to fp32 rounding (dropout off), and no library GEMM of the heads' shapes remains."""
    import bench
    from model_init import disable_dropout_
    dev = torch.device("cuda", 0)
    base = tuple(sorted(set(bench.COMMITTED_SWITCHES[precision]) - {"MDETR_HEADS"}))
    traj = {}
    try:
        for names in (base, base + ("MDETR_HEADS",)):
            step = bench.TrainStep(dev, 2, precision, size=(96, 320), switches=names)
            disable_dropout_(step.raw_model)
            traj[names] = [float(step()) for _ in range(3)]
    finally:
        bench.apply_switches(set())
    for a, b in zip(traj[base], traj[base + ("MDETR_HEADS",)]):
        assert abs(a - b) <= (1e-3 if precision == "bf16" else 1e-4) * abs(a), traj


def test_head_tail_kernels_match_the_framework_expression_on_the_gpu():
    """csrc/head_tail.hip at the training shape (3 levels x 8 images x 550 queries, the 24 x 80 depth map) against
    monodetr.py:226-253 written with the framework's operators in float64: values and all five gradients."""
    import sys, os
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from test_head_tail_emulated_cpu import reference
    from monodetr_amd import head_tail_ext as ext
    torch.manual_seed(0)
    L, B, Q, H, W = 3, 8, 550, 24, 80
    dev = "cuda"
    delta, init_ref, inter = torch.randn(L, B, Q, 6, device=dev), torch.rand(B, Q, 2, device=dev), torch.rand(L - 1, B, Q, 6, device=dev)
    size3d, depth_reg, depth_map = torch.rand(L, B, Q, 3, device=dev) + 0.5, torch.randn(L, B, Q, 2, device=dev), torch.rand(B, H, W, device=dev) * 50
    img_h, focal = torch.full((B,), 375.0, device=dev), torch.full((B,), 721.5, device=dev)
    was, ext.ENABLED = ext.ENABLED, True
    try:
        leaves = [t.clone().requires_grad_(True) for t in (delta, init_ref, size3d, depth_reg, depth_map)]
        coord, ave = ext.head_tail(leaves[0], leaves[1], inter, leaves[2], leaves[3], leaves[4], img_h, focal)
        gc, ga = torch.randn_like(coord), torch.randn_like(ave)
        ((coord * gc).sum() + (ave * ga).sum()).backward()
        again = ext.head_tail(delta, init_ref, inter, size3d, depth_reg, depth_map, img_h, focal)
        assert torch.equal(again[0], coord.detach()) and torch.equal(again[1], ave.detach())
    finally:
        ext.ENABLED = was
    ref_leaves = [t.double().clone().requires_grad_(True) for t in (delta, init_ref, size3d, depth_reg, depth_map)]
    rc, ra = reference(ref_leaves[0], ref_leaves[1], inter.double(), ref_leaves[2], ref_leaves[3], ref_leaves[4], img_h.double(), focal.double())
    ((rc * gc.double()).sum() + (ra * ga.double()).sum()).backward()
    assert (coord.double() - rc).abs().max() < 1e-6
    assert ((ave.double() - ra).abs() / (1 + ra.abs())).max() < 1e-4          # fp32 of 1 / (sigmoid + 1e-6) - 1 + f H / h2d (values up to ~1e3)
    for name, a, b in zip(("delta", "init_ref", "size3d", "depth_reg", "depth_map"), leaves, ref_leaves):
        assert ((a.grad.double() - b.grad).abs() / (1e-3 * b.grad.abs().max() + b.grad.abs())).max() < 2e-3, name


@pytest.mark.parametrize("precision", ["bf16", "fp32"])
def test_training_step_with_the_head_tail_kernels_matches_the_framework_operators(precision):
    """The committed list with and without MDETR_HEAD_TAIL: the same loss trajectory to fp32 rounding (dropout off)."""
    import bench
    from model_init import disable_dropout_
    dev = torch.device("cuda", 0)
    base = tuple(sorted(set(bench.COMMITTED_SWITCHES[precision]) - {"MDETR_HEAD_TAIL"}))
    traj = {}
    try:
        for names in (base, base + ("MDETR_HEAD_TAIL",)):
            step = bench.TrainStep(dev, 2, precision, size=(96, 320), switches=names)
            disable_dropout_(step.raw_model)
            traj[names] = [float(step()) for _ in range(3)]
    finally:
        bench.apply_switches(set())
    for a, b in zip(traj[base], traj[base + ("MDETR_HEAD_TAIL",)]):
        assert abs(a - b) <= (1e-3 if precision == "bf16" else 1e-4) * abs(a), traj
