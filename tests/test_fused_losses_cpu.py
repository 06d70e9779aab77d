"""The fused matched-pair losses (csrc/pair_losses_math.h, the arithmetic of csrc/pair_losses.hip) against
the PyTorch criterion and its autograd, on the CPU through the host build of the same functions
(tests/native).  Values: all 26 loss entries; gradients: w.r.t. all five prediction tensors of all levels."""
import pytest
import torch

import backends
import native_host
from model_init import load_cfg


def _problem(L, B, Q, K, G, seed, empty_image=True):
    g = torch.Generator().manual_seed(seed)
    r = lambda *s: torch.rand(*s, generator=g)          # noqa: E731
    n = lambda *s: torch.randn(*s, generator=g)         # noqa: E731
    preds = {
        'pred_logits': n(L, B, Q, 3),
        'pred_boxes': torch.cat((0.2 + 0.6 * r(L, B, Q, 2), 0.02 + 0.2 * r(L, B, Q, 4)), -1),
        'pred_3d_dim': 0.5 + 2.5 * r(L, B, Q, 3),
        'pred_depth': torch.cat((5 + 40 * r(L, B, Q, 1), n(L, B, Q, 1)), -1),
        'pred_angle': n(L, B, Q, 24),
    }
    num = torch.randint(1, K + 1, (B,), generator=g)
    if empty_image:
        num[-1] = 0
    gt = {
        'labels': torch.randint(0, 3, (B, K), generator=g),
        'boxes': torch.cat((0.2 + 0.6 * r(B, K, 2), 0.05 + 0.2 * r(B, K, 2)), -1),
        'boxes_3d': torch.cat((0.2 + 0.6 * r(B, K, 2), 0.02 + 0.2 * r(B, K, 4)), -1),
        'depth': 5 + 40 * r(B, K),
        'size_3d': 0.8 + 2 * r(B, K, 3),
        'heading_bin': torch.randint(0, 12, (B, K), generator=g),
        'heading_res': 0.3 * n(B, K),
        'valid': torch.arange(K)[None, :] < num[:, None],
        'num': num.to(torch.int32),
        'num_host': [int(v) for v in num],
    }
    return preds, gt


@pytest.fixture(params=backends.BACKENDS)
def criterion(request):
    from monodetr_amd import ddn_loss_ext, lsa_ext, pair_losses_ext
    from monodetr_amd.monodetr import build_monodetr
    torch.manual_seed(0)
    cfg = load_cfg()
    _, crit = build_monodetr(cfg)
    crit.train()
    pair_losses_ext._backend = ddn_loss_ext._backend = lsa_ext._backend = backends.get(request.param)
    yield crit
    pair_losses_ext._backend = ddn_loss_ext._backend = lsa_ext._backend = None


@pytest.mark.parametrize("L,B,Q,K,G,seed", [(3, 4, 110, 7, 11, 0), (3, 2, 550, 50, 11, 1), (1, 3, 20, 5, 1, 2)])
def test_fused_pair_losses_match_the_criterion(criterion, L, B, Q, K, G, seed):
    preds, gt = _problem(L, B, Q, K, G, seed)
    criterion.group_num = G
    outputs = {k: v[-1] for k, v in preds.items()}
    outputs['pred_depth_map_logits'] = torch.randn(B, 81, 6, 20, generator=torch.Generator().manual_seed(seed + 7))
    outputs['aux_outputs'] = [{k: v[i] for k, v in preds.items()} for i in range(L - 1)]

    def run(fused):
        criterion.fused_pair_losses = criterion.matcher.fused_cost = fused      # matching, pair losses, depth-map loss
        leaves = {k: v.clone().requires_grad_(True) for k, v in preds.items()}
        out = dict(outputs, _levels=leaves)
        losses = criterion(out, gt)
        total = criterion.weighted_total(losses)
        total.backward()
        return losses, {k: v.grad for k, v in leaves.items()}

    ref_losses, ref_grads = run(False)
    got_losses, got_grads = run(True)
    assert set(ref_losses) == set(got_losses)
    for k in ref_losses:
        a, b = float(ref_losses[k]), float(got_losses[k])
        assert abs(a - b) <= 2e-5 * max(1.0, abs(a)), (k, a, b)
    for k in ref_grads:
        scale = ref_grads[k].abs().max().item()
        assert scale > 0
        assert (ref_grads[k] - got_grads[k]).abs().max().item() <= 2e-5 * scale + 1e-9, k


def test_fused_pair_losses_with_a_device_normaliser_and_no_targets(criterion):
    """num_boxes as a tensor (the multi-GPU form) and a batch without any object (every loss but the focal
    background term is zero, gradients finite)."""
    from monodetr_amd.pair_losses_ext import fused_pair_losses
    preds, gt = _problem(2, 2, 22, 4, 11, 5, empty_image=False)
    gt['valid'][:] = False
    gt['num'][:] = 0
    assign = torch.full((2, 2, 11, 4), -1, dtype=torch.int64)
    leaves = {k: v.clone().requires_grad_(True) for k, v in preds.items()}
    rows = fused_pair_losses(leaves, assign, gt, torch.tensor(1.0), 0.25)
    for name in ('loss_center', 'loss_bbox', 'loss_giou', 'loss_depth', 'loss_dim', 'loss_angle'):
        assert float(rows[name].abs().sum()) == 0.0
    assert (rows['class_error'] == 100).all() and (rows['loss_ce'] > 0).all()
    sum(rows[k].sum() for k in ('loss_ce', 'loss_bbox', 'loss_dim')).backward()
    assert all(torch.isfinite(v.grad).all() for v in leaves.values())
    assert float(leaves['pred_boxes'].grad.abs().sum()) == 0.0


@pytest.mark.parametrize("backend", backends.BACKENDS)
@pytest.mark.parametrize("layout", ["contiguous", "channels_last"])
def test_fused_ddn_loss_matches_the_pytorch_ddn_loss(layout, backend):
    """csrc/ddn_loss_math.h (host build) == DDNLoss (box painting, LID bins, focal loss with the +1e-6 one-hot,
    fg/bg balancer): value and the gradient w.r.t. all depth logits; boxes that leave the image, overlap
    (nearest object wins), an image without objects, padded slots."""
    from monodetr_amd import ddn_loss_ext
    from monodetr_amd.monodetr.depth_predictor.ddn_loss import DDNLoss
    from monodetr_amd.utils import box_ops
    ddn_loss_ext._backend = backends.get(backend)
    try:
        g = torch.Generator().manual_seed(3)
        B, C, H, W, K = 3, 81, 24, 80, 6
        logits = torch.randn(B, C, H, W, generator=g)
        if layout == "channels_last":
            logits = logits.contiguous(memory_format=torch.channels_last)
        boxes = torch.cat((torch.rand(B, K, 2, generator=g), 0.05 + 0.5 * torch.rand(B, K, 2, generator=g)), -1)
        boxes[0, 0] = torch.tensor([0.02, 0.5, 0.3, 0.4])              # sticks out on the left: negative corner wraps
        boxes[0, 1] = boxes[0, 2]                                       # identical boxes, different depths
        depth = 2 + 55 * torch.rand(B, K, generator=g)
        depth[1, 0] = 75.0                                              # beyond depth_max -> the extra bin
        num = torch.tensor([K, 3, 0])
        valid = torch.arange(K)[None, :] < num[:, None]

        ref_mod = DDNLoss()
        za = logits.clone().requires_grad_(True)
        b = boxes
        xyxy = box_ops.box_cxcywh_to_xyxy(torch.stack((b[..., 0] * W, b[..., 1] * H, b[..., 2] * W, b[..., 3] * H), -1))
        xyxy = torch.where(valid[..., None], xyxy, torch.zeros_like(xyxy))
        ref = ref_mod(za, xyxy.reshape(-1, 4), K, depth.reshape(-1), valid=valid.reshape(-1))
        (ref * 1.7).backward()

        zb = logits.clone().requires_grad_(True)
        got = ddn_loss_ext.fused_ddn_loss(zb, boxes, depth, valid, ref_mod.alpha, ref_mod.balancer.fg_weight,
                                          ref_mod.balancer.bg_weight)
        (got * 1.7).backward()
        assert abs(float(ref) - float(got)) <= 2e-6 * abs(float(ref))
        assert zb.grad.stride() == zb.stride()
        assert (za.grad - zb.grad).abs().max() <= 2e-5 * za.grad.abs().max()
    finally:
        ddn_loss_ext._backend = None


@pytest.mark.parametrize("backend", backends.BACKENDS)
def test_fused_matching_cost_and_solver_match_scipy_on_the_pytorch_cost(backend):
    """pl_match_cost (the cost evaluated inside the device solver) against matcher.cost_padded: the assignments
    equal scipy's on the PyTorch cost matrix (total cost equal where an optimum is not unique).  "host": a serial
    copy of the solver around the shared cost arithmetic; "emul": the real lsa.hip kernel (wave argmin, LDS duals)
    with the fused cost, on the HIP-on-CPU shim."""
    import numpy as np
    from scipy.optimize import linear_sum_assignment
    from monodetr_amd import lsa_ext
    from monodetr_amd.monodetr import build_monodetr
    lsa_ext._backend = backends.get(backend)
    try:
        _, crit = build_monodetr(load_cfg())
        m = crit.matcher
        sizes = ((3, 4, 550, 50, 11, 0), (2, 3, 110, 7, 11, 1), (1, 2, 64, 64, 1, 2))
        if backend == "emul":
            sizes += ((2, 2, 1100, 50, 11, 3),)                       # 100 queries per group (512 x 1760 configuration): two columns per lane
        for (L, B, Q, K, G, seed) in sizes:
            preds, gt = _problem(L, B, Q, K, G, seed)
            cost = m.cost_padded(preds['pred_logits'], preds['pred_boxes'], gt).double().numpy()
            got = lsa_ext.batched_assignment_fused(preds['pred_logits'], preds['pred_boxes'], gt, G,
                                                   (m.cost_class, m.cost_bbox, m.cost_3dcenter, m.cost_giou)).numpy()
            n = Q // G
            for l in range(L):
                for b in range(B):
                    k = int(gt['num'][b])
                    for g in range(G):
                        a = got[l, b, g]
                        assert (a[k:] == -1).all()
                        if k == 0:
                            continue
                        sub = cost[l, b, g * n:(g + 1) * n, :k]
                        r, c = linear_sum_assignment(sub)
                        mine = a[:k] - g * n
                        assert len(set(mine.tolist())) == k and mine.min() >= 0 and mine.max() < n
                        ref_total, my_total = sub[r, c].sum(), sub[mine, np.arange(k)].sum()
                        assert abs(ref_total - my_total) <= 1e-5 * max(1.0, abs(ref_total))
    finally:
        lsa_ext._backend = None
