"""csrc/head_tail.hip on the HIP-on-CPU shim: the arithmetic between the heads' raw outputs and the predictions (reference
lib/models/monodetr/monodetr.py:226-253) and the decoder's reference update (depthaware_transformer.py:602-613), against the same
expressions written with the framework's operators -- values and every gradient, in float64."""
import pytest
import torch
import torch.nn.functional as F

import native_emul
from monodetr_amd.utils.misc import inverse_sigmoid


@pytest.fixture()
def ext():
    from monodetr_amd import head_tail_ext
    head_tail_ext._backend = native_emul.lib()
    was, head_tail_ext.ENABLED = head_tail_ext.ENABLED, True
    yield head_tail_ext
    head_tail_ext.ENABLED = was
    head_tail_ext._backend = None


def reference(delta, init_ref, inter_refs, size3d, depth_reg, depth_map, img_h, focal):
    """monodetr.py:226-253 on level-stacked tensors (the product's own framework path, MonoDETR.forward)."""
    L, B, Q, _ = delta.shape
    first = inverse_sigmoid(init_ref)
    if first.shape[-1] == 2:
        first = F.pad(first, (0, 4))
    ref = torch.cat((first[None], inverse_sigmoid(inter_refs)), 0)
    coord = (delta + ref).sigmoid()
    h2d = torch.clamp((coord[..., 4] + coord[..., 5]) * img_h.view(1, -1, 1), min=1.0)
    geo = size3d[..., 0] / h2d * focal.view(1, -1, 1)
    centre = ((coord[..., :2] - 0.5) * 2).detach().permute(1, 0, 2, 3).reshape(B, L * Q, 1, 2)
    m = F.grid_sample(depth_map.unsqueeze(1), centre, mode='bilinear', align_corners=True).view(B, L, Q).permute(1, 0, 2)
    ave = torch.cat([((1. / (depth_reg[..., 0:1].sigmoid() + 1e-6) - 1.) + geo.unsqueeze(-1) + m.unsqueeze(-1)) / 3, depth_reg[..., 1:2]], -1)
    return coord, ave


@pytest.mark.parametrize("nd0", [2, 6])
def test_head_tail_matches_the_framework_expression(ext, nd0):
    torch.manual_seed(nd0)
    L, B, Q, H, W = 3, 2, 37, 24, 80
    delta = torch.randn(L, B, Q, 6)
    delta[0, 0, :4, 4:] = -9.0                                       # tiny boxes: (t + b) * image height < 1 -> the clamp's flat side
    init_ref = torch.rand(B, Q, nd0)
    init_ref[0, 0] = 0.0                                             # on the clamp of inverse_sigmoid
    init_ref[0, 1] = 1.0
    inter = torch.rand(L - 1, B, Q, 6)
    size3d = torch.rand(L, B, Q, 3) + 0.5
    depth_reg = torch.randn(L, B, Q, 2)
    depth_map = torch.rand(B, H, W) * 50
    img_h, focal = torch.tensor([375.0, 370.0]), torch.tensor([721.5, 707.0])
    leaves = [t.clone().requires_grad_(True) for t in (delta, init_ref, size3d, depth_reg, depth_map)]
    coord, ave = ext.head_tail(leaves[0], leaves[1], inter, leaves[2], leaves[3], leaves[4], img_h, focal)
    gc, ga = torch.randn_like(coord), torch.randn_like(ave)
    ((coord * gc).sum() + (ave * ga).sum()).backward()
    ref_leaves = [t.double().clone().requires_grad_(True) for t in (delta, init_ref, size3d, depth_reg, depth_map)]
    rc, ra = reference(ref_leaves[0], ref_leaves[1], inter.double(), ref_leaves[2], ref_leaves[3], ref_leaves[4], img_h.double(), focal.double())
    ((rc * gc.double()).sum() + (ra * ga.double()).sum()).backward()
    assert (coord.double() - rc).abs().max() < 1e-6
    assert ((ave.double() - ra).abs() / (1 + ra.abs())).max() < 1e-5
    for name, a, b in zip(("delta", "init_ref", "size3d", "depth_reg", "depth_map"), leaves, ref_leaves):
        assert a.grad is not None and ((a.grad.double() - b.grad).abs() / (1e-3 * b.grad.abs().max() + b.grad.abs())).max() < 2e-3, name
    # only one of the two outputs used: the other's gradient is absent, not zeros to be read
    leaves2 = [t.clone().requires_grad_(True) for t in (delta, init_ref, size3d, depth_reg, depth_map)]
    c2, _ = ext.head_tail(leaves2[0], leaves2[1], inter, leaves2[2], leaves2[3], leaves2[4], img_h, focal)
    (c2 * gc).sum().backward()
    assert float(leaves2[4].grad.abs().max()) == 0.0 and float(leaves2[3].grad.abs().max()) == 0.0


@pytest.mark.parametrize("nd", [2, 6])
def test_box_refine_matches_the_decoder_expression(ext, nd):
    torch.manual_seed(5)
    delta, ref = torch.randn(2, 50, 6), torch.rand(2, 50, nd)
    ref[0, 0], ref[0, 1] = 0.0, 1.0
    got = ext.box_refine(delta, ref)
    if nd == 6:
        want = (delta.double() + inverse_sigmoid(ref.double())).sigmoid()
    else:
        want = torch.cat((delta.double()[..., :2] + inverse_sigmoid(ref.double()), delta.double()[..., 2:]), -1).sigmoid()
    assert got.shape == (2, 50, 6) and (got.double() - want).abs().max() < 1e-6
