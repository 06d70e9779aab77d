"""Host helpers that mirror the reference's utils (utils/box_ops.py, utils/misc.py): identities and
cross-checks between the pairwise and the matched-pair forms used by the criterion."""
import pytest
import torch

from monodetr_amd.utils import box_ops, misc


def _boxes(n, seed):
    g = torch.Generator().manual_seed(seed)
    c = torch.rand(n, 2, generator=g, dtype=torch.float64)
    wh = 0.05 + 0.4 * torch.rand(n, 2, generator=g, dtype=torch.float64)
    return torch.cat((c, wh), -1)


def test_box_conversions_round_trip():
    b = _boxes(50, 0)
    xyxy = box_ops.box_cxcywh_to_xyxy(b)
    assert torch.allclose(box_ops.box_xyxy_to_cxcywh(xyxy), b, atol=1e-12)
    assert torch.allclose(box_ops.box_area(xyxy), b[:, 2] * b[:, 3], atol=1e-12)
    lrtb = torch.cat((b[:, :2], b[:, 2:3] / 2, b[:, 2:3] / 2, b[:, 3:4] / 2, b[:, 3:4] / 2), -1)     # symmetric (l, r, t, b)
    assert torch.allclose(box_ops.box_cxcylrtb_to_xyxy(lrtb), xyxy, atol=1e-12)


def test_pairwise_giou_properties_and_matched_pair_form():
    a, b = box_ops.box_cxcywh_to_xyxy(_boxes(40, 1)), box_ops.box_cxcywh_to_xyxy(_boxes(40, 2))
    g = box_ops.generalized_box_iou(a, b)
    assert g.shape == (40, 40) and (g <= 1 + 1e-12).all() and (g >= -1 - 1e-12).all()
    assert torch.allclose(box_ops.generalized_box_iou(a, a).diagonal(), torch.ones(40, dtype=a.dtype), atol=1e-12)
    assert torch.allclose(g, box_ops.generalized_box_iou(b, a).t(), atol=1e-12)                 # symmetric
    assert torch.allclose(box_ops.elementwise_giou(a, b), g.diagonal(), atol=1e-12)              # what the criterion uses
    iou, union = box_ops.box_iou(a, b)
    assert (iou >= 0).all() and (iou <= 1 + 1e-12).all() and (g <= iou + 1e-12).all()
    with pytest.raises(AssertionError):
        box_ops.generalized_box_iou(a.flip(-1), b)                                               # degenerate boxes are refused


def test_inverse_sigmoid_and_accuracy():
    x = torch.linspace(-8, 8, 101, dtype=torch.float64)
    assert torch.allclose(misc.inverse_sigmoid(x.sigmoid()), x, atol=1e-6)
    assert torch.isfinite(misc.inverse_sigmoid(torch.tensor([0.0, 1.0, -0.5, 1.5]))).all()     # clamped, never inf
    logits = torch.tensor([[0.1, 2.0, 0.3], [1.5, 0.2, 0.1], [0.0, 0.1, 3.0], [0.9, 0.8, 0.7]])
    target = torch.tensor([1, 0, 1, 2])
    top1, top2 = misc.accuracy(logits, target, topk=(1, 2))
    assert float(top1) == 50.0 and float(top2) == 75.0
    assert float(misc.accuracy(logits[:0], target[:0])[0]) == 0.0


def test_single_process_distributed_helpers():
    assert misc.get_world_size() == 1 and misc.get_rank() == 0 and misc.is_main_process()
    d = {"a": torch.tensor(1.0), "b": torch.tensor(2.0)}
    assert misc.reduce_dict(d) is d
    m = torch.zeros(2, 3, 4, dtype=torch.bool)
    assert not misc.no_padding(m) and misc.no_padding(misc.mark_no_padding(m)) and misc.no_padding(None)
    assert misc.at_least_fp32(torch.ones(2, dtype=torch.bfloat16)).dtype == torch.float32
    assert misc.at_least_fp32(torch.ones(2, dtype=torch.float64)).dtype == torch.float64
