"""Box helpers used by the matcher and the losses -- mirror of /root/reference/utils/box_ops.py:13-72
(same function names and conventions; ``box_area`` is inlined because torchvision is not a
dependency here, reference :10)."""
import torch


def box_cxcywh_to_xyxy(x):
    cx, cy, w, h = x.unbind(-1)
    return torch.stack((cx - 0.5 * w, cy - 0.5 * h, cx + 0.5 * w, cy + 0.5 * h), -1)


def box_cxcylrtb_to_xyxy(x):
    """(cx, cy, l, r, t, b): projected 3D centre + distances to the 2D box sides (reference :20-24)."""
    cx, cy, l, r, t, b = x.unbind(-1)
    return torch.stack((cx - l, cy - t, cx + r, cy + b), -1)


def box_xyxy_to_cxcywh(x):
    x0, y0, x1, y1 = x.unbind(-1)
    return torch.stack(((x0 + x1) / 2, (y0 + y1) / 2, x1 - x0, y1 - y0), -1)


def box_area(boxes):
    return (boxes[..., 2] - boxes[..., 0]) * (boxes[..., 3] - boxes[..., 1])


def box_iou(boxes1, boxes2):
    """Pairwise IoU and union, boxes in xyxy: [N,4] x [M,4] -> [N,M] (reference :34-48)."""
    a1, a2 = box_area(boxes1), box_area(boxes2)
    wh = (torch.min(boxes1[:, None, 2:], boxes2[:, 2:]) - torch.max(boxes1[:, None, :2], boxes2[:, :2])).clamp(min=0)
    inter = wh[..., 0] * wh[..., 1]
    union = a1[:, None] + a2 - inter
    return inter / union, union


def generalized_box_iou(boxes1, boxes2, check=True):
    """Pairwise GIoU (reference :51-72).  ``check`` keeps the reference's degenerate-box asserts
    (each one is a device->host sync); hot callers that have already validated pass check=False."""
    if check:
        assert (boxes1[:, 2:] >= boxes1[:, :2]).all()
        assert (boxes2[:, 2:] >= boxes2[:, :2]).all()
    iou, union = box_iou(boxes1, boxes2)
    wh = (torch.max(boxes1[:, None, 2:], boxes2[:, 2:]) - torch.min(boxes1[:, None, :2], boxes2[:, :2])).clamp(min=0)
    hull = wh[..., 0] * wh[..., 1]
    return iou - (hull - union) / hull


def elementwise_giou(a, b):
    """GIoU of matched pairs a[i] vs b[i] ([N,4] xyxy each) -- equals diag(generalized_box_iou(a, b))
    without building the N x N matrix the reference builds (monodetr.py:381-383)."""
    inter_wh = (torch.min(a[:, 2:], b[:, 2:]) - torch.max(a[:, :2], b[:, :2])).clamp(min=0)
    inter = inter_wh[:, 0] * inter_wh[:, 1]
    union = box_area(a) + box_area(b) - inter
    hull_wh = (torch.max(a[:, 2:], b[:, 2:]) - torch.min(a[:, :2], b[:, :2])).clamp(min=0)
    hull = hull_wh[:, 0] * hull_wh[:, 1]
    return inter / union - (hull - union) / hull
