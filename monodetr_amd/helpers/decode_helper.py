"""Model outputs -> KITTI detections -- mirror of ``lib/helpers/decode_helper.py`` (the two functions the tester
calls; the heat-map helpers of that file belong to centre-net style heads and are not mirrored).

``extract_dets_from_outputs`` runs on the device without host synchronisation; ``decode_detections`` is the
reference's numpy step (image-plane decoding, back-projection with the frame's calibration), vectorised per image."""
import numpy as np
import torch

from ..datasets.utils import class2angle
from ..utils import box_ops


def extract_dets_from_outputs(outputs, K=50, topk=50):
    """Top-``topk`` (query, class) pairs by sigmoid score -> ``[B, topk, 37]``:
    label, score, 2-D centre x, y, 2-D width, height, depth, 24 heading logits / residuals, 3 dimensions,
    projected 3-D centre x, y, exp(-log-variance of depth)."""
    logits, boxes = outputs['pred_logits'], outputs['pred_boxes']
    B, Q, C = logits.shape
    scores, flat = torch.topk(logits.sigmoid().view(B, -1), topk, dim=1)
    query = (flat // C).unsqueeze(-1)
    labels = (flat % C).view(B, -1, 1)
    take = lambda t: torch.gather(t, 1, query.expand(-1, -1, t.shape[-1]))
    boxes = take(boxes)
    heading = take(outputs['pred_angle'])
    size_3d = take(outputs['pred_3d_dim'])
    depth = take(outputs['pred_depth'][:, :, 0:1])
    sigma = take(torch.exp(-outputs['pred_depth'][:, :, 1:2]))
    xywh = box_ops.box_xyxy_to_cxcywh(box_ops.box_cxcylrtb_to_xyxy(boxes))
    return torch.cat([labels, scores.view(B, -1, 1), xywh[:, :, 0:1], xywh[:, :, 1:2], xywh[:, :, 2:4], depth, heading, size_3d,
                      boxes[:, :, 0:1], boxes[:, :, 1:2], sigma], dim=2)


def get_heading_angle(heading):
    """24 values = 12 bin logits + 12 residuals -> angle in (-pi, pi]."""
    cls = np.argmax(heading[0:12])
    return class2angle(cls, heading[12:24][cls], to_label_format=True)


def decode_detections(dets, info, calibs, cls_mean_size, threshold):
    """``dets`` numpy ``[B, max_dets, 37]`` (layout above), ``info`` dict of numpy arrays (``img_id``, ``img_size``),
    ``calibs`` one ``Calibration`` per image -> ``{img_id: [[cls, alpha, x1, y1, x2, y2, h, w, l, x, y, z, ry, score], ...]}``."""
    results = {}
    for i in range(dets.shape[0]):
        d = dets[i][dets[i, :, 1] >= threshold]
        W, H = info['img_size'][i][0], info['img_size'][i][1]
        cls_id = d[:, 0].astype(np.int64)
        x, y, w, h = d[:, 2] * W, d[:, 3] * H, d[:, 4] * W, d[:, 5] * H
        depth = d[:, 6]
        dims = d[:, 31:34] + cls_mean_size[cls_id]
        loc = calibs[i].img_to_rect(d[:, 34] * W, d[:, 35] * H, depth)
        loc[:, 1] += dims[:, 0] / 2                                  # box origin: bottom centre
        preds = []
        for j in range(d.shape[0]):
            alpha = get_heading_angle(d[j, 7:31])
            ry = calibs[i].alpha2ry(alpha, x[j])
            preds.append([int(cls_id[j]), alpha, x[j] - w[j] / 2, y[j] - h[j] / 2, x[j] + w[j] / 2, y[j] + h[j] / 2]
                         + dims[j].tolist() + loc[j].tolist() + [ry, d[j, 1] * d[j, -1]])
        results[info['img_id'][i]] = preds
    return results
