"""Hungarian matching between queries and ground-truth objects -- mirror of
lib/models/monodetr/matcher.py (``HungarianMatcher`` :14-104, ``build_matcher`` :107-112).

Same cost (focal classification cost :62-66, L1 on the projected 3D centre :68-72, L1 on the
l/r/t/b box sides :74-78, negative GIoU :80-83, weighted :86) and the same per-group assignment
(:90-103: queries are split into ``group_num`` groups, each matched independently against all
targets of the image).  The assignment problems themselves are solved on the host like the
reference does (scipy ``linear_sum_assignment``, :96); what changes is the traffic around them:
``match_layers`` scores all decoder layers in one batched device computation and makes ONE
device->host copy per iteration instead of one per layer.
"""
import os

import numpy as np
import torch
from scipy.optimize import linear_sum_assignment
from torch import nn

from ..utils.box_ops import box_cxcylrtb_to_xyxy, generalized_box_iou


class HungarianMatcher(nn.Module):
    def __init__(self, cost_class: float = 1, cost_3dcenter: float = 1, cost_bbox: float = 1, cost_giou: float = 1):
        super().__init__()
        self.cost_class, self.cost_3dcenter, self.cost_bbox, self.cost_giou = cost_class, cost_3dcenter, cost_bbox, cost_giou
        assert cost_class != 0 or cost_bbox != 0 or cost_giou != 0, "all costs cant be 0"
        # MDETR_FUSED_LOSSES=1: matching cost evaluated inside the device solver (off until its first GPU
        # validation, like the fused losses: tests/test_fused_gpu.py)
        self.fused_cost = os.environ.get("MDETR_FUSED_LOSSES") == "1"

    @torch.no_grad()
    def cost_matrix(self, pred_logits, pred_boxes, tgt_ids, tgt_boxes):
        """pred_logits [N, C], pred_boxes [N, 6], tgt_ids [K], tgt_boxes [K, 6] -> [N, K]."""
        prob = pred_logits.sigmoid()
        alpha, gamma = 0.25, 2.0
        neg = (1 - alpha) * (prob ** gamma) * (-(1 - prob + 1e-8).log())
        pos = alpha * ((1 - prob) ** gamma) * (-(prob + 1e-8).log())
        c_class = pos[:, tgt_ids] - neg[:, tgt_ids]
        c_center = torch.cdist(pred_boxes[:, 0:2], tgt_boxes[:, 0:2], p=1)
        c_bbox = torch.cdist(pred_boxes[:, 2:6], tgt_boxes[:, 2:6], p=1)
        c_giou = -generalized_box_iou(box_cxcylrtb_to_xyxy(pred_boxes), box_cxcylrtb_to_xyxy(tgt_boxes), check=False)
        return self.cost_bbox * c_bbox + self.cost_3dcenter * c_center + self.cost_class * c_class + self.cost_giou * c_giou

    @staticmethod
    def assign(C, sizes, group_num):
        """C: host array [B, Q, K_total]; sizes: targets per image.  Per image: (query idx, target idx)
        int64 arrays, groups concatenated in order with the group's query offset added (:98-102)."""
        B, Q, _ = C.shape
        n = Q // group_num
        offs = np.concatenate(([0], np.cumsum(sizes)))
        out = []
        for i in range(B):
            rows, cols = [], []
            if sizes[i] > 0:
                block = C[i, :, offs[i]:offs[i + 1]]
                for g in range(group_num):
                    r, c = linear_sum_assignment(block[g * n:(g + 1) * n])
                    rows.append(r + g * n)
                    cols.append(c)
            out.append((np.concatenate(rows).astype(np.int64) if rows else np.zeros(0, np.int64),
                        np.concatenate(cols).astype(np.int64) if cols else np.zeros(0, np.int64)))
        return out

    @torch.no_grad()
    def match_layers(self, layer_outputs, targets, group_num=11):
        """Match several decoder layers at once.  layer_outputs: list of dicts with 'pred_logits'
        [B,Q,C] and 'pred_boxes' [B,Q,6].  Returns, per layer, the reference's list of
        (index_i, index_j) int64 CPU tensors."""
        logits = torch.stack([o["pred_logits"] for o in layer_outputs])          # [Ly, B, Q, C]
        boxes = torch.stack([o["pred_boxes"] for o in layer_outputs])
        Ly, B, Q, _ = logits.shape
        sizes = [len(v["boxes"]) for v in targets]
        tgt_ids = torch.cat([v["labels"] for v in targets]).long()
        tgt_boxes = torch.cat([v["boxes_3d"] for v in targets])
        if logits.dtype in (torch.float16, torch.bfloat16):     # autocast heads: score in fp32
            logits, boxes = logits.float(), boxes.float()
        C = self.cost_matrix(logits.flatten(0, 2), boxes.flatten(0, 2), tgt_ids, tgt_boxes.to(boxes.dtype))
        C = C.view(Ly, B, Q, -1).cpu().numpy()                                    # the one D2H copy
        return [[(torch.from_numpy(i), torch.from_numpy(j)) for i, j in self.assign(C[l], sizes, group_num)]
                for l in range(Ly)]

    @torch.no_grad()
    def cost_padded(self, logits, boxes, gt):
        """logits [L,B,Q,C], boxes [L,B,Q,6], padded ground truth -> cost [L,B,Q,K] (same terms as
        `cost_matrix`, evaluated per image against that image's K target slots only)."""
        if logits.dtype in (torch.float16, torch.bfloat16):
            logits, boxes = logits.float(), boxes.float()
        L, B, Q, _ = logits.shape
        K = gt["valid"].shape[1]
        prob = logits.sigmoid()
        alpha, gamma = 0.25, 2.0
        neg = (1 - alpha) * (prob ** gamma) * (-(1 - prob + 1e-8).log())
        pos = alpha * ((1 - prob) ** gamma) * (-(prob + 1e-8).log())
        lab = gt["labels"][None, :, None, :].expand(L, B, Q, K)
        c_class = (pos - neg).gather(-1, lab)
        tb = gt["boxes_3d"].to(boxes.dtype)[None, :, None, :, :]                     # [1,B,1,K,6]
        pb = boxes[:, :, :, None, :]                                                 # [L,B,Q,1,6]
        c_center = (pb[..., 0:2] - tb[..., 0:2]).abs().sum(-1)
        c_bbox = (pb[..., 2:6] - tb[..., 2:6]).abs().sum(-1)
        p_xyxy, t_xyxy = box_cxcylrtb_to_xyxy(pb), box_cxcylrtb_to_xyxy(tb)
        wh = (torch.min(p_xyxy[..., 2:], t_xyxy[..., 2:]) - torch.max(p_xyxy[..., :2], t_xyxy[..., :2])).clamp(min=0)
        inter = wh[..., 0] * wh[..., 1]
        area_p = (p_xyxy[..., 2] - p_xyxy[..., 0]) * (p_xyxy[..., 3] - p_xyxy[..., 1])
        area_t = (t_xyxy[..., 2] - t_xyxy[..., 0]) * (t_xyxy[..., 3] - t_xyxy[..., 1])
        union = area_p + area_t - inter
        hw = (torch.max(p_xyxy[..., 2:], t_xyxy[..., 2:]) - torch.min(p_xyxy[..., :2], t_xyxy[..., :2])).clamp(min=0)
        hull = hw[..., 0] * hw[..., 1]
        c_giou = -(inter / union - (hull - union) / hull)
        return self.cost_bbox * c_bbox + self.cost_3dcenter * c_center + self.cost_class * c_class + self.cost_giou * c_giou

    @torch.no_grad()
    def assign_padded(self, layer_outputs, gt, group_num=11):
        """Assignment of every decoder layer in static shape: [L, B, G, K] int64, the matched query of
        each target slot (-1 for padded slots).  On the GPU the problems are solved by the device
        Hungarian kernel (csrc/lsa.hip) with no host synchronisation; on CPU tensors by scipy, as the
        reference does everywhere."""
        return self.assign_stacked(torch.stack([o["pred_logits"] for o in layer_outputs]),
                                   torch.stack([o["pred_boxes"] for o in layer_outputs]), gt, group_num)

    @torch.no_grad()
    def assign_stacked(self, logits, boxes, gt, group_num=11):
        """`assign_padded` on already layer-stacked predictions: logits [L,B,Q,C], boxes [L,B,Q,6]."""
        L, B, Q, _ = logits.shape
        K = gt["valid"].shape[1]
        n = Q // group_num
        if self.fused_cost and n <= 128 and K <= min(n, 64) and n * group_num == Q:
            # the cost is evaluated inside the solver kernel: no cost matrix, one launch (lsa_ext)
            from ..lsa_ext import batched_assignment_fused
            return batched_assignment_fused(logits, boxes, gt, group_num, (self.cost_class, self.cost_bbox,
                                            self.cost_3dcenter, self.cost_giou)).long()
        C = self.cost_padded(logits, boxes, gt)
        if C.is_cuda and n <= 128 and K <= min(n, 64):
            from ..lsa_ext import batched_assignment
            return batched_assignment(C.float(), gt["num"], group_num).long()
        Ch = C.double().cpu().numpy()
        sizes = gt["num_host"] if gt.get("num_host") is not None else gt["num"].tolist()
        out = -np.ones((L, B, group_num, K), dtype=np.int64)
        for l in range(L):
            for b in range(B):
                k = int(sizes[b])
                for g in range(group_num):
                    if k:
                        r, c = linear_sum_assignment(Ch[l, b, g * n:(g + 1) * n, :k])
                        out[l, b, g, c] = r + g * n
        return torch.from_numpy(out).to(C.device)

    @torch.no_grad()
    def forward(self, outputs, targets, group_num=11):
        """outputs: {'pred_logits' [B,Q,C], 'pred_boxes' [B,Q,6]}; targets: list of dicts with 'labels',
        'boxes', 'boxes_3d'.  Returns a list (one per image) of (index_i, index_j)."""
        return self.match_layers([outputs], targets, group_num)[0]


def build_matcher(cfg):
    return HungarianMatcher(cost_class=cfg['set_cost_class'], cost_bbox=cfg['set_cost_bbox'],
                            cost_3dcenter=cfg['set_cost_3dcenter'], cost_giou=cfg['set_cost_giou'])
