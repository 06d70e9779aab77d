"""Depth-map classification loss -- mirror of depth_predictor/ddn_loss/ddn_loss.py (``DDNLoss``
:12-127).  Target: every pixel inside a ground-truth 2D box takes that object's centre depth
(nearest object wins where boxes overlap -- the reference paints far-to-near, :52-62), binned
with linear-increasing discretisation (:64-101); focal loss per pixel, fg/bg balanced.
No ``torch.cuda.current_device()`` at construction (reference :32 makes CPU construction fail)."""
import math

import torch
from torch import nn

from .balancer import Balancer, box_cover, image_index, integer_corners
from .focalloss import FocalLoss


class DDNLoss(nn.Module):
    def __init__(self, alpha=0.25, gamma=2.0, fg_weight=13, bg_weight=1, downsample_factor=1):
        super().__init__()
        self.balancer = Balancer(downsample_factor=downsample_factor, fg_weight=fg_weight, bg_weight=bg_weight)
        self.alpha, self.gamma = alpha, gamma
        self.loss_func = FocalLoss(alpha=self.alpha, gamma=self.gamma, reduction="none")

    def build_target_depth_from_3dcenter(self, depth_logits, gt_boxes2d, gt_center_depth, num_gt_per_img, valid=None):
        B, _, H, W = depth_logits.shape
        maps = torch.zeros((B, H, W), device=depth_logits.device, dtype=depth_logits.dtype)
        if gt_boxes2d.shape[0] == 0:
            return maps
        cover = box_cover(integer_corners(gt_boxes2d), H, W)                       # [K, H, W]
        if valid is not None:
            cover = cover & valid.view(-1, 1, 1)
        depth = gt_center_depth.to(maps.dtype).view(-1, 1, 1)
        painted = torch.where(cover, depth, torch.full_like(depth, float("inf")).expand_as(cover))
        nearest = torch.full((B, H, W), float("inf"), device=maps.device, dtype=maps.dtype)
        nearest.index_reduce_(0, image_index(num_gt_per_img, maps.device, B), painted, "amin")
        return torch.where(torch.isinf(nearest), maps, nearest)

    def bin_depths(self, depth_map, mode="LID", depth_min=1e-3, depth_max=60, num_bins=80, target=False):
        if mode == "UD":
            indices = (depth_map - depth_min) / ((depth_max - depth_min) / num_bins)
        elif mode == "LID":
            bin_size = 2 * (depth_max - depth_min) / (num_bins * (1 + num_bins))
            indices = -0.5 + 0.5 * torch.sqrt(1 + 8 * (depth_map - depth_min) / bin_size)
        elif mode == "SID":
            indices = num_bins * (torch.log(1 + depth_map) - math.log(1 + depth_min)) / \
                (math.log(1 + depth_max) - math.log(1 + depth_min))
        else:
            raise NotImplementedError
        if target:
            bad = (indices < 0) | (indices > num_bins) | (~torch.isfinite(indices))
            indices = torch.where(bad, torch.full_like(indices, num_bins), indices).type(torch.int64)
        return indices

    def forward(self, depth_logits, gt_boxes2d, num_gt_per_img, gt_center_depth, valid=None):
        """depth_logits [B, D+1, H, W]; gt_boxes2d [K, 4] xyxy in depth-map pixels (all images
        concatenated); num_gt_per_img: list of per-image counts, or an int K for the padded static form
        (then `valid` [B*K] marks the real boxes and padded boxes are all-zero); gt_center_depth [K]."""
        target = self.bin_depths(self.build_target_depth_from_3dcenter(
            depth_logits, gt_boxes2d, gt_center_depth, num_gt_per_img, valid), target=True)
        return self.balancer(loss=self.loss_func(depth_logits, target), gt_boxes2d=gt_boxes2d,
                             num_gt_per_img=num_gt_per_img)
