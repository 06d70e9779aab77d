"""Foreground/background balancing of the per-pixel depth loss -- mirror of
depth_predictor/ddn_loss/balancer.py (``Balancer`` :7-50, ``compute_fg_mask`` :53-81).
The per-box Python painting loops of the reference are replaced by one vectorised rasterisation
(`box_cover`); box corners follow the reference exactly: floor(x1,y1), ceil(x2,y2), then Python
slice semantics [v1:v2, u1:u2] including its treatment of negative indices."""
import torch
from torch import nn


def _slice_bounds(lo, hi, n):
    """Start/stop of the Python slice [lo:hi] on a length-n axis (negative indices wrap, then clip)."""
    start = torch.where(lo < 0, (lo + n).clamp(min=0), lo.clamp(max=n))
    stop = torch.where(hi < 0, (hi + n).clamp(min=0), hi.clamp(max=n))
    return start, stop


def box_cover(boxes_xyxy_long, H, W):
    """[K,4] integer (u1, v1, u2, v2) -> bool [K, H, W]: pixels the slice [v1:v2, u1:u2] selects."""
    u1, v1, u2, v2 = boxes_xyxy_long.unbind(-1)
    x0, x1 = _slice_bounds(u1, u2, W)
    y0, y1 = _slice_bounds(v1, v2, H)
    xs = torch.arange(W, device=boxes_xyxy_long.device).view(1, 1, W)
    ys = torch.arange(H, device=boxes_xyxy_long.device).view(1, H, 1)
    return (xs >= x0.view(-1, 1, 1)) & (xs < x1.view(-1, 1, 1)) & (ys >= y0.view(-1, 1, 1)) & (ys < y1.view(-1, 1, 1))


def integer_corners(gt_boxes2d, downsample_factor=1):
    b = gt_boxes2d / downsample_factor
    return torch.cat((torch.floor(b[:, :2]), torch.ceil(b[:, 2:])), 1).long()


def image_index(num_gt_per_img, device, num_images=None):
    """Image index of every box.  `num_gt_per_img` is the reference's per-image list of counts, or an
    int K meaning "every image has exactly K (padded) slots" -- the static-shape form, which needs
    `num_images` and involves no data-dependent sizes."""
    if isinstance(num_gt_per_img, int):
        return torch.arange(num_images, device=device).repeat_interleave(num_gt_per_img)
    counts = torch.as_tensor(num_gt_per_img, device=device)
    return torch.repeat_interleave(torch.arange(len(num_gt_per_img), device=device), counts)


def compute_fg_mask(gt_boxes2d, shape, num_gt_per_img, downsample_factor=1, device=torch.device("cpu")):
    """bool mask of `shape` [B, H, W], True inside any ground-truth box of the image."""
    B, H, W = shape
    fg = torch.zeros(shape, dtype=torch.bool, device=device)
    if gt_boxes2d.shape[0] == 0:
        return fg
    cover = box_cover(integer_corners(gt_boxes2d, downsample_factor), H, W)
    fg = torch.zeros((B, H, W), dtype=torch.int32, device=device).index_add_(
        0, image_index(num_gt_per_img, device, B), cover.to(torch.int32)) > 0
    return fg


class Balancer(nn.Module):
    def __init__(self, fg_weight, bg_weight, downsample_factor=1):
        super().__init__()
        self.fg_weight, self.bg_weight, self.downsample_factor = fg_weight, bg_weight, downsample_factor

    def forward(self, loss, gt_boxes2d, num_gt_per_img):
        """loss [B, H, W] per-pixel -> scalar: (fg_w * sum_fg + bg_w * sum_bg) / num_pixels."""
        fg = compute_fg_mask(gt_boxes2d, loss.shape, num_gt_per_img, self.downsample_factor, loss.device)
        weights = torch.where(fg, float(self.fg_weight), float(self.bg_weight)).to(loss.dtype)
        return (loss * weights).sum() / fg.numel()
