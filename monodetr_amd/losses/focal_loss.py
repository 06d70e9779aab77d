"""``sigmoid_focal_loss`` -- mirror of /root/reference/lib/losses/focal_loss.py:69-94 (RetinaNet
focal loss on logits, mean over queries then sum, normalised by num_boxes)."""
import torch.nn.functional as F


def sigmoid_focal_loss(inputs, targets, num_boxes, alpha: float = 0.25, gamma: float = 2, per_layer: bool = False):
    """inputs / targets [B, Q, C] -> scalar; with per_layer=True [L, B, Q, C] -> [L] (the same formula
    for every leading slice)."""
    p = inputs.sigmoid()
    ce = F.binary_cross_entropy_with_logits(inputs, targets, reduction="none")
    p_t = p * targets + (1 - p) * (1 - targets)
    loss = ce * (1 - p_t) ** gamma
    if alpha >= 0:
        loss = (alpha * targets + (1 - alpha) * (1 - targets)) * loss
    if per_layer:
        return loss.mean(2).flatten(1).sum(1) / num_boxes
    return loss.mean(1).sum() / num_boxes
