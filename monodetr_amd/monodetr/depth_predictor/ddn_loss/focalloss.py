"""Multi-class focal loss -- mirror of depth_predictor/ddn_loss/focalloss.py (``one_hot`` :12-55,
``focal_loss`` :58-129, ``FocalLoss`` :132-179).  Note the reference's one-hot carries +eps = 1e-6
on every class (:55), so every class contributes eps * focal_c besides the target class; kept."""
import torch
import torch.nn.functional as F
from torch import nn


def one_hot(labels, num_classes, device=None, dtype=None, eps=1e-6):
    if not labels.dtype == torch.int64:
        raise ValueError(f"labels must be of the same dtype torch.int64. Got: {labels.dtype}")
    shape = labels.shape
    hot = torch.zeros((shape[0], num_classes) + shape[1:], device=device, dtype=dtype)
    return hot.scatter_(1, labels.unsqueeze(1), 1.0) + eps


def focal_loss(input, target, alpha, gamma=2.0, reduction='none', eps=None):
    """input [N, C, *] logits, target [N, *] int64 -> [N, *] (reduction 'none')."""
    if input.size(0) != target.size(0) or target.size()[1:] != input.size()[2:]:
        raise ValueError(f'Expected target size {(input.size(0),) + input.size()[2:]}, got {target.size()}')
    logp = F.log_softmax(input, dim=1)
    focal = -alpha * torch.pow(1.0 - logp.exp(), gamma) * logp
    # sum_c (onehot_c + 1e-6) * focal_c  without materialising the one-hot tensor
    loss = focal.gather(1, target.unsqueeze(1)).squeeze(1) + 1e-6 * focal.sum(1)
    if reduction == 'none':
        return loss
    if reduction == 'mean':
        return loss.mean()
    if reduction == 'sum':
        return loss.sum()
    raise NotImplementedError(f"Invalid reduction mode: {reduction}")


class FocalLoss(nn.Module):
    def __init__(self, alpha, gamma=2.0, reduction='none', eps=None):
        super().__init__()
        self.alpha, self.gamma, self.reduction, self.eps = alpha, gamma, reduction, eps

    def forward(self, input, target):
        return focal_loss(input, target, self.alpha, self.gamma, self.reduction, self.eps)
