"""Restatement of the reference's only CPU path for MSDA -- TEST INFRASTRUCTURE ONLY.

The reference has no native CPU kernel (ops/src/cpu/ms_deform_attn_cpu.cpp:17-40 raise); its CPU
path is ``ms_deform_attn_core_pytorch`` (ops/functions/ms_deform_attn_func.py:41-61): one
``F.grid_sample(bilinear, zeros, align_corners=False)`` per level on a (B*M, D, H, W) view of
value, then an attention-weighted sum over the L*P samples.  This file re-derives that
computation (it is what bench.py times as the op-level ``cpu_baseline`` of kind "port", because
/root/reference does not exist on the GPU box).  Differentiable through autograd.
"""
import torch
import torch.nn.functional as F


def msda_grid_sample(value, spatial_shapes, sampling_locations, attention_weights):
    B, S, M, D = value.shape
    _, Lq, _, L, P, _ = sampling_locations.shape
    hw = [(int(h), int(w)) for h, w in spatial_shapes]
    grids = sampling_locations * 2 - 1                      # [0,1] -> [-1,1] (func.py:46)
    per_level = []
    start = 0
    for lvl, (H, W) in enumerate(hw):
        v = value[:, start:start + H * W]                   # [B, HW, M, D]
        start += H * W
        v = v.permute(0, 2, 3, 1).reshape(B * M, D, H, W)   # image per (batch, head)
        g = grids[:, :, :, lvl].permute(0, 2, 1, 3, 4).reshape(B * M, Lq, P, 2)
        per_level.append(F.grid_sample(v, g, mode="bilinear", padding_mode="zeros", align_corners=False))
    sampled = torch.stack(per_level, dim=-2).reshape(B * M, D, Lq, L * P)
    w = attention_weights.permute(0, 2, 1, 3, 4).reshape(B * M, 1, Lq, L * P)
    out = (sampled * w).sum(-1)                              # [B*M, D, Lq]
    return out.reshape(B, M * D, Lq).transpose(1, 2).contiguous()


class GridSampleMSDA:
    """Stand-in for the extension module object (``MultiScaleDeformableAttention``, ops/src/vision.cpp:13-16) that runs the
    reference's CPU path -- `msda_grid_sample` forward, its autograd backward -- behind the extension's two functions, so that the
    whole model can be timed on the host the way the reference would run there (bench.py's ``cpu_baseline`` leg; test
    infrastructure, never on the product path).  The forward keeps its autograd graph until the matching backward call."""

    def __init__(self):
        self._graphs = {}

    def ms_deform_attn_forward(self, value, spatial_shapes, level_start_index, sampling_loc, attn_weight, im2col_step):
        leaves = [t.detach().requires_grad_(True) for t in (value, sampling_loc, attn_weight)]
        with torch.enable_grad():
            out = msda_grid_sample(leaves[0], spatial_shapes.tolist(), leaves[1], leaves[2])
        self._graphs[(value.data_ptr(), sampling_loc.data_ptr())] = (out, leaves)
        return out.detach()

    def ms_deform_attn_backward(self, value, spatial_shapes, level_start_index, sampling_loc, attn_weight, grad_output, im2col_step):
        out, leaves = self._graphs.pop((value.data_ptr(), sampling_loc.data_ptr()))
        return list(torch.autograd.grad(out, leaves, grad_output.to(out.dtype)))
