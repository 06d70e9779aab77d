"""Autograd boundary of the MSDA operator -- mirror of
lib/models/monodetr/ops/functions/ms_deform_attn_func.py:21-38.

``MSDeformAttnFunction.apply(value, value_spatial_shapes, value_level_start_index,
sampling_locations, attention_weights, im2col_step)`` returns ``[B, Lq, M*D]`` and yields gradients
for arguments 0, 3 and 4 only, like the reference.  ``MSDA`` is the extension-module object
(reference: ``import MultiScaleDeformableAttention as MSDA``, :18); here it is
``monodetr_amd.msda_ext`` -- the gfx950 kernels behind the C ABI.  There is no Python/CPU
fallback in this file: CPU tensors raise "Not implemented on the CPU" exactly as the reference's
dispatcher does (ops/src/ms_deform_attn.h:38).  The reference's debug helper
``ms_deform_attn_core_pytorch`` (:41-61) is deliberately not re-exported from the product package;
its restatement lives in oracle/msda_torch_ref.py (test infrastructure).
"""
import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from .... import msda_ext as MSDA


class MSDeformAttnFunction(Function):
    # under autocast the projections feeding this op are bf16; the op itself computes in fp32
    # (the reference's extension is fp32/fp64 only, ms_deform_attn_cuda.cu:64, and would raise)
    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(ctx, value, value_spatial_shapes, value_level_start_index, sampling_locations,
                attention_weights, im2col_step):
        ctx.im2col_step = im2col_step
        out = MSDA.ms_deform_attn_forward(value, value_spatial_shapes, value_level_start_index,
                                          sampling_locations, attention_weights, im2col_step)
        ctx.save_for_backward(value, value_spatial_shapes, value_level_start_index,
                              sampling_locations, attention_weights)
        return out

    @staticmethod
    @once_differentiable
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(ctx, grad_output):
        value, shapes, level_start, loc, attn = ctx.saved_tensors
        g_value, g_loc, g_attn = MSDA.ms_deform_attn_backward(
            value, shapes, level_start, loc, attn, grad_output.contiguous(), ctx.im2col_step)
        return g_value, None, None, g_loc, g_attn, None
