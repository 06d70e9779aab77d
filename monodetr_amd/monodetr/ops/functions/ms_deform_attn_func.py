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
import os

import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from .... import msda_ext as MSDA


_LOW = (torch.bfloat16, torch.float16)
# MDETR_MSDA_BF16=1: a bf16 model's value / output / grad_output go through the mixed-precision kernels as they are
# (include/monodetr_amd.h, mdetr_msda_forward_bf16) instead of being widened to fp32 around the call.  Off until the
# kernels have run on the GPU.
_NATIVE_BF16 = os.environ.get("MDETR_MSDA_BF16") == "1"


class MSDeformAttnFunction(Function):
    # The operator computes in fp32 (the reference's extension is fp32/fp64 only,
    # ms_deform_attn_cuda.cu:64, and would raise on anything else).  bf16 / fp16 inputs -- under
    # autocast or in a bf16 model -- are widened on the way in; the output and the gradients come
    # back in the dtypes of the corresponding inputs.
    @staticmethod
    def forward(ctx, value, value_spatial_shapes, value_level_start_index, sampling_locations,
                attention_weights, im2col_step):
        ctx.im2col_step = im2col_step
        ctx.in_dtypes = (value.dtype, sampling_locations.dtype, attention_weights.dtype)
        ctx.native_bf16 = _NATIVE_BF16 and MSDA.bf16_supported(value, sampling_locations)
        if ctx.native_bf16:
            value, sampling_locations, attention_weights = value.contiguous(), sampling_locations.float(), attention_weights.float()
            out = MSDA.ms_deform_attn_forward_bf16(value, value_spatial_shapes, value_level_start_index,
                                                   sampling_locations, attention_weights)
            ctx.save_for_backward(value, value_spatial_shapes, value_level_start_index,
                                  sampling_locations, attention_weights)
            return out
        if value.dtype in _LOW or sampling_locations.dtype in _LOW or attention_weights.dtype in _LOW:
            value, sampling_locations, attention_weights = value.float(), sampling_locations.float(), attention_weights.float()
        elif not (value.dtype == sampling_locations.dtype == attention_weights.dtype):
            wide = torch.promote_types(torch.promote_types(value.dtype, sampling_locations.dtype), attention_weights.dtype)
            value, sampling_locations, attention_weights = value.to(wide), sampling_locations.to(wide), attention_weights.to(wide)
        out = MSDA.ms_deform_attn_forward(value, value_spatial_shapes, value_level_start_index,
                                          sampling_locations, attention_weights, im2col_step)
        ctx.save_for_backward(value, value_spatial_shapes, value_level_start_index,
                              sampling_locations, attention_weights)
        return out.to(ctx.in_dtypes[0]) if ctx.in_dtypes[0] in _LOW else out

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_output):
        value, shapes, level_start, loc, attn = ctx.saved_tensors
        dv, dl, da = ctx.in_dtypes
        if ctx.native_bf16:
            g_value, g_loc, g_attn = MSDA.ms_deform_attn_backward_bf16(
                value, shapes, level_start, loc, attn, grad_output.to(torch.bfloat16).contiguous())
            return g_value.to(dv), None, None, g_loc.to(dl), g_attn.to(da), None
        g_value, g_loc, g_attn = MSDA.ms_deform_attn_backward(
            value, shapes, level_start, loc, attn, grad_output.to(value.dtype).contiguous(), ctx.im2col_step)
        return g_value.to(dv), None, None, g_loc.to(dl), g_attn.to(da), None
