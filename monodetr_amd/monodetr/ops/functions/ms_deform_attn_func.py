"""Autograd boundary of the MSDA operator -- mirror of
lib/models/monodetr/ops/functions/ms_deform_attn_func.py:21-38.

``MSDeformAttnFunction.apply(value, value_spatial_shapes, value_level_start_index,
sampling_locations, attention_weights, im2col_step)`` returns ``[B, Lq, M*D]`` and yields gradients
for arguments 0, 3 and 4 only, like the reference.  ``MSDA`` is the extension-module object
(reference: ``import MultiScaleDeformableAttention as MSDA``, :18); here it is
``monodetr_amd.msda_ext`` -- the gfx950 kernels behind the C ABI.  There is no Python/CPU
fallback in this file: CPU tensors raise "Not implemented on the CPU" exactly as the reference's
dispatcher does (ops/src/ms_deform_attn.h:38).

``ms_deform_attn_core_pytorch(value, value_spatial_shapes, sampling_locations, attention_weights)`` is the reference's
"for debug and test only" helper (:41-61, imported by its ops/test.py:19): the same operator spelled with
``F.grid_sample``, any float dtype, CPU or GPU, differentiable through autograd.  It is part of this module's surface
like in the reference and, like there, nothing in the model calls it -- ``MSDeformAttnFunction`` never falls back to it.
"""
import os

import torch
import torch.nn.functional as F
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
        ctx.native_bf16 = _NATIVE_BF16 and value.dtype == torch.bfloat16 and MSDA.bf16_supported(value, sampling_locations)
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


def ms_deform_attn_core_pytorch(value, value_spatial_shapes, sampling_locations, attention_weights):
    """Reference :41-61.  value [N, S, M, D]; value_spatial_shapes iterable of (H, W); sampling_locations
    [N, Lq, M, L, P, 2] as (x, y) in [0, 1]; attention_weights [N, Lq, M, L, P]  ->  [N, Lq, M*D].
    One bilinear ``grid_sample`` (zero padding, align_corners=False) per level over a (N*M, D, H, W) view of that
    level's tokens, then the attention-weighted sum over the L*P samples."""
    N, S, M, D = value.shape
    Lq, L, P = sampling_locations.shape[1], sampling_locations.shape[3], sampling_locations.shape[4]
    sizes = [(int(h), int(w)) for h, w in value_spatial_shapes]
    levels = value.split([h * w for h, w in sizes], dim=1)
    grid = 2 * sampling_locations - 1                                         # grid_sample's [-1, 1] convention (:46)
    sampled = []
    for lvl, (h, w) in enumerate(sizes):
        image = levels[lvl].flatten(2).transpose(1, 2).reshape(N * M, D, h, w)                  # :49-51
        where = grid[:, :, :, lvl].transpose(1, 2).flatten(0, 1)                                 # [N*M, Lq, P, 2] (:53)
        sampled.append(F.grid_sample(image, where, mode='bilinear', padding_mode='zeros', align_corners=False))   # :55-56
    weights = attention_weights.transpose(1, 2).reshape(N * M, 1, Lq, L * P)                    # :59
    out = (torch.stack(sampled, dim=-2).flatten(-2) * weights).sum(-1).view(N, M * D, Lq)       # :60
    return out.transpose(1, 2).contiguous()
