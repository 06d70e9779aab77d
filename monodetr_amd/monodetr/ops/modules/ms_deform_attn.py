"""``MSDeformAttn`` -- mirror of lib/models/monodetr/ops/modules/ms_deform_attn.py:69-162.

Same constructor, parameter names (``sampling_offsets``, ``attention_weights``, ``value_proj``,
``output_proj``), ``_reset_parameters`` initialisation (:106-120), forward signature (:122) and
numerics; the sampling + aggregation itself is the gfx950 operator behind
``MSDeformAttnFunction``.  ``MSDeformAttn_cross`` (:164-256, unused by the model) and
``MultiheadAttention`` (:259-379, imported by depthaware_transformer.py:11 but never
instantiated) are exported for import compatibility.
"""
import math
import os
import warnings

import torch
import torch.nn.functional as F
from torch import nn

from ..functions import MSDeformAttnFunction
from ... import _cut
from ...linear import Linear, token_linear, token_linear_skip
from .... import msda_prologue_ext


# MDETR_MSDA_PROLOGUE=1: fused softmax + sampling-location kernel (off until its first GPU validation,
# tests/test_fused_gpu.py)
_FUSED_PROLOGUE = os.environ.get("MDETR_MSDA_PROLOGUE") == "1"
# with the prologue kernel: the sampling-offset and attention-weight projections as one GEMM (False: two; tests)
_PACKED_PROJECTION = True
# fp32 values for the decoder's deformable cross-attention in a bf16 model (False: bf16 values; tests)
_WIDE_CROSS_VALUE = True



def _check_token_count(spatial_shapes, S):
    """sum(H_l * W_l) == S (reference :136).  The reference evaluates this on the device tensor every call (a
    device->host sync); here once per (tensor object, version, S) -- remembered ON the tensor, not by its address,
    which the caching allocator re-issues to later tensors."""
    seen = getattr(spatial_shapes, "_mdetr_token_count", None)
    if seen != (spatial_shapes._version, S):
        assert int((spatial_shapes[:, 0] * spatial_shapes[:, 1]).sum()) == S
        try:
            spatial_shapes._mdetr_token_count = (spatial_shapes._version, S)
        except AttributeError:
            pass


def _is_power_of_2(n):
    if not isinstance(n, int) or n < 0:
        raise ValueError("invalid input for _is_power_of_2: {} (type: {})".format(n, type(n)))
    return n != 0 and (n & (n - 1)) == 0


def _star_offsets(n_heads, n_levels, n_points):
    """Bias initialisation of ``sampling_offsets``: one direction per head on the unit square's
    boundary (max-norm 1), scaled by the point index 1..P (reference :108-114)."""
    theta = torch.arange(n_heads, dtype=torch.float32) * (2.0 * math.pi / n_heads)
    d = torch.stack([theta.cos(), theta.sin()], -1)
    d = d / d.abs().max(-1, keepdim=True)[0]
    scale = torch.arange(1, n_points + 1, dtype=torch.float32).view(1, 1, n_points, 1)
    return (d.view(n_heads, 1, 1, 2) * scale).expand(n_heads, n_levels, n_points, 2).reshape(-1)


class MSDeformAttn(nn.Module):
    def __init__(self, d_model=256, n_levels=4, n_heads=8, n_points=4, conditional=False):
        super().__init__()
        if d_model % n_heads != 0:
            raise ValueError("d_model must be divisible by n_heads, but got {} and {}".format(d_model, n_heads))
        if not _is_power_of_2(d_model // n_heads):
            warnings.warn("MSDeformAttn: head dimension %d is not a power of 2; the gfx950 fast path "
                          "needs 32 channels per head, other sizes take the generic kernel." % (d_model // n_heads))
        self.im2col_step = 64
        self.d_model, self.n_levels, self.n_heads, self.n_points = d_model, n_levels, n_heads, n_points
        self.conditional = conditional
        d_value = d_model // 2 if conditional else d_model
        self.sampling_offsets = Linear(d_model, n_heads * n_levels * n_points * 2)
        self.attention_weights = Linear(d_model, n_heads * n_levels * n_points)
        self.value_proj = Linear(d_value, d_value)
        self.output_proj = Linear(d_value, d_value)
        self._reset_parameters()

    def _reset_parameters(self):
        with torch.no_grad():
            self.sampling_offsets.weight.zero_()
            self.sampling_offsets.bias = nn.Parameter(_star_offsets(self.n_heads, self.n_levels, self.n_points))
            self.attention_weights.weight.zero_()
            self.attention_weights.bias.zero_()
            nn.init.xavier_uniform_(self.value_proj.weight)
            self.value_proj.bias.zero_()
            nn.init.xavier_uniform_(self.output_proj.weight)
            self.output_proj.bias.zero_()

    def forward(self, query, reference_points, input_flatten, input_spatial_shapes,
                input_level_start_index, input_padding_mask=None, chain_input=False):
        """query [N,Lq,C]; reference_points [N,Lq,L,2] or [N,Lq,L,6] (cx,cy,l,r,t,b);
        input_flatten [N,S,C]; input_spatial_shapes [L,2] (H,W) int64; input_level_start_index [L];
        input_padding_mask [N,S] bool (True = padding).  Returns [N,Lq,C].
        chain_input (not in the reference's signature): -> (output, input_flatten'), the second == input_flatten, for the NEXT reader
        of the same tensor -- the decoder's layers all read the encoder's memory, and three readers in parallel mean two 42 MB
        sums of their gradients; read in a chain, each value projection adds the gradient arriving from the readers behind it
        inside its input-gradient product (`linear.token_linear_skip`)."""
        N, Lq, _ = query.shape
        S = input_flatten.shape[1]
        M, L, P = self.n_heads, self.n_levels, self.n_points
        _check_token_count(input_spatial_shapes, S)

        # cross-attention of a few queries into the pyramid (the decoder: 550 queries, S = 10 200) in a bf16 model: fp32 values --
        # the projection's fp32 accumulator written out unrounded -- through the fp32 operator.  Its d/d(location) differences
        # neighbouring value rows; with bf16-rounded values those were the least accurate gradients of the bf16 model.
        wide = _WIDE_CROSS_VALUE and input_flatten.is_cuda and input_flatten.dtype == torch.bfloat16 and Lq * 8 <= S
        if chain_input:
            value, input_next = token_linear_skip(input_flatten, self.value_proj.weight, self.value_proj.bias, wide_out=wide)
        else:
            value, input_next = token_linear(input_flatten, self.value_proj.weight, self.value_proj.bias, wide_out=wide), input_flatten
        if input_padding_mask is not None:
            value = value.masked_fill(input_padding_mask[..., None], 0.0)
        value = value.view(N, S, M, -1)

        if reference_points.shape[-1] not in (2, 6):
            raise ValueError("Last dim of reference_points must be 2 or 4, but get {} instead.".format(
                reference_points.shape[-1]))
        packed = offsets = logits = None
        if _FUSED_PROLOGUE and _PACKED_PROJECTION and not torch.is_autocast_enabled():
            packed = token_linear(query, *self._packed_projection())
        else:
            offsets = token_linear(query, self.sampling_offsets.weight, self.sampling_offsets.bias).view(N, Lq, M, L, P, 2)
            logits = token_linear(query, self.attention_weights.weight, self.attention_weights.bias).view(N, Lq, M, L * P)
        out = self._attend(value, packed, offsets, logits, reference_points, input_spatial_shapes, input_level_start_index, query.dtype)
        return (out, input_next) if chain_input else out

    def _packed_projection(self):
        """The two projections of the query as ONE GEMM (384 = 256 offset + 128 logit columns): one read of the query forward,
        one input-gradient GEMM and one weight-gradient GEMM backward, and no sum of two query gradients."""
        return (torch.cat((self.sampling_offsets.weight, self.attention_weights.weight), 0),
                torch.cat((self.sampling_offsets.bias, self.attention_weights.bias), 0))

    def _attend(self, value, packed, offsets, logits, reference_points, input_spatial_shapes, input_level_start_index, out_dtype):
        N, S = value.shape[:2]
        M, L, P = self.n_heads, self.n_levels, self.n_points
        Lq = (packed if packed is not None else offsets).shape[1]
        if packed is not None and not msda_prologue_ext.packed_supported(packed, reference_points, L, P):
            offsets, logits = packed[..., :M * L * P * 2].reshape(N, Lq, M, L, P, 2), packed[..., M * L * P * 2:].reshape(N, Lq, M, L * P)
            packed = None
        if packed is not None:
            locations, weights = msda_prologue_ext.msda_prologue_packed(packed, reference_points, input_spatial_shapes, M, L, P)
        elif _FUSED_PROLOGUE and msda_prologue_ext.supported(offsets, logits, reference_points):
            # softmax + sampling-location arithmetic in one fp32 launch (csrc/msda_prologue.hip)
            locations, weights = msda_prologue_ext.msda_prologue(offsets, logits, reference_points, input_spatial_shapes)
        else:
            weights = F.softmax(logits, -1).view(N, Lq, M, L, P)
            ref = reference_points[:, :, None, :, None, :]
            if reference_points.shape[-1] == 2:
                wh = input_spatial_shapes.flip(-1)                                  # (W_l, H_l), :150
                locations = ref + offsets / wh[None, None, None, :, None, :]
            else:
                extent = ref[..., 2::2] + ref[..., 3::2]                            # (l+r, t+b), :153-155
                locations = ref[..., :2] + offsets / P * extent * 0.5

        out = MSDeformAttnFunction.apply(value, input_spatial_shapes, input_level_start_index,
                                         locations, weights, self.im2col_step)
        out = _cut.at("msda", out, when_armed=True)                   # (a two-graph iteration may start its second graph here)
        if out.dtype != out_dtype and out_dtype == torch.bfloat16:
            out = out.to(out_dtype)                                    # (the fp32 operator of the wide form)
        return token_linear(out, self.output_proj.weight, self.output_proj.bias)

    def forward_self(self, src, pos, reference_points, input_spatial_shapes, input_level_start_index, input_padding_mask=None):
        """``forward(src + pos, reference_points, src, ...)`` -- self-attention over the pyramid, the encoder's call (reference
        depthaware_transformer.py:339-341) -- together with the tensor the residual connection continues from:
        -> (output, src').  `src` feeds the value projection, the query (after + pos) and the residual; with the packed
        projection both GEMMs take `linear.token_linear_skip`, whose backward adds its input gradient into the gradient
        arriving over the residual path inside the GEMM (two 126 MB elementwise sums per layer otherwise)."""
        if not (_FUSED_PROLOGUE and _PACKED_PROJECTION and src.is_cuda and not torch.is_autocast_enabled()
                and reference_points.shape[-1] in (2, 6)):
            return self.forward(src if pos is None else src + pos, reference_points, src, input_spatial_shapes,
                                input_level_start_index, input_padding_mask), _cut.at("msda", src, when_armed=True)
        N, S, _ = src.shape
        _check_token_count(input_spatial_shapes, S)
        value, src1 = token_linear_skip(src, self.value_proj.weight, self.value_proj.bias)
        if input_padding_mask is not None:
            value = value.masked_fill(input_padding_mask[..., None], 0.0)
        packed, src2 = token_linear_skip(src1, *self._packed_projection(), pos=pos)
        out = self._attend(value.view(N, S, self.n_heads, -1), packed, None, None, reference_points, input_spatial_shapes,
                           input_level_start_index, src.dtype)
        return out, _cut.at("msda", src2, when_armed=True)


class MSDeformAttn_cross(MSDeformAttn):
    """Half-width-value variant the reference defines at :164-256 and never instantiates."""

    def __init__(self, d_model=256, n_levels=4, n_heads=8, n_points=4):
        super().__init__(d_model, n_levels, n_heads, n_points, conditional=True)


# The reference vendors a copy of torch's MultiheadAttention (:259-379) that the model imports but
# never constructs; the stock module has the same parameters.
MultiheadAttention = nn.MultiheadAttention
