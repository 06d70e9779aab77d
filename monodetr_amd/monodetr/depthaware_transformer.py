"""Depth-aware transformer -- mirror of lib/models/monodetr/depthaware_transformer.py
(``DepthAwareTransformer`` :68-312, ``VisualEncoderLayer`` :315-354, ``VisualEncoder`` :357-384,
``DepthAwareDecoderLayer`` :387-515, ``DepthAwareDecoder`` :518-626, ``build_depthaware_transformer``
:644-660).  Same class names, constructor arguments, parameter names and forward signatures.

What is different underneath (MI355X-first, numerics unchanged):
  * deformable attention is the gfx950 operator (ops/), dense attention the fused core (attention.py);
  * activations stay batch-first [B, L, C]; the group fold of the decoder self-attention
    (reference :480-503: split the 11 query groups and concatenate them along the batch) becomes a
    free view [B, G*n, C] -> [B*G, n, C]; the hard-coded 50 (:481-482) is n = queries // groups;
  * level shapes travel as Python ints next to the int64 device tensor, so building reference
    points does not iterate a CUDA tensor (one device sync per element in the reference, :366);
    constant per-resolution tensors are cached;
  * padding masks known to be all-False by construction (utils.misc.no_padding) skip the
    masked_fill / valid-ratio / key-padding work;
  * ``sa_v_proj`` is kept as a parameter (checkpoints) but its GEMM, whose result the reference
    discards (:471 vs :477), is not executed.
The two-stage / DAB / DINO branches (all off in configs/monodetr.yaml:50-54) are not built.
"""
import copy

import torch
import torch.nn.functional as F
from torch import nn
from torch.nn.init import constant_, normal_, xavier_uniform_

from .. import head_tail_ext
from ..add_ln_ext import residual_layernorm
from . import _cut
from ..utils.misc import inverse_sigmoid, no_padding
from .attention import MultiheadAttention as FusedMultiheadAttention
from .heads import heads_level
from .. import _tune

# the decoder's layers read the encoder's memory in a chain (MSDeformAttn.forward, chain_input); MDETR_TUNE=decoder_chain=0: A-B runs
_CHAIN_MEMORY = _tune.get("decoder_chain", "1") != "0"
from .linear import Linear, ffn_hidden, token_linear
from .ops.modules import MSDeformAttn, MSDeformAttn_cross, MultiheadAttention  # noqa: F401  (reference :11)


class MLP(nn.Module):
    """Linear -> ReLU -> ... -> Linear."""

    def __init__(self, input_dim, hidden_dim, output_dim, num_layers):
        super().__init__()
        self.num_layers = num_layers
        dims = [input_dim] + [hidden_dim] * (num_layers - 1) + [output_dim]
        self.layers = nn.ModuleList(Linear(a, b) for a, b in zip(dims[:-1], dims[1:]))

    def forward(self, x):
        x = x.to(self.layers[0].weight.dtype)       # fp32 heads may sit behind a bf16 body
        for layer in self.layers[:-1]:
            x = F.relu(layer(x))
        return self.layers[-1](x)


def fused_first_layers(x, heads):
    """The first Linear of several heads that read the same tensor as ONE GEMM (weights concatenated along the output
    axis): `heads` are MLPs or plain nn.Linear modules.  Returns the per-head slices of the result (pre-activation).  The
    prediction heads are five tiny fp32 products per decoder level on the same [B * Q, 256] tensor; separately each costs a
    25-40 microsecond library launch three times over (forward, input gradient, weight gradient)."""
    firsts = [h.layers[0] if isinstance(h, MLP) else h for h in heads]
    x = x.to(firsts[0].weight.dtype)                # fp32 heads may sit behind a bf16 body: one cast for all of them
    widths = [l.out_features for l in firsts]
    pad = -sum(widths) % 8                          # column sums (the bias gradient, csrc/colsum.hip) want whole 16-byte vectors
    ws, bs = [l.weight for l in firsts], [l.bias for l in firsts]
    if pad:
        zw, zb = _zero_rows(ws[0], pad)
        ws.append(zw)
        bs.append(zb)
    y = token_linear(x, torch.cat(ws, 0), torch.cat(bs, 0))
    return y.split(widths + ([pad] if pad else []), -1)[:len(widths)]


_ZERO_ROWS = {}


def _zero_rows(like, rows):
    """(zeros [rows, like.shape[1]], zeros [rows]) in like's dtype on its device: constants, made once (two fill launches per call
    otherwise)."""
    key = (like.device, like.dtype, rows, like.shape[1])
    hit = _ZERO_ROWS.get(key)
    if hit is None:
        hit = _ZERO_ROWS[key] = (like.new_zeros(rows, like.shape[1]), like.new_zeros(rows))
    return hit


def mlp_rest(mlp, h):
    """The remainder of MLP.forward given the output `h` of its first Linear."""
    if mlp.num_layers == 1:
        return h
    x = F.relu(h)
    for layer in mlp.layers[1:-1]:
        x = F.relu(layer(x))
    return mlp.layers[-1](x)


class _SummedPair(torch.autograd.Function):
    """([W1 + W2; W3 + W4], [b1 + b2; b3 + b4]): two pairs of projections that act on the same input, summed in weight space and
    packed for one GEMM.  Backward: the packed gradient's two halves, each handed to BOTH members of its pair -- views, nothing is
    computed.  (As plain `+` / `cat` the same gradient tensor reaches two AccumulateGrad nodes, the first of which has to clone it;
    `chunk_sums.flush()` makes sure the values exist before anything downstream may read them.)"""

    @staticmethod
    def forward(ctx, w1, w2, w3, w4, b1, b2, b3, b4):
        ctx.rows = w1.shape[0]
        return torch.cat((w1 + w2, w3 + w4), 0), torch.cat((b1 + b2, b3 + b4), 0)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, gw, gb):
        from .. import chunk_sums
        chunk_sums.flush()
        n = ctx.rows
        return gw[:n], gw[:n], gw[n:], gw[n:], gb[:n], gb[:n], gb[n:], gb[n:]


class _LevelPos(torch.autograd.Function):
    """cat_l(flatten(pos_l) + level_embed[l]) along the token axis, as ``dtype`` (reference depthaware_transformer.py:215-218).
    The sine embeddings are constants; the only gradient is level_embed's, a sum over the batch and the level's tokens.  As
    the backward of four broadcast adds that is four generic reductions over [B, HW_l, C] slices of the concatenated
    gradient (260 + 144 + ... microseconds at B = 8); here one streaming pass sums the batch (csrc/colsum.hip on the
    [B, S * C] view) and four small column sums finish the levels."""

    @staticmethod
    def forward(ctx, level_embed, dtype, *pos):
        ctx.counts = [p.shape[-2] * p.shape[-1] for p in pos]
        ctx.embed_dtype = level_embed.dtype
        ctx.pos_meta = [(p.shape, p.dtype) for p in pos]
        return torch.cat([p.flatten(2).transpose(1, 2) + level_embed[l].view(1, 1, -1) for l, p in enumerate(pos)], 1).to(dtype)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):
        from .. import colsum_ext
        B, S, C = g.shape
        g = g.contiguous()
        flat = g.view(B, S * C)
        wide = torch.float64 if g.dtype == torch.float64 else torch.float32
        over_batch = (colsum_ext.column_sum(flat) if colsum_ext.supported(flat) else flat.to(wide).sum(0)).view(S, C)
        rows, start = [], 0
        for n in ctx.counts:
            part = over_batch[start:start + n]
            rows.append(colsum_ext.column_sum(part) if colsum_ext.supported(part) else part.sum(0))
            start += n
        # a LEARNED position embedding (position_embedding: learned / v3, position_encoding.py:58-84) needs its gradient too:
        # the slice of g that belongs to the level, back in the map's layout (the sine embedding is a constant: None)
        gpos, start = [], 0
        for l, (n, (shape, dt)) in enumerate(zip(ctx.counts, ctx.pos_meta)):
            gpos.append(g[:, start:start + n].transpose(1, 2).reshape(shape).to(dt) if ctx.needs_input_grad[2 + l] else None)
            start += n
        return (torch.stack(rows).to(ctx.embed_dtype), None) + tuple(gpos)


def _get_clones(module, N):
    return nn.ModuleList([copy.deepcopy(module) for _ in range(N)])


def _get_activation_fn(activation):
    try:
        return {"relu": F.relu, "gelu": F.gelu, "glu": F.glu}[activation]
    except KeyError:
        raise RuntimeError(f"activation should be relu/gelu, not {activation}.")


def _shape_list(spatial_shapes):
    """[(H, W), ...] as Python ints (one host copy if a device tensor is given)."""
    if torch.is_tensor(spatial_shapes):
        return [tuple(int(v) for v in hw) for hw in spatial_shapes.tolist()]
    return [tuple(int(v) for v in hw) for hw in spatial_shapes]


class VisualEncoderLayer(nn.Module):
    def __init__(self, d_model=256, d_ffn=1024, dropout=0.1, activation="relu", n_levels=4, n_heads=8, n_points=4):
        super().__init__()
        self.self_attn = MSDeformAttn(d_model, n_levels, n_heads, n_points)
        self.dropout1 = nn.Dropout(dropout)
        self.norm1 = nn.LayerNorm(d_model)
        self.linear1 = Linear(d_model, d_ffn)
        self.activation = _get_activation_fn(activation)
        self.dropout2 = nn.Dropout(dropout)
        self.linear2 = Linear(d_ffn, d_model)
        self.dropout3 = nn.Dropout(dropout)
        self.norm2 = nn.LayerNorm(d_model)

    @staticmethod
    def with_pos_embed(tensor, pos):
        return tensor if pos is None else tensor + pos

    def forward_ffn(self, src):
        # (skip=True: the GEMM's input comes back as the tensor the residual continues from -- its two gradients meet inside
        #  the input-gradient GEMM, linear.token_linear_skip)
        h, src = ffn_hidden(src, self.linear1, self.dropout2, self.activation, skip=True)   # ReLU (+ Dropout) fused behind the GEMM where possible
        ff = token_linear(h, self.linear2.weight, self.linear2.bias)
        return residual_layernorm(src, ff, self.norm2, self.dropout3)

    def forward(self, src, pos, reference_points, spatial_shapes, level_start_index, padding_mask=None):
        attn, src = self.self_attn.forward_self(src, pos, reference_points, spatial_shapes, level_start_index, padding_mask)
        return self.forward_ffn(residual_layernorm(src, attn, self.norm1, self.dropout1))


class VisualEncoder(nn.Module):
    def __init__(self, encoder_layer, num_layers):
        super().__init__()
        self.layers = _get_clones(encoder_layer, num_layers)
        self.num_layers = num_layers
        self._ref_cache = {}

    @staticmethod
    def get_reference_points(spatial_shapes, valid_ratios, device):
        """Pixel centres of every level, normalised to [0,1] and rescaled by the valid ratios:
        [B, S, L, 2] (reference :364-376)."""
        per_level = []
        for lvl, (H, W) in enumerate(_shape_list(spatial_shapes)):
            ys = torch.linspace(0.5, H - 0.5, H, dtype=torch.float32, device=device)
            xs = torch.linspace(0.5, W - 0.5, W, dtype=torch.float32, device=device)
            gy, gx = torch.meshgrid(ys, xs, indexing="ij")
            gy = gy.reshape(-1)[None] / (valid_ratios[:, None, lvl, 1] * H)
            gx = gx.reshape(-1)[None] / (valid_ratios[:, None, lvl, 0] * W)
            per_level.append(torch.stack((gx, gy), -1))
        return torch.cat(per_level, 1)[:, :, None] * valid_ratios[:, None]

    def _unpadded_reference_points(self, shapes, B, L, device):
        key = (tuple(shapes), L, str(device))
        ref = self._ref_cache.get(key)
        if ref is None:
            ones = torch.ones((1, L, 2), dtype=torch.float32, device=device)
            ref = self._ref_cache[key] = self.get_reference_points(shapes, ones, device)
        return ref.expand(B, -1, -1, -1)

    def forward(self, src, spatial_shapes, level_start_index, valid_ratios, pos=None, padding_mask=None,
                ref_token_index=None, ref_token_coord=None, shape_list=None):
        shapes = shape_list if shape_list is not None else _shape_list(spatial_shapes)
        if valid_ratios is None:      # no padding anywhere: ratios are identically 1
            reference_points = self._unpadded_reference_points(shapes, src.shape[0], len(shapes), src.device)
        else:
            reference_points = self.get_reference_points(shapes, valid_ratios, src.device)
        out = src
        for i, layer in enumerate(self.layers):
            with _cut.armed(i == len(self.layers) - 1):               # (_cut: the last layer's operator may head a second graph)
                out = layer(out, pos, reference_points, spatial_shapes, level_start_index, padding_mask)
        return out


class DepthAwareDecoderLayer(nn.Module):
    def __init__(self, d_model=256, d_ffn=1024, dropout=0.1, activation="relu", n_levels=4, n_heads=8,
                 n_points=4, group_num=1):
        super().__init__()
        # visual cross attention (deformable)
        self.cross_attn = MSDeformAttn(d_model, n_levels, n_heads, n_points)
        self.dropout1 = nn.Dropout(dropout)
        self.norm1 = nn.LayerNorm(d_model)
        # depth cross attention (dense)
        self.cross_attn_depth = FusedMultiheadAttention(d_model, n_heads, dropout=dropout)
        self.dropout_depth = nn.Dropout(dropout)
        self.norm_depth = nn.LayerNorm(d_model)
        # inter-query self attention (dense, per query group while training)
        self.self_attn = FusedMultiheadAttention(d_model, n_heads, dropout=dropout)
        self.dropout2 = nn.Dropout(dropout)
        self.norm2 = nn.LayerNorm(d_model)
        # ffn
        self.linear1 = Linear(d_model, d_ffn)
        self.activation = _get_activation_fn(activation)
        self.dropout3 = nn.Dropout(dropout)
        self.linear2 = Linear(d_ffn, d_model)
        self.dropout4 = nn.Dropout(dropout)
        self.norm3 = nn.LayerNorm(d_model)
        self.group_num = group_num
        # content / position projections feeding the self attention
        self.sa_qcontent_proj = Linear(d_model, d_model)
        self.sa_qpos_proj = Linear(d_model, d_model)
        self.sa_kcontent_proj = Linear(d_model, d_model)
        self.sa_kpos_proj = Linear(d_model, d_model)
        self.sa_v_proj = nn.Linear(d_model, d_model)      # never contributes (reference :471 vs :477)
        self.nhead = n_heads

    @staticmethod
    def with_pos_embed(tensor, pos):
        return tensor if pos is None else tensor + pos

    def forward_ffn(self, tgt):
        ff = self.linear2(ffn_hidden(tgt, self.linear1, self.dropout3, self.activation, tokenwise=False))
        return residual_layernorm(tgt, ff, self.norm3, self.dropout4)

    def _self_attention_inputs(self, x):
        """q = (Wqc + Wqp) x + b, k = (Wkc + Wkp) x + b : the four projections of the reference
        (:467-474) act on the same input, so they are summed in weight space -- ONE GEMM, not 4."""
        w, b = _SummedPair.apply(self.sa_qcontent_proj.weight, self.sa_qpos_proj.weight, self.sa_kcontent_proj.weight, self.sa_kpos_proj.weight,
                                 self.sa_qcontent_proj.bias, self.sa_qpos_proj.bias, self.sa_kcontent_proj.bias, self.sa_kpos_proj.bias)
        q, k = token_linear(x, w, b).split(x.shape[-1], -1)
        return q, k

    def forward(self, tgt, query_pos, reference_points, src, src_spatial_shapes, level_start_index,
                src_padding_mask, depth_pos_embed, mask_depth, bs, query_sine_embed=None, is_first=None,
                depth_pos_embed_ip=None, pos_embeds=None, self_attn_mask=None, query_pos_un=None, chain_src=False):
        """tgt, query_pos [B, Nq, C]; depth_pos_embed [HW/256, B, C] (sequence-first, as the reference
        passes it); mask_depth [B, HW/256] or None.  chain_src: -> (tgt, src') with src' == src for the next layer
        (`MSDeformAttn.forward`, chain_input)."""
        B, Nq, C = tgt.shape
        depth_tokens = depth_pos_embed.transpose(0, 1)
        # depth cross attention
        d = self.cross_attn_depth.forward_batch_first(tgt, depth_tokens, depth_tokens, mask_depth)
        tgt = residual_layernorm(tgt, d, self.norm_depth, self.dropout_depth)
        # self attention: keys/queries from content+position, values = tgt itself
        q, k = self._self_attention_inputs(self.with_pos_embed(tgt, query_pos))
        v = tgt
        if self.training and self.group_num > 1:
            if Nq % self.group_num != 0:
                raise ValueError("training expects num_queries * group_num queries, got %d for %d groups" % (Nq, self.group_num))
            n = Nq // self.group_num
            q, k, v = (t.reshape(B * self.group_num, n, C) for t in (q, k, v))
        s = self.self_attn.forward_batch_first(q, k, v).reshape(B, Nq, C)
        tgt = residual_layernorm(tgt, s, self.norm2, self.dropout2)
        # deformable cross attention into the visual memory
        c = self.cross_attn(self.with_pos_embed(tgt, query_pos), reference_points, src, src_spatial_shapes,
                            level_start_index, src_padding_mask, chain_input=chain_src)
        if chain_src:
            c, src = c
        tgt = residual_layernorm(tgt, c, self.norm1, self.dropout1)
        return (self.forward_ffn(tgt), src) if chain_src else self.forward_ffn(tgt)


class DepthAwareDecoder(nn.Module):
    def __init__(self, decoder_layer, num_layers, return_intermediate=False, d_model=None, use_dab=False,
                 two_stage_dino=False):
        super().__init__()
        if use_dab or two_stage_dino:
            raise NotImplementedError("use_dab / two_stage_dino are off in configs/monodetr.yaml and not built")
        self.layers = _get_clones(decoder_layer, num_layers)
        self.num_layers = num_layers
        self.return_intermediate = return_intermediate
        # set by MonoDETR for iterative box refinement (reference monodetr.py:129-131)
        self.bbox_embed = None
        self.dim_embed = None
        self.class_embed = None
        self.use_dab = use_dab
        self.two_stgae_dino = two_stage_dino
        # present in every published checkpoint, unused on the default path (reference :541-542)
        self.query_scale = MLP(d_model, d_model, d_model, 2)
        self.ref_point_head = MLP(d_model, d_model, 2, 2)

    def forward(self, tgt, reference_points, src, src_spatial_shapes, src_level_start_index, src_valid_ratios,
                query_pos=None, src_padding_mask=None, depth_pos_embed=None, mask_depth=None, bs=None,
                depth_pos_embed_ip=None, pos_embeds=None, attn_mask=None):
        output = tgt
        bs = src.shape[0]
        L = src_spatial_shapes.shape[0]
        hs, refs, dims = [], [], []
        reference_dims = None
        head_out = self.__dict__["head_outputs"] = []      # per level (box delta, depth, angle, class logits) when the heads are fused
        for lid, layer in enumerate(self.layers):
            nd = reference_points.shape[-1]
            assert nd in (2, 6)
            if src_valid_ratios is None:
                ref_in = reference_points[:, :, None].expand(-1, -1, L, -1)
            else:
                ref_in = reference_points[:, :, None] * src_valid_ratios.repeat(1, 1, nd // 2)[:, None]
            # (the layers read the memory one after the other: `src` continues from the previous reader)
            output = layer(output, query_pos, ref_in, src, src_spatial_shapes, src_level_start_index,
                           src_padding_mask, depth_pos_embed, mask_depth, bs, query_sine_embed=None,
                           is_first=(lid == 0), depth_pos_embed_ip=depth_pos_embed_ip, pos_embeds=pos_embeds,
                           self_attn_mask=attn_mask, query_pos_un=None, chain_src=_CHAIN_MEMORY)
            if _CHAIN_MEMORY:
                output, src = output
            fused = self.__dict__.get("fused_heads")      # (class_embed, depth_embed, angle_embed) lists, set by MonoDETR
            if fused is not None and self.bbox_embed is not None and self.dim_embed is not None:
                # every head that reads this level's output, first layers as one GEMM; the box and size heads are
                # finished here (the refinement needs them), the rest is handed to MonoDETR.forward -- which would
                # otherwise evaluate bbox_embed a second time on the same tensor (reference monodetr.py:226-236)
                cls, dep, ang = (f[lid] for f in fused)
                grouped = heads_level(output, self.bbox_embed[lid], self.dim_embed[lid], dep, ang, cls)
                if grouped is not None:
                    # the five heads as grouped fp32 launches (heads.py); `output` continues as the tensor handed back, so that
                    # the next layer's gradient is summed inside the heads' input-gradient launch
                    delta, grouped_dims, dep_out, ang_out, cls_out, output = grouped
                    head_out.append((delta, dep_out, ang_out, cls_out))
                else:
                    parts = fused_first_layers(output, [self.bbox_embed[lid], self.dim_embed[lid], dep, ang, cls])
                    delta = mlp_rest(self.bbox_embed[lid], parts[0])
                    head_out.append((delta, mlp_rest(dep, parts[2]), mlp_rest(ang, parts[3]),
                                     mlp_rest(cls, parts[4]) if isinstance(cls, MLP) else parts[4]))
            elif self.bbox_embed is not None:
                delta = self.bbox_embed[lid](output)
            if self.bbox_embed is not None and head_tail_ext.usable(delta, reference_points) and delta.shape[-1] == 6:
                # iterative refinement (:602-613) as one launch: sigmoid(delta + inverse_sigmoid(reference)), no gradient
                reference_points = head_tail_ext.box_refine(delta, reference_points)
            elif self.bbox_embed is not None:           # iterative refinement (:602-613)
                if nd == 6:
                    new_ref = (delta + inverse_sigmoid(reference_points)).sigmoid()
                else:
                    new_ref = torch.cat((delta[..., :2] + inverse_sigmoid(reference_points), delta[..., 2:]), -1).sigmoid()
                reference_points = new_ref.detach()
            if fused is not None and self.bbox_embed is not None and self.dim_embed is not None:
                reference_dims = grouped_dims if grouped is not None else mlp_rest(self.dim_embed[lid], parts[1])
            elif self.dim_embed is not None:
                reference_dims = self.dim_embed[lid](output)
            if self.return_intermediate:
                hs.append(output)
                refs.append(reference_points)
                dims.append(reference_dims)
        if self.return_intermediate:
            return torch.stack(hs), torch.stack(refs), torch.stack(dims)
        return output, reference_points


class DepthAwareTransformer(nn.Module):
    def __init__(self, d_model=256, nhead=8, num_encoder_layers=6, num_decoder_layers=6, dim_feedforward=1024,
                 dropout=0.1, activation="relu", return_intermediate_dec=False, num_feature_levels=4,
                 dec_n_points=4, enc_n_points=4, two_stage=False, two_stage_num_proposals=50, group_num=11,
                 use_dab=False, two_stage_dino=False):
        super().__init__()
        if two_stage or use_dab or two_stage_dino:
            raise NotImplementedError("two_stage / use_dab / two_stage_dino are off in configs/monodetr.yaml and not built")
        self.d_model, self.nhead = d_model, nhead
        self.two_stage, self.two_stage_num_proposals = two_stage, two_stage_num_proposals
        self.use_dab, self.two_stage_dino, self.group_num = use_dab, two_stage_dino, group_num

        self.encoder = VisualEncoder(
            VisualEncoderLayer(d_model, dim_feedforward, dropout, activation, num_feature_levels, nhead, enc_n_points),
            num_encoder_layers)
        self.decoder = DepthAwareDecoder(
            DepthAwareDecoderLayer(d_model, dim_feedforward, dropout, activation, num_feature_levels, nhead,
                                   dec_n_points, group_num=group_num),
            num_decoder_layers, return_intermediate_dec, d_model, use_dab=use_dab, two_stage_dino=two_stage_dino)
        self.level_embed = nn.Parameter(torch.Tensor(num_feature_levels, d_model))
        self.reference_points = nn.Linear(d_model, 2)
        self._shape_cache = {}
        self._reset_parameters()

    def _reset_parameters(self):
        for p in self.parameters():
            if p.dim() > 1:
                nn.init.xavier_uniform_(p)
        for m in self.modules():
            if isinstance(m, MSDeformAttn):
                m._reset_parameters()
        xavier_uniform_(self.reference_points.weight.data, gain=1.0)
        constant_(self.reference_points.bias.data, 0.)
        normal_(self.level_embed)

    def get_valid_ratio(self, mask):
        """Fraction of each image that is not padding, (w, h) per image."""
        _, H, W = mask.shape
        vh = torch.sum(~mask[:, :, 0], 1).float() / H
        vw = torch.sum(~mask[:, 0, :], 1).float() / W
        return torch.stack([vw, vh], -1)

    def _level_tensors(self, shapes, device):
        key = (tuple(shapes), str(device))
        hit = self._shape_cache.get(key)
        if hit is None:
            ss = torch.as_tensor(shapes, dtype=torch.long, device=device)
            hit = self._shape_cache[key] = (ss, torch.cat((ss.new_zeros((1,)), ss.prod(1).cumsum(0)[:-1])))
        return hit

    def encode(self, srcs, masks, pos_embeds):
        """Flatten + concatenate the pyramid and run the visual encoder (depthaware_transformer.py:208-229 of the
        reference): returns (memory [B, S, C], spatial_shapes, level_start_index, valid_ratios, mask_flatten, mask_depth).
        Also the whole of BASELINE configs[1] after the backbone (`bench.py --config 2`)."""
        unpadded = all(no_padding(m) for m in masks)
        shapes = [tuple(s.shape[-2:]) for s in srcs]
        # flatten every level to [B, HW, C] and concatenate along the token axis
        src_flatten = torch.cat([s.flatten(2).transpose(1, 2) for s in srcs], 1)
        lvl_pos = _LevelPos.apply(self.level_embed, src_flatten.dtype, *pos_embeds)
        spatial_shapes, level_start_index = self._level_tensors(shapes, src_flatten.device)
        if unpadded:
            mask_flatten = valid_ratios = mask_depth = None
        else:
            mask_flatten = torch.cat([m.flatten(1) for m in masks], 1)
            valid_ratios = torch.stack([self.get_valid_ratio(m) for m in masks], 1)
            mask_depth = masks[1].flatten(1)

        memory = self.encoder(src_flatten, spatial_shapes, level_start_index, valid_ratios, lvl_pos, mask_flatten,
                              shape_list=shapes)
        return memory, spatial_shapes, level_start_index, valid_ratios, mask_flatten, mask_depth

    def forward(self, srcs, masks, pos_embeds, query_embed=None, depth_pos_embed=None, depth_pos_embed_ip=None,
                attn_mask=None):
        assert query_embed is not None
        B, C = srcs[0].shape[:2]
        memory, spatial_shapes, level_start_index, valid_ratios, mask_flatten, mask_depth = self.encode(srcs, masks, pos_embeds)

        # queries: first half of the embedding is the positional part, second half the content
        query_pos, tgt = torch.split(query_embed, C, dim=1)
        reference_points = self.reference_points(query_pos.to(self.reference_points.weight.dtype)).sigmoid()
        reference_points = reference_points.unsqueeze(0).expand(B, -1, -1)
        query_pos = query_pos.to(memory.dtype).unsqueeze(0).expand(B, -1, -1)
        tgt = tgt.to(memory.dtype).unsqueeze(0).expand(B, -1, -1)
        init_reference_out = reference_points

        depth_tokens = depth_pos_embed.flatten(2).permute(2, 0, 1)
        depth_tokens_ip = depth_pos_embed_ip.flatten(2).permute(2, 0, 1)
        hs, inter_references, inter_references_dim = self.decoder(
            tgt, reference_points, memory, spatial_shapes, level_start_index, valid_ratios, query_pos,
            mask_flatten, depth_tokens, mask_depth, bs=B, depth_pos_embed_ip=depth_tokens_ip,
            pos_embeds=pos_embeds, attn_mask=attn_mask)
        return hs, init_reference_out, inter_references, inter_references_dim, None, None


def build_depthaware_transformer(cfg):
    return DepthAwareTransformer(
        d_model=cfg['hidden_dim'], dropout=cfg['dropout'], activation="relu", nhead=cfg['nheads'],
        dim_feedforward=cfg['dim_feedforward'], num_encoder_layers=cfg['enc_layers'],
        num_decoder_layers=cfg['dec_layers'], return_intermediate_dec=cfg['return_intermediate_dec'],
        num_feature_levels=cfg['num_feature_levels'], dec_n_points=cfg['dec_n_points'],
        enc_n_points=cfg['enc_n_points'], two_stage=cfg['two_stage'], two_stage_num_proposals=cfg['num_queries'],
        use_dab=cfg['use_dab'], two_stage_dino=cfg['two_stage_dino'])
