"""Depth encoder -- mirror of lib/models/monodetr/depth_predictor/transformer.py
(``TransformerEncoder`` :16-33, ``TransformerEncoderLayer`` :36-65): post-norm encoder layers over
the H/16 x W/16 depth tokens, q = k = src + pos, v = src.  Sequence-first (L, N, E) like the
reference; the attention core is the fused one of ..attention (no score matrix in HBM)."""
import copy

import torch.nn.functional as F
from torch import nn

from ...add_ln_ext import residual_layernorm
from ..attention import MultiheadAttention
from ..linear import Linear, ffn_hidden


def _get_clones(module, N):
    return nn.ModuleList([copy.deepcopy(module) for _ in range(N)])


def _get_activation_fn(activation):
    try:
        return {"relu": F.relu, "gelu": F.gelu, "glu": F.glu}[activation]
    except KeyError:
        raise RuntimeError(f"activation should be relu/gelu, not {activation}.")


class TransformerEncoderLayer(nn.Module):
    def __init__(self, d_model, nhead, dim_feedforward=2048, dropout=0.1, activation="relu"):
        super().__init__()
        self.self_attn = MultiheadAttention(d_model, nhead, dropout=dropout)
        self.linear1 = Linear(d_model, dim_feedforward)
        self.dropout = nn.Dropout(dropout)
        self.linear2 = Linear(dim_feedforward, d_model)
        self.norm1 = nn.LayerNorm(d_model)
        self.norm2 = nn.LayerNorm(d_model)
        self.dropout1 = nn.Dropout(dropout)
        self.dropout2 = nn.Dropout(dropout)
        self.activation = _get_activation_fn(activation)

    def forward(self, src, src_key_padding_mask, pos):
        qk = src if pos is None else src + pos
        src = residual_layernorm(src, self.self_attn(qk, qk, src, key_padding_mask=src_key_padding_mask)[0], self.norm1, self.dropout1)
        ff = self.linear2(ffn_hidden(src, self.linear1, self.dropout, self.activation, tokenwise=False))
        return residual_layernorm(src, ff, self.norm2, self.dropout2)


    def forward_batch_first(self, src, src_key_padding_mask, pos):
        """The same layer on [B, L, E] tensors (what a channels-last feature map is without any copy)."""
        qk = src if pos is None else src + pos
        attn = self.self_attn.forward_batch_first(qk, qk, src, key_padding_mask=src_key_padding_mask)
        src = residual_layernorm(src, attn, self.norm1, self.dropout1)
        ff = self.linear2(ffn_hidden(src, self.linear1, self.dropout, self.activation, tokenwise=False))
        return residual_layernorm(src, ff, self.norm2, self.dropout2)


class TransformerEncoder(nn.Module):
    def __init__(self, encoder_layer, num_layers, norm=None):
        super().__init__()
        self.layers = _get_clones(encoder_layer, num_layers)
        self.num_layers = num_layers
        self.norm = norm

    def forward(self, src, src_key_padding_mask, pos):
        for layer in self.layers:
            src = layer(src, src_key_padding_mask=src_key_padding_mask, pos=pos)
        return src if self.norm is None else self.norm(src)

    def forward_batch_first(self, src, src_key_padding_mask, pos):
        for layer in self.layers:
            src = layer.forward_batch_first(src, src_key_padding_mask, pos)
        return src if self.norm is None else self.norm(src)
