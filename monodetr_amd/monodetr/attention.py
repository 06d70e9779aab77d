"""Dense multi-head attention with ``torch.nn.MultiheadAttention``'s parameters and call signature.

The reference uses ``nn.MultiheadAttention`` for the depth cross-attention and the decoder
self-attention (depthaware_transformer.py:399,404) and for the depth encoder
(depth_predictor/transformer.py:40), always as ``attn(q, k, v, key_padding_mask=...)[0]`` -- with
the default ``need_weights=True`` that materialises and head-averages the full score matrix nobody
reads (944 MB for the depth encoder at B=8).  This module keeps the parameter
names/shapes (``in_proj_weight [3E,E]``, ``in_proj_bias [3E]``, ``out_proj.{weight,bias}``) so
checkpoints load, keeps the ``(L, N, E)`` sequence-first API and returns ``(output, None)``, but
never builds the averaged weights: the QK^T -> softmax -> dropout -> PV core runs fused.

``attention_core`` is the single entry point of that core (batch-first), so the backend is chosen
in one place.
"""
import os

import torch
import torch.nn.functional as F
from torch import nn

from ..attn_ext import fused_attention
from .linear import Linear, split_rows, token_linear


def _sdpa(q, k, v, num_heads, dropout_p, key_padding_mask):
    """PyTorch evaluation of the same function (CPU tensors; comparator on the GPU)."""
    B, Lq, E = q.shape
    Lk = k.shape[1]
    d = E // num_heads
    qh = q.reshape(B, Lq, num_heads, d).transpose(1, 2)
    kh = k.reshape(B, Lk, num_heads, d).transpose(1, 2)
    vh = v.reshape(B, Lk, num_heads, d).transpose(1, 2)
    mask = None
    if key_padding_mask is not None:
        mask = ~key_padding_mask.view(B, 1, 1, Lk)          # True = attend
    out = F.scaled_dot_product_attention(qh, kh, vh, attn_mask=mask, dropout_p=dropout_p)
    return out.transpose(1, 2).reshape(B, Lq, E)


def attention_core(q, k, v, num_heads, dropout_p=0.0, key_padding_mask=None):
    """softmax(q k^T / sqrt(d) + mask) v per head.
    q [B, Lq, E]; k, v [B, Lk, E] (last dim contiguous; slices of a packed projection are fine);
    key_padding_mask [B, Lk] bool (True = ignore) or None.  Returns [B, Lq, E].  Dropout (if
    dropout_p > 0) acts on the attention probabilities, as in nn.MultiheadAttention.

    GPU tensors with 32 channels per head (the model's geometry) run the gfx950 MFMA kernels of
    monodetr_amd/csrc/attn.hip -- no fallback: a missing library raises.  CPU tensors (the
    reference is plain PyTorch there too) and other head sizes use PyTorch's SDPA;
    MDETR_TUNE="attn_backend=sdpa" forces the comparator on the GPU (tests)."""
    from .. import _tune
    if q.is_cuda and q.shape[-1] == 32 * num_heads and _tune.get("attn_backend", "hip") == "hip":
        if q.dtype not in (torch.float32, torch.bfloat16):
            q, k, v = q.float(), k.float(), v.float()
        if k.dtype != q.dtype or v.dtype != q.dtype:
            k, v = k.to(q.dtype), v.to(q.dtype)
        return fused_attention(q, k, v, num_heads, dropout_p, key_padding_mask)
    return _sdpa(q, k, v, num_heads, dropout_p, key_padding_mask)


class MultiheadAttention(nn.Module):
    def __init__(self, embed_dim, num_heads, dropout=0.0):
        super().__init__()
        if embed_dim % num_heads != 0:
            raise ValueError("embed_dim must be divisible by num_heads")
        self.embed_dim, self.num_heads, self.dropout = embed_dim, num_heads, dropout
        self.head_dim = embed_dim // num_heads
        self.in_proj_weight = nn.Parameter(torch.empty(3 * embed_dim, embed_dim))
        self.in_proj_bias = nn.Parameter(torch.zeros(3 * embed_dim))
        self.out_proj = Linear(embed_dim, embed_dim)
        nn.init.xavier_uniform_(self.in_proj_weight)
        nn.init.zeros_(self.out_proj.bias)

    def project(self, query, key, value):
        """Packed in-projection; one GEMM when query is key is value, two when key is value."""
        E = self.embed_dim
        W, b = self.in_proj_weight, self.in_proj_bias
        if query is key and key is value:
            return token_linear(query, W, b).split(E, -1)
        # (row blocks through `split_rows`: one gradient assembly per parameter instead of a zero-fill, a copy and an add per block;
        #  every product through `token_linear`: the 4 400-row weight gradients take csrc/small_wgrad.hip)
        if query is key:                                     # q = k = x + pos, v = x (encoder layers)
            (Wqk, Wv), (bqk, bv) = split_rows(W, 2 * E, E), split_rows(b, 2 * E, E)
            q, k = token_linear(query, Wqk, bqk).split(E, -1)
            return q, k, token_linear(value, Wv, bv)
        if key is value:
            (Wq, Wkv), (bq, bkv) = split_rows(W, E, 2 * E), split_rows(b, E, 2 * E)
            k, v = token_linear(key, Wkv, bkv).split(E, -1)
            return token_linear(query, Wq, bq), k, v
        (Wq, Wk, Wv), (bq, bk, bv) = split_rows(W, E, E, E), split_rows(b, E, E, E)
        return token_linear(query, Wq, bq), token_linear(key, Wk, bk), token_linear(value, Wv, bv)

    def forward_batch_first(self, query, key, value, key_padding_mask=None):
        """[B, L, E] in, [B, Lq, E] out."""
        q, k, v = self.project(query, key, value)
        p = self.dropout if self.training else 0.0
        return self.out_proj(attention_core(q, k, v, self.num_heads, p, key_padding_mask))

    def forward(self, query, key, value, key_padding_mask=None, need_weights=False, attn_mask=None):
        """nn.MultiheadAttention calling convention: (L, N, E) tensors -> (output (L, N, E), None)."""
        if attn_mask is not None:
            raise NotImplementedError("attn_mask is not used on the MonoDETR path")
        qb = query.transpose(0, 1)
        kb = qb if key is query else key.transpose(0, 1)
        vb = kb if value is key else value.transpose(0, 1)
        out = self.forward_batch_first(qb, kb, vb, key_padding_mask)
        return out.transpose(0, 1), None
