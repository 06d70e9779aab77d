"""Positional encodings -- mirror of lib/models/monodetr/position_encoding.py (sine :20-56,
learned :59-89, ``build_position_encoding`` :92-99)."""
import math

import torch
from torch import nn

from ..utils.misc import no_padding


class PositionEmbeddingSine(nn.Module):
    """2-D sine/cosine encoding over the un-padded extent of each image: ``num_pos_feats`` channels
    for y followed by ``num_pos_feats`` for x, interleaved sin/cos, frequencies
    temperature**(2*(i//2)/num_pos_feats)."""

    def __init__(self, num_pos_feats=64, temperature=10000, normalize=False, scale=None):
        super().__init__()
        if scale is not None and normalize is False:
            raise ValueError("normalize should be True if scale is passed")
        self.num_pos_feats, self.temperature, self.normalize = num_pos_feats, temperature, normalize
        self.scale = 2 * math.pi if scale is None else scale
        self._cache = {}

    def forward(self, tensor_list):
        """The encoding depends on the mask only.  Masks tagged all-False by construction
        (utils.misc.mark_no_padding -- every mask this model builds) give the same tensor for a given
        shape on every call, so it is computed once (~22 kernels per pyramid level otherwise)."""
        mask = tensor_list.mask
        assert mask is not None
        if not no_padding(mask):
            return self._encode(mask)
        key = (tuple(mask.shape), mask.device)
        pos = self._cache.get(key)
        if pos is None:
            pos = self._cache[key] = self._encode(mask)
        return pos

    def _encode(self, mask):
        valid = ~mask
        y = valid.cumsum(1, dtype=torch.float32)
        x = valid.cumsum(2, dtype=torch.float32)
        if self.normalize:
            y = y / (y[:, -1:, :] + 1e-6) * self.scale
            x = x / (x[:, :, -1:] + 1e-6) * self.scale
        i = torch.arange(self.num_pos_feats, dtype=torch.float32, device=mask.device)
        freq = self.temperature ** (2 * torch.div(i, 2, rounding_mode="floor") / self.num_pos_feats)

        def encode(coord):
            ang = coord[..., None] / freq
            return torch.stack((ang[..., 0::2].sin(), ang[..., 1::2].cos()), -1).flatten(-2)

        return torch.cat((encode(y), encode(x)), -1).permute(0, 3, 1, 2)


class PositionEmbeddingLearned(nn.Module):
    """Learned 50-entry row/column tables, linearly interpolated to the feature-map size."""

    def __init__(self, num_pos_feats=256):
        super().__init__()
        self.row_embed = nn.Embedding(50, num_pos_feats)
        self.col_embed = nn.Embedding(50, num_pos_feats)

    @staticmethod
    def get_embed(coord, embed):
        lo = coord.floor()
        frac = (coord - lo).unsqueeze(-1)
        lo = lo.long()
        hi = (lo + 1).clamp(max=49)
        return embed(lo) * (1 - frac) + embed(hi) * frac

    def forward(self, tensor_list):
        x = tensor_list.tensors
        h, w = x.shape[-2:]
        col = self.get_embed(torch.arange(w, device=x.device) / w * 49, self.col_embed)
        row = self.get_embed(torch.arange(h, device=x.device) / h * 49, self.row_embed)
        pos = torch.cat([col.unsqueeze(0).expand(h, -1, -1), row.unsqueeze(1).expand(-1, w, -1)], -1)
        return pos.permute(2, 0, 1).unsqueeze(0).expand(x.shape[0], -1, -1, -1)


def build_position_encoding(cfg):
    half = cfg['hidden_dim'] // 2
    kind = cfg['position_embedding']
    if kind in ('v2', 'sine'):
        return PositionEmbeddingSine(half, normalize=True)
    if kind in ('v3', 'learned'):
        return PositionEmbeddingLearned(half)
    raise ValueError(f"not supported {kind}")
