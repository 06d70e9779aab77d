"""Foreground depth head + depth encoder -- mirror of
lib/models/monodetr/depth_predictor/depth_predictor.py:7-104.

Fuses the stride-8/16/32 feature maps at stride 16, predicts a (num_depth_bins+1)-way depth
distribution per pixel (LID bins), its expectation ``weighted_depth``, and the depth-aware tokens
the decoder cross-attends to: encoder(src, pos) + interpolated learned depth positional embedding.
"""
import torch
import torch.nn.functional as F
from torch import nn

from ...utils.misc import at_least_fp32, no_padding
from .transformer import TransformerEncoder, TransformerEncoderLayer


def _conv_gn(cin, cout, k, stride=1):
    return nn.Sequential(nn.Conv2d(cin, cout, kernel_size=(k, k), stride=(stride, stride), padding=k // 2),
                         nn.GroupNorm(32, cout))


class DepthPredictor(nn.Module):
    def __init__(self, model_cfg):
        super().__init__()
        nbins = int(model_cfg["num_depth_bins"])
        dmin, dmax = float(model_cfg["depth_min"]), float(model_cfg["depth_max"])
        self.depth_max = dmax
        # linear-increasing discretisation: bin i spans delta*(i+1); value = bin centre (:21-25)
        delta = 2 * (dmax - dmin) / (nbins * (1 + nbins))
        idx = torch.linspace(0, nbins - 1, nbins)
        centres = (idx + 0.5).pow(2) * delta / 2 - delta / 8 + dmin
        self.depth_bin_values = nn.Parameter(torch.cat([centres, torch.tensor([dmax])]), requires_grad=False)

        d = model_cfg["hidden_dim"]
        self.downsample = _conv_gn(d, d, 3, 2)
        self.proj = _conv_gn(d, d, 1)
        self.upsample = _conv_gn(d, d, 1)
        self.depth_head = nn.Sequential(
            nn.Conv2d(d, d, kernel_size=(3, 3), padding=1), nn.GroupNorm(32, num_channels=d), nn.ReLU(),
            nn.Conv2d(d, d, kernel_size=(3, 3), padding=1), nn.GroupNorm(32, num_channels=d), nn.ReLU())
        self.depth_classifier = nn.Conv2d(d, nbins + 1, kernel_size=(1, 1))
        self.depth_encoder = TransformerEncoder(TransformerEncoderLayer(d, nhead=8, dim_feedforward=256, dropout=0.1), 1)
        self.depth_pos_embed = nn.Embedding(int(dmax) + 1, 256)

    def forward(self, feature, mask, pos):
        assert len(feature) == 4
        s16 = self.proj(feature[1])
        s32 = self.upsample(F.interpolate(feature[2], size=s16.shape[-2:], mode='bilinear'))
        s8 = self.downsample(feature[0])
        src = self.depth_head((s8 + s16 + s32) / 3)

        depth_logits = self.depth_classifier(src)
        weighted_depth = (F.softmax(at_least_fp32(depth_logits), dim=1) * at_least_fp32(self.depth_bin_values).reshape(1, -1, 1, 1)).sum(dim=1)

        B, C, H, W = src.shape
        tokens = src.flatten(2).permute(2, 0, 1)
        key_mask = None if no_padding(mask) else mask.flatten(1)
        enc = self.depth_encoder(tokens, key_mask, pos.flatten(2).permute(2, 0, 1).to(tokens.dtype))
        depth_pos_embed_ip = self.interpolate_depth_embed(weighted_depth).to(src.dtype)
        depth_embed = enc.permute(1, 2, 0).reshape(B, C, H, W) + depth_pos_embed_ip
        return depth_logits, depth_embed, weighted_depth, depth_pos_embed_ip

    def interpolate_depth_embed(self, depth):
        depth = depth.clamp(min=0, max=self.depth_max)
        return self.interpolate_1d(depth, self.depth_pos_embed).permute(0, 3, 1, 2)

    def interpolate_1d(self, coord, embed):
        lo = coord.floor()
        frac = (coord - lo).unsqueeze(-1)
        lo = lo.long()
        hi = (lo + 1).clamp(max=embed.num_embeddings - 1)
        return embed(lo) * (1 - frac) + embed(hi) * frac
