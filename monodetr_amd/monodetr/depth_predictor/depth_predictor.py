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
from ...conv3x3_ext import Conv3x3
from ...group_norm_ext import GroupNorm
from ..linear import PointwiseConv2d
from .transformer import TransformerEncoder, TransformerEncoderLayer


def _conv_gn(cin, cout, k, stride=1):
    return nn.Sequential(PointwiseConv2d(cin, cout, kernel_size=(k, k), stride=(stride, stride), padding=k // 2),
                         GroupNorm(32, cout))


_resize_cache = {}


def _resize_matrix(n_in, n_out, device, dtype):
    """[n_out, n_in] matrix of F.interpolate(mode='bilinear', align_corners=False) along one axis,
    obtained by resizing the identity -- the coefficients are the library's own."""
    key = (n_in, n_out, str(device), dtype)
    m = _resize_cache.get(key)
    if m is None:
        eye = torch.eye(n_in, device=device, dtype=torch.float64).view(1, n_in, n_in, 1)
        m = F.interpolate(eye, size=(n_out, 1), mode='bilinear').view(n_in, n_out).t().contiguous().to(dtype)
        _resize_cache[key] = m
    return m


def _bilinear_resize(x, size):
    """F.interpolate(x, size, mode='bilinear') (reference depth_predictor.py:58) as two small matrix
    products (bilinear resizing is separable).  The library kernel takes 0.69 ms forward + 0.35 ms
    backward for this 12x40 -> 24x80 map in bf16 channels_last; the products take a few microseconds."""
    if not x.is_cuda:
        return F.interpolate(x, size=(int(size[0]), int(size[1])), mode='bilinear')
    return _bilinear_resize_matmul(x, size)


def _bilinear_resize_matmul(x, size):
    B, C, H, W = x.shape
    Ho, Wo = int(size[0]), int(size[1])
    Ah = _resize_matrix(H, Ho, x.device, x.dtype)
    Aw = _resize_matrix(W, Wo, x.device, x.dtype)
    t = x.permute(0, 2, 3, 1)                                   # [B, H, W, C]: a view for channels_last input
    t = torch.matmul(Aw, t.reshape(B * H, W, C))                # [B*H, Wo, C]
    t = torch.matmul(Ah, t.reshape(B, H, Wo * C))               # [B, Ho, Wo*C]
    return t.view(B, Ho, Wo, C).permute(0, 3, 1, 2)             # NCHW view with channels_last strides


class _HatTimesTable(torch.autograd.Function):
    """hat [..., n] @ table [n, C] whose table gradient hat^T dOut -- n = 61 rows contracted over B H W = 15 360 positions, which the
    library runs as one single-tile GEMM (129 us) -- takes csrc/small_wgrad.hip (`hat` in the role of dY, dOut in the role of X)."""

    @staticmethod
    def forward(ctx, hat, table):
        ctx.save_for_backward(hat, table)
        return hat @ table

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, dout):
        from ... import small_wgrad_ext
        hat, table = ctx.saved_tensors
        d_hat = dout @ table.t() if ctx.needs_input_grad[0] else None
        d_table = None
        if ctx.needs_input_grad[1]:
            h2, d2 = hat.reshape(-1, hat.shape[-1]), dout.reshape(-1, dout.shape[-1])
            if not d2.is_contiguous():
                d2 = d2.contiguous()
            if small_wgrad_ext.ENABLED and table.dtype in (torch.float32, torch.bfloat16) and small_wgrad_ext.supported(h2, d2):
                d_table = small_wgrad_ext.small_wgrad(h2, d2, table.dtype)[0]
            else:
                d_table = (h2.t() @ d2).to(table.dtype)
        return d_hat, d_table


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
            # (GroupNorm(relu=True) carries the ReLU; the Identity keeps the reference's Sequential indices -- state_dict keys)
            Conv3x3(d, d, kernel_size=(3, 3), padding=1), GroupNorm(32, d, relu=True), nn.Identity(),
            Conv3x3(d, d, kernel_size=(3, 3), padding=1), GroupNorm(32, d, relu=True), nn.Identity())
        self.depth_classifier = PointwiseConv2d(d, nbins + 1, kernel_size=(1, 1))
        self.depth_encoder = TransformerEncoder(TransformerEncoderLayer(d, nhead=8, dim_feedforward=256, dropout=0.1), 1)
        self.depth_pos_embed = nn.Embedding(int(dmax) + 1, 256)

    def forward(self, feature, mask, pos):
        assert len(feature) == 4
        s16 = self.proj(feature[1])
        s32 = self.upsample(_bilinear_resize(feature[2], s16.shape[-2:]))
        s8 = self.downsample(feature[0])
        src = self.depth_head((s8 + s16 + s32) / 3)

        depth_logits = self.depth_classifier(src)
        logits32, bins32 = at_least_fp32(depth_logits), at_least_fp32(self.depth_bin_values)
        if depth_logits.dim() == 4 and depth_logits.is_contiguous(memory_format=torch.channels_last) and depth_logits.shape[1] > 1:
            # the classifier's output is channels-last: a pixel's 81 bin logits are one row.  Softmax over the LAST dimension of the
            # [B, H, W, 81] view takes the framework's row kernel (≈ 5 us each way); over dim 1 of the NCHW shape it took the strided
            # "spatial" kernel, 59 + 48 us per iteration for a 5 MB tensor.  The expectation is a matrix-vector product.
            weighted_depth = torch.matmul(F.softmax(logits32.permute(0, 2, 3, 1), dim=-1), bins32)
        else:
            weighted_depth = (F.softmax(logits32, dim=1) * bins32.reshape(1, -1, 1, 1)).sum(dim=1)

        B, C, H, W = src.shape
        # the depth encoder batch-first: [B, HW, C] IS the channels-last map (a view), and so is its output -- the reference's
        # sequence-first (HW, B, C) layout (depth_predictor.py:63-68) costs a layout copy per projection here
        tokens = src.permute(0, 2, 3, 1).reshape(B, H * W, C)
        key_mask = None if no_padding(mask) else mask.flatten(1)
        enc = self.depth_encoder.forward_batch_first(tokens, key_mask, pos.permute(0, 2, 3, 1).reshape(B, H * W, C).to(tokens.dtype))
        depth_pos_embed_ip = self.interpolate_depth_embed(weighted_depth).to(src.dtype)
        depth_embed = enc.view(B, H, W, C).permute(0, 3, 1, 2) + depth_pos_embed_ip
        return depth_logits, depth_embed, weighted_depth, depth_pos_embed_ip

    def interpolate_depth_embed(self, depth):
        depth = depth.clamp(min=0, max=self.depth_max)
        return self.interpolate_1d(depth, self.depth_pos_embed).permute(0, 3, 1, 2)

    def interpolate_1d(self, coord, embed):
        """Linear interpolation between neighbouring embedding rows (reference :74-83:
        embed(floor) * (1 - frac) + embed(floor + 1) * frac).  Written as hat-function weights times
        the table, w_j = max(0, 1 - |coord - j|): the same two non-zero weights per position, the same
        gradients for the table and for `coord`, but one small GEMM each way instead of two embedding
        lookups whose backward is a sort + segmented scatter (1.2 ms per step on MI355X)."""
        n = embed.num_embeddings
        coord = coord.clamp(max=n - 1)
        grid = torch.arange(n, device=coord.device, dtype=coord.dtype)
        hat = (1 - (coord.unsqueeze(-1) - grid).abs()).clamp(min=0)               # [..., n]
        hat = hat.to(embed.weight.dtype)
        if hat.is_cuda and torch.is_grad_enabled() and embed.weight.requires_grad:
            return _HatTimesTable.apply(hat, embed.weight)
        return hat @ embed.weight
