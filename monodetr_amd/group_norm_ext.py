"""``GroupNorm`` (+ ReLU) for channels-last activations with 8 channels per group -- nn.GroupNorm(32, 256), the
normalisation behind every input projection (reference monodetr.py:77-99) and depth-predictor stage
(depth_predictor.py:30-56) -- in two launches forward, three backward (csrc/group_norm.hip through
``mdetr_group_norm_forward / _backward``), without the NCHW layout copies the framework's kernel needs around it."""
import os

import torch
import torch.nn.functional as F
from torch import nn

from . import _capi

_backend = None               # tests substitute the CPU emulation of the same kernel source (tests/native_emul.py)
# MDETR_GROUP_NORM=1 routes GroupNorm modules through the kernel (bench.py's committed list after its GPU validation)
ENABLED = os.environ.get("MDETR_GROUP_NORM") == "1"


def _lib():
    return _backend if _backend is not None else _capi.lib()


def supported(x, weight, bias, groups):
    if not ((x.is_cuda or _backend is not None) and x.dim() == 4 and x.dtype in (torch.float32, torch.bfloat16)
            and x.is_contiguous(memory_format=torch.channels_last) and x.numel() > 0 and x.data_ptr() % 16 == 0):
        return False
    C = x.shape[1]
    if weight is None or bias is None or C != 8 * groups or (C // 8) & (C // 8 - 1) or C // 8 > 256:
        return False
    return (weight.dtype == bias.dtype and weight.dtype in (torch.float32, x.dtype) and weight.device == x.device
            and weight.is_contiguous() and bias.is_contiguous())


def _dev(t):
    return (t.device.index, torch.cuda.current_stream(t.device).cuda_stream) if t.is_cuda else (-1, None)


def _code(t):
    return _capi.MDETR_BF16 if t.dtype == torch.bfloat16 else _capi.MDETR_F32


def _workspace(x, need):
    from . import _workspace as W
    return W.get("group_norm", x.device, need, floor=1 << 20)


class _GroupNorm(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, groups, eps, relu):
        N, C, H, W = x.shape
        lib = _lib()
        ws = _workspace(x, lib.mdetr_group_norm_workspace_bytes(N, H * W, C, groups))
        y = torch.empty_like(x)                                      # channels_last, like x
        stats = torch.empty(N, groups, 2, dtype=torch.float32, device=x.device)
        dev, stream = _dev(x)
        _capi.check(lib.mdetr_group_norm_forward(_code(x), _code(weight), x.data_ptr(), weight.data_ptr(), bias.data_ptr(), y.data_ptr(),
                                                 stats.data_ptr(), ws.data_ptr(), ws.numel(), N, H * W, C, groups, float(eps), int(relu),
                                                 dev, stream), "mdetr_group_norm_forward")
        ctx.save_for_backward(x, weight, bias, stats)
        ctx.groups, ctx.relu = groups, relu
        return y

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, dy):
        x, weight, bias, stats = ctx.saved_tensors
        N, C, H, W = x.shape
        lib = _lib()
        if not dy.is_contiguous(memory_format=torch.channels_last) or dy.dtype != x.dtype:
            dy = dy.to(x.dtype).contiguous(memory_format=torch.channels_last)
        ws = _workspace(x, lib.mdetr_group_norm_workspace_bytes(N, H * W, C, ctx.groups))
        dx = torch.empty_like(x)
        dparams = torch.empty(2, C, dtype=weight.dtype, device=x.device)
        dev, stream = _dev(x)
        _capi.check(lib.mdetr_group_norm_backward(_code(x), _code(weight), dy.data_ptr(), x.data_ptr(), weight.data_ptr(), bias.data_ptr(),
                                                  stats.data_ptr(), dx.data_ptr(), dparams.data_ptr(), ws.data_ptr(), ws.numel(),
                                                  N, H * W, C, ctx.groups, int(ctx.relu), dev, stream), "mdetr_group_norm_backward")
        return dx, dparams[0], dparams[1], None, None, None


def group_norm(x, weight, bias, groups, eps=1e-5, relu=False):
    """F.group_norm(x, groups, weight, bias, eps) followed by ReLU if `relu`; x NCHW in channels_last memory format."""
    if not supported(x, weight, bias, groups):
        raise RuntimeError("group_norm: needs a channels_last CUDA f32/bf16 activation with 8 channels per group")
    return _GroupNorm.apply(x, weight, bias, groups, eps, relu)


class GroupNorm(nn.GroupNorm):
    """nn.GroupNorm (same parameters, same state_dict keys); ``relu=True`` folds the ReLU that follows it in the model into
    the same pass.  The kernel when MDETR_GROUP_NORM=1 and the input qualifies, the framework's operators otherwise."""

    def __init__(self, num_groups, num_channels, eps=1e-5, affine=True, relu=False):
        super().__init__(num_groups, num_channels, eps=eps, affine=affine)
        self.relu = relu

    def forward(self, x):
        if ENABLED and supported(x, self.weight, self.bias, self.num_groups) and not torch.is_autocast_enabled():
            return _GroupNorm.apply(x, self.weight, self.bias, self.num_groups, self.eps, self.relu)
        y = super().forward(x)
        return F.relu(y) if self.relu else y
