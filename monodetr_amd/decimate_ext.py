"""``decimate2(x)``: x[:, :, ::2, ::2] of a channels-last activation as a dense channels-last tensor (csrc/decimate.hip through
``mdetr_decimate2``), with the adjoint as its backward -- what turns a 1x1 / stride-2 convolution into a token GEMM."""
import os

import torch

from . import _capi

_backend = None               # tests substitute the CPU emulation of the same kernel source (tests/native_emul.py)
# False (tests): the 1x1 / stride-2 projection shortcuts stay with csrc/conv_taps.hip (one-tap implicit GEMM)
ENABLED = True


def _lib():
    return _backend if _backend is not None else _capi.lib()


def supported(x):
    return ((x.is_cuda or _backend is not None) and x.dim() == 4 and x.is_contiguous(memory_format=torch.channels_last)
            and (x.shape[1] * x.element_size()) % 16 == 0 and x.data_ptr() % 16 == 0 and x.numel() > 0)


def _call(backward, src, dst, B, H, W, C):
    dev, stream = (src.device.index, torch.cuda.current_stream(src.device).cuda_stream) if src.is_cuda else (-1, None)
    _capi.check(_lib().mdetr_decimate2(backward, src.data_ptr(), dst.data_ptr(), B, H, W, C * src.element_size(), dev, stream), "mdetr_decimate2")


class _Decimate2(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        B, C, H, W = x.shape
        y = torch.empty((B, C, (H + 1) // 2, (W + 1) // 2), dtype=x.dtype, device=x.device, memory_format=torch.channels_last)
        _call(0, x, y, B, H, W, C)
        ctx.shape = (B, C, H, W)
        return y

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, dy):
        B, C, H, W = ctx.shape
        dy = dy.contiguous(memory_format=torch.channels_last)
        dx = torch.empty(ctx.shape, dtype=dy.dtype, device=dy.device, memory_format=torch.channels_last)
        _call(1, dy, dx, B, H, W, C)
        return dx


def maxpool_supported(x):
    return ((x.is_cuda or _backend is not None) and x.dim() == 4 and x.dtype == torch.bfloat16 and x.shape[1] % 8 == 0
            and x.is_contiguous(memory_format=torch.channels_last) and x.data_ptr() % 16 == 0 and x.numel() > 0
            and not (torch.is_grad_enabled() and x.requires_grad))


def maxpool3x3s2(x):
    """nn.MaxPool2d(3, 2, 1) of a channels-last bf16 activation that needs no gradient (the frozen stem's output)."""
    B, C, H, W = x.shape
    y = torch.empty((B, C, (H - 1) // 2 + 1, (W - 1) // 2 + 1), dtype=x.dtype, device=x.device, memory_format=torch.channels_last)
    dev, stream = (x.device.index, torch.cuda.current_stream(x.device).cuda_stream) if x.is_cuda else (-1, None)
    _capi.check(_lib().mdetr_maxpool3x3s2_bf16(x.data_ptr(), y.data_ptr(), B, H, W, C, dev, stream), "mdetr_maxpool3x3s2_bf16")
    return y


def decimate2(x):
    """x [B, C, H, W] channels-last -> x[:, :, ::2, ::2] as a dense channels-last tensor; backward scatters into zeros."""
    if not supported(x):
        raise RuntimeError("decimate2: needs a channels-last CUDA activation with 16-byte pixels")
    return _Decimate2.apply(x)
