"""``bias_act(x, bias, skip, relu, dropout_p)``: ``dropout(relu(x + bias + skip))`` over a channels-last activation in
one launch each way (csrc/bias_act.hip through ``mdetr_bias_act_forward / _backward``) -- the tails of the backbone's
convolutions (frozen-BN shift + ReLU, residual addition + ReLU) and of the FFN's first layer (ReLU + Dropout).  The
dropout decision is a hash of (seed, element index); the backward reads only ``dy`` and the saved output."""
import os

import torch

from . import _capi

_backend = None               # tests substitute the CPU emulation of the same kernel source (tests/native_emul.py)
# MDETR_FUSED_EPILOGUE=1 routes the sites through the kernel; on the committed list of kernel_families.py
ENABLED = os.environ.get("MDETR_FUSED_EPILOGUE") == "1"


def _lib():
    return _backend if _backend is not None else _capi.lib()


def _channel_fastest(t):
    """Dense with the channel (4-D: dim 1 of a channels_last tensor; otherwise the last dim) as the unit-stride index."""
    if t.dim() == 4:
        return t.is_contiguous(memory_format=torch.channels_last)
    return t.dim() >= 2 and t.is_contiguous()


def _cols(t):
    return t.shape[1] if t.dim() == 4 else t.shape[-1]


def supported(x, bias=None, skip=None):
    if not ((x.is_cuda or _backend is not None) and x.dtype in (torch.float32, torch.bfloat16) and _channel_fastest(x)
            and x.numel() > 0 and x.data_ptr() % 16 == 0):
        return False
    C = _cols(x)
    if C % (8 if x.dtype == torch.bfloat16 else 4) != 0:
        return False
    if bias is not None and not (bias.dim() == 1 and bias.shape[0] == C and not bias.requires_grad and bias.device == x.device
                                 and bias.dtype in (torch.float32, x.dtype)):
        return False
    if skip is not None and not (skip.shape == x.shape and skip.dtype == x.dtype and skip.device == x.device
                                 and skip.stride() == x.stride() and skip.data_ptr() % 16 == 0):
        return False
    return True


def _dev(t):
    cuda = t.is_cuda
    return (t.device.index if cuda else -1), (torch.cuda.current_stream(t.device).cuda_stream if cuda else None)


def _io(t):
    return _capi.MDETR_BF16 if t.dtype == torch.bfloat16 else _capi.MDETR_F32


class _BiasAct(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, bias, skip, relu, p, seed, seed_dev):
        C = _cols(x)
        rows = x.numel() // C
        y = torch.empty_like(x)                                      # keeps x's (channels_last) strides
        b = bias
        if b is not None and (not b.is_contiguous() or b.data_ptr() % 16 != 0):
            b = b.contiguous().clone()
        dev, stream = _dev(x)
        rc = _lib().mdetr_bias_act_forward(_io(x), _io(b) if b is not None else _capi.MDETR_F32, x.data_ptr(),
                                           b.data_ptr() if b is not None else None, skip.data_ptr() if skip is not None else None,
                                           y.data_ptr(), rows, C, 1 if relu else 0, float(p), int(seed),
                                           seed_dev.data_ptr() if seed_dev is not None else None, dev, stream)
        if rc != 0:
            _capi.check(rc, "mdetr_bias_act_forward")
        ctx.relu, ctx.scale, ctx.has_skip = bool(relu), (1.0 / (1.0 - p) if p > 0.0 else 1.0), skip is not None
        if relu:
            ctx.save_for_backward(y)
        return y

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, dy):
        if not ctx.relu:                                             # x + bias + skip (no dropout without the ReLU: see bias_act)
            return dy, None, (dy if ctx.has_skip else None), None, None, None, None
        y, = ctx.saved_tensors
        if dy.stride() != y.stride() or dy.data_ptr() % 16 != 0:
            dy = torch.empty_like(y).copy_(dy)
        dx = torch.empty_like(y)
        C = _cols(y)
        dev, stream = _dev(y)
        rc = _lib().mdetr_bias_act_backward(_io(y), dy.data_ptr(), y.data_ptr(), dx.data_ptr(), y.numel() // C, C, float(ctx.scale), dev, stream)
        if rc != 0:
            _capi.check(rc, "mdetr_bias_act_backward")
        return dx, None, (dx if ctx.has_skip else None), None, None, None, None


def act_backward(dy, y, scale=1.0):
    """dx = dy * scale where y > 0, else 0: the backward of y = dropout(relu(.)) from the saved output alone (the same launch
    `_BiasAct.backward` makes; also serves csrc/tgemm.hip's ReLU + Dropout epilogue, which makes bias_act's keep decisions)."""
    if dy.stride() != y.stride() or dy.data_ptr() % 16 != 0:
        dy = torch.empty_like(y).copy_(dy)
    dx = torch.empty_like(y)
    C = _cols(y)
    dev, stream = _dev(y)
    rc = _lib().mdetr_bias_act_backward(_io(y), dy.data_ptr(), y.data_ptr(), dx.data_ptr(), y.numel() // C, C, float(scale), dev, stream)
    if rc != 0:
        _capi.check(rc, "mdetr_bias_act_backward")
    return dx


def bias_act(x, bias=None, skip=None, relu=True, dropout_p=0.0, seed=None):
    """dropout(relu(x + bias + skip)).  ``x`` [B, C, H, W] channels_last or [..., C] contiguous (f32 / bf16); ``bias`` [C]
    without gradient (a frozen-BN shift); ``skip`` like ``x``.  Dropout needs ``relu`` (the backward recovers the mask from
    the sign of the output).  ``seed`` fixes the mask (tests); otherwise each call draws a fresh one."""
    if not supported(x, bias, skip):
        raise RuntimeError("bias_act: needs a CUDA f32/bf16 channels-last activation (16-byte aligned, C % 8 == 0 for bf16, "
                           "% 4 for f32), a bias without gradient and a skip tensor of the same layout")
    if dropout_p > 0.0 and not relu:
        raise RuntimeError("bias_act: dropout is only fused behind the ReLU")
    seed_dev = None
    if dropout_p > 0.0 and seed is None:
        if x.is_cuda and torch.cuda.is_current_stream_capturing():
            from .attn_ext import site_seed
            seed, seed_dev = site_seed(x.device)                     # a replayed graph needs a seed that lives on the device
        else:
            from .add_ln_ext import _host_seed
            seed = _host_seed()
    return _BiasAct.apply(x, bias, skip, relu, float(dropout_p), seed or 0, seed_dev)
