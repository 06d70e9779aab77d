"""``conv3x3(x, weight, shift, relu)``: 3x3 / stride 1 / pad 1 convolution of a channels_last bf16 activation with the folded
frozen-BN shift and the ReLU in its epilogue (csrc/conv3x3.hip through ``mdetr_conv3x3_forward``): forward and input gradient
run on the kernel, the weight gradient on csrc/conv_wgrad.hip (conv_wgrad_ext)."""
import os

import torch

from . import _capi

_backend = None               # tests substitute the CPU emulation of the same kernel source (tests/native_emul.py)
# MDETR_CONV3X3=1 routes the backbone's stride-1 3x3 convolutions through the kernel; on the committed bf16 list of kernel_families.py
ENABLED = os.environ.get("MDETR_CONV3X3") == "1"


def _lib():
    return _backend if _backend is not None else _capi.lib()


def supported(x, weight, stride=(1, 1), padding=(1, 1), dilation=(1, 1), groups=1):
    """x [B, C, H, W] channels_last bf16, weight [N, C, 3, 3] bf16; stride 1, padding 1, no dilation / groups."""
    return ((x.is_cuda or _backend is not None) and x.dim() == 4 and weight.dim() == 4 and x.dtype == torch.bfloat16
            and weight.dtype == torch.bfloat16 and tuple(weight.shape[2:]) == (3, 3) and tuple(stride) == (1, 1)
            and tuple(padding) == (1, 1) and tuple(dilation) == (1, 1) and groups == 1 and weight.shape[1] == x.shape[1]
            and x.shape[1] % 64 == 0 and weight.shape[0] % 32 == 0 and x.numel() > 0
            and x.is_contiguous(memory_format=torch.channels_last) and x.data_ptr() % 16 == 0)


def _ohwi(weight):
    """[N, C, 3, 3] -> contiguous [N, 3, 3, C] (a view when the weight is channels_last already)."""
    w = weight.permute(0, 2, 3, 1).contiguous()
    return w if w.data_ptr() % 16 == 0 else w.clone()


def _launch(x_cl, w_ohwi, shift, relu, mirror=False, mask=None):
    """x_cl [B, C, H, W] channels_last, w_ohwi [N, 3, 3, C] contiguous -> y [B, N, H, W] channels_last.
    mask [B, N, H, W] channels_last bf16: the result is zeroed where mask <= 0 (``mdetr_conv3x3_masked``)."""
    B, C, H, W = x_cl.shape
    N = w_ohwi.shape[0]
    y = torch.empty((B, N, H, W), dtype=torch.bfloat16, device=x_cl.device, memory_format=torch.channels_last)
    cuda = x_cl.is_cuda
    flags = (1 if relu else 0) | (2 if mirror else 0)
    dev_i, stream = x_cl.device.index if cuda else -1, torch.cuda.current_stream(x_cl.device).cuda_stream if cuda else None
    if mask is not None:
        assert mask.shape == y.shape and mask.dtype == torch.bfloat16 and mask.is_contiguous(memory_format=torch.channels_last)
        rc = _lib().mdetr_conv3x3_masked(x_cl.data_ptr(), w_ohwi.data_ptr(), shift.data_ptr() if shift is not None else None, mask.data_ptr(),
                                         y.data_ptr(), B, H, W, C, N, flags, dev_i, stream)
    else:
        rc = _lib().mdetr_conv3x3_forward(x_cl.data_ptr(), w_ohwi.data_ptr(), shift.data_ptr() if shift is not None else None, y.data_ptr(),
                                          B, H, W, C, N, flags, dev_i, stream)
    if rc != 0:
        _capi.check(rc, "mdetr_conv3x3_masked" if mask is not None else "mdetr_conv3x3_forward")
    return y


class _Conv3x3(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, shift, relu, hand_out_token=False, in_token=None):
        # in_token (linear.ReluToken): x is a ReLU output that only this convolution consumes; its producer leaves the ReLU's backward
        # mask to THIS backward, which applies it where the input gradient leaves the kernel (mdetr_conv3x3_masked) -- or, on the
        # library route, by one threshold_backward of its own
        ctx.in_token = in_token
        if in_token is not None:
            in_token.premasked = True
        w = _ohwi(weight)
        sh = None if shift is None else shift.float().contiguous()
        y = _launch(x, w, sh, relu)
        ctx.relu = bool(relu)
        ctx.shift_dtype = None if shift is None else shift.dtype
        ctx.w_ihwo = getattr(weight, "_mdetr_ihwo", None)              # [C, 3, 3, N] copy made with the weight (csrc/wfold.hip), if any
        ctx.relu_token = None
        if relu and hand_out_token:
            from .monodetr.linear import ReluToken
            ctx.relu_token = y._mdetr_relu_token = ReluToken()         # the one consumer of y may take over this ReLU's backward mask
        ctx.save_for_backward(x, weight, w, *((y,) if relu else ()))
        return y

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, dy):
        x, weight, w = ctx.saved_tensors[:3]
        if ctx.relu and not (ctx.relu_token is not None and ctx.relu_token.premasked):      # (premasked: the consumer's input gradient came masked)
            dy = torch.ops.aten.threshold_backward(dy, ctx.saved_tensors[3], 0.0)
        dy = dy.contiguous(memory_format=torch.channels_last)
        dx = dw = None
        if ctx.needs_input_grad[0]:
            # dX = conv(dY, w') with w'[c, t, s, n] = w[n, 2 - t, 2 - s, c]: the same kernel on the weight with its channel axes
            # swapped (one small copy), the taps mirrored by the kernel's addressing
            if dy.shape[1] % 64 == 0 and x.shape[1] % 32 == 0 and dy.data_ptr() % 16 == 0:
                wt = ctx.w_ihwo
                if wt is None or wt.shape != (w.shape[3], 3, 3, w.shape[0]) or not wt.is_contiguous() or wt.dtype != w.dtype:
                    wt = w.permute(3, 1, 2, 0).contiguous()
                masked = ctx.in_token is not None and x.dtype == torch.bfloat16 and x.is_contiguous(memory_format=torch.channels_last)
                dx = _launch(dy, wt, None, False, mirror=True, mask=x if masked else None)
                if ctx.in_token is not None and not masked:
                    dx = torch.ops.aten.threshold_backward(dx, x, 0.0)
            else:
                dx = torch.ops.aten.convolution_backward(dy, x, weight, None, (1, 1), (1, 1), (1, 1), False, (0, 0), 1, (True, False, False))[0]
                if ctx.in_token is not None:
                    dx = torch.ops.aten.threshold_backward(dx, x, 0.0)
        if ctx.needs_input_grad[1]:
            from . import conv_wgrad_ext
            if conv_wgrad_ext.supported(x, dy, 3, 1):                    # csrc/conv_wgrad.hip: split-K over pixel tiles on the matrix cores
                dw = conv_wgrad_ext.weight_gradient(x, dy, 3, 1, weight.dtype)
            else:
                dw = torch.ops.aten.convolution_backward(dy, x, weight, None, (1, 1), (1, 1), (1, 1), False, (0, 0), 1, (False, True, False))[1]
        ds = None
        if ctx.shift_dtype is not None and ctx.needs_input_grad[2]:
            # a trainable shift (a convolution bias): its gradient is the column sum of dY over the B*H*W pixels
            dy2 = dy.permute(0, 2, 3, 1).reshape(-1, dy.shape[1])
            if dy2.is_cuda and _backend is None:
                from .colsum_ext import column_sum, supported as colsum_ok
                od = ctx.shift_dtype if ctx.shift_dtype in (torch.float32, torch.bfloat16) else torch.float32       # (one rounding, no cast launch)
                ds = column_sum(dy2, out_dtype=od) if colsum_ok(dy2) else dy2.float().sum(0)
            else:
                ds = dy2.float().sum(0)
            ds = ds.to(ctx.shift_dtype)
        return dx, dw, ds, None, None, None


def conv3x3(x, weight, shift=None, relu=False, hand_out_token=False, in_token=None):
    """act(conv2d(x, weight, padding=1) + shift[None, :, None, None]); ``shift`` [N]: a frozen-BN shift or a trainable bias."""
    if not supported(x, weight):
        raise RuntimeError("conv3x3: needs a CUDA bf16 channels_last activation with C % 64 == 0 and a bf16 [N, C, 3, 3] weight with N % 32 == 0")
    return _Conv3x3.apply(x, weight, shift, relu, hand_out_token, in_token if x.requires_grad else None)


class Conv3x3(torch.nn.Conv2d):
    """nn.Conv2d (same parameters, same state_dict keys) whose forward takes the kernel when MDETR_CONV3X3=1 and the call
    qualifies (3x3 / stride 1 / pad 1, bf16 channels_last, C % 64 == 0, N % 32 == 0), and nn.Conv2d's otherwise."""

    def forward(self, x):
        if ENABLED and x.dtype == self.weight.dtype and not torch.is_autocast_enabled() \
                and supported(x, self.weight, self.stride, self.padding, self.dilation, self.groups) and self.padding_mode == "zeros":
            return conv3x3(x, self.weight, self.bias, relu=False)
        return super().forward(x)
