"""``conv_strided(x, weight, shift, relu)``: the stride-2 convolutions of the backbone, the fourth pyramid level and the depth
predictor -- 3x3 / stride 2 / pad 1 and 1x1 / stride 2 on a channels_last bf16 activation, with the folded frozen-BN shift (or
a trainable bias) and the ReLU in the epilogue (csrc/conv_taps.hip through ``mdetr_conv_taps``).  Forward and input gradient
run on that kernel -- the input gradient as the four pixel-parity classes of the stride-2 transpose, each a stride-1 problem
with 1x1 / 1x2 / 2x1 / 2x2 taps --, the weight gradient on csrc/conv_wgrad.hip (conv_wgrad_ext)."""
import os

import torch

from . import _capi

_backend = None               # tests substitute the CPU emulation of the same kernel source (tests/native_emul.py)
ENABLED = os.environ.get("MDETR_CONV_STRIDED") == "1"


def _lib():
    return _backend if _backend is not None else _capi.lib()


def supported(x, weight, stride=(2, 2), padding=(1, 1), dilation=(1, 1), groups=1):
    """x [B, C, H, W] channels_last bf16; weight [N, C, 3, 3] with padding 1 or [N, C, 1, 1] with padding 0; stride 2."""
    k = tuple(weight.shape[2:]) if weight.dim() == 4 else None
    return ((x.is_cuda or _backend is not None) and x.dim() == 4 and weight.dim() == 4 and x.dtype == torch.bfloat16
            and weight.dtype == torch.bfloat16 and tuple(stride) == (2, 2) and tuple(dilation) == (1, 1) and groups == 1
            and ((k == (3, 3) and tuple(padding) == (1, 1)) or (k == (1, 1) and tuple(padding) == (0, 0)))
            and weight.shape[1] == x.shape[1] and x.shape[1] % 64 == 0 and weight.shape[0] % 64 == 0 and x.numel() > 0
            and x.is_contiguous(memory_format=torch.channels_last) and x.data_ptr() % 16 == 0)


def _ohwi(weight):
    """[N, C, kh, kw] -> contiguous [N, kh, kw, C] (a view when the weight is channels_last already)."""
    w = weight.permute(0, 2, 3, 1).contiguous()
    return w if w.data_ptr() % 16 == 0 else w.clone()


def _call(x, w, shift, y, dims, relu):
    cuda = x.is_cuda
    d = torch.tensor(dims, dtype=torch.int64)
    rc = _lib().mdetr_conv_taps(x.data_ptr(), w.data_ptr(), shift.data_ptr() if shift is not None else None, y.data_ptr(), d.data_ptr(),
                                1 if relu else 0, x.device.index if cuda else -1, torch.cuda.current_stream(x.device).cuda_stream if cuda else None)
    if rc != 0:
        _capi.check(rc, "mdetr_conv_taps")


def _split_count(B, OH, OW, N, C, k, relu):
    """Contraction splits for the forward pass, 0 = none: few output tiles (4 x 32 pixels x 32 channels each) against many channel
    slabs -- split until ~1 000 workgroups exist, at least 4 slabs of 32 channels per split."""
    from . import _tune
    if relu or k != 3 or _tune.get("conv_taps_split", "1") == "0":
        return 0
    blocks = B * ((OH + 3) // 4) * ((OW + 31) // 32) * (N // 32)
    slabs = C // 32
    if blocks >= 256 or slabs < 16:
        return 0
    ks = 1
    while ks * 2 <= 64 and blocks * ks * 2 <= 1024 and slabs % (ks * 2) == 0 and slabs // (ks * 2) >= 4:
        ks *= 2
    return ks if ks >= 2 else 0


def _forward_split(x, w_ohwi, shift, ks, dims):
    """The forward pass with the contraction split ks ways (mdetr_conv_taps_split) + one sum over the splits (csrc/colsum.hip)."""
    B, OH, OW, N = dims[0], dims[4], dims[5], dims[6]
    cuda = x.is_cuda
    cols = B * OH * OW * N
    if cuda and _backend is None:
        from . import _workspace as W_
        part = W_.get("conv_taps_split", x.device, ks * cols * 4).view(torch.float32)[:ks * cols]
    else:
        part = torch.empty(ks * cols, dtype=torch.float32, device=x.device)
    d = torch.tensor(dims, dtype=torch.int64)
    rc = _lib().mdetr_conv_taps_split(x.data_ptr(), w_ohwi.data_ptr(), shift.data_ptr() if shift is not None else None, part.data_ptr(),
                                      part.numel(), d.data_ptr(), ks, x.device.index if cuda else -1,
                                      torch.cuda.current_stream(x.device).cuda_stream if cuda else None)
    if rc != 0:
        _capi.check(rc, "mdetr_conv_taps_split")
    part = part.view(ks, cols)
    if cuda and _backend is None:
        from .colsum_ext import column_sum, supported as colsum_ok
        y = column_sum(part, out_dtype=torch.bfloat16) if colsum_ok(part) else part.sum(0).to(torch.bfloat16)
    else:
        y = part.sum(0).to(torch.bfloat16)
    return y.view(B, OH, OW, N).permute(0, 3, 1, 2)                     # [B, N, OH, OW] with channels_last strides


def _forward(x, w_ohwi, shift, relu):
    """x [B, C, H, W] channels_last, w_ohwi [N, k, k, C] -> y [B, N, OH, OW] channels_last."""
    B, C, H, W = x.shape
    N, k = w_ohwi.shape[0], w_ohwi.shape[1]
    OH, OW = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    ks = _split_count(B, OH, OW, N, C, k, relu)
    if ks:
        return _forward_split(x, w_ohwi, shift, ks, [B, H, W, C, OH, OW, N, 2, k, k, 1, 1, 0, 1, 0, 1, 0, OH * OW * N, OW * N, N, k * k * C, k * C, C])
    y = torch.empty((B, N, OH, OW), dtype=torch.bfloat16, device=x.device, memory_format=torch.channels_last)
    pad = 1 if k == 3 else 0
    _call(x, w_ohwi, shift, y, [B, H, W, C, OH, OW, N, 2, k, k, pad, pad, 0, 1, 0, 1, 0, OH * OW * N, OW * N, N, k * k * C, k * C, C], relu)
    return y


def _input_gradient(dy, w_ohwi, H, W, w_ihwo=None):
    """dX [B, C, H, W] of the stride-2 convolution from dY [B, N, OH, OW] (channels_last): the four pixel-parity classes in one
    launch (``mdetr_conv_dgrad_s2``); every element of dX is written by it."""
    B, N, OH, OW = dy.shape
    k, C = w_ohwi.shape[1], w_ohwi.shape[3]
    wt = w_ihwo                                                         # [C, k, k, N]: made with the weight (csrc/wfold.hip) ...
    if wt is None or wt.shape != (C, k, k, N) or not wt.is_contiguous() or wt.dtype != w_ohwi.dtype:
        wt = w_ohwi.permute(3, 1, 2, 0).contiguous()                    # ... or one small copy here
    dx = torch.empty((B, C, H, W), dtype=torch.bfloat16, device=dy.device, memory_format=torch.channels_last)
    cuda = dy.is_cuda
    rc = _lib().mdetr_conv_dgrad_s2(dy.data_ptr(), wt.data_ptr(), dx.data_ptr(), B, OH, OW, N, H, W, C, k, dy.device.index if cuda else -1,
                                    torch.cuda.current_stream(dy.device).cuda_stream if cuda else None)
    if rc != 0:
        _capi.check(rc, "mdetr_conv_dgrad_s2")
    return dx


class _ConvStrided(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, shift, relu, hand_out_token=False):
        w = _ohwi(weight)
        sh = None if shift is None else shift.float().contiguous()
        y = _forward(x, w, sh, relu)
        ctx.relu = bool(relu)
        ctx.shift_dtype = None if shift is None else shift.dtype
        ctx.w_ihwo = getattr(weight, "_mdetr_ihwo", None)
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
        if ctx.relu and not (ctx.relu_token is not None and ctx.relu_token.premasked):
            dy = torch.ops.aten.threshold_backward(dy, ctx.saved_tensors[3], 0.0)
        dy = dy.contiguous(memory_format=torch.channels_last)
        k = weight.shape[2]
        pad = (1, 1) if k == 3 else (0, 0)
        dx = dw = ds = None
        if ctx.needs_input_grad[0]:
            dx = _input_gradient(dy, w, x.shape[2], x.shape[3], ctx.w_ihwo)
        if ctx.needs_input_grad[1]:
            from . import conv_wgrad_ext
            if conv_wgrad_ext.supported(x, dy, k, 2):
                dw = conv_wgrad_ext.weight_gradient(x, dy, k, 2, weight.dtype)
            else:
                dw = torch.ops.aten.convolution_backward(dy, x, weight, None, (2, 2), pad, (1, 1), False, (0, 0), 1, (False, True, False))[1]
        if ctx.shift_dtype is not None and ctx.needs_input_grad[2]:
            dy2 = dy.permute(0, 2, 3, 1).reshape(-1, dy.shape[1])
            if dy2.is_cuda and _backend is None:
                from .colsum_ext import column_sum, supported as colsum_ok
                od = ctx.shift_dtype if ctx.shift_dtype in (torch.float32, torch.bfloat16) else torch.float32       # (one rounding, no cast launch)
                ds = column_sum(dy2, out_dtype=od) if colsum_ok(dy2) else dy2.float().sum(0)
            else:
                ds = dy2.float().sum(0)
            ds = ds.to(ctx.shift_dtype)
        return dx, dw, ds, None, None


def conv_strided(x, weight, shift=None, relu=False, hand_out_token=False):
    """act(conv2d(x, weight, stride=2, padding=1 (3x3) / 0 (1x1)) + shift[None, :, None, None])."""
    if not supported(x, weight, padding=(1, 1) if weight.shape[2] == 3 else (0, 0)):
        raise RuntimeError("conv_strided: needs a bf16 channels_last activation with C % 64 == 0 and a bf16 [N, C, 3, 3] / [N, C, 1, 1] weight with N % 64 == 0")
    return _ConvStrided.apply(x, weight, shift, relu, hand_out_token)


class ConvStrided(torch.nn.Conv2d):
    """nn.Conv2d (same parameters, same state_dict keys) whose forward takes the kernel when MDETR_CONV_STRIDED=1 and the call
    qualifies, and nn.Conv2d's otherwise (monodetr.py:87-92, depth_predictor.py:29-31: 3x3 / stride 2 / pad 1 with a bias)."""

    def forward(self, x):
        if ENABLED and x.dtype == self.weight.dtype and not torch.is_autocast_enabled() and self.padding_mode == "zeros" \
                and supported(x, self.weight, self.stride, self.padding, self.dilation, self.groups):
            return conv_strided(x, self.weight, self.bias, relu=False)
        return super().forward(x)
