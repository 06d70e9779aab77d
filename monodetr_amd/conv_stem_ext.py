"""``conv_stem(x, weight, shift)``: the ResNet stem -- 7x7 / stride 2 / pad 3 convolution of the 3-channel image, frozen-BN shift
and ReLU in the epilogue (csrc/conv_stem.hip through ``mdetr_conv_stem``).  The stem is frozen (lib/models/monodetr/
backbone.py:71-73): forward only; the packed weight is built once per folded weight."""
import os

import torch

from . import _capi

_backend = None               # tests substitute the CPU emulation of the same kernel source (tests/native_emul.py)
ENABLED = os.environ.get("MDETR_CONV_STEM") == "1"
K_PACKED = 176


def _lib():
    return _backend if _backend is not None else _capi.lib()


def supported(x, weight, stride=(2, 2), padding=(3, 3), dilation=(1, 1), groups=1):
    return ((x.is_cuda or _backend is not None) and x.dim() == 4 and weight.dim() == 4 and x.dtype == torch.bfloat16
            and weight.dtype == torch.bfloat16 and tuple(weight.shape) == (64, 3, 7, 7) and x.shape[1] == 3 and tuple(stride) == (2, 2)
            and tuple(padding) == (3, 3) and tuple(dilation) == (1, 1) and groups == 1 and x.numel() > 0
            and x.is_contiguous(memory_format=torch.channels_last) and not x.requires_grad and not weight.requires_grad)


def pack_weight(weight):
    """[64, 3, 7, 7] -> [64, 176] bf16: row n holds, per tap row t, the 21 values w[n, ch, t, e] in (e, ch) order -- the order
    of 7 consecutive channels-last pixels -- padded to 24; 8 zeros at the end."""
    w = weight.detach().permute(0, 2, 3, 1).reshape(64, 7, 21)                  # [n, t, e * 3 + ch]
    out = torch.zeros(64, K_PACKED, dtype=torch.bfloat16, device=weight.device)
    out[:, :168].view(64, 7, 24)[:, :, :21] = w.to(torch.bfloat16)
    return out


def _launch(x, packed, shift, relu=True):
    assert relu, "the stem's epilogue is shift + ReLU"
    B, _, H, W = x.shape
    OH, OW = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    y = torch.empty((B, 64, OH, OW), dtype=torch.bfloat16, device=x.device, memory_format=torch.channels_last)
    cuda = x.is_cuda
    rc = _lib().mdetr_conv_stem(x.data_ptr(), packed.data_ptr(), shift.data_ptr() if shift is not None else None, y.data_ptr(), B, H, W,
                                x.device.index if cuda else -1, torch.cuda.current_stream(x.device).cuda_stream if cuda else None)
    if rc != 0:
        _capi.check(rc, "mdetr_conv_stem")
    return y


_packed = {}


def conv_stem(x, weight, shift):
    """relu(conv2d(x, weight, stride=2, padding=3) + shift); the packed form of `weight` is cached per (storage, version)."""
    if not supported(x, weight):
        raise RuntimeError("conv_stem: needs a bf16 channels_last [B, 3, H, W] image batch and a frozen bf16 [64, 3, 7, 7] weight")
    # keyed by the tensor OBJECT (a weak reference: a freed tensor's address and version may come back with another weight) and its
    # version counter
    hit = _packed.get("w")
    if hit is None or hit[0]() is not weight or hit[1] != weight._version or (hit[4]() is not shift if shift is not None else hit[4] is not None):
        import weakref
        hit = _packed["w"] = (weakref.ref(weight), weight._version, pack_weight(weight), None if shift is None else shift.float().contiguous(),
                              None if shift is None else weakref.ref(shift))
    return _launch(x, hit[2], hit[3])
