"""``preprocess_batch``: the training image path of the input pipeline on the device, one launch per batch
(csrc/kitti_prep.hip through ``mdetr_kitti_preprocess``): photometric distortion, flip, PIL-exact affine/bilinear warp,
normalisation and HWC -> CHW of ``lib/datasets/kitti/kitti_dataset.py:127-163``, from decoded RGB8 images.

``DESCRIPTOR`` is the numpy dtype of ``MdetrKittiImage`` (include/monodetr_amd.h); the host side of the pipeline
(``monodetr_amd/datasets``) fills one record per image."""
import os

import numpy as np
import torch

from . import _capi

_backend = None               # tests substitute the host build of the same arithmetic (tests/native)

FLIP, DISTORT, CONTRAST_FIRST, BRIGHTNESS, CONTRAST, SATURATION, HUE = 1, 2, 4, 8, 16, 32, 64
IDENTITY_PERM = 0x24

DESCRIPTOR = np.dtype([('pixel_offset', '<i8'), ('width', '<i4'), ('height', '<i4'), ('flags', '<u4'), ('perm', '<u4'),
                       ('brightness', '<f4'), ('contrast', '<f4'), ('saturation', '<f4'), ('hue', '<f4'),
                       ('inv', '<f8', (6,))], align=True)
assert DESCRIPTOR.itemsize == 88

_MEAN = (0.485, 0.456, 0.406)          # kitti_dataset.py:79-80
_STD = (0.229, 0.224, 0.225)


def _lib():
    return _backend if _backend is not None else _capi.lib()


def preprocess_batch(pixels, descriptors, out_hw=(384, 1280), dtype=torch.float32, mean=_MEAN, std=_STD, out=None,
                     channels_last=False):
    """pixels: uint8 1-D tensor (all images of the batch back to back); descriptors: uint8 tensor viewing
    ``DESCRIPTOR`` records [n * 88] on the same device -> [n, 3, H, W] ``dtype`` tensor on that device.
    ``channels_last`` lays the result out as torch's channels_last memory format (what the convolutions want).
    Runs on the current stream; no synchronisation."""
    if pixels.dtype != torch.uint8 or descriptors.dtype != torch.uint8 or not pixels.is_contiguous() or not descriptors.is_contiguous():
        raise RuntimeError("pixels and descriptors must be contiguous uint8 tensors")
    if pixels.device != descriptors.device:
        raise RuntimeError("pixels is on %s but descriptors on %s" % (pixels.device, descriptors.device))
    if descriptors.numel() % DESCRIPTOR.itemsize != 0:
        raise RuntimeError("descriptors must hold whole %d-byte records" % DESCRIPTOR.itemsize)
    if not pixels.is_cuda and _backend is None:
        raise RuntimeError("Not implemented on the CPU")
    if dtype not in (torch.float32, torch.bfloat16):
        raise RuntimeError("output dtype must be float32 or bfloat16")
    n = descriptors.numel() // DESCRIPTOR.itemsize
    H, W = out_hw
    fmt = torch.channels_last if channels_last else torch.contiguous_format
    if out is None:
        out = torch.empty((n, 3, H, W), dtype=dtype, device=pixels.device, memory_format=fmt)
    elif tuple(out.shape) != (n, 3, H, W) or out.dtype != dtype or not out.is_contiguous(memory_format=fmt) or out.device != pixels.device:
        raise RuntimeError("out must be a dense [n,3,H,W] tensor of the requested dtype and memory format on the pixels' device")
    m = np.asarray(mean, dtype=np.float32)
    s = np.asarray(std, dtype=np.float32)
    cuda = pixels.is_cuda
    rc = _lib().mdetr_kitti_preprocess(
        pixels.data_ptr(), descriptors.data_ptr(), n, out.data_ptr(),
        _capi.MDETR_F32 if dtype == torch.float32 else _capi.MDETR_BF16, H, W, 1 if channels_last else 0, m.ctypes.data, s.ctypes.data,
        pixels.device.index if cuda else -1, torch.cuda.current_stream(pixels.device).cuda_stream if cuda else None)
    if rc != 0:
        _capi.check(rc, "mdetr_kitti_preprocess")
    return out


# ---- known-answer self check ---------------------------------------------------------------------------------------
# A defect in the device image path would not crash: it would silently feed the network wrong pixels.  The first batch a
# DeviceLoader uploads on a device is therefore preceded by one tiny launch of the same kernel on a fixed 48 x 36 image
# with every stage of the chain switched on (flip, brightness, saturation, hue, contrast-last, channel permutation, a
# non-trivial affine map), whose float32 output must hash to the value below.  tests/test_kitti_pipeline_cpu.py holds that
# constant to the numpy / PIL oracle of the reference chain (oracle/kitti_pipeline.py, itself pinned on the fixture
# recorded from the reference's KITTI_Dataset).
_SELF_CHECK_SHA256 = "038e0608be770aea8ba68b60d368f5aba58f0791ea6e2574f21fc50fdab98d9a"
_self_checked = set()


def self_check_inputs():
    v = np.arange(36 * 48 * 3, dtype=np.uint32)
    img = ((v * np.uint32(2654435761)) >> np.uint32(13)).astype(np.uint8).reshape(36, 48, 3)
    d = np.zeros(1, dtype=DESCRIPTOR)
    d['width'], d['height'] = 48, 36
    d['flags'] = FLIP | DISTORT | BRIGHTNESS | CONTRAST | SATURATION | HUE
    d['perm'] = 1 | (2 << 2) | (0 << 4)
    d['brightness'], d['contrast'], d['saturation'], d['hue'] = 11.5, 1.25, 0.75, -9.0
    d['inv'] = np.array([1.17, 0.031, -1.4, -0.027, 1.43, -0.8])
    return img, d


def self_check(device):
    """Once per device and process; raises if the kernel's output for the known input is not the known answer."""
    import hashlib
    device = torch.device(device)
    if device in _self_checked:
        return
    img, d = self_check_inputs()
    px = torch.from_numpy(img.reshape(-1).copy()).to(device)
    ds = torch.from_numpy(d.view(np.uint8).reshape(-1).copy()).to(device)
    out = preprocess_batch(px, ds, out_hw=(24, 40))
    got = hashlib.sha256(out.cpu().contiguous().numpy().tobytes()).hexdigest()
    if got != _SELF_CHECK_SHA256:
        # MDETR_TUNE="prep_self_check=warn": a new compiler / architecture may legitimately differ in a last bit (the bit-exact answer
        # is pinned to the reference chain on gfx950 + ROCm 7.2); the operator of such a system decides, not a hard stop
        msg = ("mdetr_kitti_preprocess failed its known-answer check on %s (got %s): the device image path must not be "
               "trusted (MDETR_TUNE=prep_self_check=warn continues)" % (device, got[:16]))
        from . import _tune
        if _tune.get("prep_self_check", "strict") != "warn":
            raise RuntimeError(msg)
        import warnings
        warnings.warn(msg)
    _self_checked.add(device)
