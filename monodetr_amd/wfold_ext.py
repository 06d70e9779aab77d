"""``fold_weights`` / ``unfold_grads``: the frozen-BatchNorm fold of many trainable convolution weights in ONE launch each way
(csrc/wfold.hip through ``mdetr_fold_weights`` / ``mdetr_unfold_grads``).  The reference applies FrozenBatchNorm2d behind every
convolution (lib/models/monodetr/backbone.py:27-64); monodetr/backbone.py folds its scale into the weight, which for trainable
weights is per-iteration work: folded = bf16(W * scale) forward, dW = float(dfolded) * scale backward.  Weights are channels-last
(OHWI in memory); 3x3 weights also get their [C][tap][O] copy, the operand of the input-gradient kernels (csrc/conv3x3.hip with
mirrored taps, csrc/conv_taps.hip), which then need no transposing copy of their own."""
import ctypes
import os

import torch

from . import _capi

# MDETR_WFOLD=1 (kernel_families decides: committed for bf16)
ENABLED = os.environ.get("MDETR_WFOLD") == "1"
_backend = None      # tests substitute the same kernel source built for the host (tests/native_emul.py)


def _lib():
    return _backend if _backend is not None else _capi.lib()


def _ohwi_dense(t):
    """A 4-D tensor whose memory is [O][kh][kw][C] without gaps (channels-last; every layout of a 1x1 weight qualifies)."""
    return t.dim() == 4 and t.permute(0, 2, 3, 1).is_contiguous()


def supported(weights, scales, dt):
    """fp32 channels-last weights [O, C, kh, kw] with fp32 scales [O] on one device, bf16 results, O and C multiples of 8."""
    if dt != torch.bfloat16 or not weights:
        return False
    dev = weights[0].device
    if not (dev.type == "cuda" or _backend is not None):
        return False
    for w, s in zip(weights, scales):
        if not (w.dtype == torch.float32 and s.dtype == torch.float32 and w.device == dev and s.device == dev and _ohwi_dense(w)
                and s.dim() == 1 and s.is_contiguous() and s.shape[0] == w.shape[0] and w.shape[0] % 8 == 0 and w.shape[1] % 8 == 0
                and w.data_ptr() % 16 == 0 and w.numel() % 8 == 0):
            return False
    return True


def _ptr_array(tensors):
    return (ctypes.c_void_p * len(tensors))(*[t.data_ptr() if t is not None else None for t in tensors])


def _dims(weights):
    n = len(weights)
    mk = lambda vals: (ctypes.c_int * n)(*vals)
    return mk([w.shape[0] for w in weights]), mk([w.shape[1] for w in weights]), mk([w.shape[2] * w.shape[3] for w in weights])


def _where(t):
    cuda = t.is_cuda
    return (t.device.index if cuda else -1), (torch.cuda.current_stream(t.device).cuda_stream if cuda else None)


@torch.no_grad()
def fold_weights(weights, scales, transposed):
    """-> (folded, folded_t): folded[i] = bf16(weights[i] * scales[i][:, None, None, None]) with channels-last strides; folded_t[i] =
    the same values as a contiguous [C, kh, kw, O] tensor where transposed[i], else None.  Two allocations for all of them."""
    n = len(weights)
    pad = lambda k: -(-k // 8) * 8
    offs, total = [], 0
    for w in weights:
        offs.append(total)
        total += pad(w.numel())
    offs_t, total_t = [], 0
    for w, t in zip(weights, transposed):
        offs_t.append(total_t if t else None)
        if t:
            total_t += pad(w.numel())
    dev = weights[0].device
    flat = torch.empty(total, dtype=torch.bfloat16, device=dev)
    flat_t = torch.empty(total_t, dtype=torch.bfloat16, device=dev) if total_t else None
    folded, folded_t = [], []
    for w, o, ot in zip(weights, offs, offs_t):
        O, C, kh, kw = w.shape
        folded.append(flat[o:o + w.numel()].view(O, kh, kw, C).permute(0, 3, 1, 2))
        folded_t.append(flat_t[ot:ot + w.numel()].view(C, kh, kw, O) if ot is not None else None)
    Os, Cs, taps = _dims(weights)
    device, stream = _where(weights[0])
    rc = _lib().mdetr_fold_weights(n, _ptr_array(weights), _ptr_array(scales), _ptr_array(folded), _ptr_array(folded_t) if flat_t is not None else None,
                                   Os, Cs, taps, device, stream)
    if rc != 0:
        _capi.check(rc, "mdetr_fold_weights")
    return folded, folded_t


def grads_supported(grads, weights):
    return all(g is not None and g.dtype == torch.bfloat16 and g.shape == w.shape and g.device == w.device and _ohwi_dense(g)
               and g.data_ptr() % 16 == 0 for g, w in zip(grads, weights))


@torch.no_grad()
def unfold_grads(grads, scales, weights):
    """-> [float(grads[i]) * scales[i][:, None, None, None]] as fp32 tensors with the weights' (channels-last) strides; one allocation."""
    n = len(grads)
    pad = lambda k: -(-k // 8) * 8
    offs, total = [], 0
    for w in weights:
        offs.append(total)
        total += pad(w.numel())
    flat = torch.empty(total, dtype=torch.float32, device=weights[0].device)
    out = []
    for w, o in zip(weights, offs):
        O, C, kh, kw = w.shape
        out.append(flat[o:o + w.numel()].view(O, kh, kw, C).permute(0, 3, 1, 2))
    Os, Cs, taps = _dims(weights)
    device, stream = _where(weights[0])
    rc = _lib().mdetr_unfold_grads(n, _ptr_array(grads), _ptr_array(scales), _ptr_array(out), Os, Cs, taps, device, stream)
    if rc != 0:
        _capi.check(rc, "mdetr_unfold_grads")
    return out
