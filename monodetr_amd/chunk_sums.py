"""The chunk sums of the split weight-gradient kernels (csrc/twgrad.hip, csrc/conv_wgrad.hip), batched.

Those kernels cut their contraction over the tokens / pixels into chunks and leave fp32 partials ``[chunks, cols]`` that are added
in chunk order (deterministic, one rounding).  Added right away that is one small launch per weight gradient -- 159 of them in the
round-5 iteration, ~6 us each, mostly fill and drain.  Inside ``deferred()`` (``helpers/step_helper.TrainIteration`` wraps every
backward call in it) the sum is only REGISTERED: the caller gets the result tensor at once, its values arrive with ``flush()`` --
ONE launch of ``mdetr_chunk_sums`` for up to 48 gradients -- when the context closes, or earlier where something reads a deferred
result (the frozen-BN unfold of the backbone's weight gradients, monodetr/backbone.py).  The arithmetic is the same sum in the same
order.  Outside the context every sum runs immediately.

What makes the deferral safe: a weight gradient leaves its autograd function only towards the parameter's AccumulateGrad node,
which stores the tensor without reading it (the iteration clears ``.grad`` to None first); the two readers inside the backward pass
flush first.  Gradient exchange and optimizer run after the context has closed."""
import ctypes
import os
import threading

import torch

from . import _capi

# MDETR_CHUNK_SUMS=1 (kernel_families decides): batch the sums; off = every sum its own launch, as in round 5
ENABLED = os.environ.get("MDETR_CHUNK_SUMS") == "1"
# tests: a registered result is filled with NaN until its flush, so that anything reading it too early shows (a recycled buffer
# otherwise tends to hold last iteration's -- plausible -- values)
POISON = False
# tests: every sum at once, through the same kernel (the reference the batched results must equal bit for bit)
IMMEDIATE = False
_backend = None               # tests substitute the CPU emulation of the same kernel source (tests/native_emul.py)
_lock = threading.RLock()     # (registrations come from autograd's device thread, flush() from either)
_depth = 0
_pending = []                 # (part [chunks, cols] fp32, out [cols]); both stay referenced until the flush


class _Job(ctypes.Structure):
    _fields_ = [("part", ctypes.c_void_p), ("out", ctypes.c_void_p), ("cols", ctypes.c_int64), ("chunks", ctypes.c_int32), ("out_dtype", ctypes.c_int32)]


def _lib():
    return _backend if _backend is not None else _capi.lib()


def supported(part, out_dtype):
    return ((part.is_cuda or _backend is not None) and part.dim() == 2 and part.dtype == torch.float32 and part.is_contiguous()
            and part.shape[1] % 4 == 0 and part.shape[0] > 0 and part.data_ptr() % 16 == 0 and out_dtype in (torch.float32, torch.bfloat16))


def _launch(jobs):
    arr = (_Job * len(jobs))()
    for q, (part, out) in zip(arr, jobs):
        q.part, q.out, q.cols, q.chunks = part.data_ptr(), out.data_ptr(), part.shape[1], part.shape[0]
        q.out_dtype = _capi.MDETR_BF16 if out.dtype == torch.bfloat16 else _capi.MDETR_F32
    dev = jobs[0][0].device
    rc = _lib().mdetr_chunk_sums(ctypes.cast(arr, ctypes.c_void_p), len(jobs), dev.index if dev.type == "cuda" else -1,
                                 torch.cuda.current_stream(dev).cuda_stream if dev.type == "cuda" else None)
    if rc != 0:
        msg = _lib().mdetr_last_error()
        raise RuntimeError("mdetr_chunk_sums failed (code %d): %s" % (rc, msg.decode() if msg else "?"))


def chunk_sum(part, out_dtype=torch.float32):
    """part [chunks, cols] fp32 -> [cols] in out_dtype = the chunks added in order.  Inside ``deferred()`` the values arrive at the
    next ``flush()``; `part` must then be a tensor of its own (not a shared scratch buffer): it is read later."""
    if not supported(part, out_dtype):
        raise RuntimeError("chunk_sum: needs contiguous fp32 partials [chunks, cols], cols a multiple of 4, 16-byte aligned")
    out = torch.empty(part.shape[1], dtype=out_dtype, device=part.device)
    with _lock:
        if ENABLED and _depth > 0 and not IMMEDIATE:
            if POISON:
                out.fill_(float("nan"))
            _pending.append((part, out))
            return out
    _launch([(part, out)])
    return out


def deferring():
    return ENABLED and _depth > 0


def flush():
    """Compute every registered sum now (on the current stream of the calling thread -- the stream the producers ran on)."""
    with _lock:
        jobs = list(_pending)
        del _pending[:]
    if jobs:
        by_dev = {}
        for j in jobs:
            by_dev.setdefault(j[0].device, []).append(j)
        for group in by_dev.values():
            _launch(group)


class deferred:
    """``with deferred(): loss.backward()`` -- chunk sums registered inside are computed together when the block ends (also when it
    ends with an exception: no registered result stays unwritten)."""

    def __enter__(self):
        global _depth
        with _lock:
            _depth += 1
        return self

    def __exit__(self, *exc):
        global _depth
        with _lock:
            _depth -= 1
            last = _depth == 0
        if last:
            flush()
        return False
