"""Scratch buffers of the HIP kernels (the C ABI never allocates: the caller hands a workspace in).

One buffer per (kernel family, device, STREAM): calls issued on different streams may run concurrently and must not share
scratch.  A buffer that has to grow is replaced, and the superseded one is KEPT alive: a captured hipGraph
(helpers/step_helper.TrainIteration) holds the addresses it was recorded with, and a later eager call with a larger shape -- an
evaluation batch, another resolution -- must not hand that memory back to the caching allocator while replays still use it.
Growth is geometric, so what is retained is bounded by the largest request."""
import torch

_live = {}
_retired = []


def get(tag, device, nbytes, zero=False, floor=0):
    """A uint8 tensor of at least `nbytes` for kernel family `tag` on the current stream of `device`.  zero: the buffer is
    zero-filled when it is (re)allocated -- for kernels that keep an invariant in it between calls (msda_fused's ``far``
    cookie, the loss kernels' counters): freshly allocated pool memory may hold a stale, valid-looking header."""
    device = torch.device(device)
    stream = torch.cuda.current_stream(device).cuda_stream if device.type == "cuda" else 0
    key = (tag, device, stream)
    ws = _live.get(key)
    if ws is None or ws.numel() < nbytes:
        size = max(int(nbytes), int(floor), (ws.numel() * 3) // 2 if ws is not None else 0)
        if ws is not None:
            _retired.append(ws)
        ws = _live[key] = (torch.zeros if zero else torch.empty)(size, dtype=torch.uint8, device=device)
    return ws


def retained_bytes():
    return sum(t.numel() for t in _retired)
