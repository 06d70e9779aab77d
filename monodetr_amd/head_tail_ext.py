"""``head_tail(...)`` / ``box_refine(...)``: the per-query arithmetic behind the prediction heads in one launch each way
(csrc/head_tail.hip through ``mdetr_head_tail_forward / _backward`` and ``mdetr_box_refine``): boxes from deltas and references, the
geometric depth, the depth-map lookup and the three-way depth average of lib/models/monodetr/monodetr.py:226-253, and the decoder's
reference update between layers (depthaware_transformer.py:602-613)."""
import os

import torch

from . import _capi

_backend = None               # tests substitute the CPU emulation of the same kernel source (tests/native_emul.py)
# MDETR_HEAD_TAIL=1 (kernel_families decides): the fused kernels; off = the framework's elementwise operators
ENABLED = os.environ.get("MDETR_HEAD_TAIL") == "1"


def _lib():
    return _backend if _backend is not None else _capi.lib()


def _dev(t):
    return (t.device.index, torch.cuda.current_stream(t.device).cuda_stream) if t.is_cuda else (-1, None)


def _f32(t):
    return t.detach().to(torch.float32).contiguous()


def usable(*tensors):
    return ENABLED and all((t.is_cuda or _backend is not None) and t.dtype == torch.float32 for t in tensors)


def _check(rc, what):
    if rc != 0:
        msg = _lib().mdetr_last_error()
        raise RuntimeError("%s failed (code %d): %s" % (what, rc, msg.decode() if msg else "?"))


@torch.no_grad()
def box_refine(delta, ref):
    """sigmoid(delta + inverse_sigmoid(ref)) on ref's components (2 or 6), sigmoid(delta) on the rest; delta [..., 6] -> [..., 6]."""
    d, r = _f32(delta), _f32(ref)
    out = torch.empty_like(d)
    dev, st = _dev(d)
    _check(_lib().mdetr_box_refine(d.data_ptr(), r.data_ptr(), out.data_ptr(), d.numel() // 6, r.shape[-1], dev, st), "mdetr_box_refine")
    return out


class _HeadTail(torch.autograd.Function):
    @staticmethod
    def forward(ctx, delta, init_ref, inter_refs, size3d, depth_reg, depth_map, img_h, focal):
        L, B, Q, _ = delta.shape
        nd0 = init_ref.shape[-1]
        H, W = depth_map.shape[-2:]
        t = [_f32(x) for x in (delta, init_ref, inter_refs, size3d, depth_reg, depth_map, img_h, focal)]
        coord = torch.empty((L, B, Q, 6), dtype=torch.float32, device=delta.device)
        depth = torch.empty((L, B, Q, 2), dtype=torch.float32, device=delta.device)
        dev, st = _dev(delta)
        _check(_lib().mdetr_head_tail_forward(*[x.data_ptr() for x in t], coord.data_ptr(), depth.data_ptr(), L, B, Q, nd0, H, W, dev, st),
               "mdetr_head_tail_forward")
        ctx.dims = (L, B, Q, nd0, H, W)
        ctx.save_for_backward(t[1], t[3], t[4], t[6], t[7], coord)
        ctx.set_materialize_grads(False)
        return coord, depth

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g_coord, g_depth):
        init_ref, size3d, depth_reg, img_h, focal, coord = ctx.saved_tensors
        L, B, Q, nd0, H, W = ctx.dims
        gc = _f32(g_coord) if g_coord is not None else None
        gd = _f32(g_depth) if g_depth is not None else None
        new = lambda *s: torch.empty(s, dtype=torch.float32, device=coord.device)           # noqa: E731
        g_delta, g_init, g_size, g_reg = new(L, B, Q, 6), new(B, Q, nd0), new(L, B, Q, 3), new(L, B, Q, 2)
        g_map = new(B, H, W) if ctx.needs_input_grad[5] else None
        dev, st = _dev(coord)
        ptr = lambda x: x.data_ptr() if x is not None else None                              # noqa: E731
        _check(_lib().mdetr_head_tail_backward(init_ref.data_ptr(), size3d.data_ptr(), depth_reg.data_ptr(), img_h.data_ptr(), focal.data_ptr(),
                                               coord.data_ptr(), ptr(gc), ptr(gd), g_delta.data_ptr(), g_init.data_ptr(), g_size.data_ptr(),
                                               g_reg.data_ptr(), ptr(g_map), L, B, Q, nd0, H, W, dev, st), "mdetr_head_tail_backward")
        return g_delta, g_init, None, g_size, g_reg, g_map, None, None


def head_tail(delta, init_ref, inter_refs, size3d, depth_reg, depth_map, img_h, focal):
    """delta [L,B,Q,6], init_ref [B,Q,2|6], inter_refs [L-1,B,Q,6] (no gradient), size3d [L,B,Q,3], depth_reg [L,B,Q,2],
    depth_map [B,H,W], img_h / focal [B] -> (coord [L,B,Q,6], depth_ave [L,B,Q,2]); gradients to delta, init_ref, size3d, depth_reg
    and depth_map."""
    return _HeadTail.apply(delta, init_ref, inter_refs, size3d, depth_reg, depth_map, img_h, focal)
