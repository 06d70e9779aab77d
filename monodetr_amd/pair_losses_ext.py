"""``fused_pair_losses``: MonoDETR's matched-pair losses for every decoder level in one launch forward,
one backward (csrc/pair_losses.hip through ``mdetr_pair_losses_forward / _backward``).

Input: the level-stacked fp32 predictions, the device assignment ``[L, B, G, K]`` and the padded ground
truth of ``monodetr.pad_targets``.  Output: a ``[9, L]`` tensor whose rows are ``ROWS``; rows 0-6 carry
gradients to the five prediction tensors, rows 7-8 (class error, cardinality error) are metrics.
"""
import torch

from . import _capi

ROWS = ("loss_ce", "loss_center", "loss_bbox", "loss_giou", "loss_depth", "loss_dim", "loss_angle",
        "class_error", "cardinality_error")

_backend = None               # tests substitute the host build of the same arithmetic (tests/native)


def _lib():
    return _backend if _backend is not None else _capi.lib()


def _workspace(device, nbytes):
    from . import _workspace as W
    return W.get("pair_losses", device, nbytes, zero=True, floor=4096)   # zero on first use


def _stream(device):
    return torch.cuda.current_stream(device).cuda_stream if device.type == "cuda" else None


def _pack(preds, assign, gt):
    logits, boxes, dims, depths, angles = (t.contiguous().float() for t in preds)
    return (logits, boxes, dims, depths, angles, assign.to(torch.int32).contiguous(),
            gt["labels"].to(torch.int64).contiguous(), gt["boxes_3d"].float().contiguous(), gt["depth"].float().contiguous(),
            gt["size_3d"].float().contiguous(), gt["heading_bin"].to(torch.int64).contiguous(),
            gt["heading_res"].float().contiguous(), gt["valid"].to(torch.uint8).contiguous(),
            gt["num"].to(torch.int32).contiguous())


class _FusedPairLosses(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, boxes, dims, depths, angles, assign, gt, num_boxes, focal_alpha):
        t = _pack((logits, boxes, dims, depths, angles), assign, gt)
        L, B, Q, C = t[0].shape
        G, K = assign.shape[2], assign.shape[3]
        dev = t[0].device
        nb_dev = num_boxes.to(device=dev, dtype=torch.float32).reshape(1) if torch.is_tensor(num_boxes) else None
        nb_host = 0.0 if nb_dev is not None else float(num_boxes)
        out = torch.empty((len(ROWS), L), dtype=torch.float32, device=dev)
        comp = torch.empty(L, dtype=torch.float32, device=dev)
        lib = _lib()
        ws = _workspace(dev, lib.mdetr_pair_losses_workspace_bytes(L, B))
        rc = lib.mdetr_pair_losses_forward(
            *[x.data_ptr() for x in t], L, B, Q, C, G, K, float(focal_alpha), nb_host,
            nb_dev.data_ptr() if nb_dev is not None else None, out.data_ptr(), comp.data_ptr(), ws.data_ptr(),
            dev.index if dev.type == "cuda" else -1, _stream(dev))
        if rc != 0:
            _capi.check(rc, "mdetr_pair_losses_forward")
        ctx.packed, ctx.shape, ctx.nb = t[:13], (L, B, Q, C, G, K), (nb_host, nb_dev)
        ctx.alpha, ctx.comp = float(focal_alpha), comp
        ctx.in_dtypes = tuple(x.dtype for x in (logits, boxes, dims, depths, angles))
        ctx.mark_non_differentiable(comp)
        return out, comp

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, grad_out, _grad_comp):
        t = ctx.packed
        L, B, Q, C, G, K = ctx.shape
        dev = t[0].device
        grads = [torch.empty_like(x) for x in t[:5]]
        nb_host, nb_dev = ctx.nb
        go = grad_out.contiguous().float()                           # named: it must outlive the call that reads its pointer
        rc = _lib().mdetr_pair_losses_backward(
            *[x.data_ptr() for x in t], L, B, Q, C, G, K, ctx.alpha, nb_host,
            nb_dev.data_ptr() if nb_dev is not None else None, go.data_ptr(),
            ctx.comp.data_ptr(), *[g.data_ptr() for g in grads], dev.index if dev.type == "cuda" else -1, _stream(dev))
        if rc != 0:
            _capi.check(rc, "mdetr_pair_losses_backward")
        return tuple(g.to(dt) for g, dt in zip(grads, ctx.in_dtypes)) + (None, None, None, None)


def fused_pair_losses(stacked, assign, gt, num_boxes, focal_alpha):
    """stacked: dict of level-stacked predictions ('pred_logits', 'pred_boxes', 'pred_3d_dim', 'pred_depth',
    'pred_angle'); returns {row name: [L] tensor}."""
    out, _ = _FusedPairLosses.apply(stacked['pred_logits'], stacked['pred_boxes'], stacked['pred_3d_dim'],
                                    stacked['pred_depth'], stacked['pred_angle'], assign, gt, num_boxes, focal_alpha)
    return dict(zip(ROWS, out.unbind(0)))
