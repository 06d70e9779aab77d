"""``fused_ddn_loss``: MonoDETR's depth-map loss in one launch forward and one backward
(csrc/ddn_loss.hip through ``mdetr_ddn_loss_forward / _backward``).  Takes the depth logits in whatever
dense layout they have (the classifier's output is channels_last) and the padded ground truth."""
import torch

from . import _capi

_backend = None               # tests substitute the host build of the same arithmetic (tests/native)


def _lib():
    return _backend if _backend is not None else _capi.lib()


def _workspace(device):
    from . import _workspace as W
    return W.get("ddn_loss", device, 16, zero=True)                 # zero on first use


def _dense(t):
    return t.is_contiguous() or t.is_contiguous(memory_format=torch.channels_last)


class _FusedDdnLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, boxes, depth, valid, alpha, fg_weight, bg_weight, depth_min, depth_max):
        z = logits.float()
        if not _dense(z):
            z = z.contiguous()
        B, C, H, W = z.shape
        K = boxes.shape[1]
        dev = z.device
        bx, dp, va = boxes.float().contiguous(), depth.float().contiguous(), valid.to(torch.uint8).contiguous()
        out = torch.empty(1, dtype=torch.float32, device=dev)
        args = (z.data_ptr(), bx.data_ptr(), dp.data_ptr(), va.data_ptr(), B, C, H, W, K, *z.stride(),
                float(alpha), float(fg_weight), float(bg_weight), float(depth_min), float(depth_max))
        rc = _lib().mdetr_ddn_loss_forward(*args, out.data_ptr(), _workspace(dev).data_ptr(),
                                           dev.index if dev.type == "cuda" else -1,
                                           torch.cuda.current_stream(dev).cuda_stream if dev.type == "cuda" else None)
        if rc != 0:
            _capi.check(rc, "mdetr_ddn_loss_forward")
        ctx.keep, ctx.args, ctx.in_dtype = (z, bx, dp, va), args, logits.dtype
        return out[0]

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, grad_out):
        z = ctx.keep[0]
        dev = z.device
        grad = torch.empty_like(z)                       # same strides as z (dense)
        g = grad_out.reshape(1).float().contiguous()                 # a named tensor: it must outlive the call that reads its pointer
        rc = _lib().mdetr_ddn_loss_backward(*ctx.args, g.data_ptr(), grad.data_ptr(),
                                            dev.index if dev.type == "cuda" else -1,
                                            torch.cuda.current_stream(dev).cuda_stream if dev.type == "cuda" else None)
        if rc != 0:
            _capi.check(rc, "mdetr_ddn_loss_backward")
        return (grad.to(ctx.in_dtype),) + (None,) * 8


def fused_ddn_loss(logits, boxes_cxcywh, depth, valid, alpha=0.25, fg_weight=13.0, bg_weight=1.0, depth_min=1e-3,
                   depth_max=60.0):
    """logits [B, D+1, H, W]; boxes_cxcywh [B, K, 4] normalised to the image; depth, valid [B, K] -> 0-d loss."""
    return _FusedDdnLoss.apply(logits, boxes_cxcywh, depth, valid, alpha, fg_weight, bg_weight, depth_min, depth_max)
