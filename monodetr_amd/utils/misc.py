"""The few pieces of /root/reference/utils/misc.py the model path needs: ``NestedTensor``
(:287-307), ``inverse_sigmoid`` (:473-476), ``accuracy`` (:436-451) and the process-group
helpers (:381-407, :135-159).  Same names and semantics."""
import torch
import torch.distributed as dist


class NestedTensor(object):
    """A feature map and its padding mask (True = padding)."""

    def __init__(self, tensors, mask):
        self.tensors = tensors
        self.mask = mask

    def to(self, device):
        return NestedTensor(self.tensors.to(device), None if self.mask is None else self.mask.to(device))

    def decompose(self):
        return self.tensors, self.mask

    def __repr__(self):
        return str(self.tensors)


def inverse_sigmoid(x, eps=1e-5):
    x = x.clamp(min=0, max=1)
    return torch.log(x.clamp(min=eps) / (1 - x).clamp(min=eps))


@torch.no_grad()
def accuracy(output, target, topk=(1,)):
    """precision@k in percent, list of 0-d tensors."""
    if target.numel() == 0:
        return [torch.zeros([], device=output.device)]
    pred = output.topk(max(topk), 1, True, True)[1].t()
    hit = pred.eq(target.view(1, -1).expand_as(pred))
    return [hit[:k].reshape(-1).float().sum(0) * (100.0 / target.size(0)) for k in topk]


def is_dist_avail_and_initialized():
    return dist.is_available() and dist.is_initialized()


def get_world_size():
    return dist.get_world_size() if is_dist_avail_and_initialized() else 1


def get_rank():
    return dist.get_rank() if is_dist_avail_and_initialized() else 0


def is_main_process():
    return get_rank() == 0


def reduce_dict(input_dict, average=True):
    """All-reduce the values of a dict of 0-d tensors (one packed message), as :135-159."""
    world = get_world_size()
    if world < 2:
        return input_dict
    with torch.no_grad():
        names = sorted(input_dict.keys())
        packed = torch.stack([input_dict[k] for k in names], 0)
        dist.all_reduce(packed)
        if average:
            packed /= world
        return dict(zip(names, packed))


def mark_no_padding(mask):
    """Tag a padding mask that is all-False BY CONSTRUCTION (every mask this model builds is:
    reference backbone.py:88, monodetr.py:173-174).  Consumers can then skip masked_fill /
    valid-ratio / key_padding work without a device->host sync; untagged masks take the general
    path.  The tag lives on the tensor object and does not survive views, so test it before
    reshaping."""
    mask._mdetr_no_padding = True
    return mask


def no_padding(mask):
    return mask is None or getattr(mask, "_mdetr_no_padding", False)


def at_least_fp32(x):
    """bf16 / fp16 -> fp32; fp32 and fp64 untouched."""
    return x.float() if x.dtype in (torch.bfloat16, torch.float16) else x
