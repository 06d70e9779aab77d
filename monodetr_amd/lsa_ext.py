"""Host binding of the batched assignment solver (C ABI: mdetr_lsa_forward, csrc/lsa.hip)."""
import torch

from . import _capi


def batched_assignment(cost, num_targets, groups):
    """cost [L, B, Q, Kmax] fp32 (CUDA; any strides), num_targets [B] int32 -> assign [L, B, G, Kmax] int32:
    for layer l, image b, query group g, the query index in [g*n, (g+1)*n), n = Q // groups, matched to
    target t (t < num_targets[b]); -1 for padded targets.  No host synchronisation."""
    assert cost.is_cuda and cost.dtype == torch.float32 and cost.dim() == 4
    L, B, Q, K = cost.shape
    n = Q // groups
    assert n * groups == Q and n <= 128 and K <= min(n, 64), "need Q = groups * n with n <= 128 and Kmax <= min(n, 64)"
    if cost.stride(0) != B * cost.stride(1):
        cost = cost.contiguous()
    num_targets = num_targets.to(device=cost.device, dtype=torch.int32).contiguous()
    assign = torch.empty((L, B, groups, K), dtype=torch.int32, device=cost.device)
    rc = _capi.lib().mdetr_lsa_forward(
        cost.data_ptr(), num_targets.data_ptr(), assign.data_ptr(), L, B, groups, n, K,
        cost.stride(1), cost.stride(2), cost.stride(3), cost.device.index,
        torch.cuda.current_stream(cost.device).cuda_stream)
    _capi.check(rc, "mdetr_lsa_forward")
    return assign


_backend = None               # tests substitute the host build (tests/native): same cost arithmetic, serial solver


def batched_assignment_fused(logits, boxes, gt, groups, weights, focal_alpha=0.25):
    """Assignment straight from level-stacked predictions (logits [L,B,Q,C], boxes [L,B,Q,6]) and the padded
    ground truth: the cost of matcher.py:55-84 is evaluated inside the solver kernel.  weights = (class, bbox,
    3dcenter, giou).  Returns [L, B, G, K] int32 like `batched_assignment`."""
    L, B, Q, C = logits.shape
    K = gt["valid"].shape[1]
    n = Q // groups
    assert n * groups == Q and n <= 128 and K <= min(n, 64), "need Q = groups * n with n <= 128 and Kmax <= min(n, 64)"
    dev = logits.device
    lg, bx = logits.float().contiguous(), boxes.float().contiguous()
    labels, b3d = gt["labels"].to(torch.int64).contiguous(), gt["boxes_3d"].float().contiguous()
    num = gt["num"].to(device=dev, dtype=torch.int32).contiguous()
    assign = torch.empty((L, B, groups, K), dtype=torch.int32, device=dev)
    lib = _backend if _backend is not None else _capi.lib()
    rc = lib.mdetr_lsa_forward_fused(
        lg.data_ptr(), bx.data_ptr(), labels.data_ptr(), b3d.data_ptr(), num.data_ptr(), assign.data_ptr(),
        L, B, groups, n, K, C, float(weights[0]), float(weights[1]), float(weights[2]), float(weights[3]),
        float(focal_alpha), dev.index if dev.type == "cuda" else -1,
        torch.cuda.current_stream(dev).cuda_stream if dev.type == "cuda" else None)
    if rc != 0:
        _capi.check(rc, "mdetr_lsa_forward_fused")
    return assign
