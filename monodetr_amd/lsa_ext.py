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
    assert n * groups == Q and n <= 64 and K <= n, "need Q = groups * n with n <= 64 and Kmax <= n"
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
