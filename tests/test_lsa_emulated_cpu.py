"""csrc/lsa.hip (cost-matrix form, the default path's matcher) on the HIP-on-CPU shim: optimal assignments against scipy,
and termination on non-finite costs: the path search and the augmentation are bounded by n + 1 trips per row, so that
diverged predictions (NaN / inf costs) can produce a meaningless matching but never a hung GPU queue."""
import numpy as np
import pytest
import torch
from scipy.optimize import linear_sum_assignment

import native_emul


def solve(cost, num, groups, n):
    """cost [LI, Q, K] fp32, num [images] int32 -> assign [LI, groups, K] int32 through mdetr_lsa_forward."""
    LI, Q, K = cost.shape
    images = num.numel()
    assign = torch.full((LI, groups, K), -7, dtype=torch.int32)
    rc = native_emul.lib().mdetr_lsa_forward(cost.data_ptr(), num.data_ptr(), assign.data_ptr(), LI // images, images, groups, n, K,
                                             cost.stride(0), cost.stride(1), cost.stride(2), -1, None)
    assert rc == 0
    return assign


@pytest.mark.parametrize("layers,images,groups,n,K,seed", [(3, 4, 11, 50, 50, 0), (1, 2, 1, 64, 64, 1), (2, 3, 5, 17, 9, 2),
                                                          (2, 2, 11, 100, 50, 3), (1, 2, 2, 128, 64, 4), (1, 3, 3, 65, 7, 5)])    # two columns per lane
def test_assignments_are_optimal(layers, images, groups, n, K, seed):
    g = torch.Generator().manual_seed(seed)
    cost = torch.rand(layers * images, groups * n, K, generator=g) * 4 - 1
    cost[0, :, 0] = cost[0, :1, 0]                                   # a constant column: ties
    num = torch.randint(0, K + 1, (images,), generator=g, dtype=torch.int32)
    num[0] = K
    got = solve(cost, num, groups, n).numpy()
    c64 = cost.double().numpy()
    for li in range(layers * images):
        k = int(num[li % images])
        for gi in range(groups):
            a = got[li, gi]
            assert (a[k:] == -1).all()
            if k == 0:
                continue
            sub = c64[li, gi * n:(gi + 1) * n, :k]
            r, c = linear_sum_assignment(sub)
            mine = a[:k] - gi * n
            assert len(set(mine.tolist())) == k and mine.min() >= 0 and mine.max() < n
            assert abs(sub[r, c].sum() - sub[mine, np.arange(k)].sum()) <= 1e-9 * max(1.0, abs(sub[r, c].sum()))


@pytest.mark.timeout(120)
@pytest.mark.parametrize("poison", ["nan_row", "all_nan", "inf", "nan_late"])
def test_non_finite_costs_terminate(poison):
    g = torch.Generator().manual_seed(5)
    n, K, groups = 50, 50, 2
    cost = torch.rand(2, groups * n, K, generator=g)
    if poison == "nan_row":
        cost[:, :, 3] = float("nan")                                  # target 3 costs NaN against every query
    elif poison == "all_nan":
        cost[:] = float("nan")
    elif poison == "inf":
        cost[:, :, 10:] = float("inf")
    else:
        cost[:, :, 40] = float("nan")
    num = torch.tensor([K, 30], dtype=torch.int32)
    got = solve(cost, num, groups, n)                                 # must return; the values only have to be in range
    assert ((got >= -1) & (got < groups * n)).all()
    for li in range(2):
        for gi in range(groups):
            a = got[li, gi][got[li, gi] >= 0]
            assert len(set(a.tolist())) == a.numel()                  # still a matching: no query used twice
