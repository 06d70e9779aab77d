"""Device Hungarian matcher (csrc/lsa.hip) vs scipy.optimize.linear_sum_assignment, the solver the
reference calls on the host (matcher.py:96).  Random fp32 costs have unique optima, so the indices
must be identical; a tie case checks equal total cost instead."""
import numpy as np
import pytest
import torch
from scipy.optimize import linear_sum_assignment

pytestmark = pytest.mark.gpu


def scipy_assign(cost, num_targets, groups):
    L, B, Q, K = cost.shape
    n = Q // groups
    out = -np.ones((L, B, groups, K), dtype=np.int32)
    C = cost.double().cpu().numpy()
    for l in range(L):
        for b in range(B):
            k = int(num_targets[b])
            for g in range(groups):
                if k:
                    r, c = linear_sum_assignment(C[l, b, g * n:(g + 1) * n, :k])
                    out[l, b, g, c] = r + g * n
    return out


@pytest.mark.parametrize("L,B,G,n,K", [(3, 8, 11, 50, 50), (1, 4, 1, 50, 50), (2, 3, 5, 7, 7), (1, 2, 2, 64, 64), (1, 1, 1, 1, 1),
                                       (3, 8, 11, 100, 50), (1, 2, 2, 128, 64)])       # > 64 queries per group: two columns per lane
def test_matches_scipy(L, B, G, n, K):
    from monodetr_amd.lsa_ext import batched_assignment
    g = torch.Generator().manual_seed(L * 100 + n)
    cost = (torch.randn(L, B, G * n, K, generator=g) * 3).cuda()
    num = torch.randint(0, K + 1, (B,), generator=g)
    num[0] = K                                           # a full problem
    if B > 1:
        num[1] = 0                                       # an image without objects
    got = batched_assignment(cost, num.int().cuda(), G).cpu().numpy()
    want = scipy_assign(cost.cpu(), num, G)
    assert np.array_equal(got, want)


def test_strided_cost_and_ties():
    from monodetr_amd.lsa_ext import batched_assignment
    # non-contiguous view (target axis sliced out of a wider tensor)
    g = torch.Generator().manual_seed(5)
    wide = torch.randn(2, 3, 100, 80, generator=g).cuda()
    cost = wide[..., 10:60]
    num = torch.tensor([50, 13, 1], dtype=torch.int32)
    got = batched_assignment(cost, num.cuda(), 2).cpu().numpy()
    assert np.array_equal(got, scipy_assign(cost.cpu(), num, 2))
    # ties: integer costs -> many optimal assignments; the total cost must be optimal and rows distinct
    cost = torch.randint(0, 3, (1, 2, 20, 20), generator=g).float().cuda()
    num = torch.tensor([20, 9], dtype=torch.int32)
    got = batched_assignment(cost, num.cuda(), 1).cpu().numpy()
    want = scipy_assign(cost.cpu(), num, 1)
    C = cost.cpu().numpy()
    for b in range(2):
        k = int(num[b])
        rows_g, rows_w = got[0, b, 0, :k], want[0, b, 0, :k]
        assert len(set(rows_g.tolist())) == k and (rows_g >= 0).all()
        assert abs(C[0, b, rows_g, np.arange(k)].sum() - C[0, b, rows_w, np.arange(k)].sum()) < 1e-9
        assert (got[0, b, 0, k:] == -1).all()
