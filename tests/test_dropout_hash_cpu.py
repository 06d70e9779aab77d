"""Statistical quality of the attention kernels' stateless dropout hash (csrc/attn.hip keep_elem), restated
with integer numpy arithmetic (the same restatement the GPU test uses to predict the kernels' mask): drop
rate, independence of neighbouring elements along every coordinate, binomial row / column counts."""
import numpy as np
import pytest

M = 0xFFFFFFFF


def strong32(v):
    """murmur3's 32-bit finaliser (attn.hip strong32)."""
    v = v ^ (v >> np.uint64(16)); v = (v * 0x85EBCA6B) & M; v = v ^ (v >> np.uint64(13)); v = (v * 0xC2B2AE35) & M
    return v ^ (v >> np.uint64(16))


def keep_mask(seed, B, H, Lq, Lk, p):
    b = np.arange(B, dtype=np.uint64).reshape(B, 1, 1, 1)
    h = np.arange(H, dtype=np.uint64).reshape(1, H, 1, 1)
    q = np.arange(Lq, dtype=np.uint64).reshape(1, 1, Lq, 1)
    k = np.arange(Lk, dtype=np.uint64).reshape(1, 1, 1, Lk)
    qconst = (seed & M) ^ ((((seed >> 32) & M) + ((b * 131 + h) * 0xC2B2AE3D & M)) & M)
    qs = strong32(((q * 0x9E3779B1) & M) ^ qconst) | np.uint64(1)
    ks = strong32((((k + 0x7F4A7C15) & M) * 0x85EBCA77) & M)
    x = ((qs & 0xFFFFFF) * (ks & 0xFFFFFF)) & M                     # v_mul_u32_u24: low 32 bits of the 24 x 24 bit product
    return x >= int(p * 4294967296.0)


def corr(a, b):
    a = a - a.mean()
    b = b - b.mean()
    return float((a * b).mean() / np.sqrt((a * a).mean() * (b * b).mean()))


@pytest.mark.parametrize("seed", [0x1234567890ABCDEF, 1])
def test_hash_dropout_is_unbiased_and_uncorrelated(seed):
    p = 0.1
    drop = 1.0 - keep_mask(seed, 2, 8, 550, 1920, p).astype(np.float64)      # 16.9 M elements
    n = drop.size
    assert abs(drop.mean() - p) < 4 * np.sqrt(p * (1 - p) / n)                # 4 sigma
    tol = 5 / np.sqrt(n)                                                       # correlation noise level ~ 1/sqrt(n)
    assert abs(corr(drop[..., :-1], drop[..., 1:])) < tol                      # neighbouring keys
    assert abs(corr(drop[:, :, :-1], drop[:, :, 1:])) < tol                    # neighbouring queries
    assert abs(corr(drop[:, :-1], drop[:, 1:])) < tol                          # neighbouring heads
    assert abs(corr(drop[:-1], drop[1:])) < tol * 1.2                          # neighbouring images
    assert abs(corr(drop[..., :-32], drop[..., 32:])) < tol                    # one MFMA tile apart
    assert abs(corr(drop[:, :, :-1, :-1], drop[:, :, 1:, 1:])) < tol           # diagonal
    rows, cols = drop.sum(-1), drop.sum(-2)
    assert abs(rows.std() / np.sqrt(1920 * p * (1 - p)) - 1) < 0.02            # binomial spread per query row
    assert abs(cols.std() / np.sqrt(550 * p * (1 - p)) - 1) < 0.02             # and per key column


def test_seed_sequence_of_the_module_gives_independent_masks():
    """Successive calls draw their seeds from a Weyl sequence (attn_ext._next_seed: + the 64-bit golden ratio):
    masks of consecutive calls are uncorrelated.  (Rounds 1-2 mixed the seed with one multiply, and seeds that differed only in
    the lowest bit gave correlated masks (-0.11) -- which is why the counter does not step by 1; the query term now goes through
    a full finaliser and adjacent seeds are independent as well.)"""
    from monodetr_amd.attn_ext import _WEYL
    assert _WEYL % (1 << 64) == 0x9E3779B97F4A7C15
    s, m64 = 12345, (1 << 64) - 1
    base = keep_mask(s, 2, 8, 550, 1920, 0.1).astype(np.float64)
    for i in (1, 2, 3):
        other = keep_mask((s + i * 0x9E3779B97F4A7C15) & m64, 2, 8, 550, 1920, 0.1).astype(np.float64)
        assert abs(corr(base, other)) < 5e-3               # measured 1.3e-3, 2e-4, 3e-4: negligible for dropout
    adjacent = keep_mask(s + 1, 2, 8, 550, 1920, 0.1).astype(np.float64)
    assert abs(corr(base, adjacent)) < 5e-3            # (was -0.11 with the one-multiply hash of rounds 1-2)
