"""LDS bank-conflict model (MI355X_MICROARCH.md, "LDS": lane groups and bank functions per instruction) applied
to the access patterns of csrc/attn.hip and csrc/tgemm.hip.  Documents, and guards against regressions of,
the padding choices: transposed tile 68 bf16 per row (72 was 2-way on reads: measured 52 % conflict cycles),
row-major tile 40 per row, token-GEMM weight rows K + 8 and slab rows 72; and evaluates the staging-thread
remap (compile-time MDETR_ATTN_STAGE_REMAP in attn.hip, off until validated)."""
import pytest

B128_READ_GROUPS = [[0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27],
                    [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31]]
B128_READ_GROUPS += [[l + 32 for l in g] for g in B128_READ_GROUPS]


def worst_conflict(byte_addr, width, groups, nbanks):
    """Max number of DISTINCT dwords mapped to one bank within a lane group (1 = conflict-free)."""
    worst = 1
    for g in groups:
        per_bank = {}
        for lane in g:
            a = byte_addr(lane)
            if a is None:
                continue
            for d in range(a // 4, (a + width + 3) // 4):
                per_bank.setdefault(d % nbanks, set()).add(d)
        worst = max(worst, max(len(v) for v in per_bank.values()))
    return worst


HALVES = [list(range(32)), list(range(32, 64))]
OCTETS = [list(range(8 * i, 8 * i + 8)) for i in range(8)]


def staging(t, remap):
    """(row, chunk) of staging thread t of a 256-thread workgroup (attn.hip tile_load / tile_store)."""
    if remap:
        return (t & 15) + 16 * (t >> 6), (t >> 4) & 3
    return t >> 2, t & 3


@pytest.mark.parametrize("tpad,expect", [(72, 2), (68, 1)])
def test_transposed_tile_reads(tpad, expect):
    # frag_cols: lane reads 8 bytes at ((lane & 31) * tpad + 4 * (lane >> 5)) elements (+ 8 elements for the second read)
    for extra in (0, 8):
        w = worst_conflict(lambda l: 2 * ((l & 31) * tpad + 4 * (l >> 5) + extra), 8, HALVES, 64)
        assert w == expect


def test_row_major_tile_reads_are_conflict_free():
    # frag_rows: lane reads 16 bytes at ((lane & 31) * 40 + 8 * (lane >> 5)) elements
    assert worst_conflict(lambda l: 2 * ((l & 31) * 40 + 8 * (l >> 5)), 16, B128_READ_GROUPS, 64) == 1


def test_tgemm_slab_fragment_reads_and_tile_parking_are_conflict_free():
    # csrc/tgemm.hip: slab rows of 72 bf16; a lane reads 16 bytes at ((lane & 31) * 72 + 8 * (lane >> 5)) elements
    assert worst_conflict(lambda l: 2 * ((l & 31) * 72 + 8 * (l >> 5)), 16, B128_READ_GROUPS, 64) == 1


@pytest.mark.parametrize("remap,expect_tr,expect_rm", [(False, 2, 2), (True, 1, 1)])
def test_staging_stores_with_and_without_the_remap(remap, expect_tr, expect_rm):
    """Wave w of the workgroup stages threads 64 w .. 64 w + 63.  Transposed copy: 2-byte stores at
    ((8 chunk + i) * 68 + row) elements; row-major copy: 16-byte stores at (row * 40 + 8 chunk) elements."""
    worst_tr = worst_rm = 1
    for wave in range(4):
        for i in range(8):
            def addr(l, i=i, wave=wave):
                row, chunk = staging(64 * wave + l, remap)
                return 2 * ((8 * chunk + i) * 68 + row)
            worst_tr = max(worst_tr, worst_conflict(addr, 2, HALVES, 32))

        def addr_rm(l, wave=wave):
            row, chunk = staging(64 * wave + l, remap)
            return 2 * (row * 40 + 8 * chunk)
        worst_rm = max(worst_rm, worst_conflict(addr_rm, 16, OCTETS, 32))
    assert worst_tr == expect_tr and worst_rm == expect_rm


def test_remap_is_a_permutation_of_the_tile():
    for remap in (False, True):
        assert sorted(staging(t, remap) for t in range(256)) == sorted((r, c) for r in range(64) for c in range(4))


@pytest.mark.parametrize("remap,expect", [(False, 2), (True, 1)])
def test_conv_taps_staging_stores(remap, expect):
    """csrc/conv_taps.hip: 16-byte staging stores into LDS rows of 40 bf16 (32-channel slab + 8 pad = 80 bytes).  Round 3's first
    mapping gave consecutive lanes the four pieces of one row (two rows per 8-lane store group: row 1's last piece wraps onto
    row 0's first banks) -- rocprofv3 counted 27 - 38 % of the kernels' LDS cycles as bank conflicts
    (profiles/r03_pmc_conv_strided_after.json); the remap gives a 16-lane group 16 consecutive rows at one piece: none
    (profiles/r03_pmc_conv_strided_remap.json)."""
    def task(p):
        if remap:
            return (p // 16) % 4, p // 64 * 16 + p % 16            # (piece, row)
        return p % 4, p // 4
    worst = 1
    for wave in range(4):
        def addr(l, wave=wave):
            piece, row = task(64 * wave + l)
            return 2 * (row * 40 + piece * 8)
        worst = max(worst, worst_conflict(addr, 16, OCTETS, 32))
    assert worst == expect
    assert sorted(task(p) for p in range(256)) == sorted((c, r) for r in range(64) for c in range(4))    # both cover 64 rows x 4 pieces
