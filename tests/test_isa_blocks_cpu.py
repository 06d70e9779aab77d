"""tools/isa_blocks.py: the static instruction mix of one kernel of an assembly listing (no GPU, no compiler: a hand-made listing)."""
import collections

from monodetr_amd.tools import isa_blocks

LISTING = """
	.text
_ZN5mdetr6otherEv:
	s_endpgm
.Lfunc_end0:
_ZN5mdetr6kernelILi4EEEvPf: ; @_ZN5mdetr6kernelILi4EEEvPf
; %bb.0:
	s_load_dwordx2 s[0:1], s[4:5], 0x0
	v_mul_lo_u32 v1, v0, s2
	v_mul_hi_u32 v2, v0, s2
	s_waitcnt lgkmcnt(0)
	s_cbranch_scc1 .LBB1_2
.LBB1_1:
	global_load_dwordx4 v[4:7], v[2:3], off
	v_dot2c_f32_bf16_e32 v8, v4, v5
	ds_add_u64 v9, v[4:5]
	ds_read_b128 v[4:7], v9
	v_pk_fma_f32 v[4:5], v[4:5], v[6:7], v[4:5]
	v_mfma_f32_32x32x16_bf16 a[0:15], v[0:3], v[4:7], a[0:15]
.LBB1_2:
	global_atomic_add_f32 v[2:3], v4, off
	global_store_dword v[2:3], v4, off
	s_endpgm
.Lfunc_end1:
	.size	x, 4
; NumVgprs: 10
; Occupancy: 8
"""


def test_kernel_is_found_and_split_into_blocks():
    name, blocks, meta = isa_blocks.kernel_blocks(LISTING.split("\n"), "kernelILi4E")
    assert name.startswith("_ZN5mdetr6kernel")
    assert [b[0] for b in blocks] == ["entry", ".LBB1_1", ".LBB1_2"]
    total = collections.Counter(isa_blocks.kind(i) for _, ins in blocks for i in ins)
    assert total == {"salu": 2, "imul32": 2, "wait": 1, "branch": 1, "load": 1, "dot2": 1, "lds_atomic": 1, "lds": 1, "valu": 1, "mfma": 1,
                     "atomic": 1, "store": 1}
    assert any("NumVgprs" in m for m in meta) and any("Occupancy" in m for m in meta)


def test_unknown_kernel_is_an_error():
    try:
        isa_blocks.kernel_blocks(LISTING.split("\n"), "nothing_like_it")
    except SystemExit as e:
        assert "no kernel" in str(e)
    else:
        raise AssertionError("expected SystemExit")
