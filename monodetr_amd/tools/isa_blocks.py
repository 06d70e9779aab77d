"""Static instruction mix of one kernel of a gfx950 assembly listing (`hipcc -S --cuda-device-only`), whole and per basic block:
how many VALU / SALU / LDS / memory instructions, and how many of the VALU ones are the quarter-rate 32-bit integer multiplies
(`v_mul_lo_u32`, `v_mul_hi_u32`: what a runtime integer division expands into).  No GPU needed.

    python -m monodetr_amd.tools.isa_blocks listing.s 'msda_bwd_fusedI14__hip_bfloat16S2_Li1024ELi2ELi4ELb0' [--blocks 8]
"""
import argparse
import collections
import re

QUARTER = ("v_mul_lo_u32", "v_mul_hi_u32", "v_mul_hi_i32", "v_mul_lo_i32")


def kind(op):
    if op in QUARTER:
        return "imul32"
    if op.startswith("v_dot2"):
        return "dot2"
    if op.startswith(("ds_add", "ds_pk_add", "ds_max", "ds_min")):
        return "lds_atomic"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_load", "buffer_load", "flat_load", "scratch_load")):
        return "load"
    if op.startswith(("global_atomic", "buffer_atomic", "flat_atomic")):
        return "atomic"
    if op.startswith(("global_store", "buffer_store", "flat_store", "scratch_store")):
        return "store"
    if op.startswith("v_mfma") or op.startswith("v_smfma"):
        return "mfma"
    if op.startswith("v_"):
        return "valu"
    if op.startswith(("s_cbranch", "s_branch")):
        return "branch"
    if op.startswith("s_waitcnt"):
        return "wait"
    if op.startswith("s_"):
        return "salu"
    return "other"


def kernel_blocks(lines, pattern):
    start = next((i for i, l in enumerate(lines) if re.match(r"^_Z\S*:", l) and pattern in l), None)
    if start is None:
        raise SystemExit("no kernel matching %r" % pattern)
    end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
    blocks, cur = [], ("entry", [])
    for l in lines[start + 1:end]:
        s = l.strip()
        if not s or s.startswith((";", "/")):
            continue
        if s.endswith(":") or re.match(r"^\.LBB\S*:", s):
            blocks.append(cur)
            cur = (s.split(":")[0], [])
            continue
        if s.startswith("."):
            continue
        cur[1].append(s.split()[0])
    blocks.append(cur)
    meta = [l.strip() for l in lines[end:end + 120] if any(k in l for k in ("; NumVgprs", "; NumSgprs", "; ScratchSize", "; Occupancy", "; LDSByteSize", "; NumAgprs"))]
    return lines[start].split(":")[0], blocks, meta


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("listing")
    ap.add_argument("kernel", help="substring of the mangled kernel name")
    ap.add_argument("--blocks", type=int, default=10, help="basic blocks with the most quarter-rate multiplies to list")
    a = ap.parse_args()
    name, blocks, meta = kernel_blocks(open(a.listing).read().split("\n"), a.kernel)
    total = collections.Counter()
    for _, ins in blocks:
        total.update(kind(i) for i in ins)
    n = sum(total.values())
    print(name)
    print("  %d instructions in %d basic blocks; %s" % (n, len(blocks), "  ".join(meta)))
    for k, v in total.most_common():
        print("  %7d  %s" % (v, k))
    # issue slots if every instruction ran once: 1 per VALU, 4 per quarter-rate multiply
    print("  quarter-rate multiplies: %d of %d VALU instructions = %.0f %% of the VALU issue slots of one straight pass"
          % (total["imul32"], total["imul32"] + total["valu"] + total["dot2"], 100.0 * 4 * total["imul32"] / max(1, 4 * total["imul32"] + total["valu"] + total["dot2"])))
    heavy = sorted(blocks, key=lambda b: -sum(1 for i in b[1] if i in QUARTER))[:a.blocks]
    for label, ins in heavy:
        c = collections.Counter(kind(i) for i in ins)
        if c["imul32"]:
            print("  %-12s %4d instr  %s" % (label, len(ins), dict(c)))


if __name__ == "__main__":
    main()
