#!/usr/bin/env python3
"""Instruction mix of one kernel in a gfx950 assembly listing.

  hipcc --offload-arch=gfx950 -O3 -std=c++17 -S --cuda-device-only quinoa_amd/csrc/qdg_kernels.hip -o /tmp/k.s
  python tools/isa_stats.py /tmp/k.s k_rhs_p1tILb0ELb1ELi1ELb0E

Prints per basic block (label) the number of instructions by class, and the kernel's
register / LDS / scratch usage from its .amdhsa metadata.  Static counts: weigh the
blocks by their trip counts yourself."""
import collections
import re
import sys


def classify(op):
    if op.startswith("v_") and ("f64" in op):
        if op.startswith(("v_rcp", "v_rsq", "v_sqrt", "v_div")):
            return "valu_f64_trans"
        return "valu_f64"
    if op.startswith("v_cndmask") or op.startswith("v_mov") or op.startswith("v_accvgpr"):
        return "valu_mov/sel"
    if op.startswith("v_cmp") or op.startswith("v_cmpx"):
        return "valu_cmp"
    if op.startswith("v_"):
        return "valu_other"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    if op.startswith("s_waitcnt"):
        return "waitcnt"
    if op.startswith("s_barrier"):
        return "barrier"
    if op.startswith(("s_load", "s_buffer_load")):
        return "smem"
    if op.startswith(("s_cbranch", "s_branch")):
        return "branch"
    if op.startswith("s_"):
        return "salu"
    return "other"


def main():
    path, pat = sys.argv[1], sys.argv[2]
    lines = open(path).read().split("\n")
    start = None
    for i, l in enumerate(lines):
        if re.match(r"^_ZN.*%s.*:\s" % re.escape(pat), l) or re.match(r"^_ZN.*%s.*:$" % re.escape(pat), l):
            start = i
            break
    if start is None:
        raise SystemExit("kernel not found")
    name = lines[start].split(":")[0]
    blocks = collections.OrderedDict()
    cur = "entry"
    blocks[cur] = collections.Counter()
    total = collections.Counter()
    i = start + 1
    while i < len(lines) and not lines[i].startswith(".Lfunc_end"):
        l = lines[i].strip()
        i += 1
        if not l or l.startswith(";") or l.startswith("."):
            m = re.match(r"^(\.LBB\d+_\d+):", l)
            if m:
                cur = m.group(1)
                blocks[cur] = collections.Counter()
            continue
        op = l.split()[0]
        c = classify(op)
        blocks[cur][c] += 1
        total[c] += 1
    print(name)
    keys = ["valu_f64", "valu_f64_trans", "valu_mov/sel", "valu_cmp", "valu_other", "lds", "vmem", "smem", "salu",
            "waitcnt", "barrier", "branch", "other"]
    print("%-12s" % "block" + "".join("%9s" % k[:9] for k in keys) + "%9s" % "sum")
    for b, c in blocks.items():
        n = sum(c.values())
        if n >= 12:
            print("%-12s" % b + "".join("%9d" % c[k] for k in keys) + "%9d" % n)
    print("%-12s" % "TOTAL" + "".join("%9d" % total[k] for k in keys) + "%9d" % sum(total.values()))
    # metadata
    for j in range(i, min(i + 400, len(lines))):
        l = lines[j].strip()
        if any(k in l for k in (".amdhsa_next_free_vgpr", ".amdhsa_next_free_sgpr", ".amdhsa_accum_offset",
                                ".amdhsa_group_segment_fixed_size", ".amdhsa_private_segment_fixed_size",
                                "; Occupancy", "; NumVgprs", "; NumAgprs", "; ScratchSize", "; LDSByteSize")):
            print("   ", l)
        if l.startswith(".end_amdhsa_kernel"):
            break


if __name__ == "__main__":
    main()
