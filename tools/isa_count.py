#!/usr/bin/env python3
"""Instruction-class census of one kernel in a hipcc -save-temps .s file (static counts per basic block).

usage: isa_count.py file.s <substring of the mangled kernel name> [--blocks]
"""
import re
import sys
from collections import Counter


def classify(op):
    if op.startswith("v_mfma"):
        return "mfma"
    if op.startswith(("v_exp", "v_rcp", "v_log", "v_rsq", "v_sqrt", "v_sin", "v_cos")):
        return "valu_trans"
    if op.startswith(("v_mad_u64", "v_mul_lo", "v_mul_hi", "v_mad_i64")):
        return "valu_quarter"
    if op.startswith(("v_readlane", "v_writelane", "v_readfirstlane")):
        return "valu_lane"
    if op.startswith(("v_accvgpr",)):
        return "valu_acc"
    if op.startswith("v_"):
        return "valu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("buffer_", "global_", "flat_", "scratch_")):
        return "vmem_scratch" if op.startswith("scratch_") else "vmem"
    if op.startswith("s_waitcnt"):
        return "waitcnt"
    if op.startswith("s_barrier"):
        return "barrier"
    if op.startswith(("s_load", "s_buffer")):
        return "smem"
    if op.startswith("s_"):
        return "salu"
    return "other"


def main():
    path, key = sys.argv[1], sys.argv[2]
    blocks = "--blocks" in sys.argv
    by_barrier = "--barriers" in sys.argv
    lines = open(path).read().split("\n")
    start = None
    for i, l in enumerate(lines):
        if re.match(r"^[_A-Za-z0-9]+:", l) and key in l.split(":")[0]:
            start = i
            break
    if start is None:
        sys.exit("kernel not found")
    total = Counter()
    ops = Counter()
    cur, cur_name, per_block = Counter(), "entry", []
    for l in lines[start + 1:]:
        s = l.strip()
        if s.startswith(".Lfunc_end") or s.startswith(".end_amdhsa_kernel"):
            break
        if s.endswith(":") and s.startswith(".LBB"):
            per_block.append((cur_name, cur))
            cur, cur_name = Counter(), s[:-1]
            continue
        if not s or s.startswith((";", ".")):
            continue
        op = s.split()[0]
        if not re.match(r"^[a-z]", op):
            continue
        c = classify(op)
        if by_barrier and c == "barrier":
            per_block.append((cur_name, cur))
            cur, cur_name = Counter(), "after barrier %d" % len(per_block)
        total[c] += 1
        cur[c] += 1
        ops[op] += 1
    per_block.append((cur_name, cur))
    print("total:", dict(total))
    if blocks or by_barrier:
        for name, c in per_block:
            if c.get("mfma", 0) >= 24 or by_barrier:
                print(name, dict(c))
    else:
        for op, n in ops.most_common(60):
            print("%6d  %s" % (n, op))


main()
