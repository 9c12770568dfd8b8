#!/usr/bin/env python3
"""Safety check of the counted waits in front of the slab barriers (Pipe6::advance<N>): in the -save-temps .s of a kernel,
every `s_waitcnt vmcnt(N)` that directly precedes an `s_barrier` must have AT LEAST N vector-memory instructions between
the last LDS-DMA (`... lds`) before it and itself -- otherwise the wait could leave a DMA in flight across the barrier.

usage: check_vmcnt.py file.s <substring of the mangled kernel name>
"""
import re
import sys

path, key = sys.argv[1], sys.argv[2]
lines = open(path).read().split("\n")
start = next(i for i, l in enumerate(lines) if re.match(r"^[_A-Za-z0-9]+:", l) and key in l.split(":")[0])
ops = []
for l in lines[start + 1:]:
    s = l.strip()
    if s.startswith(".Lfunc_end") or s.startswith("s_endpgm"):
        break
    if not s or s.startswith((";", ".")):
        continue
    ops.append(s)
bad = n_checked = 0
hist = {}
for i, s in enumerate(ops):
    if not s.startswith("s_barrier"):
        continue
    # the wait in front of this barrier (skip s_nop / salu)
    j = i - 1
    while j >= 0 and not ops[j].startswith("s_waitcnt") and j > i - 4:
        j -= 1
    m = re.search(r"vmcnt\((\d+)\)", ops[j]) if j >= 0 and ops[j].startswith("s_waitcnt") else None
    if not m:
        hist["none"] = hist.get("none", 0) + 1
        continue
    n = int(m.group(1))
    young = 0
    k = j - 1
    while k >= 0:
        o = ops[k]
        first = o.split()[0]
        if first.startswith(("buffer_", "global_", "scratch_", "flat_")):
            if " lds" in o or "_lds_" in first:
                break
            young += 1
        if first.startswith("s_barrier"):
            young = 1 << 30        # no DMA in this step at all: any count is safe
            break
        k -= 1
    n_checked += 1
    hist[n] = hist.get(n, 0) + 1
    if young < n:
        bad += 1
        print("UNSAFE: barrier #%d waits vmcnt(%d) with only %d younger vector-memory ops behind the last LDS-DMA" % (n_checked, n, young))
print("checked %d barrier waits, counts %s, unsafe %d" % (n_checked, hist, bad))
sys.exit(1 if bad else 0)
