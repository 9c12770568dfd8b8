#!/usr/bin/env python
"""Registers, spills, scratch, LDS and occupancy of every kernel instantiation in the given sources, one line per kernel:

    python tools/resource_usage.py pinn_x6.hip pinn_x6_train.hip pinn_x6_wgrad.hip > profiles/rNN/resource_usage_x6.txt

Compiles each file for gfx950 with the library's own flags plus -Rpass-analysis=kernel-resource-usage (no GPU needed) and
reduces the remarks.  Kernel names are demangled and stripped of namespaces and argument lists."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pinn_amd import _build  # noqa: E402


def main():
    names = sys.argv[1:] or ["pinn_x6.hip", "pinn_x6_train.hip", "pinn_x6_wgrad.hip"]
    extra = dict((os.path.basename(s), e) for s, e in _build.SOURCES)
    print("# hipcc %s -Rpass-analysis=kernel-resource-usage on %s (one line per kernel instantiation)" % (" ".join(_build.BASE_FLAGS), ", ".join(names)))
    for n in names:
        src = os.path.join(_build.CSRC, n)
        cmd = [_build._hipcc()] + _build.BASE_FLAGS + list(extra.get(n, [])) + ["-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", "/dev/null"]
        txt = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT).stdout.decode(errors="replace")
        cur = None
        rows = {}
        for line in txt.splitlines():
            m = re.search(r"remark: (?:Function Name: )(\S+)", line)
            if m:
                cur = m.group(1)
                rows[cur] = {}
                continue
            m = re.search(r"remark:\s+([A-Za-z ]+[A-Za-z])(?: \[[^\]]*\])?: (\d+)", line)
            if m and cur:
                rows[cur][m.group(1).strip()] = int(m.group(2))
        for k, r in rows.items():
            name = subprocess.run(["c++filt", k], stdout=subprocess.PIPE).stdout.decode().strip()
            name = re.sub(r"\(.*$", "", name).replace("void ", "").replace("pinn::x6::", "").replace("pinn::", "")
            name = name.replace("(anonymous namespace)::", "")
            print("%-58s VGPR %3d AGPR %3d  VGPR-spill %3d  SGPR-spill %3d  scratch %3d B/lane  waves/SIMD %d  LDS(static) %6d" % (
                name, r.get("VGPRs", -1), r.get("AGPRs", -1), r.get("VGPRs Spill", -1), r.get("SGPRs Spill", -1),
                r.get("ScratchSize", -1), r.get("Occupancy", -1), r.get("LDS Size", -1)))


if __name__ == "__main__":
    main()
