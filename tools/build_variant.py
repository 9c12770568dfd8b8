#!/usr/bin/env python3
"""Build a variant of libpinn_hip.so into tools/exp/<name>/ (git-ignored, travels with gpurun) for A/B timing:

    tools/build_variant.py NAME "EXTRA FLAGS" file1.hip [file2.hip ...]

The named sources are recompiled with the extra flags; every other object comes from the in-tree build.
Load it with PINN_HIP_LIB=tools/exp/NAME/libpinn_hip.so.
"""
import os
import subprocess
import sys

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [R]
import pinn_amd  # noqa: E402,F401
from pinn_amd import _build  # noqa: E402

name, flags, files = sys.argv[1], sys.argv[2].split(), sys.argv[3:]
_build.build()
out = os.path.join(R, "tools", "exp", name)
os.makedirs(out, exist_ok=True)
objs, procs = [], []
for src, extra in _build.SOURCES:
    sp = os.path.join(_build.CSRC, src)
    if src in files:
        obj = os.path.join(out, src[:-4] + ".o")
        cmd = [_build._hipcc()] + _build.BASE_FLAGS + ["-c", sp, "-o", obj] + extra + flags
        procs.append((cmd, subprocess.Popen(cmd)))
    else:
        obj = sp[:-4] + ".o"
    objs.append(obj)
for cmd, p in procs:
    if p.wait() != 0:
        sys.exit("failed: " + " ".join(cmd))
lib = os.path.join(out, "libpinn_hip.so")
subprocess.check_call([_build._hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs)
print(lib)
