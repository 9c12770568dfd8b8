"""Build libpinn_hip.so (the C-ABI library of include/pinn_hip.h) for gfx950 with hipcc.

In-tree build: objects and the .so land next to the sources in csrc/ (git-ignored, but
shipped to the GPU box with the snapshot).  hipcc cross-compiles without a GPU.
"""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(CSRC, "libpinn_hip.so")
ARCH = "gfx950"

# per-file extra flags: the residual kernel mirrors torch's separately rounded float ops
SOURCES = [
    ("pinn_residuals.hip", ["-ffp-contract=off"]),
    ("pinn_results.hip", ["-ffp-contract=off"]),
    ("pinn_mlp.hip", []),
    ("pinn_train.hip", []),
    ("pinn_bf16.hip", []),
    ("pinn_x6.hip", []),
    ("pinn_x6_train.hip", []),
    ("pinn_x6_wgrad.hip", []),
    ("pinn_wide.hip", []),
    ("pinn_optim.hip", []),
]
HEADERS = ["pinn_mlp_core.h", "pinn_bf16_core.h", "pinn_x6_core.h", "pinn_wgrad_args.h", os.path.join("..", "..", "include", "pinn_hip.h")]


def _hipcc():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: the HIP library cannot be built (set HIPCC=/path/to/hipcc)")


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def build(force=False, verbose=False):
    """Compile every HIP source for gfx950 and link csrc/libpinn_hip.so. Returns its path."""
    hipcc = _hipcc()
    hdrs = [os.path.join(CSRC, h) for h in HEADERS]
    objs = []
    procs = []
    for src, extra in SOURCES:
        sp = os.path.join(CSRC, src)
        if not os.path.exists(sp):
            continue
        obj = sp[:-4] + ".o"
        objs.append(obj)
        if force or _stale(obj, [sp] + hdrs):
            cmd = [hipcc, "--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-c", sp, "-o", obj] + extra
            if verbose:
                print(" ".join(cmd), file=sys.stderr)
            procs.append((cmd, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    for cmd, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError("hipcc failed: %s\n%s" % (" ".join(cmd), out.decode(errors="replace")))
    if force or _stale(LIB, objs):
        cmd = [hipcc, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", LIB] + objs
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
        if r.returncode != 0:
            raise RuntimeError("link failed: %s\n%s" % (" ".join(cmd), r.stdout.decode(errors="replace")))
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
