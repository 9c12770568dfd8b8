"""Build libpinn_hip.so (the C-ABI library of include/pinn_hip.h) for gfx950 with hipcc.

In-tree build: objects and the .so land next to the sources in csrc/ (git-ignored, but
shipped to the GPU box with the snapshot).  hipcc cross-compiles without a GPU.

Staleness is decided by CONTENT, not by mtime: every object carries a side file with the
SHA-256 of its source, of every header and of its command line, and the library one with the
hashes of its objects -- a checkout, a copy to the GPU box or a touched file neither forces nor
hides a rebuild.  One process builds at a time (fcntl lock; every rank of a torchrun job imports
the package), outputs are written under a temporary name and renamed, and a failed compile or
link raises: a stale library is never left looking current.
"""
import fcntl
import hashlib
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(CSRC, "libpinn_hip.so")
LOCK = os.path.join(CSRC, ".build.lock")
ARCH = "gfx950"
BASE_FLAGS = ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC"]

# per-file extra flags: the residual kernel mirrors torch's separately rounded float ops
SOURCES = [
    ("pinn_residuals.hip", ["-ffp-contract=off"]),
    ("pinn_results.hip", ["-ffp-contract=off"]),
    ("pinn_mlp.hip", []),
    ("pinn_train.hip", []),
    ("pinn_bf16.hip", []),
    # no SLP packing in the chain kernels: v_pk_*_f32 do not co-issue with the other wave's MFMAs (MC-dropout + 1.7 %)
    ("pinn_x6.hip", ["-fno-slp-vectorize"]),
    ("pinn_x6_train.hip", ["-fno-slp-vectorize"]),
    ("pinn_x6_wgrad.hip", []),
    ("pinn_wide.hip", []),
    ("pinn_optim.hip", []),
]
HEADERS = ["pinn_mlp_core.h", "pinn_bf16_core.h", "pinn_x6_core.h", "pinn_wgrad_args.h", os.path.join("..", "..", "include", "pinn_hip.h")]


class BuildError(RuntimeError):
    """hipcc ran and failed (as opposed to: hipcc is not installed)."""


class NoCompiler(RuntimeError):
    pass


def _hipcc():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise NoCompiler("hipcc not found: the HIP library cannot be built (set HIPCC=/path/to/hipcc)")


def _sha(paths, extra=()):
    h = hashlib.sha256()
    for p in paths:
        with open(p, "rb") as f:
            h.update(hashlib.sha256(f.read()).digest())
    for e in extra:
        h.update(e.encode())
        h.update(b"\0")
    return h.hexdigest()


def _current(target, digest):
    try:
        with open(target + ".sha256") as f:
            return os.path.exists(target) and f.read().strip() == digest
    except OSError:
        return False


def _stamp(target, digest):
    tmp = "%s.sha256.tmp%d" % (target, os.getpid())
    with open(tmp, "w") as f:
        f.write(digest + "\n")
    os.replace(tmp, target + ".sha256")


def source_digest():
    """One hash over everything the library is built from (sources, headers, flags)."""
    hdrs = [os.path.join(CSRC, h) for h in HEADERS]
    srcs = [os.path.join(CSRC, s) for s, _ in SOURCES]
    return _sha(srcs + hdrs, BASE_FLAGS + [" ".join(e) for _, e in SOURCES])


def is_current():
    """True when csrc/libpinn_hip.so was built from exactly the sources in the tree."""
    hdrs = [os.path.join(CSRC, h) for h in HEADERS]
    digests = []
    for src, extra in SOURCES:
        sp = os.path.join(CSRC, src)
        d = _sha([sp] + hdrs, BASE_FLAGS + extra)
        if not _current(sp[:-4] + ".o", d):
            return False
        digests.append(d)
    return _current(LIB, hashlib.sha256("".join(digests).encode()).hexdigest())


def build(force=False, verbose=False):
    """Compile every HIP source for gfx950 and link csrc/libpinn_hip.so. Returns its path.

    Raises NoCompiler when hipcc is absent and BuildError when a compile or the link fails."""
    if not force and is_current():
        return LIB
    hipcc = _hipcc()
    with open(LOCK, "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if not force and is_current():          # another process built it while we waited
                return LIB
            hdrs = [os.path.join(CSRC, h) for h in HEADERS]
            objs, digests, procs = [], [], []
            for src, extra in SOURCES:
                sp = os.path.join(CSRC, src)
                obj = sp[:-4] + ".o"
                d = _sha([sp] + hdrs, BASE_FLAGS + extra)
                objs.append(obj)
                digests.append(d)
                if force or not _current(obj, d):
                    tmp = "%s.tmp%d" % (obj, os.getpid())
                    cmd = [hipcc] + BASE_FLAGS + ["-c", sp, "-o", tmp] + extra
                    if verbose:
                        print(" ".join(cmd), file=sys.stderr)
                    procs.append((cmd, obj, tmp, d, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
            failed = None
            for cmd, obj, tmp, d, p in procs:
                out, _ = p.communicate()
                if p.returncode != 0:
                    failed = failed or "hipcc failed: %s\n%s" % (" ".join(cmd), out.decode(errors="replace"))
                    if os.path.exists(tmp):
                        os.remove(tmp)
                else:
                    os.replace(tmp, obj)
                    _stamp(obj, d)
            if failed:
                raise BuildError(failed)
            lib_digest = hashlib.sha256("".join(digests).encode()).hexdigest()
            if force or procs or not _current(LIB, lib_digest):
                tmp = "%s.tmp%d" % (LIB, os.getpid())
                cmd = [hipcc, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", tmp] + objs
                r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
                if r.returncode != 0:
                    if os.path.exists(tmp):
                        os.remove(tmp)
                    raise BuildError("link failed: %s\n%s" % (" ".join(cmd), r.stdout.decode(errors="replace")))
                os.replace(tmp, LIB)
                _stamp(LIB, lib_digest)
            return LIB
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
