"""Builds librecman_hip.so in-tree (recman_amd/csrc/) with hipcc for gfx950.

The .so is git-ignored but travels to the GPU box with the gpurun snapshot.
Rebuilds only when a source is newer than the library (or force=True).
"""
import glob
import os
import shutil
import subprocess
import sys

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
LIB = os.path.join(CSRC, "librecman_hip.so")
ARCH = "gfx950"


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def _stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = sources() + glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(
        os.path.join(CSRC, "..", "..", "include", "*.h"))
    return any(os.path.getmtime(p) > t for p in deps)


def hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: cannot build librecman_hip.so")
    return exe


def build(force=False, verbose=True, extra_flags=()):
    if not force and not _stale():
        return LIB
    objs = []
    procs = []
    for src in sources():
        obj = src[:-4] + ".o"
        cmd = [hipcc(), "-O3", "-std=c++17", f"--offload-arch={ARCH}", "-fPIC", "-c", src, "-o", obj,
               "-Wall", "-Wno-unused-function", *extra_flags]
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((src, subprocess.Popen(cmd)))
        objs.append(obj)
    for src, p in procs:
        if p.wait() != 0:
            raise RuntimeError(f"hipcc failed on {src}")
    cmd = [hipcc(), "-shared", "-fPIC", f"--offload-arch={ARCH}", *objs, "-o", LIB]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB)
