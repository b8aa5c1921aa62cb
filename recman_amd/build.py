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


def clean_experiments(verbose=True):
    """Removes experiment builds (librecman_*.so other than the product library) from csrc/: they
    would travel to the GPU box with every gpurun push.  Experiment builds belong under build/."""
    for p in glob.glob(os.path.join(CSRC, "librecman_*.so")):
        if os.path.abspath(p) != os.path.abspath(LIB):
            if verbose:
                print(f"[build] removing stray experiment build {p}", flush=True)
            os.remove(p)


def build(force=False, verbose=True, extra_flags=(), out=None, only=None):
    """out: another output path (an experiment build, e.g. build/librecman_x.so, loaded through
    RECMAN_HIP_LIB; its objects go next to it) - the product library is `LIB`.  only: with `out`,
    the source stems (e.g. ["cross"]) that are recompiled with extra_flags; every other
    translation unit is taken from the product build's objects."""
    clean_experiments(verbose)
    if out is None and not force and not _stale():
        return LIB
    if out is not None:
        if only:
            build(force=False, verbose=verbose)  # the product objects must be current
        return _build_variant(out, extra_flags, verbose, only)
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


def _build_variant(out, extra_flags, verbose, only=None):
    out = os.path.abspath(out)
    os.makedirs(os.path.dirname(out), exist_ok=True)
    tag = os.path.splitext(os.path.basename(out))[0]
    objs, procs = [], []
    for src in sources():
        if only and os.path.basename(src)[:-4] not in only:
            objs.append(src[:-4] + ".o")
            continue
        obj = os.path.join(os.path.dirname(out), f"{tag}_{os.path.basename(src)[:-4]}.o")
        cmd = [hipcc(), "-O3", "-std=c++17", f"--offload-arch={ARCH}", "-fPIC", "-c", src, "-o", obj,
               "-Wall", "-Wno-unused-function", *extra_flags]
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((src, subprocess.Popen(cmd)))
        objs.append(obj)
    for src, p in procs:
        if p.wait() != 0:
            raise RuntimeError(f"hipcc failed on {src}")
    subprocess.check_call([hipcc(), "-shared", "-fPIC", f"--offload-arch={ARCH}", *objs, "-o", out])
    return out


if __name__ == "__main__":
    # python -m recman_amd.build [--force] [--out build/librecman_x.so [--only cross,embed] -DFLAG ...]
    argv = sys.argv[1:]
    out = argv[argv.index("--out") + 1] if "--out" in argv else None
    flags = [a for a in argv if a.startswith("-D")]
    only = argv[argv.index("--only") + 1].split(",") if "--only" in argv else None
    print(build(force="--force" in argv, out=out, extra_flags=flags, only=only))
