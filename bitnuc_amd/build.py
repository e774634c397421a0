"""In-tree build of libbitnuc_hip.so (gfx950) with hipcc.

`python -m bitnuc_amd.build` or `build_library()`; the .so lands next to this
file so that it travels to the GPU box with the repo snapshot.
"""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libbitnuc_hip.so")
LIB_SWEEP = os.path.join(HERE, "libbitnuc_hip_sweep.so")  # evidence build: all 47 codec variants + the ballot formulation
# one translation unit per kernel family + the runtime + the RCCL layer (csrc/runtime.h says who owns what)
UNITS = ["runtime", "codec", "kmer", "batch", "analysis", "comm"]
SOURCES = [os.path.join(CSRC, u + ".hip") for u in UNITS]
import glob

DEPS = SOURCES + glob.glob(os.path.join(CSRC, "*.h")) + [os.path.join(HERE, "..", "include", "bitnuc_hip.h")]
CXXFLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function", "-fno-gpu-rdc"]


def hipcc_path():
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: libbitnuc_hip.so cannot be built (there is no CPU fallback)")


def is_stale(lib=LIB):
    if not os.path.exists(lib):
        return True
    t = os.path.getmtime(lib)
    return any(os.path.getmtime(d) > t for d in DEPS)


def build_library(force=False, verbose=True, extra_flags=(), sweep=False, jobs=None, out=None):
    """sweep=False: the product (libbitnuc_hip.so, the shipped codec variants).  sweep=True: the evidence build
    (libbitnuc_hip_sweep.so, -DBITNUC_SWEEP_VARIANTS) that tools/sweep*.py and the all-variants parity test load.
    The units are compiled side by side (`jobs` at a time, default: the CPUs of this process, at most one per unit) and linked
    into one shared library; objects live in a private temporary directory, the library appears atomically."""
    import concurrent.futures
    import tempfile
    lib = out or (LIB_SWEEP if sweep else LIB)  # out: an experiment's library (tools/ab_*.py), never the product's path
    if not force and not is_stale(lib):
        return lib
    hipcc = hipcc_path()
    flags = CXXFLAGS + list(extra_flags) + (["-DBITNUC_SWEEP_VARIANTS"] if sweep else [])
    if jobs is None:
        jobs = max(1, min(len(UNITS), len(os.sched_getaffinity(0))))
    tmp = f"{lib}.tmp.{os.getpid()}"  # other processes (bench ranks) only ever see a complete library
    with tempfile.TemporaryDirectory(prefix="bitnuc_obj_") as objdir:
        def compile_unit(unit):
            obj = os.path.join(objdir, unit + ".o")
            cmd = [hipcc, *flags, "-c", os.path.join(CSRC, unit + ".hip"), "-o", obj]
            if verbose:
                print(" ".join(cmd), file=sys.stderr)
            subprocess.run(cmd, check=True, cwd=HERE)
            return obj
        with concurrent.futures.ThreadPoolExecutor(max_workers=jobs) as ex:
            objs = list(ex.map(compile_unit, UNITS))
        link = [hipcc, "--offload-arch=gfx950", "-fPIC", "-shared", "-fno-gpu-rdc", "-o", tmp, *objs, "-ldl", "-lpthread"]
        if verbose:
            print(" ".join(link).replace(tmp, lib), file=sys.stderr)
        try:
            subprocess.run(link, check=True, cwd=HERE)
            os.replace(tmp, lib)
        finally:
            if os.path.exists(tmp):
                os.unlink(tmp)
    if not sweep and not out:
        write_build_info()
    return lib


BUILD_INFO = os.path.join(HERE, "BUILD_INFO.json")


def write_build_info():
    """Where the library came from, for machines without the repository's history (the GPU box gets a snapshot without .git):
    the commit the tree was at when the product library was built, and whether the tree had uncommitted changes."""
    import datetime
    import json
    root = os.path.dirname(HERE)
    try:
        head = subprocess.run(["git", "-C", root, "rev-parse", "--short=12", "HEAD"], capture_output=True, text=True, timeout=10).stdout.strip()
        dirty = bool(subprocess.run(["git", "-C", root, "status", "--porcelain", "--", "bitnuc_amd/csrc", "include"], capture_output=True, text=True, timeout=10).stdout.strip())
    except Exception:  # noqa: BLE001
        head, dirty = "", False
    if not head:
        return  # no history here either: keep whatever travelled with the snapshot
    with open(BUILD_INFO, "w") as f:
        json.dump({"commit": head, "csrc_dirty": dirty, "built": datetime.datetime.now().isoformat(timespec="seconds")}, f)


def ensure_built(sweep=False):
    """Build the library only if it is missing (a fresh checkout: the .so is git-ignored).  An existing
    library is used as is -- file times do not survive every copy, so staleness is `build_library`'s business."""
    lib = LIB_SWEEP if sweep else LIB
    if not os.path.exists(lib):
        build_library(force=True, sweep=sweep)
    return lib


if __name__ == "__main__":
    build_library(force="--force" in sys.argv, sweep="--sweep" in sys.argv)
    print(LIB_SWEEP if "--sweep" in sys.argv else LIB)
