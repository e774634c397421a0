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
SOURCES = [os.path.join(CSRC, "bitnuc_hip.hip")]
import glob

DEPS = SOURCES + glob.glob(os.path.join(CSRC, "*.h")) + [os.path.join(HERE, "..", "include", "bitnuc_hip.h")]


def hipcc_path():
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: libbitnuc_hip.so cannot be built (there is no CPU fallback)")


def is_stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(d) > t for d in DEPS)


def build_library(force=False, verbose=True, extra_flags=()):
    if not force and not is_stale():
        return LIB
    tmp = f"{LIB}.tmp.{os.getpid()}"  # other processes (bench ranks) only ever see a complete library
    cmd = [hipcc_path(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
           "-Wall", "-Wno-unused-function", "-fno-gpu-rdc", *extra_flags, "-o", tmp, *SOURCES, "-ldl"]
    if verbose:
        print(" ".join(cmd).replace(tmp, LIB), file=sys.stderr)
    try:
        subprocess.run(cmd, check=True, cwd=HERE)
        os.replace(tmp, LIB)
    finally:
        if os.path.exists(tmp):
            os.unlink(tmp)
    return LIB


def ensure_built():
    """Build libbitnuc_hip.so only if it is missing (a fresh checkout: the .so is git-ignored).  An existing
    library is used as is -- file times do not survive every copy, so staleness is `build_library`'s business."""
    if not os.path.exists(LIB):
        build_library(force=True)
    return LIB


if __name__ == "__main__":
    build_library(force="--force" in sys.argv)
    print(LIB)
