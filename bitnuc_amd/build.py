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

DEPS = SOURCES + glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(CSRC, "evidence", "*.h")) + [os.path.join(HERE, "..", "include", "bitnuc_hip.h")]
# -amdgpu-mfma-vgpr-form: the matrix-core scan (csrc/scan_mfma_device.h) keeps its accumulator in VGPRs; without it hipcc places it in AGPRs and
# pays a v_accvgpr_read per result and a v_accvgpr_write per initial value (32 moves per 1024 windows)
CXXFLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function", "-fno-gpu-rdc", "-mllvm", "-amdgpu-mfma-vgpr-form=1"]


def hipcc_path():
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: libbitnuc_hip.so cannot be built (there is no CPU fallback)")


def code_only(text):
    """C / C++ / HIP source text without what the compiler does not see: // and /* */ comments removed (string and character
    literals kept intact), runs of blanks collapsed, blank lines dropped.  Line structure is kept (preprocessor directives end at
    a newline), so moving code between lines still counts as a change -- only comments, indentation and blank lines do not."""
    lines, cur, i, n = [], [], 0, len(text)  # cur: the pieces of the current line; blanks outside literals collapse as they arrive

    def blank():
        if cur and cur[-1] != " ":
            cur.append(" ")

    def newline():
        while cur and cur[-1] == " ":
            cur.pop()
        if cur:
            lines.append("".join(cur))
        cur.clear()
    while i < n:
        c = text[i]
        if c == "/" and i + 1 < n and text[i + 1] == "/":
            j = text.find("\n", i)
            while j != -1 and j > 0 and text[j - 1] == "\\":  # a line comment that ends in a backslash continues on the next line
                j = text.find("\n", j + 1)
            i = n if j == -1 else j
        elif c == "/" and i + 1 < n and text[i + 1] == "*":
            j = text.find("*/", i + 2)
            blank()
            i = n if j == -1 else j + 2
        elif c in "\"'":
            j = i + 1
            while j < n and text[j] != c:
                j += 2 if text[j] == "\\" else 1
            cur.append(text[i:j + 1])  # a literal is code, blanks and all
            i = j + 1
        elif c == "\n":
            newline()
            i += 1
        elif c in " \t\r\f\v":
            blank()
            i += 1
        else:
            cur.append(c)
            i += 1
    newline()
    return "\n".join(lines)


def identity_flags(extra_flags=(), sweep=False):
    """The part of the command line that changes the code: folded into csrc_sha16 so that a library built with other flags (an ablation
    -D, another optimisation level) is not taken for 'the sources' library' (ADVICE r4)."""
    return sorted(CXXFLAGS + list(extra_flags) + (["-DBITNUC_SWEEP_VARIANTS"] if sweep else []))


def csrc_sha16(extra_flags=(), sweep=None):
    """Identity of what the library is compiled from: the CODE of csrc/*.h, csrc/*.hip, csrc/evidence/*.h and include/bitnuc_hip.h
    (comments, indentation and blank lines do not count: code_only) + the compiler flags that change code (identity_flags; the
    product and the evidence build share one identity: -DBITNUC_SWEEP_VARIANTS is reported by bitnuc_version() on its own).  Compiled
    into every build as -DBITNUC_CSRC_SHA and returned by bitnuc_version(), so that a binary can be held against the sources beside
    it -- and an edit to a comment leaves both the library and the profiles keyed by this hash (profiles/hbm_traffic.json) current."""
    import hashlib
    h = hashlib.sha256()
    try:
        for path in sorted(glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(CSRC, "*.hip")) + glob.glob(os.path.join(CSRC, "evidence", "*.h"))) + [os.path.join(HERE, "..", "include", "bitnuc_hip.h")]:
            h.update(os.path.basename(path).encode())
            h.update(code_only(open(path, "r", encoding="utf-8", errors="surrogateescape").read()).encode("utf-8", errors="surrogateescape"))
    except FileNotFoundError as e:
        raise RuntimeError(f"bitnuc_amd: the library's identity needs the sources beside it, and {e.filename} is missing (a library built elsewhere -- a CMake or "
                           "cargo build without -DBITNUC_CSRC_SHA -- is accepted with BITNUC_ALLOW_STALE_LIB=1)") from None
    h.update(" ".join(identity_flags(extra_flags)).encode())
    return h.hexdigest()[:16]


_SHA_MARK = b" gfx950 csrc:"  # runtime.hip: bitnuc_version() == "bitnuc_hip <ver> gfx950 csrc:<16 hex digits>[ sweep]"


def library_sha16(lib=LIB):
    """The csrc_sha16 a built library carries, read from the file (no dlopen: a stale library must not enter the process that
    is about to replace it).  None for a missing file or a library from before the identity was compiled in."""
    try:
        data = open(lib, "rb").read()
    except OSError:
        return None
    i = data.find(_SHA_MARK)
    if i < 0:
        return None
    sha = data[i + len(_SHA_MARK): i + len(_SHA_MARK) + 16]
    return sha.decode() if len(sha) == 16 and all(c in b"0123456789abcdef" for c in sha) else None


def is_stale(lib=LIB):
    """True when `lib` is missing or was not compiled from the sources on disk (by content, not by file time: times do not
    survive every copy of a tree)."""
    return library_sha16(lib) != csrc_sha16()


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
    # (an experiment's flags are part of ITS identity: such a library never passes for the product's)
    flags = CXXFLAGS + list(extra_flags) + (["-DBITNUC_SWEEP_VARIANTS"] if sweep else []) + [f'-DBITNUC_CSRC_SHA="{csrc_sha16(extra_flags)}"']
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


def under_profiler():
    """True in a process started under rocprofv3 (its tool library is preloaded and initialises the GPU before the program
    starts): such a process must not start a compiler chain -- every exec in it is the exec-after-GPU-init this pool forbids."""
    if any("rocprof" in os.environ.get(k, "").lower() for k in ("LD_PRELOAD", "ROCP_TOOL_LIBRARIES", "HSA_TOOLS_LIB")):
        return True
    return "ROCPROFILER_LIBRARY_CTOR" in os.environ  # set by rocprofv3 for the application it starts (/opt/rocm/bin/rocprofv3)


LAST_ACTION = {}  # library path -> "as shipped" | "rebuilt on this box" (what ensure_built did in this process; bench.py reports it)


def ensure_built(sweep=False, build=True, lib=None):
    """The library for the sources on disk.  A library that is missing (a fresh checkout: the .so is git-ignored) or that carries
    another csrc_sha16 than the sources beside it (it was built before an edit and travelled with the tree) is REBUILT in place
    when `build` is true; with build=False -- processes started under rocprofv3, whose children must not be a compiler chain --
    that is an error telling the caller to build first.  Never silently uses a binary that is not the sources'."""
    lib = lib or (LIB_SWEEP if sweep else LIB)
    if not is_stale(lib):
        LAST_ACTION.setdefault(lib, "as shipped")
        return lib
    have = library_sha16(lib)
    what = f"{lib} is missing" if not os.path.exists(lib) else f"{lib} was built from other sources (library csrc:{have}, sources csrc:{csrc_sha16()})"
    if not build or under_profiler() or os.environ.get("BITNUC_NO_BUILD"):
        raise RuntimeError(what + ": run `python3 -m bitnuc_amd.build" + (" --sweep" if sweep else "") + "` first (this process may not start a compiler: it runs under a profiler or BITNUC_NO_BUILD is set)")
    print(f"[bitnuc_amd.build] {what}: rebuilding", file=sys.stderr)
    build_library(force=True, sweep=sweep, verbose=False, out=None if lib in (LIB, LIB_SWEEP) else lib)
    if is_stale(lib):
        raise RuntimeError(f"{lib}: still not the sources' library after a rebuild")
    LAST_ACTION[lib] = "rebuilt on this box"
    return lib


if __name__ == "__main__":
    build_library(force="--force" in sys.argv, sweep="--sweep" in sys.argv)
    print(LIB_SWEEP if "--sweep" in sys.argv else LIB)
