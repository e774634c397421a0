"""CPU tests of the drop-in boundary: libbitnuc_hip.so loads, exports every symbol
include/bitnuc_hip.h declares, and fails loudly (no CPU fallback) without a GPU."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "bitnuc_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(bitnuc_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_the_reference_surface():
    syms = declared_symbols()
    # one entry point per reference function on the path (src/lib.rs:214-220)
    for name in ["bitnuc_as_2bit", "bitnuc_from_2bit", "bitnuc_encode", "bitnuc_decode",
                 "bitnuc_hdist_scalar", "bitnuc_hdist", "bitnuc_encode_dev", "bitnuc_decode_dev",
                 "bitnuc_as_2bit_batch", "bitnuc_kmer_hdist_scan"]:
        assert name in syms


@pytest.fixture(scope="module", autouse=True)
def _built():
    from bitnuc_amd import build
    build.ensure_built()  # hipcc cross-compiles gfx950 without a GPU; a fresh checkout has no .so


def test_library_exports_every_declared_symbol():
    from bitnuc_amd import _lib
    lib = _lib.load()
    for name in declared_symbols():
        assert hasattr(lib, name), f"{name} declared in include/bitnuc_hip.h but not exported"
    assert set(declared_symbols()) == set(_lib.SIGNATURES), "ctypes table and header disagree"
    assert b"gfx950" in lib.bitnuc_version()


def test_library_carries_the_hash_of_its_sources():
    """bitnuc_version() ends in the csrc_sha16 the library was compiled from; the loader holds it against the sources on disk."""
    from bitnuc_amd import _lib, build
    ver = _lib.load().bitnuc_version().decode()
    assert ver.split("csrc:")[1].split()[0] == build.csrc_sha16() == build.library_sha16(build.LIB), ver
    assert not build.is_stale(build.LIB)


def test_a_stale_library_is_refused_and_rebuilt(tmp_path, monkeypatch):
    """Plant a library that was built from OTHER sources where the product library is expected (a copy whose embedded hash differs, as
    if csrc/ had been edited after the build and the .so had travelled with the tree): the loader refuses it, ensure_built(build=False)
    and a process under a profiler say "build first" instead of compiling, ensure_built() rebuilds it in place and reports that."""
    import shutil
    import subprocess
    import sys
    from bitnuc_amd import _lib, build
    stale = str(tmp_path / "libbitnuc_hip.so")
    shutil.copy(build.LIB, stale)
    data = bytearray(open(stale, "rb").read())
    i = data.find(b" gfx950 csrc:") + len(b" gfx950 csrc:")
    data[i:i + 16] = b"0123456789abcdef"
    open(stale, "wb").write(bytes(data))
    assert build.library_sha16(stale) == "0123456789abcdef" and build.is_stale(stale)
    with pytest.raises(RuntimeError, match="built from other sources"):
        _lib.load(stale)
    with pytest.raises(RuntimeError, match="bitnuc_amd.build"):
        build.ensure_built(build=False, lib=stale)
    monkeypatch.setenv("LD_PRELOAD", "/opt/rocm/lib/rocprofiler-sdk/librocprofiler-sdk-tool.so")
    assert build.under_profiler()
    with pytest.raises(RuntimeError, match="under a profiler"):
        build.ensure_built(lib=stale)
    monkeypatch.delenv("LD_PRELOAD")
    assert not build.under_profiler()
    assert build.ensure_built(lib=stale) == stale
    assert build.LAST_ACTION[stale] == "rebuilt on this box" and build.library_sha16(stale) == build.csrc_sha16()
    # a fresh process loads the rebuilt file and reports the sources' hash
    r = subprocess.run([sys.executable, "-c", f"import sys; sys.path.insert(0, {ROOT!r}); from bitnuc_amd import _lib; print(_lib.load({stale!r}).bitnuc_version().decode())"],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and build.csrc_sha16() in r.stdout, (r.stdout, r.stderr[-2000:])
    # a missing library is built as well
    missing = str(tmp_path / "sub" / "libbitnuc_hip.so")
    os.makedirs(os.path.dirname(missing))
    with pytest.raises(RuntimeError, match="is missing"):
        build.ensure_built(build=False, lib=missing)


def test_identity_ignores_comments_and_blank_lines_but_not_code(tmp_path, monkeypatch):
    """csrc_sha16 hashes the CODE (build.code_only): a planted comment, re-indentation or blank line leaves the library current (and
    profiles/hbm_traffic.json valid); a planted `+ 0`, a changed string literal or another code-changing compiler flag does not."""
    import shutil
    from bitnuc_amd import build
    work = tmp_path / "pkg"
    shutil.copytree(build.CSRC, work / "bitnuc_amd" / "csrc")
    shutil.copytree(os.path.join(ROOT, "include"), work / "include")
    monkeypatch.setattr(build, "CSRC", str(work / "bitnuc_amd" / "csrc"))
    monkeypatch.setattr(build, "HERE", str(work / "bitnuc_amd"))
    base = build.csrc_sha16()
    assert base == build.library_sha16(build.LIB) or build.is_stale(build.LIB)  # (the copy hashes like the tree it was copied from)
    f = work / "bitnuc_amd" / "csrc" / "device_prims.h"
    orig = f.read_text()
    marker = "    return (bad & 0xFCFCFCFCu) != 0u;" if "    return (bad & 0xFCFCFCFCu) != 0u;" in orig else None
    anchor = "__device__ __forceinline__ bool residue_is_bad(uint32_t bad) { return (bad & 0xFCFCFCFCu) != 0u; }"
    assert anchor in orig
    for harmless in (orig.replace(anchor, anchor + "  // a remark"), orig.replace(anchor, "/* a block\n   comment */\n" + anchor),
                     orig.replace(anchor, "\n\n    " + anchor.replace("{ return", "{   return")), "// leading comment\n" + orig):
        f.write_text(harmless)
        assert build.csrc_sha16() == base
    for change in (orig.replace("(bad & 0xFCFCFCFCu) != 0u", "(bad & 0xFCFCFCFCu) + 0 != 0u"), orig.replace("0xFCFCFCFCu", "0xFCFCFCFDu")):
        assert change != orig
        f.write_text(change)
        assert build.csrc_sha16() != base
    f.write_text(orig)
    g = work / "bitnuc_amd" / "csrc" / "runtime.hip"
    text = g.read_text()
    assert '"bitnuc_hip ' in text
    g.write_text(text.replace('"bitnuc_hip ', '"bitnuc_hip  ', 1))  # inside a string literal: code
    assert build.csrc_sha16() != base
    g.write_text(text)
    assert build.csrc_sha16() == base
    assert build.csrc_sha16(extra_flags=["-DSOME_ABLATION=1"]) != base  # a library built with other flags is not the sources' library
    # literals survive the comment stripper
    assert build.code_only('a = "// no /* comment */"; // c\nb = \'"\'; /* x */ c') == 'a = "// no /* comment */";\nb = \'"\'; c'
    del marker


def test_product_library_holds_no_evidence_kernels():
    """The kernels that lost their A/B live in bitnuc_amd/csrc/evidence/*.h, which the kernel headers include only under
    -DBITNUC_SWEEP_VARIANTS: the product library must not contain one of them (the evidence build must contain all of them)."""
    import glob
    import subprocess
    from bitnuc_amd import build
    names = set()
    for f in glob.glob(os.path.join(ROOT, "bitnuc_amd", "csrc", "evidence", "*.h")):
        names |= set(re.findall(r"^(\w+_kernel)\(", open(f).read(), flags=re.M))
    assert {"encode_quad_kernel", "encode_ballot_kernel", "decode_x2_kernel", "encode_batch2_kernel", "decode_batch2_kernel", "block_owner_kernel",
            "decode_fixed_strip_kernel", "decode_batch_plan_lines_kernel", "kmer_scan3_kernel", "probe_win_shape_kernel",
            "kmer_scan_mfma_kernel", "kmer_count_mfma_kernel", "scan_count_finish_kernel"} <= names, names
    product = subprocess.run(["nm", "-C", build.ensure_built()], capture_output=True, text=True).stdout
    assert "encode_kernel" in product and "kmer_scan2_kernel" in product
    # the matrix-core scan ships in exactly one instantiation (distance bytes: one trip per wave) and the fused count in one (three channels per base, a
    # resident grid); their other operand / pack / trip / tiling / channel forms are evidence
    assert set(re.findall(r"kmer_scan_seg_mfma_kernel<([^>]*)>", product)) == {"3, 4, 64"} and "kmer_scan_mfma_kernel<" not in product
    assert set(re.findall(r"kmer_count3_mfma_kernel<([^>]*)>", product)) == {"4, true"} and "kmer_count_mfma_kernel<" not in product
    leaked = [n for n in names if n + "<" in product or n + "(" in product]
    assert not leaked, leaked
    if os.path.exists(build.LIB_SWEEP) and not build.is_stale(build.LIB_SWEEP):
        sweep = subprocess.run(["nm", "-C", build.LIB_SWEEP], capture_output=True, text=True).stdout
        missing = [n for n in names if n not in sweep]
        assert not missing, missing


def test_no_oracle_in_product():
    # the product must not route through the oracle or any CPU fallback
    for dirpath, _, files in os.walk(os.path.join(ROOT, "bitnuc_amd")):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp", ".hpp")):
                src = open(os.path.join(dirpath, f)).read()
                assert "oracle_py" not in src and "bitnuc_oracle" not in src and "orc_" not in src, f
    hpp = open(os.path.join(ROOT, "include", "bitnuc.hpp")).read()
    assert "oracle" not in hpp.lower()


def test_struct_layout_matches_header():
    from bitnuc_amd import _lib
    assert C.sizeof(_lib.BitnucErr) == 32
    assert _lib.BitnucErr.value.offset == 8 and _lib.BitnucErr.index.offset == 16 and _lib.BitnucErr.byte.offset == 24


def test_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import bitnuc_amd
    with pytest.raises(bitnuc_amd.BackendError):
        bitnuc_amd.Context(0)


def _hip_runtime_choice(env_extra, preimport_torch=False):
    import json
    import subprocess
    import sys
    code = ("import sys, json; sys.path.insert(0, %r)\n" % ROOT
            + ("import torch\n" if preimport_torch else "")
            + "from bitnuc_amd import _lib\n_lib.load()\nprint(json.dumps(_lib.hip_runtime_choice))\n")
    env = {k: v for k, v in os.environ.items() if k not in ("BITNUC_NO_TORCH_HIP_PRELOAD", "BITNUC_LOG")}
    env.update(env_extra)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    return json.loads(r.stdout.strip().splitlines()[-1]), r.stderr


def test_hip_runtime_preload_is_logged_and_can_be_opted_out():
    """load() preloads torch's bundled libamdhip64.so so that `import bitnuc_amd; import torch` share ONE HIP
    runtime; a plain C-ABI consumer opts out with BITNUC_NO_TORCH_HIP_PRELOAD, a process that already imported torch is left
    alone, and BITNUC_LOG=1 says on stderr which runtime was chosen."""
    (reason, path), err = _hip_runtime_choice({"BITNUC_NO_TORCH_HIP_PRELOAD": "1", "BITNUC_LOG": "1"})
    assert "BITNUC_NO_TORCH_HIP_PRELOAD" in reason and path is None
    assert "bitnuc_amd: HIP runtime: system runtime" in err
    (reason, path), err = _hip_runtime_choice({}, preimport_torch=True)
    assert "torch already imported" in reason and path is None and "bitnuc_amd:" not in err
    (reason, path), err = _hip_runtime_choice({"BITNUC_LOG": "1"})
    assert ("preloaded torch's bundled runtime" in reason and path and path.endswith("libamdhip64.so")) or "system runtime" in reason
    assert "bitnuc_amd: HIP runtime:" in err


# ---- one ABI, three declarations: the C header, the ctypes table (bitnuc_amd/_lib.py) and the Rust shim (rust/src/ffi.rs) --------
# The Rust shim cannot be compiled in this image (no toolchain), so nothing but this test keeps its `extern "C"` block in step with
# the header: every exported function must be declared in all three places with the same number of arguments and the same
# argument classes (pointer / size / 64-bit integer / int), and the error struct and status codes must agree.
def _strip_comments(text):
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    return re.sub(r"//[^\n]*", " ", text)


def _c_class(t):
    t = t.strip()
    if "*" in t or "[" in t:
        return "ptr"
    base = re.sub(r"\b(const|unsigned)\b", "", t).split()
    name = base[0] if base else ""
    if t.replace("const", "").strip() in ("unsigned", "unsigned int"):
        return "int"
    return {"size_t": "size", "uint64_t": "u64", "int": "int", "double": "f64", "void": "void"}.get(name, name)


def _header_prototypes():
    text = _strip_comments(open(os.path.join(ROOT, "include", "bitnuc_hip.h")).read())
    protos = {}
    for ret, name, args in re.findall(r"(?m)^\s*((?:const\s+)?[A-Za-z_][A-Za-z0-9_]*\s*\**)\s*(bitnuc_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;", text):
        args = " ".join(args.split())
        alist = [] if args in ("", "void") else [a.strip() for a in args.split(",")]
        # drop the parameter name: the class only needs the type part
        classes = []
        for a in alist:
            m = re.match(r"(.*?)([A-Za-z_][A-Za-z0-9_]*)?(\[[^\]]*\])?$", a)
            classes.append(_c_class((m.group(1) or "") + (m.group(3) or "")))
        protos[name] = (_c_class(ret), classes)
    return protos


def _rust_class(t):
    t = t.strip()
    if t.startswith("*"):
        return "ptr"
    return {"usize": "size", "u64": "u64", "c_int": "int", "c_uint": "int", "f64": "f64", "i32": "int"}.get(t, t)


def _rust_prototypes():
    text = re.sub(r"//[^\n]*", " ", open(os.path.join(ROOT, "rust", "src", "ffi.rs")).read())
    protos = {}
    for name, args, ret in re.findall(r"pub fn (bitnuc_[a-z0-9_]+)\s*\(([^)]*)\)\s*(?:->\s*([^;]+))?;", text):
        alist = [a.strip() for a in " ".join(args.split()).split(",") if a.strip()]
        protos[name] = (_rust_class(ret) if ret.strip() else "void", [_rust_class(a.split(":", 1)[1]) for a in alist])
    return protos


def _ctypes_prototypes():
    import ctypes as C
    from bitnuc_amd import _lib as L

    def cls(t):
        if t is None:
            return "void"
        if t in (C.c_void_p, C.c_char_p) or hasattr(t, "contents") or (isinstance(t, type) and issubclass(t, C._Pointer)):
            return "ptr"
        return {C.c_size_t: "size", C.c_uint64: "u64", C.c_int: "int", C.c_uint: "int", C.c_double: "f64"}.get(t, getattr(t, "__name__", str(t)))
    table = next(v for k, v in vars(L).items() if isinstance(v, dict) and "bitnuc_encode" in v)
    return {name: (cls(ret), [cls(a) for a in args]) for name, (ret, args) in table.items()}


def test_header_ctypes_and_rust_declare_the_same_functions():
    header, rust, py = _header_prototypes(), _rust_prototypes(), _ctypes_prototypes()
    assert len(header) >= 60 and "bitnuc_encode_sharded_allgather_overlapped_dev" in header
    # on this platform size_t and uint64_t are both 64-bit integers: the classes may differ in name only where the header says so
    problems = []
    for name, (ret, args) in sorted(header.items()):
        for other, table in (("rust/src/ffi.rs", rust), ("bitnuc_amd/_lib.py", py)):
            if name not in table:
                problems.append(f"{name}: not declared in {other}")
                continue
            oret, oargs = table[name]
            if other.endswith("_lib.py"):  # ctypes.c_size_t IS ctypes.c_uint64 on this platform: one class there
                args_cmp = ["u64" if a == "size" else a for a in args]
            else:
                args_cmp = args
            if len(oargs) != len(args):
                problems.append(f"{name}: {len(args)} arguments in the header, {len(oargs)} in {other}")
            elif oargs != args_cmp:
                problems.append(f"{name}: argument classes {args} in the header, {oargs} in {other}")
            ret_cmp = "u64" if (other.endswith("_lib.py") and ret == "size") else ret
            if ret_cmp != oret:
                problems.append(f"{name}: returns {ret} in the header, {oret} in {other}")
    for other, table in (("rust/src/ffi.rs", rust), ("bitnuc_amd/_lib.py", py)):
        for name in sorted(set(table) - set(header)):
            problems.append(f"{name}: declared in {other} but not in the header")
    assert not problems, "\n".join(problems)


def test_rust_error_struct_and_status_codes_match_the_header():
    header = _strip_comments(open(os.path.join(ROOT, "include", "bitnuc_hip.h")).read())
    rust = open(os.path.join(ROOT, "rust", "src", "ffi.rs")).read()
    c_consts = {k: int(v) for k, v in re.findall(r"\b(BITNUC_[A-Z_]+)\s*=\s*(\d+)", header)}
    c_consts.update({k: int(v) for k, v in re.findall(r"#define\s+(BITNUC_[A-Z_]+)\s+(\d+)", header)})
    r_consts = {k: int(v) for k, v in re.findall(r"pub const (BITNUC_[A-Z_]+)\s*:\s*\w+\s*=\s*(\d+)", rust)}
    assert c_consts and set(r_consts) <= set(c_consts) and all(c_consts[k] == v for k, v in r_consts.items()), (c_consts, r_consts)
    for must in ("BITNUC_OK", "BITNUC_INVALID_BASE", "BITNUC_SEQUENCE_TOO_LONG", "BITNUC_INVALID_LENGTH", "BITNUC_BACKEND_ERROR", "BITNUC_UNIQUE_ID_BYTES"):
        assert must in r_consts, must
    c_fields = re.search(r"typedef struct bitnuc_err\s*\{(.*?)\}", header, flags=re.S).group(1)
    c_fields = [(t.strip(), n) for t, n in re.findall(r"([A-Za-z0-9_ ]+?)\s+([a-z_]+)\s*(?:\[\d+\])?\s*;", c_fields)]
    r_fields = re.search(r"pub struct bitnuc_err\s*\{(.*?)\}", rust, flags=re.S).group(1)
    r_fields = [(n, t.strip()) for n, t in re.findall(r"pub ([a-z_]+)\s*:\s*([^,]+),", r_fields)]
    width = {"int32_t": "i32", "int": "i32", "uint64_t": "u64", "uint8_t": "u8"}
    want = [(n, width[t]) for t, n in c_fields if not n.startswith("_") and n != "pad"]
    got = [(n, t) for n, t in r_fields if not n.startswith("_")]
    assert want == got, (want, got)


def test_header_is_plain_c_and_the_example_links():
    """include/bitnuc_hip.h is a C header (the boundary is a C ABI): examples/roundtrip.c compiles as strict C11 with gcc and links
    against the library (no C++ runtime, no HIP headers needed by the caller)."""
    import subprocess
    import tempfile
    from bitnuc_amd import build
    lib = build.ensure_built()
    with tempfile.TemporaryDirectory() as d:
        exe = os.path.join(d, "roundtrip")
        r = subprocess.run(["gcc", "-std=c11", "-Wall", "-Wextra", "-Werror", "-pedantic", "-I" + os.path.join(ROOT, "include"),
                            os.path.join(ROOT, "examples", "roundtrip.c"), "-L" + os.path.dirname(lib), "-lbitnuc_hip",
                            "-Wl,-rpath," + os.path.dirname(lib), "-o", exe], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-3000:]


@pytest.mark.gpu
def test_c_example_runs():
    import subprocess
    import tempfile
    from bitnuc_amd import build
    lib = build.ensure_built()
    with tempfile.TemporaryDirectory() as d:
        exe = os.path.join(d, "roundtrip")
        subprocess.run(["gcc", "-std=c11", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", "roundtrip.c"),
                        "-L" + os.path.dirname(lib), "-lbitnuc_hip", "-Wl,-rpath," + os.path.dirname(lib), "-o", exe], check=True, capture_output=True)
        r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, (r.stdout, r.stderr)
    assert "36 bases -> 2 words" in r.stdout and "round trip ok" in r.stdout and "InvalidBase('N') at index 3" in r.stdout, r.stdout


def test_every_declared_symbol_is_reached_by_some_test():
    """A symbol counts as reached when a test source (Python, C, C++), bench.py, __graft_entry__.py or the tested C++ mirror names it,
    or when an api.py / dist.py / sequence.py method that calls it is called from one of those."""
    import glob
    srcs = glob.glob(os.path.join(ROOT, "tests", "*.py")) + glob.glob(os.path.join(ROOT, "tests", "c", "*.c*")) + glob.glob(os.path.join(ROOT, "tests", "cpp", "*.cpp"))
    srcs += [os.path.join(ROOT, f) for f in ("bench.py", "__graft_entry__.py", os.path.join("include", "bitnuc.hpp"))]
    tests = "".join(open(f).read() for f in srcs)
    api = "".join(open(os.path.join(ROOT, "bitnuc_amd", f)).read() for f in ("api.py", "dist.py", "sequence.py"))
    methods = re.findall(r"def (\w+)\(self[^)]*\):(.*?)(?=\n    def |\n    @|\nclass |\Z)", api, flags=re.S)
    unreached = []
    for sym in declared_symbols():
        if sym in tests:
            continue
        callers = [m for m, body in methods if re.search(r"\b" + sym + r"\b", body)]
        if not any(re.search(r"\." + m + r"\b", tests) for m in callers):
            unreached.append((sym, callers))
    assert not unreached, unreached
