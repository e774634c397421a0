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
