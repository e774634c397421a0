"""CPU-only: the oracle (scalar + AVX2 restatements) under ASan + UBSan with exact-size heap
buffers.  GPU AddressSanitizer is not available on this pool; the device side is covered by
guard-word checks in tests/test_gpu_parity.py."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_oracle_under_asan_ubsan(tmp_path):
    exe = str(tmp_path / "oracle_sanitize")
    cmd = ["gcc", "-O1", "-g", "-std=c11", "-D_POSIX_C_SOURCE=200809L", "-march=x86-64-v3",
           "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
           os.path.join(ROOT, "tests", "c", "oracle_sanitize.c"),
           os.path.join(ROOT, "oracle", "bitnuc_oracle.c"), os.path.join(ROOT, "oracle", "bitnuc_avx2.c"), "-o", exe]
    subprocess.run(cmd, check=True, capture_output=True)
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300,
                         env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0"))
    assert out.returncode == 0, out.stdout + out.stderr
    assert "sanitizer harness ok" in out.stdout


def _build_and_run_host_harness(tmp_path, tag, san_flags, env_extra):
    exe = str(tmp_path / f"host_sanitize_{tag}")
    obj = str(tmp_path / f"oracle_{tag}.o")
    subprocess.run(["gcc", "-O1", "-g", "-std=c11", "-D_POSIX_C_SOURCE=200809L", *san_flags, "-c",
                    os.path.join(ROOT, "oracle", "bitnuc_oracle.c"), "-o", obj], check=True, capture_output=True)
    cmd = ["g++", "-O1", "-g", "-std=c++17", "-Wall", *san_flags, os.path.join(ROOT, "tests", "c", "host_sanitize.cpp"), obj, "-o", exe, "-lpthread"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    out = subprocess.run([exe], capture_output=True, text=True, timeout=900, env=dict(os.environ, **env_extra))
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-6000:]
    assert "host sanitizer harness ok" in out.stdout


def test_product_host_code_under_asan_ubsan(tmp_path):
    """The product's own CPU code -- csrc/host_word.h (single words, below-cutoff bulk calls) and csrc/host_pool.h (the staging
    pool of the pipelined host-pointer path) -- under AddressSanitizer + UBSan: exact-size heap buffers, every length, the
    pool in encode_pipelined's call pattern.  Both headers compile without HIP."""
    _build_and_run_host_harness(tmp_path, "asan", ["-fsanitize=address,undefined", "-fno-sanitize-recover=undefined"],
                                {"ASAN_OPTIONS": "detect_leaks=1:abort_on_error=0"})


def test_product_host_pool_under_tsan(tmp_path):
    """The same harness under ThreadSanitizer: the pool's mutex / condition-variable protocol (blocking and asynchronous jobs,
    1..9 threads, destruction with a job outstanding) has no data race."""
    _build_and_run_host_harness(tmp_path, "tsan", ["-fsanitize=thread"], {"TSAN_OPTIONS": "halt_on_error=1:second_deadlock_stack=1"})
