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
