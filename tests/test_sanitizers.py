"""AddressSanitizer + UndefinedBehaviorSanitizer over the CPU-runnable native code (GPU ASan is not available on the pool):
  * tests/host/host_emul.cpp = the per-lane device functions of csrc/grm_device_fns.h (parser classification and packing, rolling
    k-mers, minimizer words, run heads, records, their decoder) compiled for the host -- shifts by computed amounts, indices into
    unrolled arrays;
  * tests/host/deflate_emul.cpp = the format functions of the device-side zlib encoder (csrc/grm_deflate_fns.h: symbol arithmetic, length-limited
    Huffman codes, header, token -> bits) in the kernels' lockstep order;
  * oracle/grm_oracle.c (the checker itself: an out-of-bounds read there would make every parity claim worthless).
The sanitized libraries are loaded into a child interpreter that has libasan preloaded; any report aborts it."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _asan_runtime():
    p = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    return p if os.path.isabs(p) and os.path.exists(p) else None


@pytest.mark.skipif(_asan_runtime() is None, reason="no libasan in this toolchain")
def test_device_functions_and_oracle_under_sanitizers():
    env = dict(os.environ, LD_PRELOAD=_asan_runtime(), ASAN_OPTIONS="detect_leaks=0:abort_on_error=1", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1",
               GRM_HOST_EMUL_SANITIZED="1", GRM_ORACLE_SANITIZED="1")
    p = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-p", "no:cacheprovider", os.path.join(ROOT, "tests", "test_host_emul.py"),
                        os.path.join(ROOT, "tests", "test_deflate_emul.py"), os.path.join(ROOT, "tests", "test_oracle_micro.py"),
                        os.path.join(ROOT, "tests", "test_oracle_golden.py")],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    tail = (p.stdout + p.stderr)[-3000:]
    assert p.returncode == 0, tail
    assert "passed" in p.stdout and "AddressSanitizer" not in tail and "runtime error" not in tail, tail
