"""The host-side C++ of the product (in-repo LU, symbolic analysis of the sparse Cholesky, option registry) built
with AddressSanitizer + UndefinedBehaviorSanitizer and run on the CPU: sanitizers belong on the CPU build (GPU ASan
is not available on the pool).  No GPU and no HIP runtime are needed."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "paropt_amd", "csrc")


@pytest.mark.skipif(shutil.which("g++") is None or not os.path.isdir("/opt/rocm/include"),
                    reason="needs g++ and the HIP headers")
def test_host_code_under_asan_ubsan():
    subprocess.check_call(["make", "-C", CSRC, "sanitize"], stdout=subprocess.DEVNULL)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    out = subprocess.run([os.path.join(CSRC, "_build", "host_sanitize")], env=env, capture_output=True, text=True,
                         timeout=600)
    report = out.stdout + out.stderr
    assert out.returncode == 0, report[-3000:]
    assert "host_sanitize: ok" in out.stdout
    assert "AddressSanitizer" not in report and "runtime error" not in report and "LeakSanitizer" not in report
