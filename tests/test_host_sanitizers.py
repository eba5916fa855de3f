"""`make asan` / `make tsan` of the host-only translation unit, run on the CPU ("ASan for host lib").

front.hip holds the library's host-side concurrency — the thread pool that prepares query rows, the binding of
numpy's BLAS, the streaming sessions' slot and ticket bookkeeping.  Rounds 4 and 5 each found a host-side lifetime /
ordering bug by accident; this builds that unit as plain C++ against tinyknn_amd/csrc/hoststub (a synchronous CPU
stand-in for the HIP calls it makes, and a fake pipelined index) with AddressSanitizer + UBSan and with
ThreadSanitizer, and runs the driver: concurrent parallel regions from two caller threads, a pool resized between
regions, sessions with more batches in flight than slots, waits out of order, a destroy with batches outstanding.
GPU sanitizers do not exist on the pool: the device code is covered by parity tests instead."""
import os
import shutil
import subprocess

import pytest

from tinyknn_amd import _front

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "tinyknn_amd", "csrc")


def blas_path():
    for p in _front._numpy_blas_candidates():
        if os.path.exists(p):
            return p
    pytest.skip("no BLAS shared object found beside numpy")


@pytest.mark.parametrize("kind", ["asan", "tsan"])
def test_host_unit_under_sanitizer(kind):
    if not shutil.which("g++") or not shutil.which("make"):
        pytest.skip("no host toolchain")
    r = subprocess.run(["make", "-C", CSRC, kind], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    env = dict(os.environ, OPENBLAS_NUM_THREADS="1",
               ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", TSAN_OPTIONS="halt_on_error=1",
               UBSAN_OPTIONS="print_stacktrace=1")
    exe = os.path.join(CSRC, "hoststub", "host_" + kind)
    cmd = [exe, blas_path()]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env)
    if kind == "tsan" and r.returncode != 0 and "unexpected memory mapping" in r.stderr and shutil.which("setarch"):
        # (ThreadSanitizer under some kernels' address-space randomisation: same binary without it)
        r = subprocess.run(["setarch", os.uname().machine, "-R"] + cmd, capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0 and "host sanitize: OK" in r.stdout, r.stdout[-2000:] + r.stderr[-6000:]
    assert "ERROR: AddressSanitizer" not in r.stderr and "WARNING: ThreadSanitizer" not in r.stderr
    assert "runtime error" not in r.stderr
