"""CPU-side sanitizers: the plain-C rasteriser oracle (oracle/c/raster_oracle.c) and the g++ build of
the product's per-Gaussian math header (tests/host_harness/harness.cpp <- csrc/splat_math.hpp, so_rng.hpp) compiled with
-fsanitize=address,undefined, and their own test files run against those builds in a child process (LD_PRELOAD=libasan:
the interpreter itself is not instrumented).  Never on the GPU side (no sanitizer exists on this pool)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _runtime(name):
    p = subprocess.run(["gcc", f"-print-file-name={name}"], capture_output=True, text=True).stdout.strip()
    return p if os.path.isabs(p) and os.path.exists(p) else None


def test_oracle_and_host_harness_are_clean_under_asan_and_ubsan():
    asan = _runtime("libasan.so")
    if asan is None:
        pytest.skip("gcc's libasan.so is not installed")
    env = dict(os.environ, SPLAT_ONE_AMD_SANITIZE="1", LD_PRELOAD=asan,
               # leaks: the interpreter's own, not ours; halt on the first real finding; python allocates huge regions lazily
               ASAN_OPTIONS="detect_leaks=0:halt_on_error=1:abort_on_error=0:allocator_may_return_null=1",
               UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1", OMP_NUM_THREADS="4")
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-p", "no:cacheprovider", "tests/test_oracle_c.py",
                        "tests/test_host_math.py"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    out = r.stdout + r.stderr
    assert r.returncode == 0 and "passed" in r.stdout, out[-4000:]
    assert "AddressSanitizer" not in out and "runtime error" not in out, out[-4000:]
    # the sanitized libraries are the ones that ran
    for lib in ("oracle/_build/liboracle_f64_san.so", "tests/host_harness/_build/libhh_san.so"):
        assert os.path.exists(os.path.join(ROOT, lib)), lib
