"""Sanitizer builds of the host-side native code ("race detection / sanitizers"; CPU only -- GPU ASan is not
available on this pool):

  * oracle/metdet_oracle.c            `make -C oracle san`            -> the oracle's golden tests on the ASan + UBSan build
  * csrc/metdet_host.hpp              `make -C metdetpy_amd/csrc san` -> lineset_nms + the native MeteorCollector (collector.py:420-528,
    (metdet_collector.hpp,               utils.py:780-839) alone, plain g++: the collector's golden / differential tests and the g4
     mdh_lineset_nms, mdh_collector_*)   NMS cases through MDH_LIB

Both are built with -fno-sanitize-recover=all, so any finding aborts the child pytest and fails the test here.
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _asan_runtime():
    p = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    return os.path.realpath(p) if p and os.path.sep in p else None


def _run_child(tests, extra_env):
    asan = _asan_runtime()
    assert asan and os.path.exists(asan), "gcc's libasan.so not found"
    env = dict(os.environ, LD_PRELOAD=asan, PYTHONDONTWRITEBYTECODE="1",
               # (CPython itself is not leak-clean; interposed allocators must not trip over it)
               ASAN_OPTIONS="detect_leaks=0:abort_on_error=1:allocator_may_return_null=1", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1",
               **extra_env)
    out = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "not gpu", "-p", "no:cacheprovider"] + tests, cwd=ROOT, env=env,
                         capture_output=True, text=True, timeout=1500)
    tail = (out.stdout + out.stderr)[-3000:]
    assert out.returncode == 0, tail
    assert "passed" in out.stdout and "AddressSanitizer" not in tail and "runtime error" not in tail, tail
    return out.stdout


def test_oracle_c_port_is_clean_under_asan_and_ubsan():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "-s", "san"])
    so = os.path.join(ROOT, "oracle", "libmetdet_oracle_san.so")
    out = _run_child(["tests/test_oracle_golden.py", "tests/test_sanitizers.py::test_child_really_runs_the_sanitized_library"],
                     {"MDH_ORACLE_LIB": so, "MDH_EXPECT_SANITIZED": so})
    assert " passed" in out


def test_host_side_of_the_library_is_clean_under_asan_and_ubsan():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "metdetpy_amd", "csrc"), "-s", "san"])
    so = os.path.join(ROOT, "metdetpy_amd", "libmetdet_host_san.so")
    out = _run_child(["tests/test_collector_golden.py", "tests/test_host_and_abi.py::test_g4_native_lineset_nms",
                      "tests/test_sanitizers.py::test_child_really_runs_the_sanitized_library"], {"MDH_LIB": so, "MDH_EXPECT_SANITIZED": so})
    assert " passed" in out


def test_child_really_runs_the_sanitized_library():
    """Run inside the children above (a no-op otherwise): the library the binding loaded IS the sanitizer build, ASan is live in the
    process, and a GPU entry point of the host-only build refuses loudly."""
    want = os.environ.get("MDH_EXPECT_SANITIZED")
    if not want:
        return
    import ctypes
    assert ctypes.CDLL(None).__asan_init is not None  # the ASan runtime is in the process
    if "MDH_ORACLE_LIB" in os.environ:
        from oracle import c_oracle
        c_oracle.lib()
        maps = open("/proc/self/maps").read()
        assert "libmetdet_oracle_san.so" in maps and "libmetdet_oracle.so" not in maps
    if "MDH_LIB" in os.environ:
        import numpy as np
        import pytest
        from metdetpy_amd import _lib
        assert _lib.LIB_PATH == want
        L = _lib.load()
        maps = open("/proc/self/maps").read()
        assert "libmetdet_host_san.so" in maps and "libmetdet_hip.so" not in maps
        with pytest.raises(_lib.MetDetHipError, match="host-only"):
            L.mdh_create(0, 8, 8, 2, None)
        lines = np.array([[0, 0, 10, 10], [1, 1, 9, 9], [50, 50, 60, 50]], np.int32)
        kept, prob = _lib.lineset_nms(lines)
        assert len(kept) == 2 and len(prob) == 2
