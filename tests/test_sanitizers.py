"""AddressSanitizer + UndefinedBehaviorSanitizer on the CPU builds (GPU sanitizers are not available on this pool):
  * the oracle (oracle/*.c, `make -C oracle asan`) under its own golden-vector tests,
  * the host half of the PRODUCT -- csrc/rgc_host.cpp: fusion solve, IMU filter, ground gate, pose composition, TUM / PCD writers, ~550
    lines of pointer-taking C++ that need no HIP -- built alone with g++ -fsanitize=address,undefined and run under the host-stage tests.
Each runs in a child process with libasan preloaded (an instrumented shared object cannot be loaded into an uninstrumented python
otherwise); any report fails the child (halt_on_error, -fno-sanitize-recover).  No GPU needed."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SAN = ["-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-fno-omit-frame-pointer"]


def _libasan():
    p = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    if not os.path.isabs(p) or not os.path.exists(p):
        pytest.skip("libasan.so not found next to gcc")
    return os.path.realpath(p)


def _env(extra):
    env = dict(os.environ)
    env.update({"LD_PRELOAD": _libasan(), "ASAN_OPTIONS": "detect_leaks=0:halt_on_error=1:abort_on_error=0:exitcode=97",
                "UBSAN_OPTIONS": "print_stacktrace=1:halt_on_error=1", "OMP_NUM_THREADS": "2"})
    env.update(extra)
    return env


def _run_pytest(files, env):
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-p", "no:cacheprovider"] + files, cwd=ROOT, env=env, capture_output=True,
                       text=True, timeout=1500)
    tail = (r.stdout + r.stderr)[-4000:]
    assert r.returncode == 0, tail
    assert "ERROR: AddressSanitizer" not in tail and "runtime error:" not in tail, tail
    return tail


def test_oracle_under_asan_ubsan():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "-s", "asan"])
    out = _run_pytest(["tests/test_oracle_golden.py"], _env({"RGC_ORACLE_ASAN": "1"}))
    assert " passed" in out, out


def test_product_host_stages_under_asan_ubsan(tmp_path):
    lib = str(tmp_path / "librgc_host_asan.so")
    subprocess.check_call(["g++", "-std=c++17", "-fPIC", "-shared", "-Wall", "-DRGC_BUILD"] + SAN +
                          [os.path.join(ROOT, "rgc-slam_amd", "csrc", "rgc_host.cpp"), "-o", lib])
    out = _run_pytest(["tests/test_host_stages.py"], _env({"RGC_HIP_LIB": lib, "RGC_HIP_LIB_PARTIAL": "1"}))
    assert " passed" in out, out
