"""The boundary as a C and a C++ consumer sees it, without a GPU: include/rgc_hip.h is a C header (C99, -pedantic -Werror: plain pointers and sizes, no C++
in the signatures), and the host layer in the reference's language (rgc-slam_amd/cpp/*.hpp, the programs under tests/cpp/ that the -m gpu tests run)
compiles and links against the in-tree librgc_hip.so with -Wall -Wextra -Werror.  Running them needs the GPU; a header that no longer compiles should
not wait for it."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "rgc-slam_amd")


def test_the_header_is_a_c_header(tmp_path):
    src = tmp_path / "consumer.c"
    src.write_text('#include "rgc_hip.h"\nint main(void) { rgc_params p; rgc_default_params(&p); return p.max_iterations > 0 ? 0 : 1; }\n')
    for std in ("c99", "c11"):
        subprocess.check_call(["gcc", "-std=" + std, "-pedantic", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(ROOT, "include"), "-c", str(src), "-o", str(tmp_path / "consumer.o")])
    # ... and links as C against the library (rgc_default_params needs no device)
    exe = tmp_path / "consumer"
    subprocess.check_call(["gcc", "-std=c99", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe), "-L", PKG, "-lrgc_hip", "-Wl,-rpath," + PKG])
    assert subprocess.run([str(exe)]).returncode == 0


@pytest.mark.parametrize("program", ["tests/cpp/test_adaptor.cpp", "tests/cpp/test_pipelined.cpp", "tests/cpp/test_dependent.cpp", "tests/cpp/test_odometry_node.cpp",
                                     "rgc-slam_amd/cpp/sequences_per_gpu.cpp"])
def test_the_cpp_host_layer_compiles_and_links(tmp_path, program):
    out = tmp_path / "a.out"
    subprocess.check_call(["g++", "-std=c++14", "-O0", "-Wall", "-Wextra", "-Werror", "-pthread", os.path.join(ROOT, program), "-o", str(out), "-L", PKG, "-lrgc_hip",
                           "-Wl,-rpath," + PKG])
    assert out.exists()
