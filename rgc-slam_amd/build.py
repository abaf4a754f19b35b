"""Builds librgc_hip.so (the product: HIP kernels + C-ABI) in-tree with hipcc for gfx950, and beside it librgc_seq.so (the C++ host layer's
dependent frame loop behind one extern "C" call, cpp/dependent_sequence_c.cpp: host code only, linked against librgc_hip.so).

    python rgc-slam_amd/build.py [--force]
    RGC_EXTRA_FLAGS="-DRGC_LM_POST=0" RGC_LIB_OUT=/tmp/librgc_alt.so python rgc-slam_amd/build.py   # an A/B build beside the product
                                                                                                     # (RGC_HIP_LIB=/tmp/librgc_alt.so loads it)

hipcc cross-compiles without a GPU.  -ffp-contract=off is REQUIRED: the exact-kNN parity with the CPU path
depends on the fp32 squared distance not being FMA-contracted (see csrc/rgc_kernels.hip header).
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.environ.get("RGC_LIB_OUT") or os.path.join(HERE, "librgc_hip.so")
OBJDIR = CSRC if not os.environ.get("RGC_LIB_OUT") else os.path.splitext(LIB)[0] + "_obj"
SRCS = ["rgc_api.hip", "rgc_kernels.hip", "rgc_pre.hip", "rgc_frontend.hip", "rgc_host.cpp"]
DEPS = SRCS + ["rgc_kernels.h", "rgc_lm.h", os.path.join("..", "..", "include", "rgc_hip.h")]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wall", "-Wno-unused-function",
         "-fvisibility=hidden", "-DRGC_BUILD"] + os.environ.get("RGC_EXTRA_FLAGS", "").split()  # e.g. -DRGC_LAB: developer-only exports


SEQ_LIB = os.path.join(os.path.dirname(LIB), "librgc_seq.so" if not os.environ.get("RGC_LIB_OUT") else os.path.splitext(os.path.basename(LIB))[0] + "_seq.so")
SEQ_SRC = os.path.join(HERE, "cpp", "dependent_sequence_c.cpp")


def build_seq(force: bool = False, verbose: bool = False) -> str:
    """the host layer's frame loop (C++, no device code): g++-compatible, built with the same driver for one toolchain"""
    deps = [SEQ_SRC, os.path.join(HERE, "cpp", "fast_vgicp_hip.hpp"), os.path.join(HERE, "..", "include", "rgc_hip.h"), LIB]
    if not force and os.path.exists(SEQ_LIB) and all(os.path.getmtime(d) <= os.path.getmtime(SEQ_LIB) for d in deps):
        return SEQ_LIB
    cmd = [HIPCC, "-x", "c++", "-O2", "-std=c++17", "-fPIC", "-shared", "-Wall", "-fvisibility=hidden", SEQ_SRC, "-o", SEQ_LIB,
           "-L", os.path.dirname(LIB), "-l:" + os.path.basename(LIB), "-Wl,-rpath,$ORIGIN"]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return SEQ_LIB


def build(force: bool = False, verbose: bool = False) -> str:
    lib = build_hip(force, verbose)
    build_seq(force, verbose)
    return lib


def build_hip(force: bool = False, verbose: bool = False) -> str:
    deps = [os.path.join(CSRC, d) for d in DEPS] + [os.path.abspath(__file__)]
    if not force and os.path.exists(LIB) and all(os.path.getmtime(d) <= os.path.getmtime(LIB) for d in deps):
        return LIB
    objs = []
    for src in SRCS:
        os.makedirs(OBJDIR, exist_ok=True)
        obj = os.path.join(OBJDIR, os.path.splitext(src)[0] + ".o")
        cmd = [HIPCC] + (FLAGS if src.endswith(".hip") else FLAGS[1:]) + ["-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
        objs.append(obj)
    cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
