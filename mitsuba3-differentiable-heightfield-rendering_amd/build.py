"""Builds libhf.so (HIP kernels + C ABI) for gfx950 in-tree with hipcc.

The shared library is the product: nothing in this package works without it and
there is no CPU / eager fallback (loading fails loudly instead).
"""
import os
import shutil
import subprocess

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG_DIR, "csrc")
LIB_PATH = os.path.join(PKG_DIR, "libhf.so")
SOURCES = ["hf_kernels.hip", "hf_capi.cpp"]
HEADERS = ["hf_device.h", "hf_launch.h", os.path.join("..", "..", "include", "hf.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
         # spec arithmetic: only the fma calls written in the source may fuse
         "-ffp-contract=off", "-fno-fast-math", "-Wno-bitwise-instead-of-logical", "-Wno-unused-function",
         # no SLP vectoriser: its v_pk_* pairs need aligned register pairs and cost these kernels 2-10 spilled registers
         # (traversal) and 20-30 VGPRs (adjoint 88 -> 66, SI 76 -> 59); same IEEE results (profiles/r04_ab/r04_item3)
         "-fno-slp-vectorize"]


def _hipcc():
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", shutil.which("hipcc")):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (set HIPCC)")


def needs_build():
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    if not force and not needs_build():
        return LIB_PATH
    cmd = [_hipcc()] + FLAGS + ["-I", os.path.join(PKG_DIR, "..", "include")] + \
          [os.path.join(CSRC, s) for s in SOURCES] + ["-ldl", "-o", LIB_PATH]   # dl: RCCL is bound at first use
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB_PATH


HOST_EXAMPLE_SRC = os.path.join(PKG_DIR, "..", "examples", "host_loop.cpp")
HOST_EXAMPLE_BIN = os.path.join(PKG_DIR, "..", "examples", "host_loop")


def build_host_example(verbose=False):
    """examples/host_loop: the C ABI driven from plain C++ (links libhf.so, no Python / torch)."""
    build()
    cmd = [_hipcc(), "--offload-arch=gfx950", "-O2", "-std=c++17", HOST_EXAMPLE_SRC, "-o", HOST_EXAMPLE_BIN,
           "-L", PKG_DIR, "-lhf", "-Wl,-rpath,$ORIGIN/../" + os.path.basename(PKG_DIR)]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return HOST_EXAMPLE_BIN


if __name__ == "__main__":
    print(build(force=True, verbose=True))
    print(build_host_example(verbose=True))
