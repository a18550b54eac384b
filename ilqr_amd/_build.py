"""Builds libilqr_amd.so (HIP kernels + C ABI) for gfx950 with hipcc, in-tree."""
import os
import subprocess

PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG, "csrc")
LIB = os.path.join(PKG, "lib", "libilqr_amd.so")
SOURCES = [os.path.join(CSRC, "capi.hip")]
HEADERS = [os.path.join(CSRC, f) for f in ("common.hpp", "models.hpp", "boxqp.hpp", "kernels.hpp")] + \
          [os.path.join(os.path.dirname(PKG), "include", "ilqr_amd.h")]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared"]


def stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.exists(f) and os.path.getmtime(f) > t for f in SOURCES + HEADERS)


def build(force=False, verbose=False):
    """Compile the extension if missing or older than its sources. Returns the .so path."""
    if force or stale():
        os.makedirs(os.path.dirname(LIB), exist_ok=True)
        cmd = [HIPCC] + FLAGS + ["-o", LIB] + SOURCES
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force=True, verbose=True))
