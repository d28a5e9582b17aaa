"""Build libpdeip.so (HIP, gfx950) in-tree.  Used by __graft_entry__.build() and the Makefile-less dev loop.

    python pde-based-image-processing_amd/build.py [--force]
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libpdeip.so")
SOURCES = ["pdeip_capi.hip"]
HEADERS = ["pdeip_alr.hpp", "pdeip_fas.hpp", "pdeip_sym.hpp", "pdeip_pyr.hpp", "pdeip_flow.hpp", "pdeip_tv.hpp", "pdeip_models.hpp", "pdeip_pointwise.hpp", "pdeip_sor_exact.hpp", "pdeip_sor_pde8.hpp",
           "pdeip_sor_rb.hpp", os.path.join("..", "..", "include", "pdeip.h")]
# -ffp-contract=off is part of the parity contract (the reference is FMA-free C).
FLAGS = ["--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-fPIC", "-shared", "-std=c++17",
         "-Wall", "-Wno-unused-function"]


def _stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in SOURCES + HEADERS] + [os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    if not force and not _stale():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc] + FLAGS + ["-o", LIB] + [os.path.join(CSRC, s) for s in SOURCES]
    if verbose:
        print("[pdeip] " + " ".join(cmd), flush=True)
    subprocess.run(cmd, check=True, cwd=CSRC)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB)
