"""Build libpdeip.so (HIP, gfx950) in-tree.  Used by __graft_entry__.build() and the Makefile-less dev loop.

    python pde-based-image-processing_amd/build.py [--force] [--jobs N]

One object per translation unit (csrc/*.hip), compiled in parallel and only when the unit or a header it
includes changed, then linked into one shared library.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "_obj")
LIB = os.path.join(HERE, "libpdeip.so")
PUBLIC = os.path.join("..", "..", "include", "pdeip.h")
COMMON = ["pdeip_ctx.hpp", PUBLIC]
# translation unit -> the headers it includes (besides COMMON)
UNITS = {
    "pdeip_ctx.hip": [],
    "pdeip_sor5.hip": ["pdeip_models.hpp", "pdeip_pointwise.hpp", "pdeip_sor_exact.hpp", "pdeip_walk_host.hpp", "pdeip_sor_rb.hpp", "pdeip_sor_rbp.hpp", "pdeip_sor_small.hpp", "pdeip_persist_host.hpp"],
    "pdeip_walk5.hip": ["pdeip_models.hpp", "pdeip_sor_exact.hpp", "pdeip_walk_host.hpp", "pdeip_sor_walk.hpp"],
    "pdeip_sor9.hip": ["pdeip_models.hpp", "pdeip_pointwise.hpp", "pdeip_sor_pde8.hpp", "pdeip_sor_pde8_persist.hpp", "pdeip_sor_exact.hpp", "pdeip_sor_rb.hpp", "pdeip_persist_host.hpp"],
    "pdeip_line.hip": ["pdeip_alr.hpp", "pdeip_models.hpp"],
    "pdeip_stages.hip": ["pdeip_models.hpp", "pdeip_pointwise.hpp", "pdeip_flow.hpp", "pdeip_fas.hpp", "pdeip_sym.hpp", "pdeip_pyr.hpp",
                         "pdeip_tv.hpp"],
    "pdeip_host.hip": [],
    "pdeip_drivers.hip": [],
    "pdeip_multi.hip": [],
}
# -ffp-contract=off is part of the parity contract (the reference is FMA-free C).
CFLAGS = ["--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-fno-slp-vectorize", "-fPIC", "-std=c++17", "-Wall", "-Wno-unused-function"]


def _mtime(path):
    return os.path.getmtime(path) if os.path.exists(path) else 0.0


def _units():
    return {u: h for u, h in UNITS.items() if os.path.exists(os.path.join(CSRC, u))}


def _obj(unit):
    return os.path.join(OBJ, unit.replace(".hip", ".o"))


def _unit_stale(unit, headers):
    t = _mtime(_obj(unit))
    deps = [os.path.join(CSRC, f) for f in [unit] + headers + COMMON if os.path.exists(os.path.join(CSRC, f))] + [os.path.abspath(__file__)]
    return t == 0.0 or any(_mtime(d) > t for d in deps)


def build(force=False, verbose=True, jobs=None):
    units = _units()
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    os.makedirs(OBJ, exist_ok=True)
    todo = [u for u, h in units.items() if force or _unit_stale(u, h)]

    def compile_one(unit):
        cmd = [hipcc] + CFLAGS + ["-c", "-o", _obj(unit), os.path.join(CSRC, unit)]
        if verbose:
            print("[pdeip] " + " ".join(cmd), flush=True)
        subprocess.run(cmd, check=True, cwd=CSRC)

    if todo:
        with ThreadPoolExecutor(max_workers=jobs or min(len(todo), os.cpu_count() or 4)) as ex:
            list(ex.map(compile_one, todo))
    if todo or not os.path.exists(LIB) or any(_mtime(_obj(u)) > _mtime(LIB) for u in units):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + [_obj(u) for u in units]
        if verbose:
            print("[pdeip] " + " ".join(cmd), flush=True)
        subprocess.run(cmd, check=True, cwd=CSRC)
    return LIB


if __name__ == "__main__":
    jobs = int(sys.argv[sys.argv.index("--jobs") + 1]) if "--jobs" in sys.argv else None
    build(force="--force" in sys.argv, jobs=jobs)
    print(LIB)
