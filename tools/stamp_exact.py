"""Diagnostic: builds libpdeip_stamps.so (-DPDEIP_EXACT_STAMPS) and prints where one exact-order tile spends its cycles."""
import ctypes, os, subprocess, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
src = os.path.join(ROOT, "pde-based-image-processing_amd", "csrc", "pdeip_capi.hip")
so = os.path.join(ROOT, "tools", "libpdeip_stamps.so")
if not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(os.path.join(os.path.dirname(src), f)) for f in os.listdir(os.path.dirname(src))):
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-fPIC", "-shared", "-std=c++17",
                    "-DPDEIP_EXACT_STAMPS", "-o", so, src], check=True)
lib = ctypes.CDLL(so)
U, V, coef = bench.make_planes(torch, torch.device("cuda"), bench.NROWS, bench.NCOLS)
args = [ctypes.c_void_p(t.data_ptr()) for t in [U, V] + coef]
for _ in range(3):
    rc = lib.pdeip_oflow_sor_elin4_dev(None, *args, bench.NROWS, bench.NCOLS, 4, ctypes.c_float(1.9), 0, 0)
    assert rc == 0
st = (ctypes.c_ulonglong * 64)()
lib.pdeip_debug_read_stamps(st)
names = ["entry", "prologue loads+fetch0 landed", "stash0"]
for k in range(4):
    names += ["c%d fetch issued(+landed: stamp waits)" % k, "c%d compute" % k, "c%d store out" % k, "c%d stash next" % k]
prev = st[0]
for n, name in enumerate(names):
    print("%-42s %8d cycles (+%d)" % (name, st[n] - st[0], st[n] - prev))
    prev = st[n]
