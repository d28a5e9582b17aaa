"""Time the line-relaxation kernels (solver 2) against the CPU oracle's.  Usage: python tools/time_alr.py [nrows ncols iters]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np, torch
import pdeip_amd as pk
from pdeip_amd import capi, device as dev
import problems as pb, oracle_lib as orc

nrows, ncols, iters = (int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (2160, 3840, 2)
p = pb.elin4(7, nrows, ncols)
for k in ("M", "Cu", "Cv"):
    p[k] *= 0.3
names = list(p.keys())
t0 = time.time(); want = orc.oflow_alr_elin4(*p.values(), 1, 1.5); cpu = time.time() - t0
print("cpu oracle ALR lex: %.3f s / iteration (%dx%d)" % (cpu, nrows, ncols), flush=True)
d = {k: dev.to_device(v) for k, v in p.items()}
for mode, name in ((1, "zebra"), (0, "exact")):
    U, V = d["U"].clone(), d["V"].clone()
    args = [d[k] for k in names[2:]]
    dev.oflow_alr_elin4(U, V, *args, 1, 1.5, mode); torch.cuda.synchronize()
    if mode == 0:
        got = dev.to_matlab(U)
        print("  exact bit-equal to oracle:", pb.bit_equal(got, want[0]), flush=True)
    U, V = d["U"].clone(), d["V"].clone()
    torch.cuda.synchronize(); t0 = time.time()
    reps = iters * (8 if mode == 1 else 1)
    dev.oflow_alr_elin4(U, V, *args, reps, 1.5, mode); torch.cuda.synchronize()
    dt = (time.time() - t0) / reps
    print("gpu %s: %.3f ms / iteration  (%.1fx the CPU oracle)" % (name, dt * 1e3, cpu / dt), flush=True)
