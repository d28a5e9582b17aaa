"""Where a sweep wave's step goes in k_sor_rbp<ModelElin4, 4> at 4K: builds a diagnostic copy of the library with s_memtime stamps
(-DPDEIP_RBP_STAMPS -DRBP_STAMP_SWEEP=n) under gpurun_out/, runs three calls, prints per step (cycles): LDS reads landed, both
half-sweeps issued, hand-off written, LDS writes acknowledged, barrier passed, and the gap to the next step's entry."""
import ctypes, glob, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sweep = sys.argv[1] if len(sys.argv) > 1 else "1"
out = os.path.join(ROOT, "gpurun_out", "libpdeip_stamps%s.so" % sweep)
os.makedirs(os.path.dirname(out), exist_ok=True)
src = sorted(glob.glob(os.path.join(ROOT, "pde-based-image-processing_amd", "csrc", "*.hip")))
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-fno-slp-vectorize", "-fPIC", "-std=c++17", "-shared",
       "-DPDEIP_RBP_STAMPS", "-DRBP_STAMP_SWEEP=" + sweep, "-o", out] + src
pre = os.path.join(ROOT, "pde-based-image-processing_amd", "libpdeip_st%s.so" % sweep)  # tools/build_variant.py st<n> -DPDEIP_RBP_STAMPS -DRBP_STAMP_SWEEP=<n>
if os.path.exists(pre):
    out = pre
else:
    subprocess.run(cmd, check=True, cwd=os.path.join(ROOT, "pde-based-image-processing_amd", "csrc"))
import torch
lib = ctypes.CDLL(out)
nr, nc = 2160, 3840
g = torch.Generator(device="cuda").manual_seed(1)
P = lambda lo, hi: torch.empty((nc, nr), device="cuda").uniform_(lo, hi, generator=g)
U, V, U2, V2 = P(-1, 1), P(-1, 1), P(0, 1), P(0, 1)
coef = [P(-0.5, 0.5) for _ in range(3)] + [P(0.1, 1), P(0.1, 1)] + [P(0.5, 5) for _ in range(4)]
fn = lib.pdeip_oflow_sor_elin4_dev_to
fn.argtypes = [ctypes.c_void_p] * 14 + [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_float, ctypes.c_int, ctypes.c_int]
for _ in range(3):
    rc = fn(None, U.data_ptr(), V.data_ptr(), U2.data_ptr(), V2.data_ptr(), *[c.data_ptr() for c in coef], nr, nc, 4, 1.0, 1, 0)
    assert rc == 0
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * 256)()
assert lib.pdeip_debug_read_rbp_stamps(buf) == 0
rows = [[buf[t * 8 + k] for k in range(6)] for t in range(24)]
print("sweep wave %s: step | reads landed | phases issued | hand-off issued | writes acked | barrier | next entry  (cycles)" % sweep)
for t in range(23):
    r, nxt = rows[t], rows[t + 1][0]
    if not r[0] or not nxt:
        continue
    print("  t=%3d  %5d %5d %5d %5d %5d %5d   step %5d" % (60 + t, r[1] - r[0], r[2] - r[1], r[3] - r[2], r[4] - r[3], r[5] - r[4], nxt - r[5], nxt - r[0]))
