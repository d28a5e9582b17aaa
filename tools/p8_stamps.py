"""Which wave paces a walker of k_pde8_exact_persist: builds a diagnostic copy of the library (-DPDEIP_P8_STAMPS) under gpurun_out/,
runs the 9-point solver in the reference's order and prints, per workgroup (strip b, sweep t) and role, the time the wave worked
between two barriers (summed over the walk) against the walk's length."""
import ctypes, glob, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = os.path.join(ROOT, "gpurun_out", "libpdeip_p8stamps.so")
os.makedirs(os.path.dirname(out), exist_ok=True)
src = sorted(glob.glob(os.path.join(ROOT, "pde-based-image-processing_amd", "csrc", "*.hip")))
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-fno-slp-vectorize", "-fPIC", "-std=c++17", "-shared",
       "-DPDEIP_P8_STAMPS", "-o", out] + src
subprocess.run(cmd, check=True, cwd=os.path.join(ROOT, "pde-based-image-processing_amd", "csrc"))
import torch
lib = ctypes.CDLL(out)
nr, nc, it = 2160, int(sys.argv[1]) if len(sys.argv) > 1 else 3840, int(sys.argv[2]) if len(sys.argv) > 2 else 1
g = torch.Generator(device="cuda").manual_seed(3)
P = lambda lo, hi: torch.empty((nc, nr), device="cuda").uniform_(lo, hi, generator=g)
X, TR, Bp = P(0, 1), P(2, 3), P(0, 1)
W = [P(0.05, 0.25) for _ in range(8)]
fn = lib.pdeip_pde_sor8_dev
fn.argtypes = [ctypes.c_void_p] * 12 + [ctypes.c_int] * 4 + [ctypes.c_float, ctypes.c_int, ctypes.c_int]
for _ in range(3):
    assert fn(None, X.data_ptr(), TR.data_ptr(), Bp.data_ptr(), *[w.data_ptr() for w in W], nr, nc, 1, it, 1.0, 0, 0) == 0
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * 4096)()
assert lib.pdeip_debug_read_p8_stamps(buf) == 0
NC = (nr + 127 + 15) // 16
names = ["compute", "loader", "storer", "west"]
print("%d x %d, iter %d, %d chunks; per chunk: busy ns (of the walk's ns per chunk)" % (nr, nc, it, NC))
for tk in range(256):
    rows = [[buf[(tk * 4 + r) * 4 + k] for k in range(4)] for r in range(4)]
    if not rows[0][1]:
        continue
    b, t = rows[0][3] & 0xffff, rows[0][3] >> 16
    if b not in (0, 1, 2, 10, 30, 58, 59):
        continue
    cells = []
    for r in range(4):
        busy, total, real = rows[r][0], rows[r][1], rows[r][2]
        total, real = rows[0][1], rows[0][2]
        ns_per_tick = real * 10.0 / max(total, 1)
        cells.append("%s %5.0f" % (names[r], busy * ns_per_tick / NC))
    print("  b=%2d t=%d  walk %6.0f ns/chunk   %s" % (b, t, rows[0][2] * 10.0 / NC, "   ".join(cells)))
