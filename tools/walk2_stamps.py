"""Which wave paces a walker of k_sor_walk<ModelElin4>: builds a diagnostic copy of the library (-DPDEIP_P8_STAMPS) under
gpurun_out/, runs the 5-point solver in the reference's order at 2160 x ncols and prints, per workgroup (strip b, sweep t) and
role, the time the wave worked between two barriers per chunk against the walk's length per chunk, and the strips' start times.

    python tools/walk2_stamps.py [ncols] [iter]"""
import ctypes, glob, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = os.path.join(ROOT, "gpurun_out", "libpdeip_p8stamps.so")
os.makedirs(os.path.dirname(out), exist_ok=True)
src = sorted(glob.glob(os.path.join(ROOT, "pde-based-image-processing_amd", "csrc", "*.hip")))
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-fno-slp-vectorize", "-fPIC", "-std=c++17", "-shared",
       "-DPDEIP_P8_STAMPS", "-o", out] + src
if not os.environ.get("PDEIP_STAMPS_BUILT"):
    subprocess.run(cmd, check=True, cwd=os.path.join(ROOT, "pde-based-image-processing_amd", "csrc"))
import torch
lib = ctypes.CDLL(out)
nr, nc, it = 2160, int(sys.argv[1]) if len(sys.argv) > 1 else 3840, int(sys.argv[2]) if len(sys.argv) > 2 else 1
g = torch.Generator(device="cuda").manual_seed(1)
P = lambda lo, hi: torch.empty((nc, nr), device="cuda").uniform_(lo, hi, generator=g)
U, V = P(-1, 1), P(-1, 1)
coef = [P(-0.5, 0.5) for _ in range(3)] + [P(0.1, 1), P(0.1, 1)] + [P(0.5, 5) for _ in range(4)]
fn = lib.pdeip_oflow_sor_elin4_dev
fn.argtypes = [ctypes.c_void_p] * 12 + [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_float, ctypes.c_int]
for _ in range(3):
    assert fn(None, U.data_ptr(), V.data_ptr(), *[c.data_ptr() for c in coef], nr, nc, it, 1.0, 0) == 0
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * 4096)()
assert lib.pdeip_debug_read_walk2_stamps(buf) == 0
NC = (nr - 2 + 63 + 15) // 16
names = ["storer", "compute0", "loader", "compute1", "poller"]
LOAD, C0 = 2, 1
starts = {}
for tk in range(128):
    if buf[(tk * 8 + C0) * 4 + 1]:
        bt = buf[(tk * 8 + C0) * 4 + 3]
        starts[(bt & 0xffff, bt >> 16)] = (buf[(tk * 8 + LOAD) * 4 + 1] * 10.0, buf[(tk * 8 + LOAD) * 4 + 2] * 10.0)  # start ns, duration ns (loader wave)
if (0, 0) in starts:
    t00 = starts[(0, 0)][0]
    for t in range(it):
        print("sweep %d  strip: start us (lag to the west strip) | end us (lag)" % t)
        prev = None
        for b in range(0, 64):
            if (b, t) not in starts:
                continue
            st, du = starts[(b, t)]
            cur = ((st - t00) / 1e3, (st + du - t00) / 1e3)
            if b % 12 == 0 or b >= 58 or b <= 2 or os.environ.get("WALK_STAMPS_ALL"):
                print("  b=%2d  start %8.1f (%5.1f)   end %8.1f (%5.1f)   walk %6.1f" % (b, cur[0], cur[0] - prev[0] if prev else 0.0, cur[1], cur[1] - prev[1] if prev else 0.0, du / 1e3))
            prev = cur
print("%d x %d, iter %d, %d chunks; per chunk: busy ns (of the walk's ns per chunk)" % (nr, nc, it, NC))
for tk in range(128):
    rows = [[buf[(tk * 8 + r) * 4 + k] for k in range(4)] for r in range(5)]
    if not rows[C0][1]:
        continue
    b, t = rows[C0][3] & 0xffff, rows[C0][3] >> 16
    if b not in (0, 1, 2, 10, 30, 58, 59):
        continue
    cells = []
    for r in range(5):
        busy, total, real = rows[r][0], rows[C0][1], rows[C0][2]  # cycles -> ns by the first compute wave's clock pair
        cells.append("%s %5.0f" % (names[r], busy * (real * 10.0 / max(total, 1)) / NC))
    print("  b=%2d t=%d  walk %6.0f ns/chunk  clock %.2f GHz   %s" % (b, t, rows[C0][2] * 10.0 / NC, rows[C0][1] / max(rows[C0][2] * 10.0, 1), "   ".join(cells)))
# the first 16 intervals of a few walkers: per role, busy ns; compute0's end of work (us since the loader of strip 0 began)
tr = (ctypes.c_ulonglong * (128 * 8 * 16))()
assert lib.pdeip_debug_read_walk_trace(tr) == 0
t_zero = None
for tk in range(128):
    if buf[(tk * 8 + C0) * 4 + 1] and (buf[(tk * 8 + C0) * 4 + 3] & 0xffff, buf[(tk * 8 + C0) * 4 + 3] >> 16) == (0, 0):
        t_zero = buf[(tk * 8 + LOAD) * 4 + 1]
for tk in range(128):
    if not buf[(tk * 8 + C0) * 4 + 1]:
        continue
    bt = buf[(tk * 8 + C0) * 4 + 3]
    b, t = bt & 0xffff, bt >> 16
    if (b, t) not in ((0, 0), (1, 0), (2, 0), (10, 0), (1, 1)):
        continue
    clk = buf[(tk * 8 + C0) * 4 + 1] / max(buf[(tk * 8 + C0) * 4 + 2] * 10.0, 1)  # cycles per ns
    print("walker b=%d t=%d: interval: end of compute0's work us | busy ns %s" % (b, t, " ".join(names)))
    for k in range(16):
        cells = []
        for r in range(5):
            w = tr[(tk * 8 + r) * 16 + k]
            cells.append("%5.0f" % ((w & 0xffffff) / clk))
        w0 = tr[(tk * 8 + C0) * 16 + k]
        print("   k=%2d  %8.2f | %s" % (k, ((w0 >> 24) - (t_zero or 0)) / 100.0, " ".join(cells)))
