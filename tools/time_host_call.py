"""The host-pointer call a MEX stub makes (pdeip_oflow_sor_elin4, 2160 x 3840, iter = 4): wall time with the slab overlap of the
upload / sweeps / download (default) and without (PDEIP_HOST_OVERLAP=0), both orderings, against the link: one plane up with
hipMemcpy (torch) timed alone."""
import ctypes, importlib, os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("pde-based-image-processing_amd")
capi = pkg.capi
lib = capi.load()
NR, NC, IT = 2160, 3840, 4
rng = np.random.default_rng(0)
P = lambda lo, hi: rng.uniform(lo, hi, (NC, NR)).astype(np.float32)
U, V = P(-1, 1), P(-1, 1)
coef = [P(-0.25, 0.25) for _ in range(3)] + [P(0, 0.25), P(0, 0.25)] + [P(0.5, 5) for _ in range(4)]
ou, ov = np.empty_like(U), np.empty_like(V)
d = torch.empty((NC, NR), device="cuda")
h = torch.from_numpy(U)
for _ in range(3): d.copy_(h)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10): d.copy_(h)
torch.cuda.synchronize()
up = (time.perf_counter() - t0) / 10
print("one plane up (pageable): %.3f ms = %.1f GB/s; 13 up + 2 down serially at that rate: %.2f ms, 13 up alone: %.2f ms" % (up * 1e3, U.nbytes / up / 1e9, 15 * up * 1e3, 13 * up * 1e3))
res = {}
for mode, name in ((1, "red_black"), (0, "exact_order")):
    capi.set_mode(mode)
    for ov_on in ("1", "0"):
        os.environ["PDEIP_HOST_OVERLAP"] = ov_on
        ts = []
        for k in range(9):
            t0 = time.perf_counter()
            rc = lib.pdeip_oflow_sor_elin4(U.ctypes.data, V.ctypes.data, *[a.ctypes.data for a in coef], NR, NC, 1, IT, ctypes.c_float(1.0), 1, ou.ctypes.data, ov.ctypes.data, None, None)
            capi.check(rc)
            if k >= 2: ts.append(time.perf_counter() - t0)
        res[(name, ov_on)] = (sorted(ts)[len(ts) // 2] * 1e3, ou.copy(), ov.copy())
        print("%-11s overlap=%s  %.3f ms" % (name, ov_on, res[(name, ov_on)][0]))
    a, b = res[(name, "1")], res[(name, "0")]
    print("  same bits with and without the overlap:", bool(np.array_equal(a[1], b[1], equal_nan=True) and np.array_equal(a[2], b[2], equal_nan=True)))
capi.set_mode(0)
