"""Exact-order point SOR per call (in place, resident): us per call and sweeps/s for a few models / sizes / iter."""
import importlib, sys, time
import torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
dev = importlib.import_module("pde-based-image-processing_amd.device")
cases = [(2160, 3840, 4), (2160, 3840, 20), (1080, 1920, 4), (388, 584, 20), (540, 960, 4), (68, 120, 4)]
for nr, nc, it in cases:
    g = torch.Generator(device="cuda").manual_seed(1)
    P = lambda lo, hi: torch.empty((nc, nr), device="cuda").uniform_(lo, hi, generator=g)
    U, V, dU, dV = P(-1, 1), P(-1, 1), P(-0.1, 0.1), P(-0.1, 0.1)
    coef = [P(-0.5, 0.5) for _ in range(3)] + [P(0.1, 1), P(0.1, 1)] + [P(0.5, 5) for _ in range(4)]
    calls = {"elin4": lambda: dev.oflow_sor_elin4(U, V, *coef, it, 1.0, 0), "llin4": lambda: dev.oflow_sor_llin4(U, V, dU, dV, *coef, it, 1.0, 0),
             "disp4": lambda: dev.disp_sor_llin4(U, dU, coef[1], coef[3], *coef[5:], it, 1.0, 0)}
    row = []
    for name, fn in calls.items():
        for _ in range(2): fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10): fn()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 10
        row.append("%s %8.1f us %7.0f/s" % (name, dt * 1e6, it / dt))
    print("%9s iter %2d  %s" % ("%dx%d" % (nr, nc), it, "  ".join(row)), flush=True)
dev.sync_check()
