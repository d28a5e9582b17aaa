"""Exact-order elin4, one sweep, frames of 1 / 2 / 4 / 8 / 16 strips of 64 columns and 2160 or 4320 rows: the chunk time (rows) and
the strip hand-off lag (columns) of the persistent wavefront kernel, separated."""
import importlib, sys, time
import torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
dev = importlib.import_module("pde-based-image-processing_amd.device")
import os
for nr, nc in [tuple(int(x) for x in v.split('x')) for v in os.environ.get('SHAPES', '2160x1026,2160x2050,2160x3840').split(',')]:
    if True:
        for it in (1, 4):
            g = torch.Generator(device="cuda").manual_seed(1)
            P = lambda lo, hi: torch.empty((nc, nr), device="cuda").uniform_(lo, hi, generator=g)
            U, V = P(-1, 1), P(-1, 1)
            coef = [P(-0.5, 0.5) for _ in range(3)] + [P(0.1, 1), P(0.1, 1)] + [P(0.5, 5) for _ in range(4)]
            fn = lambda: dev.oflow_sor_elin4(U, V, *coef, it, 1.0, 0)
            for _ in range(3): fn()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(20): fn()
            torch.cuda.synchronize()
            print("%5d x %5d iter %d: %8.1f us" % (nr, nc, it, (time.perf_counter() - t0) / 20 * 1e6), flush=True)
dev.sync_check()
