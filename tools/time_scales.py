"""Per-scale cost of the solver calls of the FAS driver (which scales are launch-latency bound)."""
import importlib
import sys
import time

import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
pk = importlib.import_module("pde-based-image-processing_amd")
dev = importlib.import_module("pde-based-image-processing_amd.device")
capi = importlib.import_module("pde-based-image-processing_amd.capi")

nr, nc = 2160, 3840
print("%12s %10s %10s %10s %10s %10s" % ("scale", "rb_sor", "exact_sor", "zebra_alr", "exact_alr", "weights"))
while True:
    g = torch.Generator(device="cuda").manual_seed(1)
    P = lambda lo, hi: torch.empty((nc, nr), device="cuda").uniform_(lo, hi, generator=g)
    U, V = P(-1, 1), P(-1, 1)
    coef = [P(-0.5, 0.5) for _ in range(3)] + [P(0.1, 1), P(0.1, 1)] + [P(0.5, 5) for _ in range(4)]
    row = []
    for fn, mode, om in ((dev.oflow_sor_elin4, 1, 1.0), (dev.oflow_sor_elin4, 0, 1.0), (dev.oflow_alr_elin4, 1, 1.5), (dev.oflow_alr_elin4, 0, 1.5)):
        reps = 1 if (mode == 0 and fn is dev.oflow_alr_elin4 and nr > 600) else 5
        fn(U, V, *coef, 4, om, mode)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn(U, V, *coef, 4, om, mode)
        torch.cuda.synchronize()
        row.append((time.perf_counter() - t0) / reps * 1e3)
    w = [torch.empty_like(U) for _ in range(4)]
    dev.flow_opdiffweights(U, V, None, None, *w)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        dev.flow_opdiffweights(U, V, None, None, *w)
    torch.cuda.synchronize()
    row.append((time.perf_counter() - t0) / 5 * 1e3)
    print("%12s " % ("%dx%d" % (nr, nc)) + " ".join("%10.3f" % r for r in row), flush=True)
    if nr <= 10 or nc <= 10:
        break
    nr, nc = (nr + 1) // 2, (nc + 1) // 2
