"""9-point point SOR in the reference's order: ms per call and sweeps/s, persistent kernel vs one launch per front
(PDEIP_PDE8_PERSIST is read once per process: run twice)."""
import importlib, os, sys, time
import numpy as np
import torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
dev = importlib.import_module("pde-based-image-processing_amd.device")
capi = importlib.import_module("pde-based-image-processing_amd.capi")
shapes = [(2160, 3840, 4), (2160, 3840, 20), (1080, 1920, 4), (540, 960, 4), (135, 240, 4)]
print("PDEIP_PDE8_PERSIST=%s" % os.environ.get("PDEIP_PDE8_PERSIST", "1"))
for nr, nc, it in shapes:
    g = torch.Generator(device="cuda").manual_seed(3)
    P = lambda lo, hi: torch.empty((nc, nr), device="cuda").uniform_(lo, hi, generator=g)
    X, TR, Bp = P(0, 1), P(2, 3), P(0, 1)
    W = [P(0.05, 0.25) for _ in range(8)]
    fn = lambda: dev.pde_sor8(X, TR, Bp, *W, it, 1.0, capi.MODE_EXACT_ORDER)
    for _ in range(2): fn()
    torch.cuda.synchronize()
    n = 10
    t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    print("%5dx%-5d iter=%-3d %8.3f ms/call  %8.1f sweeps/s  finite=%s" % (nr, nc, it, dt * 1e3, it / dt, bool(torch.isfinite(X).all())), flush=True)
err = capi.call("pdeip_persist_error")
print("persist_error:", err)
