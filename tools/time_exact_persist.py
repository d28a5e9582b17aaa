"""Exact-order point SOR: launch-per-front vs persistent kernel by iter and frame size (run twice: PDEIP_EXACT_PERSIST=0 / 1)."""
import importlib, os, sys, time
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__)))); import torch
dev = importlib.import_module("pde-based-image-processing_amd.device")
print("PDEIP_EXACT_PERSIST =", os.environ.get("PDEIP_EXACT_PERSIST"))
for nr, nc in ((2160, 3840), (1080, 1920), (388, 584), (135, 240), (34, 60)):
    g = torch.Generator(device="cuda").manual_seed(1)
    P = lambda lo, hi: torch.empty((nc, nr), device="cuda").uniform_(lo, hi, generator=g)
    U, V = P(-1, 1), P(-1, 1)
    coef = [P(-0.5, 0.5) for _ in range(3)] + [P(0.1, 1), P(0.1, 1)] + [P(0.5, 5) for _ in range(4)]
    row = []
    for it in (1, 2, 3, 4, 6):
        dev.oflow_sor_elin4(U, V, *coef, it, 1.0, 0); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            dev.oflow_sor_elin4(U, V, *coef, it, 1.0, 0)
        torch.cuda.synchronize()
        row.append((time.perf_counter() - t0) / 5 * 1e3)
    print("%10s " % ("%dx%d" % (nr, nc)) + " ".join("iter%d %.3f" % (it, r) for it, r in zip((1, 2, 3, 4, 6), row)), flush=True)
