import sys, time, importlib
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__)))); import torch
dev = importlib.import_module("pde-based-image-processing_amd.device")
for nr, nc in ((2160, 3840), (540, 960), (135, 240), (34, 60)):
    g = torch.Generator(device="cuda").manual_seed(1)
    P = lambda lo, hi: torch.empty((nc, nr), device="cuda").uniform_(lo, hi, generator=g)
    X, B = P(0, 1), P(0, 1); w8 = [P(0.5, 5) for _ in range(8)]; TR = 1 + sum(w8)
    dev.pde_sor8(X, TR, B, *w8, 4, 1.5, 1); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10): dev.pde_sor8(X, TR, B, *w8, 4, 1.5, 1)
    torch.cuda.synchronize(); print("%dx%d pde8 rb iter4: %.3f ms" % (nr, nc, (time.perf_counter() - t0) / 10 * 1e3))
