"""Does a whole FAS-FMG run replay as a HIP graph (torch.cuda.graph capture of the library's launches)?"""
import sys, time, importlib
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__)))); import numpy as np, torch
fas = importlib.import_module("pde-based-image-processing_amd.fas"); dev = importlib.import_module("pde-based-image-processing_amd.device")
nr, nc = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (2160, 3840)
jj, ii = np.meshgrid(np.arange(nc), np.arange(nr))
big = lambda di, dj: ((np.sin(0.021 * (ii + di)) * np.cos(0.017 * (jj + dj)) + 0.3 * np.sin(0.11 * (ii + di) + 0.07 * (jj + dj)) + 1.5) * 80).astype(np.float32)
d0, d1 = dev.to_device(big(0, 0)[:, :, None]), dev.to_device(big(0.7, -0.4)[:, :, None])
for name, prm, mode in (("rb", dict(solver=1, omega=1.0), 1), ("zebra", dict(solver=2, omega=1.5), 1), ("exact", dict(solver=1, omega=1.0), 0)):
    drv = fas.FasFmgFlow(prm, mode=mode)
    U, V = drv.run(d0, d1); torch.cuda.synchronize()
    ref = (U.clone(), V.clone())
    t0 = time.perf_counter(); drv.run(d0, d1); torch.cuda.synchronize(); eager = (time.perf_counter() - t0) * 1e3
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    g = torch.cuda.CUDAGraph()
    t0 = time.perf_counter()
    with torch.cuda.graph(g, stream=side):
        gU, gV = drv.run(d0, d1)
    torch.cuda.synchronize(); cap = (time.perf_counter() - t0) * 1e3
    g.replay(); torch.cuda.synchronize()
    t0 = time.perf_counter(); g.replay(); torch.cuda.synchronize(); rep = (time.perf_counter() - t0) * 1e3
    print("%s: eager %.2f ms, capture %.1f ms, replay %.2f ms, same bits: %s" % (name, eager, cap, rep, bool(torch.equal(gU, ref[0]) and torch.equal(gV, ref[1]))), flush=True)
