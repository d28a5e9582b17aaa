"""One FAS-FMG run (red-black SOR, 2160x3840) for rocprofv3 --kernel-trace --stats: which kernels the coarse scales spend their time in."""
import sys, importlib
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__)))); import numpy as np, torch
fas = importlib.import_module("pde-based-image-processing_amd.fas"); dev = importlib.import_module("pde-based-image-processing_amd.device")
nr, nc = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (2160, 3840)
jj, ii = np.meshgrid(np.arange(nc), np.arange(nr))
big = lambda di, dj: ((np.sin(0.021 * (ii + di)) * np.cos(0.017 * (jj + dj)) + 0.3 * np.sin(0.11 * (ii + di) + 0.07 * (jj + dj)) + 1.5) * 80).astype(np.float32)
d0, d1 = dev.to_device(big(0, 0)[:, :, None]), dev.to_device(big(0.7, -0.4)[:, :, None])
drv = fas.FasFmgFlow(dict(solver=1, omega=1.0), mode=1)
drv.run(d0, d1); torch.cuda.synchronize()
drv.run(d0, d1); torch.cuda.synchronize()
