"""Two FlowEminND_llin_2D_v10 runs at 1080p (red-black SOR) for rocprofv3 --kernel-trace --stats: the driver's kernel mix."""
import importlib, sys
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__)))); import numpy as np, torch
drivers = importlib.import_module("pde-based-image-processing_amd.drivers"); capi = importlib.import_module("pde-based-image-processing_amd").capi
jj, ii = np.meshgrid(np.arange(1920), np.arange(1080))
tex = lambda di, dj, c: (np.sin(0.021 * (ii + di) + c) * np.cos(0.017 * (jj + dj) - c) + 0.3 * np.sin(0.11 * (ii + di) + 0.07 * (jj + dj))).astype(np.float32)
I0 = np.stack([tex(0, 0, c) for c in range(3)], axis=2); I1 = np.stack([tex(0.7, -0.4, c) for c in range(3)], axis=2)
Iseq = np.concatenate([(I0 + 1.3) * 98.0, (I1 + 1.3) * 98.0], axis=2).astype(np.float32)
kw = dict(mode=capi.MODE_RED_BLACK, solver=int(sys.argv[1]) if len(sys.argv) > 1 else 1, omega=1.5)  # argv[1] = 2: zebra line relaxation
for _ in range(2):
    drivers.FlowEminND_llin_2D_v10(Iseq, 3, "grad", "gradmag", **kw); torch.cuda.synchronize()
