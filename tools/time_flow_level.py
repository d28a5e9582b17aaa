import importlib, sys, time
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); import numpy as np, torch
fl = importlib.import_module("pde-based-image-processing_amd.flow_level"); dev = importlib.import_module("pde-based-image-processing_amd.device"); capi = importlib.import_module("pde-based-image-processing_amd").capi
jj, ii = np.meshgrid(np.arange(1920), np.arange(1080))
tex = lambda di, dj, c: (np.sin(0.021 * (ii + di) + c) * np.cos(0.017 * (jj + dj) - c) + 0.3 * np.sin(0.11 * (ii + di) + 0.07 * (jj + dj))).astype(np.float32)
I0 = np.stack([tex(0, 0, c) for c in range(3)], axis=2); I1 = np.stack([tex(0.7, -0.4, c) for c in range(3)], axis=2)
Z = np.zeros((1080, 1920), dtype=np.float32)
dI0, dI1, dZ = dev.to_device(I0), dev.to_device(I1), dev.to_device(Z)
dG0, dG1 = dev.rgb2grad(dI0), dev.rgb2grad(dI1)
lp = dict(firstLoop=1, secondLoop=4, iter=4, omega=1.9, alpha=0.15, b1=0.25, b2=0.72, sndTerm="gradmag", solver=1)
for mode in (0, 1, 0):
    lv = fl.FlowLlinLevel(lp, mode=mode)
    lv.run(dG0, dG1, dZ, dZ, dI0, dI1); torch.cuda.synchronize()
    for rep in range(2):
        t0 = time.perf_counter()
        for _ in range(3): lv.run(dG0, dG1, dZ, dZ, dI0, dI1)
        torch.cuda.synchronize()
        print("mode", mode, "%.3f ms" % ((time.perf_counter() - t0) / 3 * 1e3), flush=True)
