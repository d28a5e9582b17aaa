import importlib, os, sys, ctypes
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench, oracle_lib
dev = importlib.import_module("pde-based-image-processing_amd.device")
capi = importlib.import_module("pde-based-image-processing_amd").capi
U0, V0, coef = bench.make_planes(torch, torch.device("cuda"), bench.NROWS, bench.NCOLS)
print("inputs finite:", [bool(torch.isfinite(t).all()) for t in [U0, V0] + coef])
Up, Vp = U0.clone(), V0.clone()
dev.oflow_sor_elin4(Up, Vp, *coef, 4, 1.9, capi.MODE_EXACT_ORDER)
torch.cuda.synchronize()
print("gpu exact finite:", bool(torch.isfinite(Up).all()), float(Up.abs().max()))
Ur, Vr = U0.clone(), V0.clone()
dev.oflow_sor_elin4(Ur, Vr, *coef, 4, 1.9, capi.MODE_RED_BLACK)
print("gpu rb finite:", bool(torch.isfinite(Ur).all()), float(Ur.abs().max()))
rate, first = bench.cpu_baseline(U0, V0, coef, 1)
print("cpu finite:", np.isfinite(first[0]).all(), np.abs(first[0]).max(), "nan count", np.isnan(first[0]).sum())
d = Up.cpu().numpy() - first[0]
print("diff nan:", np.isnan(d).sum(), "max", np.nanmax(np.abs(d)))
bad = np.argwhere(np.isnan(first[0]))
print("first nan (col,row):", bad[:5])
