"""A few red-black elin4 sweeps at 4K for rocprofv3 --pmc runs (FETCH_SIZE / WRITE_SIZE per launch)."""
import importlib, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
dev = importlib.import_module("pde-based-image-processing_amd.device")
capi = importlib.import_module("pde-based-image-processing_amd").capi
U, V, coef = bench.make_planes(torch, torch.device("cuda"), bench.NROWS, bench.NCOLS)
mode = capi.MODE_EXACT_ORDER if (len(sys.argv) > 1 and sys.argv[1] == "exact") else capi.MODE_RED_BLACK
for _ in range(3):
    dev.oflow_sor_elin4(U, V, *coef, 4, 1.9, mode)
torch.cuda.synchronize()
