"""One exact-order call at 4K under rocprofv3 --kernel-trace: prints per-launch durations in launch order."""
import importlib, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
dev = importlib.import_module("pde-based-image-processing_amd.device")
capi = importlib.import_module("pde-based-image-processing_amd").capi
it = int(sys.argv[1]) if len(sys.argv) > 1 else 4
U0, V0, coef = bench.make_planes(torch, torch.device("cuda"), bench.NROWS, bench.NCOLS)
for _ in range(3):
    U, V = U0.clone(), V0.clone()
    dev.oflow_sor_elin4(U, V, *coef, it, 1.9, capi.MODE_EXACT_ORDER)
torch.cuda.synchronize()
