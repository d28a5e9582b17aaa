import importlib, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import problems as pb
dev = importlib.import_module("pde-based-image-processing_amd.device")
slab = importlib.import_module("pde-based-image-processing_amd.slab")
nrows, ncols = 64, 96
p = pb.elin4(501, nrows, ncols)
it_g = [dev.to_device(p[k]) for k in ("U", "V")]
cf_g = [dev.to_device(p[k]) for k in ("M", "Cu", "Cv", "Du", "Dv", "wW", "wN", "wE", "wS")]
sweep = slab.HIP_SWEEPS["elin4"]
for k in (1, 2, 3, 4):
    ref = [t.clone() for t in it_g]
    sweep(ref, cf_g, k, 1.7, 0)
    for lo, hi in ((0, 54), (42, 96), (41, 96), (20, 70)):
        loc = [t[lo:hi].clone() for t in it_g]
        sweep(loc, [t[lo:hi].contiguous() for t in cf_g], k, 1.7, lo)
        torch.cuda.synchronize()
        bad = (loc[0] != ref[0][lo:hi]).any(dim=1).cpu().numpy()
        print("k=%d slab [%d,%d): wrong local columns %s" % (k, lo, hi, np.nonzero(bad)[0].tolist()))
