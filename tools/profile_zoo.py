"""Three calls (iter = 4) of every point-SOR kernel family at 2160x3840, both orderings, for rocprofv3 passes
(kernel-trace --stats, --pmc FETCH_SIZE, --pmc WRITE_SIZE): tools/profile_zoo.sh."""
import importlib, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
dev = importlib.import_module("pde-based-image-processing_amd.device")
capi = importlib.import_module("pde-based-image-processing_amd").capi
g = torch.Generator(device="cuda").manual_seed(0)
nr, nc = 2160, 3840
IT = int(os.environ.get("ZOO_ITER", "4"))

def planes(n, lo=0.5, hi=5.0):
    return [torch.empty((nc, nr), device="cuda").uniform_(lo, hi, generator=g) for _ in range(n)]

which = sys.argv[1].split(",") if len(sys.argv) > 1 else ["elin4", "llin4", "disp4", "pde4", "pde8"]
modes = [capi.MODE_RED_BLACK] if os.environ.get("ZOO_RB_ONLY") else [capi.MODE_RED_BLACK, capi.MODE_EXACT_ORDER]
a, b, c = planes(3, -0.5, 0.5)
M, Cu, Cv, Du, Dv = a * b, -a * c, -b * c, a * a + 0.05, b * b + 0.05
w = planes(4)
U, V = planes(2, -1, 1)
dU, dV = planes(2, -0.1, 0.1)
X, B = planes(2, 0, 1)
w8 = w + planes(4)
TR4, TR8 = 1 + sum(w), 1 + sum(w8)
calls = {
    "elin4": lambda m: dev.oflow_sor_elin4(U, V, M, Cu, Cv, Du, Dv, *w, IT, 1.0, m),
    "llin4": lambda m: dev.oflow_sor_llin4(U, V, dU, dV, M, Cu, Cv, Du, Dv, *w, IT, 1.0, m),
    "disp4": lambda m: dev.disp_sor_llin4(U, dU, Cu, Du, *w, IT, 1.0, m),
    "pde4": lambda m: dev.pde_sor4(X, TR4, B, *w, IT, 1.0, m),
    "pde8": lambda m: dev.pde_sor8(X, TR8, B, *w8, IT, 1.0, m),
}
for name in which:
    for m in modes:
        for _ in range(3):
            calls[name](m)
        torch.cuda.synchronize()
print("done", flush=True)
