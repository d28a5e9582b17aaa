"""Persistent exact-order kernel (PDEIP_EXACT_PERSIST=1): parity vs the oracle + abort-word check, small to large."""
import importlib, os, sys, time
import numpy as np
os.environ["PDEIP_EXACT_PERSIST"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as orc, problems as pb
pkg = importlib.import_module("pde-based-image-processing_amd")
api, capi = pkg.mex_api, pkg.capi
api.set_mode(0)
sizes = [(32, 48), (97, 131), (64, 200), (131, 70), (3, 3), (5, 300), (260, 7), (388, 584), (1080, 1920)]
if len(sys.argv) > 1:
    sizes = [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:]]
ok = True
for shape in sizes:
    for it in (1, 4, 7):
        p = pb.elin4(11, *shape, nan_frac=0.01)
        t0 = time.time()
        got = api.Oflow_sor_elin4_2d(*p.values(), np.float32(it), np.float32(1.9), np.float32(1))
        rc = capi.load().pdeip_persist_error()
        want = orc.Oflow_sor_elin4_2d(*p.values(), it, 1.9)
        good = rc == 0 and pb.bit_equal(got[0], want[0]) and pb.bit_equal(got[1], want[1])
        print("elin4 %s it=%d: %s (%.3fs)%s" % (shape, it, "OK" if good else "MISMATCH rc=%d %s" % (rc, pb.describe_mismatch(got[0], want[0])), time.time() - t0,
              "" if rc == 0 else " " + capi.last_error()), flush=True)
        ok &= good
        if rc != 0:
            sys.exit(2)
for shape in [(97, 131), (388, 584)]:
    q = pb.llin4(21, *shape, nan_frac=0.02)
    got = api.Oflow_sor_llin4_2d(*q.values(), np.float32(4), np.float32(1.9), np.float32(1))
    want = orc.Oflow_sor_llin4_2d(*q.values(), 4, 1.9)
    good = capi.load().pdeip_persist_error() == 0 and pb.bit_equal(got[0], want[0]) and pb.bit_equal(got[1], want[1])
    print("llin4 %s: %s" % (shape, "OK" if good else "MISMATCH"), flush=True); ok &= good
    d = pb.disp4(41, *shape, nan_frac=0.02)
    good = pb.bit_equal(api.Disp_sor_llin4_2d(*d.values(), np.float32(6), np.float32(1.9), np.float32(1)), orc.Disp_sor_llin4_2d(*d.values(), 6, 1.9)) and capi.load().pdeip_persist_error() == 0
    print("disp4 %s: %s" % (shape, "OK" if good else "MISMATCH"), flush=True); ok &= good
    e = pb.pde4(51, *shape, nframes=3, nan_frac=0.02)
    good = pb.bit_equal(api.PDEsolver4(*e.values(), np.float32(5), np.float32(1.75), np.float32(1)), orc.PDEsolver4(*e.values(), 5, 1.75)) and capi.load().pdeip_persist_error() == 0
    print("pde4 F=3 %s: %s" % (shape, "OK" if good else "MISMATCH"), flush=True); ok &= good
print("ALL OK" if ok else "FAILURES")
sys.exit(0 if ok else 1)
