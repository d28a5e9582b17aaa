"""Regenerates tests/golden/*.npz: inputs + expected outputs of every gateway on the hot path.

PROVENANCE: the expected outputs are produced by this repository's CPU oracle (oracle/pdeip_oracle.c,
a restatement of the reference's C library), NOT by the reference itself -- the reference cannot be
built in this image (it needs MATLAB's mex.h/matrix.h) and ships no golden data.  The fixtures pin the
oracle against accidental change and give the GPU tests committed vectors; they do not pin the oracle
to the reference ("parity unpinned").

    python tests/golden/make_golden.py          # rewrites the fixtures
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import oracle_lib as orc  # noqa: E402
import problems as pb  # noqa: E402

# (fixture name, gateway, problem factory, gateway kwargs)
CASES = [
    ("elin4_a", "Oflow_sor_elin4_2d", lambda: pb.elin4(101, 24, 40), dict(it=4, omega=1.9, nargout=4)),
    ("elin4_nan_frames", "Oflow_sor_elin4_2d", lambda: pb.elin4(102, 33, 29, nframes=3, nan_frac=0.06), dict(it=7, omega=1.7, nargout=4)),
    ("elin4_iter0", "Oflow_sor_elin4_2d", lambda: pb.elin4(103, 16, 20, nframes=2), dict(it=0, omega=1.9, nargout=4)),
    ("elin4_iter20", "Oflow_sor_elin4_2d", lambda: pb.elin4(104, 40, 24), dict(it=20, omega=1.9, nargout=2)),
    ("llin4_a", "Oflow_sor_llin4_2d", lambda: pb.llin4(111, 24, 40), dict(it=4, omega=1.9, nargout=4)),
    ("llin4_nan_frames", "Oflow_sor_llin4_2d", lambda: pb.llin4(112, 33, 29, nframes=3, nan_frac=0.06), dict(it=5, omega=1.5, nargout=4)),
    ("llin8_a", "Oflow_sor_llin8_2d", lambda: pb.llin8(121, 33, 29, nan_frac=0.03), dict(it=4, omega=1.9, nargout=4)),
    ("disp4_a", "Disp_sor_llin4_2d", lambda: pb.disp4(131, 24, 40), dict(it=4, omega=1.9, nargout=2)),
    ("disp4_nan", "Disp_sor_llin4_2d", lambda: pb.disp4(132, 33, 29, nan_frac=0.06), dict(it=6, omega=1.9, nargout=1)),
    ("dispsym4_nan", "Disp_sor_llin_sym4_2d", lambda: pb.dispsym4(133, 33, 29, nan_frac=0.05), dict(it=4, omega=1.9, nargout=2)),
    ("alr_dispsym4", "Disp_sor_llin_sym4_2d", lambda: pb.dispsym4(207, 24, 40, nan_frac=0.03), dict(it=2, omega=1.4, nargout=2, solver=2)),
    ("pde4_a", "PDEsolver4", lambda: pb.pde4(141, 24, 40), dict(it=4, omega=1.75)),
    ("pde4_frames_nan", "PDEsolver4", lambda: pb.pde4(142, 33, 29, nframes=3, nan_frac=0.06), dict(it=5, omega=1.75)),
    ("pde8_a", "PDEsolver8", lambda: pb.pde8(151, 24, 40), dict(it=4, omega=1.75)),
    ("pde8_frames_nan", "PDEsolver8", lambda: pb.pde8(152, 33, 29, nframes=3, nan_frac=0.06), dict(it=5, omega=1.75)),
    # solver = 2: alternating line relaxation ('lex' = the reference's line order, 'colour' = zebra)
    ("alr_elin4_nan", "Oflow_sor_elin4_2d", lambda: pb.elin4(201, 33, 29, nframes=2, nan_frac=0.05), dict(it=3, omega=1.5, nargout=4, solver=2)),
    ("alr_llin4_a", "Oflow_sor_llin4_2d", lambda: pb.llin4(202, 24, 40, nan_frac=0.03), dict(it=2, omega=1.4, nargout=2, solver=2)),
    ("alr_llin8_a", "Oflow_sor_llin8_2d", lambda: pb.llin8(203, 33, 29, nan_frac=0.03), dict(it=2, omega=1.4, nargout=2, solver=2)),
    ("alr_disp4_a", "Disp_sor_llin4_2d", lambda: pb.disp4(204, 24, 40, nan_frac=0.03), dict(it=3, omega=1.4, nargout=1, solver=2)),
    ("alr_pde4_frames", "PDEsolver4", lambda: pb.pde4(205, 33, 29, nframes=2, nan_frac=0.05), dict(it=3, omega=1.3, solver=2)),
    ("alr_pde8_frames", "PDEsolver8", lambda: pb.pde8(206, 33, 29, nframes=2, nan_frac=0.05), dict(it=3, omega=1.3, solver=2)),
    ("diffweights_a", "DdiffWeights", lambda: dict(pb.diffweights(161, 24, 40), eps=np.float32(1e-5)), dict()),
    ("diffweights_frames", "DdiffWeights", lambda: dict(pb.diffweights(162, 33, 29, nframes=3), eps=np.float32(1e-3)), dict()),
    ("warp_a", "BilinInterp_2d", lambda: pb.warp(171, 24, 40), dict()),
    ("warp_frames", "BilinInterp_2d", lambda: pb.warp(172, 33, 29, nframes=4, max_disp=8.0), dict()),
    ("fstderiv_a", "FstDerivatives5", lambda: pb.image_pair(191, 24, 40), dict()),
    ("fstderiv_frames_small", "FstDerivatives5", lambda: pb.image_pair(192, 4, 7, nframes=3), dict()),
    ("sndderiv_a", "SndDerivatives5", lambda: pb.image_pair(193, 33, 29, nframes=2), dict()),
]
LHS_CASES = [
    ("lhs_elin4_frames", "oflow_lhs_elin4", lambda: pb.elin4(181, 33, 29, nframes=3, nan_frac=0.05, nan_mode="D"),
     ("U", "V", "M", "Du", "Dv", "wW", "wN", "wE", "wS")),
    ("lhs_llin4_frames", "oflow_lhs_llin4", lambda: pb.llin4(182, 33, 29, nframes=3, nan_frac=0.05, nan_mode="D"),
     ("U", "V", "dU", "dV", "M", "Du", "Dv", "wW", "wN", "wE", "wS")),
]
ORDERED = {"Oflow_sor_elin4_2d", "Oflow_sor_llin4_2d", "Oflow_sor_llin8_2d", "Disp_sor_llin4_2d", "Disp_sor_llin_sym4_2d", "PDEsolver4",
           "PDEsolver8"}


def run_case(gateway, p, kw, order):
    fn = getattr(orc, gateway)
    args = list(p.values())
    if gateway in ORDERED:
        out = fn(*args, kw["it"], kw["omega"], solver=kw.get("solver", 1), order=order,
                 **({"nargout": kw["nargout"]} if "nargout" in kw else {}))
    else:
        out = fn(*args)
    return out if isinstance(out, tuple) else (out,)


def main():
    for name, gateway, factory, kw in CASES:
        p = factory()
        blob = {"in_%02d_%s" % (k, key): v for k, (key, v) in enumerate(p.items())}
        orders = [("lex", orc.LEX), ("colour", orc.COLOUR)] if gateway in ORDERED else [("any", 0)]
        for tag, order in orders:
            for k, o in enumerate(run_case(gateway, p, kw, order)):
                blob["out_%s_%d" % (tag, k)] = o
        blob["meta"] = np.array(json.dumps(dict(gateway=gateway, **{k: (float(v) if k == "omega" else v) for k, v in kw.items()})))
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **blob)
    for name, fn, factory, keys in LHS_CASES:
        p = factory()
        args = [p[k] for k in keys]
        blob = {"in_%02d_%s" % (k, key): v for k, (key, v) in enumerate(zip(keys, args))}
        for k, o in enumerate(getattr(orc, fn)(*args)):
            blob["out_any_%d" % k] = o
        blob["meta"] = np.array(json.dumps(dict(gateway={"oflow_lhs_elin4": "Oflow_lhs_elin4_2d", "oflow_lhs_llin4": "Oflow_lhs_llin4_2d"}[fn])))
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **blob)
    print("wrote %d fixtures to %s" % (len(CASES) + len(LHS_CASES), HERE))


if __name__ == "__main__":
    main()
