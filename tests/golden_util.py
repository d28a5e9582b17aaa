"""Loading of tests/golden/*.npz (see golden/make_golden.py for provenance)."""
import glob
import json
import os

import numpy as np

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
ORDERED = {"Oflow_sor_elin4_2d", "Oflow_sor_llin4_2d", "Oflow_sor_llin8_2d", "Disp_sor_llin4_2d", "Disp_sor_llin_sym4_2d", "PDEsolver4",
           "PDEsolver8"}


def names():
    return sorted(os.path.basename(f)[:-4] for f in glob.glob(os.path.join(GOLDEN_DIR, "*.npz")))


def load(name):
    z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"), allow_pickle=False)
    meta = json.loads(str(z["meta"]))
    in_keys = sorted(k for k in z.files if k.startswith("in_"))
    inputs = [np.asfortranarray(z[k]) for k in in_keys]
    outs = {}
    for k in z.files:
        if k.startswith("out_"):
            _, tag, idx = k.split("_")
            outs.setdefault(tag, {})[int(idx)] = z[k]
    outs = {tag: tuple(d[i] for i in sorted(d)) for tag, d in outs.items()}
    return meta, inputs, outs


def call(api, meta, inputs, single):
    """Invoke gateway `meta['gateway']` of `api` (mex_api or oracle_lib).  `single`: wrap scalars as float32."""
    fn = getattr(api, meta["gateway"])
    sc = (lambda v: np.float32(v)) if single else (lambda v: v)
    args = list(inputs)
    if meta["gateway"] in ORDERED:
        args += [sc(meta["it"]), sc(meta["omega"])]
        if single:
            args.append(np.float32(meta.get("solver", 1)))
    elif meta["gateway"] == "DdiffWeights":
        args[1] = sc(float(np.asarray(args[1]).reshape(())))
    kw = {"nargout": meta["nargout"]} if "nargout" in meta else {}
    if not single and meta["gateway"] in ORDERED:
        kw["solver"] = meta.get("solver", 1)
    return fn, args, kw
