"""Find a library call that leaves HIP in a state torch cannot initialise from (development aid)."""
import ctypes, importlib, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np
import golden_util as gu
pkg = importlib.import_module("pde-based-image-processing_amd")
hip = ctypes.CDLL("libamdhip64.so")
hip.hipPeekAtLastError.restype = ctypes.c_int
cnt = ctypes.c_int(0)
def state(tag):
    e = hip.hipPeekAtLastError(); r = hip.hipGetDeviceCount(ctypes.byref(cnt))
    print("%-28s last=%d getDeviceCount rc=%d n=%d" % (tag, e, r, cnt.value), flush=True)
state("start")
only = sys.argv[1:] or gu.names()
for name in only:
    meta, inputs, outs = gu.load(name)
    fn, args, kw = gu.call(pkg.mex_api, meta, inputs, single=True)
    for tag in outs:
        pkg.mex_api.set_mode({"lex": 0, "colour": 1, "any": 0}[tag])
        fn(*args, **kw)
    state(name)
import torch
print("torch sees", torch.cuda.device_count(), flush=True)
torch.cuda.init(); print("torch init ok")
