"""The drop-in MEX gateway stubs (pde-based-image-processing_amd/mex/*.c) against a mock MEX runtime.

CPU: every stub compiles and links against libpdeip.so, and its argument checks fire (wrong count, wrong
type, too few outputs) before anything touches the GPU.  GPU: a stub called like MATLAB would call it
returns what the oracle's gateway semantics say."""
import ctypes
import glob
import os
import subprocess

import numpy as np
import pytest

import problems as pb

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MEX_DIR = os.path.join(ROOT, "pde-based-image-processing_amd", "mex")
MOCK_DIR = os.path.join(ROOT, "tests", "mexmock")
BUILD_DIR = os.path.join(MOCK_DIR, "_build")
STUBS = sorted(os.path.basename(f)[:-2] for f in glob.glob(os.path.join(MEX_DIR, "*.c")))
SINGLE, DOUBLE = 7, 6


def build_stub(name, pdeip):
    os.makedirs(BUILD_DIR, exist_ok=True)
    so = os.path.join(BUILD_DIR, name + ".so")
    srcs = [os.path.join(MEX_DIR, name + ".c"), os.path.join(MOCK_DIR, "mexmock.c")]
    deps = srcs + [os.path.join(MEX_DIR, "pdeip_mex_util.h"), os.path.join(MOCK_DIR, "mex.h"), pdeip.capi.LIB_PATH]
    if not os.path.exists(so) or any(os.path.getmtime(d) > os.path.getmtime(so) for d in deps):
        libdir = os.path.dirname(pdeip.capi.LIB_PATH)
        subprocess.run(["gcc", "-O1", "-Wall", "-Wextra", "-Werror", "-Wno-unused-function", "-shared", "-fPIC", "-I" + MOCK_DIR, "-I" + MEX_DIR,
                        "-I" + os.path.join(ROOT, "include"), "-o", so] + srcs + ["-L" + libdir, "-lpdeip", "-Wl,-rpath," + libdir],
                       check=True)
    lib = ctypes.CDLL(so)
    lib.mock_make.restype = ctypes.c_void_p
    lib.mock_make.argtypes = [ctypes.c_int, ctypes.POINTER(ctypes.c_long), ctypes.c_int, ctypes.c_void_p]
    lib.mock_free.argtypes = [ctypes.c_void_p]
    lib.mock_data.restype = ctypes.c_void_p
    lib.mock_data.argtypes = [ctypes.c_void_p]
    lib.mock_ndim.argtypes = [ctypes.c_void_p]
    lib.mock_dim.restype = ctypes.c_long
    lib.mock_dim.argtypes = [ctypes.c_void_p, ctypes.c_int]
    lib.mock_last_error.restype = ctypes.c_char_p
    lib.mock_call.argtypes = [ctypes.c_int, ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, ctypes.POINTER(ctypes.c_void_p)]
    return lib


def to_mx(lib, a):
    a = np.asarray(a)
    a = a.reshape(1, 1) if a.ndim == 0 else a
    a = np.asfortranarray(a)
    dims = (ctypes.c_long * a.ndim)(*a.shape)
    return lib.mock_make(a.ndim, dims, SINGLE if a.dtype == np.float32 else DOUBLE, a.ctypes.data)


def call(lib, nlhs, args):
    """Returns (error message or None, outputs as numpy arrays in MATLAB shape)."""
    prhs = (ctypes.c_void_p * len(args))(*[to_mx(lib, a) for a in args])
    plhs = (ctypes.c_void_p * max(nlhs, 1))()
    rc = lib.mock_call(nlhs, plhs, len(args), prhs)
    outs = []
    if rc == 0:
        for k in range(nlhs):
            shape = tuple(lib.mock_dim(plhs[k], d) for d in range(lib.mock_ndim(plhs[k])))
            n = int(np.prod(shape))
            buf = np.ctypeslib.as_array(ctypes.cast(lib.mock_data(plhs[k]), ctypes.POINTER(ctypes.c_float)), shape=(n,)).copy()
            outs.append(buf.reshape(shape, order="F"))
    for p in list(prhs) + [q for q in plhs if q]:
        lib.mock_free(p)
    return (lib.mock_last_error().decode() if rc else None), outs


def test_all_gateways_have_a_stub():
    assert STUBS == sorted(["Oflow_sor_elin4_2d", "Oflow_sor_llin4_2d", "Oflow_sor_llin8_2d", "Oflow_lhs_elin4_2d",
                            "Oflow_lhs_llin4_2d", "Disp_sor_llin4_2d", "Disp_sor_llin_sym4_2d", "PDEsolver4", "PDEsolver8", "DdiffWeights",
                            "BilinInterp_2d", "FstDerivatives5", "SndDerivatives5",
                            "FlowEminND_llin_2D_v10_gpu", "DispEminND_llin_2D_gpu", "TVdenoise8_gpu", "TVdenoise4_gpu", "FlowEminHS_elin_2D_v10_gpu", "DispEminND_llin_sym_2D_gpu", "FlowEminAD_llin_2D_v10_gpu", "FlowEminNDFASFMG_elin_2D_v10_gpu"])   # the last eight: whole drivers, resident (section 2b of INTEGRATION.md)


@pytest.mark.parametrize("name", STUBS)
def test_stub_compiles_and_checks_arity(pdeip, name):
    lib = build_stub(name, pdeip)
    err, _ = call(lib, 1, [np.zeros((4, 4), np.float32)] * 1)
    assert err is not None and ("wrong number of input parameters" in err or "proper function call" in err)


def test_stub_argument_errors(pdeip):
    lib = build_stub("Oflow_sor_elin4_2d", pdeip)
    p = pb.elin4(1, 8, 8)
    scal = [np.float32(1), np.float32(1.9), np.float32(1)]
    args = list(p.values()) + scal
    bad = list(args)
    bad[3] = bad[3].astype(np.float64)
    err, _ = call(lib, 2, bad)
    assert err == "Oflow_sor_elin4_2d: 'Cu' must be a noncomplex single-valued matrix."
    err, _ = call(lib, 1, args)
    assert "insufficient number of outputs" in err
    err, _ = call(lib, 2, list(p.values()) + [np.float32(1), np.float32(1.9), np.float32(9)])
    assert "no such solver" in err  # the C-ABI refuses before any HIP call


@pytest.mark.gpu
def test_stubs_run_like_matlab_would_call_them(pdeip, oracle):
    pdeip.capi.set_mode(pdeip.MODE_EXACT_ORDER)
    scal = [np.float32(4), np.float32(1.9), np.float32(1)]
    p = pb.elin4(2, 40, 56, nframes=2, nan_frac=0.03)
    err, outs = call(build_stub("Oflow_sor_elin4_2d", pdeip), 4, list(p.values()) + scal)
    assert err is None
    for g, w in zip(outs, oracle.Oflow_sor_elin4_2d(*p.values(), 4, 1.9, nargout=4)):
        assert pb.bit_equal(g, w)
    q = pb.disp4(3, 33, 47)
    err, outs = call(build_stub("Disp_sor_llin4_2d", pdeip), 2, list(q.values()) + scal)
    assert err is None and pb.bit_equal(outs[0], oracle.Disp_sor_llin4_2d(*q.values(), 4, 1.9)) and not outs[1].any()
    r = pb.pde8(4, 30, 34, nframes=3)
    err, outs = call(build_stub("PDEsolver8", pdeip), 1, list(r.values()) + [np.float32(3), np.float32(1.75), np.float32(1)])
    assert err is None and pb.bit_equal(outs[0], oracle.PDEsolver8(*r.values(), 3, 1.75))
    d = pb.diffweights(5, 20, 24, nframes=2)
    err, outs = call(build_stub("DdiffWeights", pdeip), 4, [d["D"], np.float32(1e-5)])
    assert err is None
    for g, w in zip(outs, oracle.DdiffWeights(d["D"], 1e-5)):
        assert pb.bit_equal(g, w)
    w_ = pb.warp(6, 24, 31, nframes=2)
    err, outs = call(build_stub("BilinInterp_2d", pdeip), 1, [w_["Iin"], w_["X"], w_["Y"]])
    assert err is None and pb.bit_equal(outs[0], oracle.BilinInterp_2d(w_["Iin"], w_["X"], w_["Y"]))
    # solver = 2, the drivers' default (alternating line relaxation), through the same stub
    err, outs = call(build_stub("Oflow_sor_elin4_2d", pdeip), 2, list(p.values()) + [np.float32(2), np.float32(1.5), np.float32(2)])
    assert err is None
    for g, w in zip(outs, oracle.Oflow_sor_elin4_2d(*p.values(), 2, 1.5, solver=2)):
        assert pb.bit_equal(g, w)
    ip = pb.image_pair(7, 21, 26, nframes=2)
    err, outs = call(build_stub("FstDerivatives5", pdeip), 3, [ip["It0"], ip["It1"]])
    assert err is None
    for g, w in zip(outs, oracle.FstDerivatives5(ip["It0"], ip["It1"])):
        assert pb.bit_equal(g, w)
    err, outs = call(build_stub("SndDerivatives5", pdeip), 5, [ip["It0"], ip["It1"]])
    assert err is None
    for g, w in zip(outs, oracle.SndDerivatives5(ip["It0"], ip["It1"])):
        assert pb.bit_equal(g, w)
