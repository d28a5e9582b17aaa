"""CPU: argument checking of the gateway mirror (what mexErrMsgTxt reports), before anything reaches the GPU."""
import numpy as np
import pytest

import problems as pb


def test_rejects_double_inputs(pdeip):
    p = pb.elin4(1, 8, 8)
    args = list(p.values())
    args[3] = args[3].astype(np.float64)
    with pytest.raises(pdeip.mex_api.MexError, match="'Cu' must be a noncomplex single-valued matrix"):
        pdeip.mex_api.Oflow_sor_elin4_2d(*args, np.float32(1), np.float32(1.9), np.float32(1))


def test_rejects_non_single_scalars(pdeip):
    p = pb.elin4(1, 8, 8)
    with pytest.raises(pdeip.mex_api.MexError, match="'iter' must be a noncomplex, single-type scalar"):
        pdeip.mex_api.Oflow_sor_elin4_2d(*p.values(), 4, np.float32(1.9), np.float32(1))


def test_insufficient_outputs(pdeip):
    p = pb.elin4(1, 8, 8)
    with pytest.raises(pdeip.mex_api.MexError, match="insufficient number of outputs"):
        pdeip.mex_api.Oflow_sor_elin4_2d(*p.values(), np.float32(1), np.float32(1.9), np.float32(1), nargout=1)
    with pytest.raises(pdeip.mex_api.MexError, match="insufficient number of outputs"):
        pdeip.mex_api.DdiffWeights(pb.diffweights(1, 8, 8)["D"], np.float32(1e-5), nargout=3)


def test_shape_mismatch(pdeip):
    p = pb.disp4(1, 8, 8)
    args = list(p.values())
    args[4] = np.asfortranarray(np.ones((8, 9), dtype=np.float32))
    with pytest.raises(pdeip.mex_api.MexError, match="'wW' is"):
        pdeip.mex_api.Disp_sor_llin4_2d(*args, np.float32(1), np.float32(1.9), np.float32(1))


def test_c_abi_argument_errors_need_no_gpu(pdeip):
    """NULL pointers / tiny images / unknown solvers are refused by the library before any HIP call."""
    capi = pdeip.capi
    lib = capi.load()
    z = np.zeros((4, 4), dtype=np.float32, order="F")
    ptr = z.ctypes.data
    rc = lib.pdeip_pde_sor4(ptr, ptr, ptr, ptr, ptr, ptr, ptr, 2, 4, 1, 1, 1.0, 1, ptr)
    assert rc == capi.PDEIP_ERR_ARG and "at least 3x3" in capi.last_error()
    rc = lib.pdeip_pde_sor4(ptr, None, ptr, ptr, ptr, ptr, ptr, 4, 4, 1, 1, 1.0, 1, ptr)
    assert rc == capi.PDEIP_ERR_ARG and "TRACE" in capi.last_error()
    rc = lib.pdeip_pde_sor4(ptr, ptr, ptr, ptr, ptr, ptr, ptr, 4, 4, 1, 1, 1.0, 3, ptr)
    assert rc == capi.PDEIP_ERR_SOLVER and "no such solver" in capi.last_error()
    # solver 2 (alternating line relaxation) is a device path like solver 1: without a GPU it fails loudly
    rc = lib.pdeip_pde_sor4(ptr, ptr, ptr, ptr, ptr, ptr, ptr, 4, 4, 1, 1, 1.0, 2, ptr)
    assert rc in (capi.PDEIP_OK, capi.PDEIP_ERR_DEVICE)
