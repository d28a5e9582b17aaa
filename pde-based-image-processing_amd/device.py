"""Device-resident calls: the `_dev` entry points of libpdeip.so on torch tensors.

torch is plumbing here (device memory, streams); all arithmetic is in the HIP library.

Layout.  A MATLAB `single` array [nrows x ncols (x F)] in column-major order is, byte for byte, a
C-contiguous array [(F x) ncols x nrows].  Device planes are therefore torch float32 tensors of
shape [ncols, nrows] or [F, ncols, nrows]; `to_device` / `to_matlab` convert from/to the numpy
arrays `mex_api` uses.  Kernels run on torch's current stream.
"""
import numpy as np
import torch

from . import capi


def to_device(a, device="cuda"):
    """numpy [nrows, ncols(, F)] (any order) -> torch [(F,) ncols, nrows] on `device`, same bytes as MATLAB."""
    a = np.asarray(a, dtype=np.float32)
    t = a.transpose(2, 1, 0) if a.ndim == 3 else a.T
    if t.flags.c_contiguous:   # column-major input (what MATLAB holds): its bytes are the device layout already
        return torch.from_numpy(t).to(device)
    # row-major numpy input: upload the bytes as they are and change the layout on the device (a strided host copy of a
    # 1080p colour pair takes longer than the whole resident run)
    d = torch.from_numpy(np.ascontiguousarray(a)).to(device)
    return (d.permute(2, 1, 0) if a.ndim == 3 else d.T).contiguous()


def div_scalar(t, s):
    """t ./ s in float32 with a true division (a Python-scalar divisor would be turned into a multiplication by 1/s)."""
    return t / torch.tensor(s, dtype=torch.float32, device=t.device)


def to_matlab(t):
    """torch [(F,) ncols, nrows] -> numpy [nrows, ncols(, F)] column-major."""
    a = t.detach().cpu().numpy()
    return np.asfortranarray(a.transpose(2, 1, 0) if a.ndim == 3 else a.T)


def _chk(*tensors):
    for t in tensors:
        if t.dtype != torch.float32 or not t.is_contiguous() or not t.is_cuda:
            raise capi.PdeipError(capi.PDEIP_ERR_ARG, "device planes must be contiguous float32 CUDA tensors")


def _dims(t):
    return int(t.shape[-1]), int(t.shape[-2]), (int(t.shape[0]) if t.dim() == 3 else 1)  # nrows, ncols, F


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _stream():
    """Raw handle of torch's current stream on the current device (the library only enqueues there).  The private accessor
    skips building a Stream object per call -- a solver call on a coarse pyramid scale is shorter than that took."""
    if _raw_stream is not None:
        return _raw_stream(torch.cuda.current_device())
    return torch.cuda.current_stream().cuda_stream


def _p(*ts):
    return [t.data_ptr() for t in ts]


def sync_check():
    """Wait for the device and raise if a bounded dependency wait of the persistent exact-order kernel timed out since
    the last check (the abort word is sticky: later calls cannot clear it).  The resident drivers call this wherever
    they hand results back to the host."""
    torch.cuda.synchronize()
    capi.call("pdeip_persist_error")


def oflow_sor_elin4(U, V, M, Cu, Cv, Du, Dv, wW, wN, wE, wS, iter, omega, mode=capi.MODE_EXACT_ORDER, col0=0, out=None):
    """In place on U, V (GS_SOR_elin4_2d, opticalflowSolvers.c:41); with out=(U2, V2): U, V are only read and the relaxed
    iterate goes to U2, V2 (no device-to-device copy in the red-black mode; callers alternate between two sets)."""
    _chk(U, V, M, Cu, Cv, Du, Dv, wW, wN, wE, wS)
    nrows, ncols, _ = _dims(U)
    if out is None:
        capi.call("pdeip_oflow_sor_elin4_dev", _stream(), *_p(U, V, M, Cu, Cv, Du, Dv, wW, wN, wE, wS), nrows, ncols,
                  int(iter), float(omega), int(mode), int(col0))
    else:
        _chk(*out)
        capi.call("pdeip_oflow_sor_elin4_dev_to", _stream(), *_p(U, V, out[0], out[1], M, Cu, Cv, Du, Dv, wW, wN, wE, wS), nrows,
                  ncols, int(iter), float(omega), int(mode), int(col0))


def oflow_sor_llin4(U, V, dU, dV, M, Cu, Cv, Du, Dv, wW, wN, wE, wS, iter, omega, mode=capi.MODE_EXACT_ORDER, col0=0):
    """In place on dU, dV (GS_SOR_llin4_2d, opticalflowSolvers.c:504)."""
    _chk(U, V, dU, dV, M, Cu, Cv, Du, Dv, wW, wN, wE, wS)
    nrows, ncols, _ = _dims(U)
    capi.call("pdeip_oflow_sor_llin4_dev", _stream(), *_p(U, V, dU, dV, M, Cu, Cv, Du, Dv, wW, wN, wE, wS), nrows,
              ncols, int(iter), float(omega), int(mode), int(col0))


def disp_sor_llin4(U, dU, Cu, Du, wW, wN, wE, wS, iter, omega, mode=capi.MODE_EXACT_ORDER, col0=0):
    """In place on dU (disparitySolvers.c:41)."""
    _chk(U, dU, Cu, Du, wW, wN, wE, wS)
    nrows, ncols, _ = _dims(U)
    capi.call("pdeip_disp_sor_llin4_dev", _stream(), *_p(U, dU, Cu, Du, wW, wN, wE, wS), nrows, ncols, int(iter),
              float(omega), int(mode), int(col0))


def pde_sor4(X, TRACE, B, wW, wN, wE, wS, iter, omega, mode=capi.MODE_EXACT_ORDER, col0=0):
    """In place on X (GS_SOR_4_2d, pdeSolvers.c:44)."""
    _chk(X, TRACE, B, wW, wN, wE, wS)
    nrows, ncols, F = _dims(X)
    capi.call("pdeip_pde_sor4_dev", _stream(), *_p(X, TRACE, B, wW, wN, wE, wS), nrows, ncols, F, int(iter),
              float(omega), int(mode), int(col0))


def pde_sor8(X, TRACE, B, wW, wNW, wN, wNE, wE, wSE, wS, wSW, iter, omega, mode=capi.MODE_EXACT_ORDER, col0=0):
    """In place on X (GS_SOR_8_2d, pdeSolvers.c:153)."""
    _chk(X, TRACE, B, wW, wNW, wN, wNE, wE, wSE, wS, wSW)
    nrows, ncols, F = _dims(X)
    capi.call("pdeip_pde_sor8_dev", _stream(), *_p(X, TRACE, B, wW, wNW, wN, wNE, wE, wSE, wS, wSW), nrows, ncols, F,
              int(iter), float(omega), int(mode), int(col0))


# ---- alternating line relaxation (solver 2); mode EXACT_ORDER = reference line order, RED_BLACK = zebra ----

def oflow_alr_elin4(U, V, M, Cu, Cv, Du, Dv, wW, wN, wE, wS, iter, omega, mode=capi.MODE_EXACT_ORDER):
    """In place on U, V (GS_ALR_SOR_elin4_2d, opticalflowSolvers.c:196)."""
    _chk(U, V, M, Cu, Cv, Du, Dv, wW, wN, wE, wS)
    nrows, ncols, _ = _dims(U)
    capi.call("pdeip_oflow_alr_elin4_dev", _stream(), *_p(U, V, M, Cu, Cv, Du, Dv, wW, wN, wE, wS), nrows, ncols,
              int(iter), float(omega), int(mode))


def oflow_alr_llin4(U, V, dU, dV, M, Cu, Cv, Du, Dv, wW, wN, wE, wS, iter, omega, mode=capi.MODE_EXACT_ORDER):
    """In place on dU, dV (GS_ALR_SOR_llin4_2d, opticalflowSolvers.c:690)."""
    _chk(U, V, dU, dV, M, Cu, Cv, Du, Dv, wW, wN, wE, wS)
    nrows, ncols, _ = _dims(U)
    capi.call("pdeip_oflow_alr_llin4_dev", _stream(), *_p(U, V, dU, dV, M, Cu, Cv, Du, Dv, wW, wN, wE, wS), nrows,
              ncols, int(iter), float(omega), int(mode))


def oflow_alr_llin8(U, V, dU, dV, M, Cu, Cv, Du, Dv, wW, wNW, wN, wNE, wE, wSE, wS, wSW, iter, omega,
                    mode=capi.MODE_EXACT_ORDER):
    """In place on dU, dV (GS_ALR_SOR_llin8_2d, opticalflowSolvers.c:1677)."""
    _chk(U, V, dU, dV, M, Cu, Cv, Du, Dv, wW, wNW, wN, wNE, wE, wSE, wS, wSW)
    nrows, ncols, _ = _dims(U)
    capi.call("pdeip_oflow_alr_llin8_dev", _stream(), *_p(U, V, dU, dV, M, Cu, Cv, Du, Dv, wW, wNW, wN, wNE, wE, wSE, wS, wSW),
              nrows, ncols, int(iter), float(omega), int(mode))


def disp_alr_llin4(U, dU, Cu, Du, wW, wN, wE, wS, iter, omega, mode=capi.MODE_EXACT_ORDER):
    """In place on dU (disparitySolvers.c:154)."""
    _chk(U, dU, Cu, Du, wW, wN, wE, wS)
    nrows, ncols, _ = _dims(U)
    capi.call("pdeip_disp_alr_llin4_dev", _stream(), *_p(U, dU, Cu, Du, wW, wN, wE, wS), nrows, ncols, int(iter),
              float(omega), int(mode))


def pde_alr4(X, TRACE, B, wW, wN, wE, wS, iter, omega, mode=capi.MODE_EXACT_ORDER):
    """In place on X (GS_ALR_SOR_4_2d, pdeSolvers.c:277)."""
    _chk(X, TRACE, B, wW, wN, wE, wS)
    nrows, ncols, F = _dims(X)
    capi.call("pdeip_pde_alr4_dev", _stream(), *_p(X, TRACE, B, wW, wN, wE, wS), nrows, ncols, F, int(iter),
              float(omega), int(mode))


def pde_alr8(X, TRACE, B, wW, wNW, wN, wNE, wE, wSE, wS, wSW, iter, omega, mode=capi.MODE_EXACT_ORDER):
    """In place on X; one iteration whatever `iter` is (GS_ALR_SOR_8_2d, pdeSolvers.c:344)."""
    _chk(X, TRACE, B, wW, wNW, wN, wNE, wE, wSE, wS, wSW)
    nrows, ncols, F = _dims(X)
    capi.call("pdeip_pde_alr8_dev", _stream(), *_p(X, TRACE, B, wW, wNW, wN, wNE, wE, wSE, wS, wSW), nrows, ncols, F,
              int(iter), float(omega), int(mode))


def oflow_res_elin4(RU, RV, U, V, M, Cu, Cv, Du, Dv, wW, wN, wE, wS):
    _chk(RU, RV, U, V, M, Cu, Cv, Du, Dv, wW, wN, wE, wS)
    nrows, ncols, _ = _dims(U)
    capi.call("pdeip_oflow_res_elin4_dev", _stream(), *_p(RU, RV, U, V, M, Cu, Cv, Du, Dv, wW, wN, wE, wS), nrows,
              ncols, _dims(M)[2])


def oflow_lhs_elin4(AU, AV, U, V, M, Du, Dv, wW, wN, wE, wS):
    _chk(AU, AV, U, V, M, Du, Dv, wW, wN, wE, wS)
    nrows, ncols, _ = _dims(U)
    capi.call("pdeip_oflow_lhs_elin4_dev", _stream(), *_p(AU, AV, U, V, M, Du, Dv, wW, wN, wE, wS), nrows, ncols,
              _dims(M)[2])


def diffweights6(D, eps, wW, wN, wE, wS):
    _chk(D, wW, wN, wE, wS)
    nrows, ncols, F = _dims(D)
    capi.call("pdeip_diffweights6_dev", _stream(), D.data_ptr(), nrows, ncols, F, float(eps), *_p(wW, wN, wE, wS))


def warp_bilinear(Iin, X, Y, Iout):
    _chk(Iin, X, Y, Iout)
    nrows, ncols, F = _dims(Iin)
    capi.call("pdeip_warp_bilinear_dev", _stream(), *_p(Iin, X, Y), nrows, ncols, F, Iout.data_ptr())


def fst_derivatives5(It0, It1, Idt, Idx, Idy):
    _chk(It0, It1, Idt, Idx, Idy)
    nrows, ncols, F = _dims(It0)
    capi.call("pdeip_fst_derivatives5_dev", _stream(), *_p(It0, It1), nrows, ncols, F, *_p(Idt, Idx, Idy))


def snd_derivatives5(It0, It1, Idxt, Idyt, Idxx, Idyy, Idxy):
    _chk(It0, It1, Idxt, Idyt, Idxx, Idyy, Idxy)
    nrows, ncols, F = _dims(It0)
    capi.call("pdeip_snd_derivatives5_dev", _stream(), *_p(It0, It1), nrows, ncols, F, *_p(Idxt, Idyt, Idxx, Idyy, Idxy))


# ---- MATLAB-side stages of a late-linearisation pyramid level (csrc/pdeip_flow.hpp) ----------------------

def flow_coords(U, V, X, Y):
    _chk(U, V, X, Y)
    nrows, ncols, _ = _dims(U)
    capi.call("pdeip_flow_coords_dev", _stream(), *_p(U, V), nrows, ncols, *_p(X, Y))


def flow_warp(U, V, I1, W1, I2=None, W2=None):
    """W1 = warp(I1) and optionally W2 = warp(I2) at (X+U, Y+V) in one launch (flow_coords + warp_bilinear x 2).
    V = None: the disparity drivers' warp along x only (Y = the row grid itself)."""
    _chk(U, I1, W1)
    v = None
    if V is not None:
        _chk(V)
        v = V.data_ptr()
    nrows, ncols, C1 = _dims(I1)
    if I2 is None:
        capi.call("pdeip_flow_warp_dev", _stream(), U.data_ptr(), v, I1.data_ptr(), C1, None, 0, nrows, ncols, W1.data_ptr(), None)
    else:
        _chk(I2, W2)
        capi.call("pdeip_flow_warp_dev", _stream(), U.data_ptr(), v, I1.data_ptr(), C1, I2.data_ptr(), _dims(I2)[2], nrows, ncols, *_p(W1, W2))


def flow_assemble(term1, term2, dU, dV, alpha, MGd, CuGd, CvGd, DuGd, DvGd):
    """term = (It, Ix, Iy, b) with [C, ncols, nrows] derivative arrays; term2 may be None, or the gradient-magnitude
    term (Ixt, Iyt, Ixx, Iyy, Ixy, b) built from snd_derivatives5."""
    It1, Ix1, Iy1, b1 = term1
    _chk(It1, Ix1, Iy1, dU, dV, MGd, CuGd, CvGd, DuGd, DvGd)
    nrows, ncols, C1 = _dims(It1)
    if term2 is not None and len(term2) == 6:
        _chk(*term2[:5])
        capi.call("pdeip_flow_assemble_gradmag_dev", _stream(), *_p(It1, Ix1, Iy1), C1, float(b1), *_p(*term2[:5]), _dims(term2[0])[2],
                  float(term2[5]), *_p(dU, dV), float(alpha), nrows, ncols, *_p(MGd, CuGd, CvGd, DuGd, DvGd))
        return
    if term2 is None:
        p2, C2, b2 = [None, None, None], 0, 0.0
    else:
        It2, Ix2, Iy2, b2 = term2
        _chk(It2, Ix2, Iy2)
        p2, C2 = _p(It2, Ix2, Iy2), _dims(It2)[2]
    capi.call("pdeip_flow_assemble_dev", _stream(), *_p(It1, Ix1, Iy1), C1, float(b1), *p2, C2, float(b2), *_p(dU, dV),
              float(alpha), nrows, ncols, *_p(MGd, CuGd, CvGd, DuGd, DvGd))


def flow_assemble_weights(term1, term2, U, V, dU, dV, alpha, MGd, CuGd, CvGd, DuGd, DvGd, wW, wN, wS, wE):
    """flow_assemble + flow_opdiffweights(U, V, dU, dV) in one launch (both only read the iterate)."""
    It1, Ix1, Iy1, b1 = term1
    _chk(It1, Ix1, Iy1, U, V, dU, dV, MGd, CuGd, CvGd, DuGd, DvGd, wW, wN, wS, wE)
    nrows, ncols, C1 = _dims(It1)
    if term2 is None:
        p2, C2, b2 = [None] * 5, 0, 0.0
    elif len(term2) == 6:
        _chk(*term2[:5])
        p2, C2, b2 = _p(*term2[:5]), _dims(term2[0])[2], term2[5]
    else:
        _chk(*term2[:3])
        p2, C2, b2 = _p(*term2[:3]) + [None, None], _dims(term2[0])[2], term2[3]
    capi.call("pdeip_flow_assemble_weights_dev", _stream(), *_p(It1, Ix1, Iy1), C1, float(b1), *p2, C2, float(b2), *_p(U, V, dU, dV),
              float(alpha), nrows, ncols, *_p(MGd, CuGd, CvGd, DuGd, DvGd, wW, wN, wS, wE))


def flow_opdiffweights(U, V, dU, dV, wW, wN, wS, wE):
    """dU = dV = None: OPdiffWeights(U, V)."""
    _chk(U, V, wW, wN, wS, wE) if dU is None else _chk(U, V, dU, dV, wW, wN, wS, wE)
    nrows, ncols, _ = _dims(U)
    d = [None, None] if dU is None else _p(dU, dV)
    capi.call("pdeip_flow_opdiffweights_dev", _stream(), *_p(U, V), *d, nrows, ncols, *_p(wW, wN, wS, wE))


def median3(A, B, out):
    """out = medfilt2(A + B, [3 3], 'symmetric'); B may be None."""
    _chk(A, out) if B is None else _chk(A, B, out)
    nrows, ncols, _ = _dims(A)
    capi.call("pdeip_median3_dev", _stream(), A.data_ptr(), None if B is None else B.data_ptr(), nrows, ncols, out.data_ptr())


def median3_pair(A0, B0, out0, A1, B1, out1):
    """out0 = medfilt2(A0 + B0), out1 = medfilt2(A1 + B1) in one launch."""
    _chk(A0, B0, out0, A1, B1, out1)
    nrows, ncols, _ = _dims(A0)
    capi.call("pdeip_median3_pair_dev", _stream(), *_p(A0, B0, A1, B1), nrows, ncols, *_p(out0, out1))


def disp_assemble(term1, term2, dU, alpha, CuGd, DuGd):
    """term = (It, Ix, b); term2 may be None (DispEminND_llin_2D.m:258-293) or the gradient-magnitude term (Ixt, Iyt, Ixx, Ixy, b)."""
    It1, Ix1, b1 = term1
    _chk(It1, Ix1, dU, CuGd, DuGd)
    nrows, ncols, C1 = _dims(It1)
    if term2 is not None and len(term2) == 5:
        _chk(*term2[:4])
        capi.call("pdeip_disp_assemble_gradmag_dev", _stream(), *_p(It1, Ix1), C1, float(b1), *_p(*term2[:4]), _dims(term2[0])[2],
                  float(term2[4]), dU.data_ptr(), float(alpha), nrows, ncols, *_p(CuGd, DuGd))
        return
    if term2 is None:
        p2, C2, b2 = [None, None], 0, 0.0
    else:
        It2, Ix2, b2 = term2
        _chk(It2, Ix2)
        p2, C2 = _p(It2, Ix2), _dims(It2)[2]
    capi.call("pdeip_disp_assemble_dev", _stream(), *_p(It1, Ix1), C1, float(b1), *p2, C2, float(b2), dU.data_ptr(), float(alpha),
              nrows, ncols, *_p(CuGd, DuGd))


def add(A, B, out):
    _chk(A, B, out)
    nrows, ncols, _ = _dims(A)
    capi.call("pdeip_add_dev", _stream(), *_p(A, B), nrows, ncols, out.data_ptr())


def tv_assemble(Iout, Iin, alpha, TRACE, B, w8):
    """TVdenoise8.m:80-86: ADdiffWeights(Iout), PsiData, TRACE, B; w8 = [aW, aNW, aN, aNE, aE, aSE, aS, aSW] (alpha-scaled)."""
    _chk(Iout, Iin, TRACE, B, *w8)
    nrows, ncols, F = _dims(Iout)
    capi.call("pdeip_tv_assemble_dev", _stream(), *_p(Iout, Iin), nrows, ncols, F, float(alpha), *_p(TRACE, B, *w8))


def hs_assemble(It0, It1, b1, b2, MGd, CuGd, CvGd, DuGd, DvGd):
    """FlowEminHS_elin_2D_v10.m:133-164: data terms of one scale from the frames [C, ncols, nrows] (or [ncols, nrows])."""
    _chk(It0, It1, MGd, CuGd, CvGd, DuGd, DvGd)
    nrows, ncols, C = _dims(It0)
    capi.call("pdeip_hs_assemble_dev", _stream(), *_p(It0, It1), C, float(b1), float(b2), nrows, ncols, *_p(MGd, CuGd, CvGd, DuGd, DvGd))


# ---- stages of the FAS full-multigrid flow driver (csrc/pdeip_fas.hpp) ------------------------------------

def _half(n):
    return (n + 1) // 2


def fas_gauss5(I, G):
    """imfilter(I, G, 'replicate', 'conv'); G: 5x5 numpy kernel."""
    import numpy as np
    _chk(I)
    nrows, ncols, F = _dims(I)
    g = np.asfortranarray(G, dtype=np.float32)
    out = torch.empty_like(I)
    capi.call("pdeip_fas_gauss5_dev", _stream(), I.data_ptr(), nrows, ncols, F, g.ctypes.data, out.data_ptr())
    return out


def fas_down(I):
    """[C, ncols, nrows] -> [C, ceil(ncols/2), ceil(nrows/2)] (lpf twice, 1:2:end)"""
    _chk(I)
    nrows, ncols, F = _dims(I)
    out = torch.empty(I.shape[:-2] + (_half(ncols), _half(nrows)), dtype=I.dtype, device=I.device)
    capi.call("pdeip_fas_down_dev", _stream(), I.data_ptr(), nrows, ncols, F, out.data_ptr())
    return out


def fas_prepare(It0, It1, b1, b2):
    """-> planes [13, C, ncols, nrows]: Idt, Idx, Idy, Idxx, Idyy, Idxy, Idxt, Idyt, M, Cu, Cv, Du, Dv"""
    _chk(It0, It1)
    nrows, ncols, F = _dims(It0)
    planes = torch.empty((13, F, ncols, nrows), dtype=It0.dtype, device=It0.device)
    capi.call("pdeip_fas_prepare_dev", _stream(), *_p(It0, It1), nrows, ncols, F, float(b1), float(b2), planes.data_ptr())
    return planes


def fas_assemble(planes, Cu, Cv, U, V, b1, b2, k, per_frame, MGd, CuGd, CvGd, DuGd, DvGd, gd=None):
    _chk(planes, U, V, MGd, DuGd, DvGd)
    nrows, ncols, _ = _dims(U)
    F = planes.shape[1]
    opt = lambda t: None if t is None else (_chk(t), t.data_ptr())[1]
    capi.call("pdeip_fas_assemble_dev", _stream(), planes.data_ptr(), opt(Cu), opt(Cv), *_p(U, V), nrows, ncols, F, float(b1),
              float(b2), float(k), int(bool(per_frame)), MGd.data_ptr(), opt(CuGd), opt(CvGd), DuGd.data_ptr(), DvGd.data_ptr(), opt(gd))


def fas_assemble_weights(planes, Cu, Cv, U, V, b1, b2, k, MGd, CuGd, CvGd, DuGd, DvGd, wW, wN, wS, wE):
    """fas_assemble(per_frame=False) + flow_opdiffweights(U, V, None, None) in one launch."""
    _chk(planes, U, V, MGd, DuGd, DvGd, wW, wN, wS, wE)
    nrows, ncols, _ = _dims(U)
    opt = lambda t: None if t is None else (_chk(t), t.data_ptr())[1]
    capi.call("pdeip_fas_assemble_weights_dev", _stream(), planes.data_ptr(), opt(Cu), opt(Cv), *_p(U, V), nrows, ncols, planes.shape[1],
              float(b1), float(b2), float(k), MGd.data_ptr(), opt(CuGd), opt(CvGd), DuGd.data_ptr(), DvGd.data_ptr(), *_p(wW, wN, wS, wE))


def fas_restrict(A, scale):
    _chk(A)
    nrows, ncols, F = _dims(A)
    out = torch.empty(A.shape[:-2] + (_half(ncols), _half(nrows)), dtype=A.dtype, device=A.device)
    capi.call("pdeip_fas_restrict_dev", _stream(), A.data_ptr(), nrows, ncols, F, float(scale), out.data_ptr())
    return out


def fas_rhs(R, A, gd):
    _chk(R, A, gd)
    nrows, ncols, F = _dims(R)
    out = torch.empty_like(R)
    capi.call("pdeip_fas_rhs_dev", _stream(), *_p(R, A, gd), nrows, ncols, F, out.data_ptr())
    return out


def fas_prolong_add(U, Uc, Ures, inv_scale):
    """In place on U."""
    _chk(U, Uc, Ures)
    nrows, ncols, _ = _dims(U)
    nrows_c, ncols_c, _ = _dims(Uc)
    capi.call("pdeip_fas_prolong_add_dev", _stream(), U.data_ptr(), nrows, ncols, *_p(Uc, Ures), nrows_c, ncols_c, float(inv_scale))


def fas_upscale(U, mul, nrows_out, ncols_out):
    """imresize(U.*mul, [nrows_out ncols_out]) (bicubic, enlarging) -> new [ncols_out, nrows_out] plane"""
    _chk(U)
    nrows, ncols, _ = _dims(U)
    out = torch.empty((ncols_out, nrows_out), dtype=U.dtype, device=U.device)
    capi.call("pdeip_fas_upscale_dev", _stream(), U.data_ptr(), nrows, ncols, float(mul), nrows_out, ncols_out, out.data_ptr())
    return out


def ad_weights(D, quantile, w8):
    """ADdiffWeights(D, quantile) of FlowEminAD_llin_2D_v10.m; w8 = [wW, wNW, wN, wNE, wE, wSE, wS, wSW] planes [ncols, nrows]."""
    _chk(D, *w8)
    nrows, ncols, F = _dims(D)
    capi.call("pdeip_ad_weights_dev", _stream(), D.data_ptr(), nrows, ncols, F, float(quantile), *_p(*w8))


def tv4_assemble(Iout, Iin, alpha, TRACE, B, w4):
    """TVdenoise4.m:84-98: DiffWeights(Iout), PsiData, TRACE, B; w4 = [aW, aN, aE, aS] (alpha-scaled)."""
    _chk(Iout, Iin, TRACE, B, *w4)
    nrows, ncols, F = _dims(Iout)
    capi.call("pdeip_tv4_assemble_dev", _stream(), *_p(Iout, Iin), nrows, ncols, F, float(alpha), *_p(TRACE, B, *w4))


def rgb2grad(I):
    """[C, ncols, nrows] (or [ncols, nrows]) -> [2C, ncols, nrows]: the x / y differences of every frame (fstTerm 'grad')."""
    _chk(I)
    nrows, ncols, F = _dims(I)
    out = torch.empty((2 * F, ncols, nrows), dtype=I.dtype, device=I.device)
    capi.call("pdeip_rgb2grad_dev", _stream(), I.data_ptr(), nrows, ncols, F, out.data_ptr())
    return out


# ---- symmetric stereo driver (csrc/pdeip_sym.hpp); float64 planes are MATLAB doubles ----------------------

def _chk64(*tensors):
    for t in tensors:
        if t.dtype != torch.float64 or not t.is_contiguous() or not t.is_cuda:
            raise capi.PdeipError(capi.PDEIP_ERR_ARG, "expected contiguous float64 CUDA tensors")


def disp_sor_llin_sym4(U0, dU0, Cu0, Du0, w0, U1, dU1, Cu1, Du1, w1, iter, omega, solver, mode=capi.MODE_EXACT_ORDER):
    """In place on dU0, dU1 (Disp_sor_llin_sym4_2d: GS_SOR_llinsym4_2d / GS_ALR_SOR_llinsym4_2d); w = [wW, wN, wE, wS]."""
    _chk(U0, dU0, Cu0, Du0, *w0, U1, dU1, Cu1, Du1, *w1)
    nrows, ncols, _ = _dims(U0)
    capi.call("pdeip_disp_sor_llin_sym4_dev", _stream(), *_p(U0, dU0, Cu0, Du0, *w0, U1, dU1, Cu1, Du1, *w1), nrows, ncols, int(iter),
              float(omega), int(solver), int(mode), 0)


def sym_warp_flow(U, Uq):
    """interp2(X, Y, U, X+Uq, Y) -> float64 plane"""
    _chk(U, Uq)
    nrows, ncols, _ = _dims(U)
    out = torch.empty(U.shape, dtype=torch.float64, device=U.device)
    capi.call("pdeip_sym_warp_flow_dev", _stream(), *_p(U, Uq), nrows, ncols, out.data_ptr())
    return out


def sym_flow_terms(U, Uw):
    """-> (Udt, Udx, CuS, DuS) float64 planes"""
    _chk(U)
    _chk64(Uw)
    nrows, ncols, _ = _dims(U)
    outs = [torch.empty_like(Uw) for _ in range(4)]
    capi.call("pdeip_sym_flow_terms_dev", _stream(), U.data_ptr(), Uw.data_ptr(), nrows, ncols, *_p(*outs))
    return outs


def sym_assemble(d, sym, dU, b1, b2, alpha, kS, sr2, first, CuG, DuG):
    """d = (Idt, Idx, Idxt, Idyt, Idxx, Idxy) [C, ncols, nrows]; sym = (Udt, Udx, CuS, DuS) float64."""
    _chk(*d, dU, CuG, DuG)
    _chk64(*sym)
    nrows, ncols, C = _dims(d[0])
    capi.call("pdeip_sym_assemble_dev", _stream(), *_p(*d), C, *_p(*sym), dU.data_ptr(), float(b1), float(b2), float(alpha), float(kS),
              float(sr2), int(bool(first)), nrows, ncols, *_p(CuG, DuG))


def flow_apriori(Us, U, dU, gammaS, alpha, as_diff, u_double, du_double, CGd, DGd):
    """Adds the spatial a-priori slice (FlowEminND_llin_2D_v10.m:301-325) to CGd / DGd in place; Us float64."""
    _chk(U, dU, CGd, DGd)
    _chk64(Us)
    nrows, ncols, _ = _dims(U)
    capi.call("pdeip_flow_apriori_dev", _stream(), Us.data_ptr(), U.data_ptr(), dU.data_ptr(), float(gammaS), float(alpha), float(as_diff),
              int(bool(u_double)), int(bool(du_double)), nrows, ncols, CGd.data_ptr(), DGd.data_ptr())


def disp_apriori(Us, U, dU, gammaS, alpha, as_diff, u_double, du_double, CGd, DGd):
    """Adds the disparity driver's spatial a-priori slice (DispEminND_llin_2D.m:277-292, exp influence function) to CGd / DGd."""
    _chk(U, dU, CGd, DGd)
    _chk64(Us)
    nrows, ncols, _ = _dims(U)
    capi.call("pdeip_disp_apriori_dev", _stream(), Us.data_ptr(), U.data_ptr(), dU.data_ptr(), float(gammaS), float(alpha), float(as_diff),
              int(bool(u_double)), int(bool(du_double)), nrows, ncols, CGd.data_ptr(), DGd.data_ptr())


# ---- the drivers' image pyramid on the device (csrc/pdeip_pyr.hpp; definitions in pyramid.py) -------------

def pyr_resize(I, nrows_out, ncols_out, method="bilinear"):
    """pyramid.resize on the device: [(C,) ncols, nrows] -> [(C,) ncols_out, nrows_out]"""
    _chk(I)
    nrows, ncols, F = _dims(I)
    out = torch.empty(I.shape[:-2] + (ncols_out, nrows_out), dtype=I.dtype, device=I.device)
    capi.call("pdeip_pyr_resize_dev", _stream(), I.data_ptr(), nrows, ncols, F, int(nrows_out), int(ncols_out), int(method == "bicubic"),
              out.data_ptr())
    return out


def pyr_smooth(I, G):
    """pyramid.smooth on the device; G: odd square numpy mask (float64)"""
    _chk(I)
    nrows, ncols, F = _dims(I)
    g = np.ascontiguousarray(G, dtype=np.float64)
    out = torch.empty_like(I)
    capi.call("pdeip_pyr_smooth_dev", _stream(), I.data_ptr(), nrows, ncols, F, g.ctypes.data, int(g.shape[0]), out.data_ptr())
    return out
