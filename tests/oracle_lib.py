"""numpy/ctypes access to the CPU oracle (oracle/liboracle.so) for the tests.

TEST INFRASTRUCTURE: the checker, never the thing measured or shipped.  The wrappers named
after the MEX gateways reproduce the gateways' semantics (copy-in, iter<=0 -> zero outputs,
residuals of the INPUT iterate, unfilled residual outputs) on top of the oracle's library-level
functions, so a parity test reads  `mex_api.X(args) == oracle.X(args)`.
"""
import ctypes
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB_PATH = os.path.join(ORACLE_DIR, "liboracle.so")

LEX = 0     # the reference's lexicographic order  (product: PDEIP_MODE_EXACT_ORDER)
COLOUR = 1  # red-black / four-colour order        (product: PDEIP_MODE_RED_BLACK)

_P, _I, _F = ctypes.c_void_p, ctypes.c_int, ctypes.c_float
_SIGS = {
    "orc_oflow_sor_elin4": [_P] * 11 + [_I, _I, _I, _F, _I],
    "orc_oflow_sor_elin4_rb_omp": [_P] * 11 + [_I, _I, _I, _F, _I],
    "orc_plane_copy_omp": [_P, _P, _I, _I, _I],
    "orc_oflow_sor_llin4": [_P] * 13 + [_I, _I, _I, _F, _I],
    "orc_oflow_res_elin4": [_P] * 13 + [_I, _I, _I],
    "orc_oflow_lhs_elin4": [_P] * 11 + [_I, _I, _I],
    "orc_oflow_res_llin4": [_P] * 15 + [_I, _I, _I],
    "orc_oflow_lhs_llin4": [_P] * 13 + [_I, _I, _I],
    "orc_disp_sor_llin4": [_P] * 8 + [_I, _I, _I, _F, _I],
    "orc_disp_res_llin4": [_P] * 9 + [_I, _I],
    "orc_disp_sor_llinsym4": [_P] * 16 + [_I, _I, _I, _F, _I],
    "orc_pde_sor4": [_P] * 7 + [_I, _I, _I, _I, _F, _I],
    "orc_pde_sor8": [_P] * 11 + [_I, _I, _I, _I, _F, _I],
    "orc_diffweights6": [_P] * 5 + [_I, _I, _I, _F],
    "orc_warp_bilinear": [_P] * 4 + [_I, _I, _I],
    "orc_fst_derivatives5": [_P] * 5 + [_I, _I, _I],
    "orc_snd_derivatives5": [_P] * 7 + [_I, _I, _I],
    "orc_oflow_alr_elin4": [_P] * 11 + [_I, _I, _I, _F, _I],
    "orc_oflow_alr_llin4": [_P] * 13 + [_I, _I, _I, _F, _I],
    "orc_oflow_alr_llin8": [_P] * 17 + [_I, _I, _I, _F, _I],
    "orc_disp_alr_llin4": [_P] * 8 + [_I, _I, _I, _F, _I],
    "orc_pde_alr4": [_P] * 7 + [_I, _I, _I, _I, _F, _I],
    "orc_pde_alr8": [_P] * 11 + [_I, _I, _I, _I, _F, _I],
}
_lib = None


def build(force=False):
    src = [os.path.join(ORACLE_DIR, f) for f in ("pdeip_oracle.c", "pdeip_oracle_alr.c", "pdeip_oracle.h", "Makefile")]
    stale = force or not os.path.exists(LIB_PATH) or any(os.path.getmtime(s) > os.path.getmtime(LIB_PATH) for s in src)
    if stale:
        subprocess.run(["make", "-C", ORACLE_DIR, "-B", "liboracle.so"], check=True, stdout=subprocess.DEVNULL)
    return LIB_PATH


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(LIB_PATH)
        for name, sig in _SIGS.items():
            fn = getattr(_lib, name)
            fn.argtypes = sig
            fn.restype = ctypes.c_int if name.endswith("_omp") else None
    return _lib


def F(a):
    """float32, column-major, owned copy."""
    return np.array(a, dtype=np.float32, order="F", copy=True)


def _p(a):
    assert a.dtype == np.float32 and a.flags["F_CONTIGUOUS"], "oracle wants column-major float32"
    return a.ctypes.data


def _frames(a):
    return a.shape[2] if a.ndim == 3 else 1


# ---- library-level functions (in place, like the reference's library) -----------------------------

def oflow_sor_elin4(U, V, M, Cu, Cv, Du, Dv, wW, wN, wE, wS, it, omega, order=LEX):
    U, V = F(U), F(V)
    ins = [F(a) for a in (M, Cu, Cv, Du, Dv, wW, wN, wE, wS)]
    lib().orc_oflow_sor_elin4(_p(U), _p(V), *[_p(a) for a in ins], U.shape[0], U.shape[1], int(it), float(omega), order)
    return U, V


def oflow_sor_elin4_rb_omp(U, V, M, Cu, Cv, Du, Dv, wW, wN, wE, wS, it, omega, nthreads=0):
    """Red-black order on `nthreads` host threads (0 = OpenMP default); returns (U, V, threads used)."""
    U, V = F(U), F(V)
    ins = [F(a) for a in (M, Cu, Cv, Du, Dv, wW, wN, wE, wS)]
    used = lib().orc_oflow_sor_elin4_rb_omp(_p(U), _p(V), *[_p(a) for a in ins], U.shape[0], U.shape[1], int(it), float(omega), int(nthreads))
    return U, V, used


def oflow_sor_llin4(U, V, dU, dV, M, Cu, Cv, Du, Dv, wW, wN, wE, wS, it, omega, order=LEX):
    U, V, dU, dV = F(U), F(V), F(dU), F(dV)
    ins = [F(a) for a in (M, Cu, Cv, Du, Dv, wW, wN, wE, wS)]
    lib().orc_oflow_sor_llin4(_p(U), _p(V), _p(dU), _p(dV), *[_p(a) for a in ins], U.shape[0], U.shape[1], int(it),
                              float(omega), order)
    return dU, dV


def oflow_res_elin4(U, V, M, Cu, Cv, Du, Dv, wW, wN, wE, wS):
    ins = [F(a) for a in (U, V, M, Cu, Cv, Du, Dv, wW, wN, wE, wS)]
    RU, RV = np.zeros_like(ins[2]), np.zeros_like(ins[2])
    lib().orc_oflow_res_elin4(_p(RU), _p(RV), *[_p(a) for a in ins], ins[0].shape[0], ins[0].shape[1], _frames(ins[2]))
    return RU, RV


def oflow_lhs_elin4(U, V, M, Du, Dv, wW, wN, wE, wS):
    ins = [F(a) for a in (U, V, M, Du, Dv, wW, wN, wE, wS)]
    AU, AV = np.zeros_like(ins[2]), np.zeros_like(ins[2])
    lib().orc_oflow_lhs_elin4(_p(AU), _p(AV), *[_p(a) for a in ins], ins[0].shape[0], ins[0].shape[1], _frames(ins[2]))
    return AU, AV


def oflow_res_llin4(U, V, dU, dV, M, Cu, Cv, Du, Dv, wW, wN, wE, wS):
    ins = [F(a) for a in (U, V, dU, dV, M, Cu, Cv, Du, Dv, wW, wN, wE, wS)]
    RU, RV = np.zeros_like(ins[4]), np.zeros_like(ins[4])
    lib().orc_oflow_res_llin4(_p(RU), _p(RV), *[_p(a) for a in ins], ins[0].shape[0], ins[0].shape[1], _frames(ins[4]))
    return RU, RV


def oflow_lhs_llin4(U, V, dU, dV, M, Du, Dv, wW, wN, wE, wS):
    ins = [F(a) for a in (U, V, dU, dV, M, Du, Dv, wW, wN, wE, wS)]
    AU, AV = np.zeros_like(ins[4]), np.zeros_like(ins[4])
    lib().orc_oflow_lhs_llin4(_p(AU), _p(AV), *[_p(a) for a in ins], ins[0].shape[0], ins[0].shape[1], _frames(ins[4]))
    return AU, AV


def disp_sor_llin4(U, dU, Cu, Du, wW, wN, wE, wS, it, omega, order=LEX):
    U, dU = F(U), F(dU)
    ins = [F(a) for a in (Cu, Du, wW, wN, wE, wS)]
    lib().orc_disp_sor_llin4(_p(U), _p(dU), *[_p(a) for a in ins], U.shape[0], U.shape[1], int(it), float(omega), order)
    return dU


def disp_res_llin4(U, dU, Cu, Du, wW, wN, wE, wS):
    ins = [F(a) for a in (U, dU, Cu, Du, wW, wN, wE, wS)]
    RU = np.zeros_like(ins[0])
    lib().orc_disp_res_llin4(_p(RU), *[_p(a) for a in ins], ins[0].shape[0], ins[0].shape[1])
    return RU


def pde_sor4(X, TRACE, B, wW, wN, wE, wS, it, omega, order=LEX):
    X = F(X)
    ins = [F(a) for a in (TRACE, B, wW, wN, wE, wS)]
    lib().orc_pde_sor4(_p(X), *[_p(a) for a in ins], X.shape[0], X.shape[1], _frames(X), int(it), float(omega), order)
    return X


def pde_sor8(X, TRACE, B, wW, wNW, wN, wNE, wE, wSE, wS, wSW, it, omega, order=LEX):
    X = F(X)
    ins = [F(a) for a in (TRACE, B, wW, wNW, wN, wNE, wE, wSE, wS, wSW)]
    lib().orc_pde_sor8(_p(X), *[_p(a) for a in ins], X.shape[0], X.shape[1], _frames(X), int(it), float(omega), order)
    return X


def diffweights6(D, eps):
    D = F(D)
    outs = [np.zeros(D.shape[:2], dtype=np.float32, order="F") for _ in range(4)]
    lib().orc_diffweights6(*[_p(o) for o in outs], _p(D), D.shape[0], D.shape[1], _frames(D), float(eps))
    return outs


def warp_bilinear(Iin, X, Y):
    Iin, X, Y = F(Iin), F(X), F(Y)
    out = np.zeros_like(Iin)
    lib().orc_warp_bilinear(_p(out), _p(Iin), _p(X), _p(Y), Iin.shape[0], Iin.shape[1], _frames(Iin))
    return out


# ---- alternating line relaxation (solver = 2), library level ----------------------------------------

def oflow_alr_elin4(U, V, M, Cu, Cv, Du, Dv, wW, wN, wE, wS, it, omega, order=LEX):
    U, V = F(U), F(V)
    ins = [F(a) for a in (M, Cu, Cv, Du, Dv, wW, wN, wE, wS)]
    lib().orc_oflow_alr_elin4(_p(U), _p(V), *[_p(a) for a in ins], U.shape[0], U.shape[1], int(it), float(omega), order)
    return U, V


def oflow_alr_llin4(U, V, dU, dV, M, Cu, Cv, Du, Dv, wW, wN, wE, wS, it, omega, order=LEX):
    U, V, dU, dV = F(U), F(V), F(dU), F(dV)
    ins = [F(a) for a in (M, Cu, Cv, Du, Dv, wW, wN, wE, wS)]
    lib().orc_oflow_alr_llin4(_p(U), _p(V), _p(dU), _p(dV), *[_p(a) for a in ins], U.shape[0], U.shape[1], int(it),
                              float(omega), order)
    return dU, dV


def oflow_alr_llin8(U, V, dU, dV, M, Cu, Cv, Du, Dv, wW, wNW, wN, wNE, wE, wSE, wS, wSW, it, omega, order=LEX):
    U, V, dU, dV = F(U), F(V), F(dU), F(dV)
    ins = [F(a) for a in (M, Cu, Cv, Du, Dv, wW, wNW, wN, wNE, wE, wSE, wS, wSW)]
    lib().orc_oflow_alr_llin8(_p(U), _p(V), _p(dU), _p(dV), *[_p(a) for a in ins], U.shape[0], U.shape[1], int(it),
                              float(omega), order)
    return dU, dV


def disp_alr_llin4(U, dU, Cu, Du, wW, wN, wE, wS, it, omega, order=LEX):
    U, dU = F(U), F(dU)
    ins = [F(a) for a in (Cu, Du, wW, wN, wE, wS)]
    lib().orc_disp_alr_llin4(_p(U), _p(dU), *[_p(a) for a in ins], U.shape[0], U.shape[1], int(it), float(omega), order)
    return dU


def pde_alr4(X, TRACE, B, wW, wN, wE, wS, it, omega, order=LEX):
    X = F(X)
    ins = [F(a) for a in (TRACE, B, wW, wN, wE, wS)]
    lib().orc_pde_alr4(_p(X), *[_p(a) for a in ins], X.shape[0], X.shape[1], _frames(X), int(it), float(omega), order)
    return X


def pde_alr8(X, TRACE, B, wW, wNW, wN, wNE, wE, wSE, wS, wSW, it, omega, order=LEX):
    X = F(X)
    ins = [F(a) for a in (TRACE, B, wW, wNW, wN, wNE, wE, wSE, wS, wSW)]
    lib().orc_pde_alr8(_p(X), *[_p(a) for a in ins], X.shape[0], X.shape[1], _frames(X), int(it), float(omega), order)
    return X


# ---- gateway-level wrappers (mexFunction semantics) ---------------------------------------------------

def Oflow_sor_elin4_2d(U, V, M, Cu, Cv, Du, Dv, wW, wN, wE, wS, it, omega, solver=1, nargout=2, order=LEX):
    assert solver in (1, 2)
    if it > 0:  # Oflow_sor_elin4_2d.c:341-346
        M0, Cu0, Cv0, Du0, Dv0 = [a[..., 0] if np.ndim(a) == 3 else a for a in (M, Cu, Cv, Du, Dv)]
        fn = oflow_sor_elin4 if solver == 1 else oflow_alr_elin4
        Uo, Vo = fn(U, V, M0, Cu0, Cv0, Du0, Dv0, wW, wN, wE, wS, it, omega, order)
    else:
        Uo, Vo = np.zeros_like(F(U)), np.zeros_like(F(V))
    if nargout >= 4:  # residuals of the input iterate (:349-350)
        return (Uo, Vo) + tuple(oflow_res_elin4(U, V, M, Cu, Cv, Du, Dv, wW, wN, wE, wS))
    return Uo, Vo


def Oflow_sor_llin4_2d(U, V, dU, dV, M, Cu, Cv, Du, Dv, wW, wN, wE, wS, it, omega, solver=1, nargout=2, order=LEX,
                       fill_residuals=True):
    assert solver in (1, 2)
    if it > 0:  # Oflow_sor_llin4_2d.c:376-381
        M0, Cu0, Cv0, Du0, Dv0 = [a[..., 0] if np.ndim(a) == 3 else a for a in (M, Cu, Cv, Du, Dv)]
        fn = oflow_sor_llin4 if solver == 1 else oflow_alr_llin4
        o0, o1 = fn(U, V, dU, dV, M0, Cu0, Cv0, Du0, Dv0, wW, wN, wE, wS, it, omega, order)
    else:
        o0, o1 = np.zeros_like(F(dU)), np.zeros_like(F(dV))
    if nargout >= 4:
        if fill_residuals:
            return (o0, o1) + tuple(oflow_res_llin4(U, V, dU, dV, M, Cu, Cv, Du, Dv, wW, wN, wE, wS))
        return o0, o1, np.zeros_like(F(M)), np.zeros_like(F(M))
    return o0, o1


def Oflow_sor_llin8_2d(U, V, dU, dV, M, Cu, Cv, Du, Dv, wW, wNW, wN, wNE, wE, wSE, wS, wSW, it, omega, solver=1,
                       nargout=2, order=LEX):
    # residual outputs never filled (Oflow_sor_llin8_2d.c:466-488)
    if solver == 2:  # the line solvers are the only place the diagonal weights act
        if it > 0:
            M0, Cu0, Cv0, Du0, Dv0 = [a[..., 0] if np.ndim(a) == 3 else a for a in (M, Cu, Cv, Du, Dv)]
            o0, o1 = oflow_alr_llin8(U, V, dU, dV, M0, Cu0, Cv0, Du0, Dv0, wW, wNW, wN, wNE, wE, wSE, wS, wSW, it, omega, order)
        else:
            o0, o1 = np.zeros_like(F(dU)), np.zeros_like(F(dV))
        return (o0, o1) if nargout < 4 else (o0, o1, np.zeros_like(F(M)), np.zeros_like(F(M)))
    # diagonal weights unused by the point solver
    return Oflow_sor_llin4_2d(U, V, dU, dV, M, Cu, Cv, Du, Dv, wW, wN, wE, wS, it, omega, solver, nargout, order,
                              fill_residuals=False)


def Disp_sor_llin4_2d(U, dU, Cu, Du, wW, wN, wE, wS, it, omega, solver=1, nargout=1, order=LEX):
    assert solver in (1, 2)
    fn = disp_sor_llin4 if solver == 1 else disp_alr_llin4
    out = fn(U, dU, Cu, Du, wW, wN, wE, wS, it, omega, order) if it > 0 else np.zeros_like(F(dU))
    return out if nargout < 2 else (out, np.zeros_like(F(U)))  # RU allocated, never computed


def Disp_sor_llin_sym4_2d(U0, dU0, Cu0, Du0, wW0, wN0, wE0, wS0, U1, dU1, Cu1, Du1, wW1, wN1, wE1, wS1, it, omega, solver=1, nargout=2,
                          order=LEX):
    """Gateway semantics of Disp_sor_llin_sym4_2d.c: copy-in, solve unconditionally."""
    assert solver in (1, 2)
    a = [F(x) for x in (U0, dU0, Cu0, Du0, wW0, wN0, wE0, wS0, U1, dU1, Cu1, Du1, wW1, wN1, wE1, wS1)]
    if solver == 1:
        lib().orc_disp_sor_llinsym4(*[_p(x) for x in a], a[0].shape[0], a[0].shape[1], max(int(it), 0), float(omega), order)
        return a[1], a[9]
    # GS_ALR_SOR_llinsym4_2d (disparitySolvers.c:503-540): per iteration the plain line solvers on field 0, then on field 1;
    # the fields never read each other, so all iterations of one field first is the same computation
    o0 = disp_alr_llin4(a[0], a[1], a[2], a[3], a[4], a[5], a[6], a[7], max(int(it), 0), omega, order)
    o1 = disp_alr_llin4(a[8], a[9], a[10], a[11], a[12], a[13], a[14], a[15], max(int(it), 0), omega, order)
    return o0, o1


def PDEsolver4(X, TRACE, B, wW, wN, wE, wS, it, omega, solver=1, order=LEX):
    assert solver in (1, 2)
    fn = pde_sor4 if solver == 1 else pde_alr4
    return fn(X, TRACE, B, wW, wN, wE, wS, max(int(it), 0), omega, order)


def PDEsolver8(X, TRACE, B, wW, wNW, wN, wNE, wE, wSE, wS, wSW, it, omega, solver=1, order=LEX):
    assert solver in (1, 2)
    fn = pde_sor8 if solver == 1 else pde_alr8  # ALR-8 runs exactly one iteration whatever `it` is (pdeSolvers.c:362)
    return fn(X, TRACE, B, wW, wNW, wN, wNE, wE, wSE, wS, wSW, max(int(it), 0), omega, order)


def DdiffWeights(D, eps):
    D = F(D)
    outs = []
    for w in diffweights6(D, eps):
        o = np.zeros(D.shape, dtype=np.float32, order="F")  # outputs carry D's dims; frame 0 only
        if D.ndim == 3:
            o[:, :, 0] = w
        else:
            o[...] = w
        outs.append(o)
    return tuple(outs)


def BilinInterp_2d(Iin, X, Y):
    return warp_bilinear(Iin, X, Y)


def Oflow_lhs_elin4_2d(U, V, M, Du, Dv, wW, wN, wE, wS):
    return oflow_lhs_elin4(U, V, M, Du, Dv, wW, wN, wE, wS)


def Oflow_lhs_llin4_2d(U, V, dU, dV, M, Du, Dv, wW, wN, wE, wS):
    return oflow_lhs_llin4(U, V, dU, dV, M, Du, Dv, wW, wN, wE, wS)


def FstDerivatives5(It0, It1):
    """[Idt, Idx, Idy] = FstDerivatives5(It0, It1)  (FstDerivatives5.c + fstSimoncelli_c)"""
    It0, It1 = F(It0), F(It1)
    outs = [np.zeros_like(It0) for _ in range(3)]
    lib().orc_fst_derivatives5(*[_p(o) for o in outs], _p(It0), _p(It1), It0.shape[0], It0.shape[1], _frames(It0))
    return tuple(outs)


def SndDerivatives5(It0, It1):
    """[Idxt, Idyt, Idxx, Idyy, Idxy] = SndDerivatives5(It0, It1)  (SndDerivatives5.c + sndSimoncelli_c)"""
    It0, It1 = F(It0), F(It1)
    outs = [np.zeros_like(It0) for _ in range(5)]
    lib().orc_snd_derivatives5(*[_p(o) for o in outs], _p(It0), _p(It1), It0.shape[0], It0.shape[1], _frames(It0))
    return tuple(outs)
