"""Host-side mirror of the reference's MEX gateways (mex/source/*.c) over libpdeip.so.

Each function has the gateway's name, argument order and meaning; arrays are numpy float32
with MATLAB's shape convention ([nrows, ncols] or [nrows, ncols, nframes]) and are handed to
the C-ABI in column-major (Fortran) order, which is exactly the memory a MATLAB `single` array
has.  `nargout` plays the role of MATLAB's nlhs.  Errors the gateways raise through
mexErrMsgTxt (wrong type, too few outputs, unknown solver) surface as MexError with the same
kind of message.

All arithmetic happens in libpdeip.so on the GPU; this file only checks, packs and unpacks.

Reference: Oflow_sor_elin4_2d.c, Oflow_sor_llin4_2d.c, Oflow_sor_llin8_2d.c, Oflow_lhs_elin4_2d.c,
Oflow_lhs_llin4_2d.c, Disp_sor_llin4_2d.c, PDEsolver4.c, PDEsolver8.c, DdiffWeights.c,
BilinInterp_2d.c, FstDerivatives5.c, SndDerivatives5.c (all under mex/source/).
"""
import numpy as np

from . import capi


class MexError(capi.PdeipError):
    """What MATLAB would report after mexErrMsgTxt."""


def _single(name, who, a):
    """mxIsSingle check of the gateways (e.g. Oflow_sor_elin4_2d.c:117-119)."""
    if not isinstance(a, np.ndarray) or a.dtype != np.float32:
        raise MexError(capi.PDEIP_ERR_ARG, "%s: '%s' must be a noncomplex single-valued matrix." % (who, name))
    if a.ndim == 1:
        a = a.reshape(-1, 1)
    if a.ndim not in (2, 3):
        raise MexError(capi.PDEIP_ERR_ARG, "%s: '%s' must have 2 or 3 dimensions." % (who, name))
    return np.asfortranarray(a)


def _scalar(name, who, v):
    """Scalars cross the MEX boundary as 1x1 singles (Oflow_sor_elin4_2d.c:260-283)."""
    a = np.asarray(v)
    if a.dtype != np.float32 or a.size != 1:
        raise MexError(capi.PDEIP_ERR_ARG, "%s: '%s' must be a noncomplex, single-type scalar" % (who, name))
    return float(a.reshape(()))


def _same_plane(who, ref, **arrays):
    for name, a in arrays.items():
        if a.shape[:2] != ref.shape[:2]:
            raise MexError(capi.PDEIP_ERR_ARG, "%s: '%s' is %s but the image is %s" %
                           (who, name, a.shape[:2], ref.shape[:2]))


def _ptr(a):
    return a.ctypes.data if a is not None else None


def _frames(a):
    return a.shape[2] if a.ndim == 3 else 1


def _run(fn, *args):
    try:
        capi.call(fn, *args)
    except capi.PdeipError as exc:
        raise MexError(exc.code, str(exc)) from None


def _out_like(a):
    return np.zeros(a.shape, dtype=np.float32, order="F")  # mxCreateNumericArray zero-fills


def _oflow_sor(who, fn, planes, weights, it, omega, solver, nargout, iterate_names):
    if nargout < 2:
        raise MexError(capi.PDEIP_ERR_ARG, "%s insufficient number of outputs." % who)
    arrs = [_single(n, who, a) for n, a in planes]
    wts = [_single(n, who, a) for n, a in weights]
    it = _scalar("iter", who, it)
    omega = _scalar("omega", who, omega)
    solver = int(_scalar("solver", who, solver))
    named = dict(zip([n for n, _ in planes], arrs))
    ref = named[iterate_names[0]]
    _same_plane(who, ref, **dict(zip([n for n, _ in planes + weights], arrs + wts)))
    M = named["M"]
    F = _frames(M)
    for n in ("Cu", "Cv", "Du", "Dv"):
        if _frames(named[n]) != F:
            raise MexError(capi.PDEIP_ERR_ARG, "%s: '%s' must have as many frames as 'M'" % (who, n))
    o0, o1 = _out_like(named[iterate_names[0]]), _out_like(named[iterate_names[1]])
    RU = RV = None
    if nargout >= 4:  # residual outputs take M's dimensions (Oflow_sor_elin4_2d.c:309-325)
        RU, RV = _out_like(M), _out_like(M)
    nrows, ncols = ref.shape[:2]
    _run(fn, *[_ptr(a) for a in arrs + wts], nrows, ncols, F, int(it), omega, solver,
         _ptr(o0), _ptr(o1), _ptr(RU), _ptr(RV))
    return (o0, o1) if nargout < 4 else (o0, o1, RU, RV)


def Oflow_sor_elin4_2d(U, V, M, Cu, Cv, Du, Dv, wW, wN, wE, wS, iter, omega, solver, nargout=2):
    """[U,V(,RU,RV)] = Oflow_sor_elin4_2d(...)  -- mex/source/Oflow_sor_elin4_2d.c:64-352."""
    return _oflow_sor("Oflow_sor_elin4_2d", "pdeip_oflow_sor_elin4",
                      [("U", U), ("V", V), ("M", M), ("Cu", Cu), ("Cv", Cv), ("Du", Du), ("Dv", Dv)],
                      [("wW", wW), ("wN", wN), ("wE", wE), ("wS", wS)], iter, omega, solver, nargout, ("U", "V"))


def Oflow_sor_llin4_2d(U, V, dU, dV, M, Cu, Cv, Du, Dv, wW, wN, wE, wS, iter, omega, solver, nargout=2):
    """[dU,dV(,RU,RV)] = Oflow_sor_llin4_2d(...)  -- mex/source/Oflow_sor_llin4_2d.c:66-386."""
    return _oflow_sor("Oflow_sor_llin4_2d", "pdeip_oflow_sor_llin4",
                      [("U", U), ("V", V), ("dU", dU), ("dV", dV), ("M", M), ("Cu", Cu), ("Cv", Cv), ("Du", Du),
                       ("Dv", Dv)],
                      [("wW", wW), ("wN", wN), ("wE", wE), ("wS", wS)], iter, omega, solver, nargout, ("dU", "dV"))


def Oflow_sor_llin8_2d(U, V, dU, dV, M, Cu, Cv, Du, Dv, wW, wNW, wN, wNE, wE, wSE, wS, wSW, iter, omega, solver,
                       nargout=2):
    """[dU,dV(,RU,RV)] = Oflow_sor_llin8_2d(...)  -- mex/source/Oflow_sor_llin8_2d.c:71-489."""
    return _oflow_sor("Oflow_sor_llin8_2d", "pdeip_oflow_sor_llin8",
                      [("U", U), ("V", V), ("dU", dU), ("dV", dV), ("M", M), ("Cu", Cu), ("Cv", Cv), ("Du", Du),
                       ("Dv", Dv)],
                      [("wW", wW), ("wNW", wNW), ("wN", wN), ("wNE", wNE), ("wE", wE), ("wSE", wSE), ("wS", wS),
                       ("wSW", wSW)], iter, omega, solver, nargout, ("dU", "dV"))


def _oflow_lhs(who, fn, planes, weights, nargout):
    if nargout < 2:
        raise MexError(capi.PDEIP_ERR_ARG, "%s insufficient number of outputs." % who)
    arrs = [_single(n, who, a) for n, a in planes]
    wts = [_single(n, who, a) for n, a in weights]
    named = dict(zip([n for n, _ in planes], arrs))
    ref = named["U"]
    _same_plane(who, ref, **dict(zip([n for n, _ in planes + weights], arrs + wts)))
    M = named["M"]
    F = _frames(M)
    AU, AV = _out_like(M), _out_like(M)  # outputs take M's dimensions (Oflow_lhs_elin4_2d.c:207-227)
    nrows, ncols = ref.shape[:2]
    _run(fn, *[_ptr(a) for a in arrs + wts], nrows, ncols, F, _ptr(AU), _ptr(AV))
    return AU, AV


def Oflow_lhs_elin4_2d(U, V, M, Du, Dv, wW, wN, wE, wS, nargout=2):
    """[AU,AV] = Oflow_lhs_elin4_2d(...)  -- mex/source/Oflow_lhs_elin4_2d.c:56-231."""
    return _oflow_lhs("Oflow_lhs_elin4_2d", "pdeip_oflow_lhs_elin4",
                      [("U", U), ("V", V), ("M", M), ("Du", Du), ("Dv", Dv)],
                      [("wW", wW), ("wN", wN), ("wE", wE), ("wS", wS)], nargout)


def Oflow_lhs_llin4_2d(U, V, dU, dV, M, Du, Dv, wW, wN, wE, wS, nargout=2):
    """[AU,AV] = Oflow_lhs_llin4_2d(...)  -- mex/source/Oflow_lhs_llin4_2d.c:59-260."""
    return _oflow_lhs("Oflow_lhs_llin4_2d", "pdeip_oflow_lhs_llin4",
                      [("U", U), ("V", V), ("dU", dU), ("dV", dV), ("M", M), ("Du", Du), ("Dv", Dv)],
                      [("wW", wW), ("wN", wN), ("wE", wE), ("wS", wS)], nargout)


def Disp_sor_llin4_2d(U, dU, Cu, Du, wW, wN, wE, wS, iter, omega, solver, nargout=1):
    """[dU(,RU)] = Disp_sor_llin4_2d(...)  -- mex/source/Disp_sor_llin4_2d.c:59-282."""
    who = "Disp_sor_llin4_2d"
    if nargout < 1:
        raise MexError(capi.PDEIP_ERR_ARG, "%s insufficient number of outputs." % who)
    names = ["U", "dU", "Cu", "Du", "wW", "wN", "wE", "wS"]
    arrs = [_single(n, who, a) for n, a in zip(names, [U, dU, Cu, Du, wW, wN, wE, wS])]
    it, omega, solver = _scalar("iter", who, iter), _scalar("omega", who, omega), int(_scalar("solver", who, solver))
    _same_plane(who, arrs[0], **dict(zip(names, arrs)))
    out = _out_like(arrs[1])
    RU = _out_like(arrs[0]) if nargout >= 2 else None
    nrows, ncols = arrs[0].shape[:2]
    _run("pdeip_disp_sor_llin4", *[_ptr(a) for a in arrs], nrows, ncols, int(it), omega, solver, _ptr(out), _ptr(RU))
    return out if nargout < 2 else (out, RU)


def Disp_sor_llin_sym4_2d(U0, dU0, Cu0, Du0, wW0, wN0, wE0, wS0, U1, dU1, Cu1, Du1, wW1, wN1, wE1, wS1, iter, omega, solver,
                          nargout=2):
    """[dU0 dU1] = Disp_sor_llin_sym4_2d(...)  -- mex/source/Disp_sor_llin_sym4_2d.c:82-440."""
    who = "Disp_sor_llin_sym4_2d"
    if nargout < 2:
        raise MexError(capi.PDEIP_ERR_ARG, "%s insufficient number of outputs." % who)
    names = ["U_in0", "dU_in0", "Cu0", "Du0", "wW0", "wN0", "wE0", "wS0", "U_in1", "dU_in1", "Cu1", "Du1", "wW1", "wN1", "wE1", "wS1"]
    arrs = [_single(n, who, a) for n, a in zip(names, [U0, dU0, Cu0, Du0, wW0, wN0, wE0, wS0, U1, dU1, Cu1, Du1, wW1, wN1, wE1, wS1])]
    it, omega, solver = _scalar("iter", who, iter), _scalar("omega", who, omega), int(_scalar("solver", who, solver))
    _same_plane(who, arrs[0], **dict(zip(names, arrs)))
    o0, o1 = _out_like(arrs[1]), _out_like(arrs[9])
    nrows, ncols = arrs[0].shape[:2]
    _run("pdeip_disp_sor_llin_sym4", *[_ptr(a) for a in arrs], nrows, ncols, int(it), omega, solver, _ptr(o0), _ptr(o1))
    return o0, o1


def _pde(who, fn, names, arrays, it, omega, solver, nargout):
    if nargout < 1:
        raise MexError(capi.PDEIP_ERR_ARG, "%s: error insufficient number of outputs." % who)
    arrs = [_single(n, who, a) for n, a in zip(names, arrays)]
    it, omega, solver = _scalar("iter", who, it), _scalar("omega", who, omega), int(_scalar("solver", who, solver))
    X = arrs[0]
    for n, a in zip(names, arrs):
        if a.shape != X.shape:
            raise MexError(capi.PDEIP_ERR_ARG, "%s: '%s' is %s but X is %s" % (who, n, a.shape, X.shape))
    out = _out_like(X)
    nrows, ncols = X.shape[:2]
    _run(fn, *[_ptr(a) for a in arrs], nrows, ncols, _frames(X), int(it), omega, solver, _ptr(out))
    return out


def PDEsolver4(X, TRACE, B, wW, wN, wE, wS, iter, omega, solver, nargout=1):
    """X = PDEsolver4(...)  -- mex/source/PDEsolver4.c:54-249."""
    return _pde("PDEsolver4", "pdeip_pde_sor4", ["X", "TRACE", "B", "wW", "wN", "wE", "wS"],
                [X, TRACE, B, wW, wN, wE, wS], iter, omega, solver, nargout)


def PDEsolver8(X, TRACE, B, wW, wNW, wN, wNE, wE, wSE, wS, wSW, iter, omega, solver, nargout=1):
    """X = PDEsolver8(...)  -- mex/source/PDEsolver8.c:54-309."""
    return _pde("PDEsolver8", "pdeip_pde_sor8",
                ["X", "TRACE", "B", "wW", "wNW", "wN", "wNE", "wE", "wSE", "wS", "wSW"],
                [X, TRACE, B, wW, wNW, wN, wNE, wE, wSE, wS, wSW], iter, omega, solver, nargout)


def DdiffWeights(D, eps, nargout=4):
    """[wW,wN,wE,wS] = DdiffWeights(D,eps)  -- mex/source/DdiffWeights.c:50-140."""
    who = "DdiffWeights"
    if nargout < 4:
        raise MexError(capi.PDEIP_ERR_ARG, "diffusion6_2d error insufficient number of outputs. Outputs from this "
                                            "function are 'wW', 'wN', 'wE' and 'wS'")
    D = _single("D", who, D)
    eps = _scalar("eps", who, eps)
    outs = [_out_like(D) for _ in range(4)]
    nrows, ncols = D.shape[:2]
    _run("pdeip_diffweights6", _ptr(D), nrows, ncols, _frames(D), eps, *[_ptr(o) for o in outs])
    return tuple(outs)


def BilinInterp_2d(Iin, X, Y, nargout=1):
    """Iout = BilinInterp_2d(Iin,X,Y)  -- mex/source/BilinInterp_2d.c:41-124."""
    who = "BilinInterp_2d"
    if nargout < 1:
        raise MexError(capi.PDEIP_ERR_ARG, "insufficient number of outputs. Outputs from this function is 'Iout'")
    Iin, X, Y = _single("Iin", who, Iin), _single("X", who, X), _single("Y", who, Y)
    _same_plane(who, Iin, X=X, Y=Y)
    out = _out_like(Iin)
    nrows, ncols = Iin.shape[:2]
    _run("pdeip_warp_bilinear", _ptr(Iin), _ptr(X), _ptr(Y), nrows, ncols, _frames(Iin), _ptr(out))
    return out


def _derivatives(who, fn, It0, It1, nout, nargout, msg):
    if nargout < nout:
        raise MexError(capi.PDEIP_ERR_ARG, msg)
    It0, It1 = _single("It0", who, It0), _single("It1", who, It1)
    if It0.shape != It1.shape:
        raise MexError(capi.PDEIP_ERR_ARG, "%s: 'It1' is %s but 'It0' is %s" % (who, It1.shape, It0.shape))
    outs = [_out_like(It0) for _ in range(nout)]
    nrows, ncols = It0.shape[:2]
    _run(fn, _ptr(It0), _ptr(It1), nrows, ncols, _frames(It0), *[_ptr(o) for o in outs])
    return tuple(outs)


def FstDerivatives5(It0, It1, nargout=3):
    """[Idt,Idx,Idy] = FstDerivatives5(It0,It1)  -- mex/source/FstDerivatives5.c:50-145."""
    return _derivatives("fstDerivatives", "pdeip_fst_derivatives5", It0, It1, 3, nargout,
                        "fstDerivatives: insufficient number of outputs...outputs from this function are 'Idt', 'Idx' and 'Idy'.")


def SndDerivatives5(It0, It1, nargout=5):
    """[Idxt,Idyt,Idxx,Idyy,Idxy] = SndDerivatives5(It0,It1)  -- mex/source/SndDerivatives5.c:51-174."""
    return _derivatives("sndDerivatives", "pdeip_snd_derivatives5", It0, It1, 5, nargout,
                        "sndDerivatives: insufficient number of outputs.")


def set_mode(mode):
    """PDEIP_MODE_EXACT_ORDER (0, default: the reference's sweep order) or PDEIP_MODE_RED_BLACK (1)."""
    capi.set_mode(mode)
