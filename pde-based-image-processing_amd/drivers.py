"""The MATLAB drivers as Python functions: same names, arguments and default parameters, every pyramid level resident on the
device (flow_level.py, fas.py) and so is the pyramid around them (pyramid.py states our definitions of imresize / imfilter /
fspecial, there being no IPT to compare with; csrc/pdeip_pyr.hpp computes exactly those): the frames go up once, the result
comes down once.  Inputs and outputs are MATLAB-shaped numpy arrays ([nrows, ncols(, C)],
images in 0..255 as the drivers expect); `mode` selects the ordering (capi.MODE_EXACT_ORDER reproduces the reference's order,
capi.MODE_RED_BLACK the parallel one).

    FlowEminND_llin_2D_v10      matlab/optical_flow/FlowEminND_llin_2D_v10.m      isotropic late-linearisation flow
    FlowEminAD_llin_2D_v10      matlab/optical_flow/FlowEminAD_llin_2D_v10.m      anisotropic diffusion
    FlowEminHS_elin_2D_v10      matlab/optical_flow/FlowEminHS_elin_2D_v10.m      Horn-Schunck, early linearisation
    FlowEminNDFASFMG_elin_2D_v10 matlab/optical_flow/FlowEminNDFASFMG_elin_2D_v10.m FAS full multigrid
    DispEminND_llin_2D          matlab/disparity/DispEminND_llin_2D.m             stereo disparity
    DispEminND_llin_sym_2D      matlab/disparity/DispEminND_llin_sym_2D.m         symmetric stereo
    TVdenoise8 / TVdenoise4     matlab/denoising/TVdenoise{8,4}.m                 total-variation denoising

`Us=`, `Vs=` (param.Us / param.Vs: spatial a-priori fields, double, NaN = no constraint) and `scales=` (param.scales) are taken
by the late-linearisation flow drivers and the disparity driver as the reference's drivers take them.
All eight also exist behind the C-ABI (pdeip_flow_nd_llin, pdeip_flow_ad_llin, pdeip_flow_hs_elin, pdeip_flow_fas_fmg_elin,
pdeip_disp_nd_llin, pdeip_disp_nd_llin_sym, pdeip_tvdenoise8 / 4: csrc/pdeip_drivers.hip) for callers that are not Python -- the
MATLAB session the toolbox runs in; the `capi_*` functions at the end of this file call those.
`graph=True` (the llin flow drivers and the FAS driver, parallel orderings): the run's launches are captured into a HIP graph
on the first call for a frame size and replayed afterwards (graphs.py) -- same kernels and bits, no per-launch host work.
"""
import math

import numpy as np
import torch

from . import capi, device as dev, fas, flow_level as fl, graphs, pyramid


def _up(t, inv, nrows, ncols, method="bilinear"):
    """imresize(t.*inv, [nrows ncols], method) on the device"""
    return dev.pyr_resize(t * np.float32(inv), nrows, ncols, method)


def _zeros_like_plane(t):
    return torch.zeros(t.shape[-2:], dtype=torch.float32, device=t.device)


def _frames(Iin, channels):
    """cat(3, frame0, frame1) -> the two frames on the device, [C, ncols, nrows] each (one upload, sliced there)."""
    I = np.asarray(Iin, dtype=np.float32)
    d = dev.to_device(I if I.ndim == 3 else I[:, :, None])
    return d[:channels].contiguous(), d[channels:2 * channels].contiguous()


def _c3(I):
    I = np.asarray(I, dtype=np.float32)
    return dev.to_device(I if I.ndim == 3 else I[:, :, None])


def _terms(d0, d1, fst, snd):
    """First / second constancy images of one scale on the device (:133-170 of the ND driver)."""
    fst, snd = fst.upper(), snd.upper()
    if fst not in ("RGB", "GRAD") or snd not in ("NONE", "RGB", "GRADMAG"):
        raise ValueError("No such fstTerm / sndTerm")
    I1 = (d0, d1) if fst == "RGB" else (dev.rgb2grad(d0), dev.rgb2grad(d1))
    I2 = (None, None) if snd == "NONE" else (d0, d1)
    return I1, I2


_GRAPHS = {}   # (driver, frame shape, parameters) -> graphs.GraphedRun of its device part


def _apriori_pyramid(field, shapes, scl_factor):
    """param.Us down the pyramid (FlowEminND_llin_2D_v10.m:162-184): NaN -> 0, USap{scl} = imresize(USap{scl-1} .* scl_factor), doubles.
    shapes: (rows, cols) per scale.  Returns the host arrays (finest first)."""
    us = np.array(field, dtype=np.float64)
    us[np.isnan(us)] = 0.0
    out = [us]
    for rows, cols in shapes[1:]:
        out.append(pyramid.resize(out[-1] * scl_factor, rows, cols, "bilinear", out_dtype=np.float64))
    return out


def _flow_llin(level_cls, Iin, channels, fstTerm, sndTerm, defaults, mode, param):
    param = dict(param)
    graph = bool(param.pop("graph", False))   # exact order too since round 3: the walkers' schedule table is built on the stream
    Us, Vs, scales = param.pop("Us", None), param.pop("Vs", None), param.pop("scales", None)
    p = dict(defaults, **param)
    p["sndTerm"] = sndTerm.lower()
    I0, I1 = _frames(Iin, channels)
    if (Us is not None or Vs is not None) and graph:
        raise ValueError("graph=True replays a captured run: not with a-priori fields")

    def device_part(f0, f1):
        P0, P1 = pyramid.build_dev(f0, f1, p["scl_factor"], 20, max_scales=scales)
        level = level_cls(p, mode=mode)
        U = _zeros_like_plane(P0[-1])
        V = torch.zeros_like(U)
        shapes = [(t.shape[-1], t.shape[-2]) for t in P0]
        ap = {}
        for name, field in (("Us", Us), ("Vs", Vs)):
            if field is not None:
                ap[name] = _apriori_pyramid(field, shapes, p["scl_factor"])
        if "Us" in ap:   # U = USap{scales} (:178): MATLAB's double array; here its float32 rounding, as in the disparity driver
            U = dev.to_device(ap["Us"][-1].astype(np.float32))
        if "Vs" in ap:
            V = dev.to_device(ap["Vs"][-1].astype(np.float32))
        for scl in range(len(P0) - 1, -1, -1):
            d0, d1 = P0[scl], P1[scl]
            (a0, a1), (b0, b1) = _terms(d0, d1, fstTerm, sndTerm)
            args = (a0, a1, U, V) + ((d0,) if level_cls is fl.FlowAdLevel else ()) + (b0, b1)
            kw = {}
            if ap:
                on_dev = lambda a: torch.from_numpy(np.ascontiguousarray(a.T)).to(U.device)
                kw = dict(Us=on_dev(ap["Us"][scl]) if "Us" in ap else None, Vs=on_dev(ap["Vs"][scl]) if "Vs" in ap else None,
                          as_diff=2.0 * p["scl_factor"] ** scl, u_double=(scl == len(P0) - 1))
            U, V = level.run(*args, **kw)
            if scl > 0:
                cols, rows = P0[scl - 1].shape[-2:]
                U, V = _up(U, 1.0 / p["scl_factor"], rows, cols), _up(V, 1.0 / p["scl_factor"], rows, cols)
        return U, V

    f0, f1 = dev.div_scalar(I0, 255.0), dev.div_scalar(I1, 255.0)
    if graph:   # the few thousand launches of the run replayed as one HIP graph (graphs.py); same kernels, same results
        key = (level_cls.__name__, tuple(I0.shape), fstTerm.lower(), sndTerm.lower(), int(mode), scales, tuple(sorted((k, repr(v)) for k, v in p.items())))
        if key not in _GRAPHS:
            _GRAPHS[key] = graphs.GraphedRun(device_part)
        U, V = _GRAPHS[key](f0, f1)
    else:
        U, V = device_part(f0, f1)
    dev.sync_check()  # results leave the device: a timed-out dependency wait of the exact-order kernel must not pass silently
    return dev.to_matlab(U), dev.to_matlab(V)


ND_DEFAULTS = dict(alpha=0.042, omega=1.9, gammaS=0.01, firstLoop=4, secondLoop=4, iter=4, b1=1.4843, b2=0.2915, scl_factor=0.75, solver=2)
AD_DEFAULTS = dict(ND_DEFAULTS, quantile=0.9, diffusion="image")


def FlowEminND_llin_2D_v10(Iin, channels, fstTerm="rgb", sndTerm="none", mode=capi.MODE_EXACT_ORDER, **param):
    """[U V] = FlowEminND_llin_2D_v10(Iin, channels, fstTerm, sndTerm, ...): Iin = cat(3, frame0, frame1)."""
    return _flow_llin(fl.FlowLlinLevel, Iin, channels, fstTerm, sndTerm, ND_DEFAULTS, mode, param)


def FlowEminAD_llin_2D_v10(Iin, channels, fstTerm="rgb", sndTerm="none", mode=capi.MODE_EXACT_ORDER, **param):
    return _flow_llin(fl.FlowAdLevel, Iin, channels, fstTerm, sndTerm, AD_DEFAULTS, mode, param)


HS_DEFAULTS = dict(alpha=0.2, omega=1.9, iter=20, b1=0.25, b2=0.75, scl_factor=0.75, solver=2)


def FlowEminHS_elin_2D_v10(Iin, channels, mode=capi.MODE_EXACT_ORDER, **param):
    p = dict(HS_DEFAULTS, **param)
    I0, I1 = _frames(Iin, channels)
    P0, P1 = pyramid.build_dev(dev.div_scalar(I0, 255.0), dev.div_scalar(I1, 255.0), p["scl_factor"], 20)
    level = fl.FlowHsLevel(p, mode=mode)
    U = _zeros_like_plane(P0[-1])
    V = torch.zeros_like(U)
    for scl in range(len(P0) - 1, -1, -1):
        U, V = level.run(P0[scl], P1[scl], U, V)
        if scl > 0:   # imresize(medfilt2(U.*(1/scl_factor), [3 3], 'symmetric'), 'OutputSize', ...): the default, bicubic, method (:189-190)
            cols, rows = P0[scl - 1].shape[-2:]
            inv = np.float32(1.0 / p["scl_factor"])
            mU, mV = torch.empty_like(U), torch.empty_like(V)
            dev.median3(U * inv, None, mU)
            dev.median3(V * inv, None, mV)
            U, V = dev.pyr_resize(mU, rows, cols, "bicubic"), dev.pyr_resize(mV, rows, cols, "bicubic")
    dev.sync_check()  # results leave the device: a timed-out dependency wait of the exact-order kernel must not pass silently
    return dev.to_matlab(U), dev.to_matlab(V)


def FlowEminNDFASFMG_elin_2D_v10(Iin, channels, mode=capi.MODE_EXACT_ORDER, **param):
    I0, I1 = _frames(Iin, channels)
    param = dict(param)
    if param.pop("graph", False):
        key = ("FasFmgFlow", tuple(I0.shape), int(mode), tuple(sorted((k, repr(v)) for k, v in param.items())))
        if key not in _GRAPHS:
            _GRAPHS[key] = fas.FasFmgFlow(param, mode=mode)
        gU, gV = _GRAPHS[key].run_graph(I0, I1)
    else:
        gU, gV = fas.FasFmgFlow(param, mode=mode).run(I0, I1)
    dev.sync_check()
    return dev.to_matlab(gU), dev.to_matlab(gV)


DISP_DEFAULTS = dict(alpha=0.042, gammaS=0.005, omega=1.9, firstLoop=4, secondLoop=6, iter=4, b1=1.48, b2=0.29, scl_factor=0.75, solver=2)   # DispEminND_llin_2D.m:51-63


def DispEminND_llin_2D(Il, Ir, fstTerm="rgb", sndTerm="none", mode=capi.MODE_EXACT_ORDER, Us=None, scales=None, **param):
    """Us: param.Us of the reference, a spatial a-priori disparity map at full resolution (double; NaN = no constraint there,
    zeroed as DispEminND_llin_2D.m:170 does).  It is scaled down with the pyramid (:172-176), starts the coarsest scale (:178;
    there MATLAB's U is that double array itself -- here its float32 rounding, a 1e-8 relative difference in the first
    firstLoop of the coarsest scale) and enters every assembly through the exp influence function (:277-292)."""
    p = dict(DISP_DEFAULTS, **param)
    p["sndTerm"] = sndTerm.lower()
    P0, P1 = pyramid.build_dev(dev.div_scalar(_c3(Il), 255.0), dev.div_scalar(_c3(Ir), 255.0), p["scl_factor"], 10, max_scales=scales)
    level = fl.DispLlinLevel(p, mode=mode)
    U = _zeros_like_plane(P0[-1])
    USap = None
    if Us is not None:
        us = np.array(Us, dtype=np.float64)
        us[np.isnan(us)] = 0.0
        USap = [us]
        for scl in range(1, len(P0)):
            cols, rows = P0[scl].shape[-2:]
            USap.append(pyramid.resize(USap[-1] * p["scl_factor"], rows, cols, "bilinear", out_dtype=np.float64))
        U = dev.to_device(USap[-1].astype(np.float32))
    for scl in range(len(P0) - 1, -1, -1):
        (a0, a1), (b0, b1) = _terms(P0[scl], P1[scl], fstTerm, sndTerm)
        if USap is None:
            U = level.run(a0, a1, U, b0, b1)
        else:
            us64 = torch.from_numpy(np.ascontiguousarray(USap[scl].T)).to(U.device)
            U = level.run(a0, a1, U, b0, b1, Us=us64, as_diff=1.75 * p["scl_factor"] ** scl, u_double=(scl == len(P0) - 1))
        if scl > 0:
            cols, rows = P0[scl - 1].shape[-2:]
            U = _up(U, 1.0 / p["scl_factor"], rows, cols)
    dev.sync_check()
    return dev.to_matlab(U)


SYM_DEFAULTS = dict(alpha=0.035, beta=0.4, omega=1.9, firstLoop=3, secondLoop=4, iter=4, b1=0.25, b2=0.72, scl_factor=0.75, solver=2)


def DispEminND_llin_sym_2D(Il, Ir, mode=capi.MODE_EXACT_ORDER, **param):
    """-> U [nrows, ncols, 2] (left-to-right and right-to-left disparity)."""
    p = dict(SYM_DEFAULTS, **param)
    # no /255 in this driver (:81-82); its coarsest scale stays unsmoothed (:94-98)
    P0, P1 = pyramid.build_dev(_c3(Il), _c3(Ir), p["scl_factor"], 10, pyramid.gaussian(3, 1.0), smooth_last=False)
    level = fl.DispSymLevel(p, mode=mode)
    U0 = _zeros_like_plane(P0[-1])
    U1 = torch.zeros_like(U0)
    for scl in range(len(P0) - 1, -1, -1):
        sr = 2.0 * (1.0 / p["scl_factor"]) ** (-scl)              # srDiff = 2*(1/scl_factor)^-(scl-1), scl 1-based there
        U0, U1 = level.run(P0[scl], P1[scl], U0, U1, sr)
        if scl > 0:
            cols, rows = P0[scl - 1].shape[-2:]
            U0, U1 = _up(U0, 1.0 / p["scl_factor"], rows, cols), _up(U1, 1.0 / p["scl_factor"], rows, cols)
    dev.sync_check()
    return np.stack([dev.to_matlab(U0), dev.to_matlab(U1)], axis=2)


def _tv(I_in, level_cls, p, G, smooth_last):
    Iin = [dev.to_device(np.asarray(I_in, dtype=np.float32))]
    cols, rows = Iin[0].shape[-2:]
    ds_rows, ds_cols = math.ceil(rows * p["scl"]), math.ceil(cols * p["scl"])
    while True:
        c, r = Iin[-1].shape[-2:]
        nr, nc = int(math.ceil(r * p["scl_factor"])), int(math.ceil(c * p["scl_factor"]))
        Iin.append(dev.pyr_resize(Iin[-1], nr, nc))
        Iin[-2] = dev.pyr_smooth(Iin[-2], G)
        if nr <= ds_rows or nc <= ds_cols:
            if smooth_last:
                Iin[-1] = dev.pyr_smooth(Iin[-1], G)
            break
    level = level_cls(p, mode=p["mode"])
    Iout = Iin[-1]
    for scl in range(len(Iin) - 1, -1, -1):
        Iout = level.run(Iin[scl], Iout)
        if scl > 0:
            c, r = Iin[scl - 1].shape[-2:]
            Iout = dev.pyr_resize(Iout, r, c)
    dev.sync_check()
    return dev.to_matlab(Iout)


def TVdenoise8(I_in, mode=capi.MODE_EXACT_ORDER, **param):
    """TVdenoise8.m; I_in in 0..1.  The coarsest scale is not smoothed: the driver writes that result to a misspelt variable (:72)."""
    p = dict(dict(alpha=500.0, omega=1.75, outer_iter=20, inner_iter=4, solver=2, scl=0.75, scl_factor=0.75), mode=mode, **param)
    return _tv(I_in, fl.TvLevel, p, pyramid.gaussian(5, 1.25), smooth_last=False)


def TVdenoise4(I_in, mode=capi.MODE_EXACT_ORDER, **param):
    p = dict(dict(alpha=5.0, omega=1.75, outer_iter=10, inner_iter=5, solver=2, scl=0.5, scl_factor=0.75), mode=mode, **param)
    return _tv(I_in, fl.Tv4Level, p, pyramid.gaussian(7, 2.0), smooth_last=True)


# ---- the same two drivers through the C-ABI (what the MEX stubs call): host arrays in, host arrays out ----------------------
_TERM = {"NONE": 0, "RGB": 1, "GRAD": 2, "GRADMAG": 3}


def _c_params(param):
    import ctypes

    class P(ctypes.Structure):
        _fields_ = [(k, ctypes.c_double) for k in ("alpha", "omega", "gammaS", "b1", "b2", "scl_factor")] + \
                   [(k, ctypes.c_int) for k in ("firstLoop", "secondLoop", "iter", "solver", "scales")]
    s = P()
    for k, _ in P._fields_:
        setattr(s, k, type(getattr(s, k))(param.get(k, 0) or 0))
    return s


def _f_single(a):
    a = np.asarray(a, dtype=np.float32)
    return np.asfortranarray(a if a.ndim == 3 else a[:, :, None])


def _f_double(a):
    return None if a is None else np.asfortranarray(np.asarray(a, dtype=np.float64))


def capi_FlowEminND_llin_2D_v10(Iin, channels, fstTerm="rgb", sndTerm="none", mode=capi.MODE_EXACT_ORDER, Us=None, Vs=None, **param):
    """pdeip_flow_nd_llin on MATLAB-shaped numpy arrays: the C++ twin of FlowEminND_llin_2D_v10 above, same bits."""
    import ctypes
    I = _f_single(Iin)
    rows, cols = I.shape[:2]
    U, V = np.zeros((rows, cols), np.float32, order="F"), np.zeros((rows, cols), np.float32, order="F")
    us, vs = _f_double(Us), _f_double(Vs)
    prm = _c_params(param)
    old = capi.get_mode()
    capi.set_mode(mode)
    try:
        capi.call("pdeip_flow_nd_llin", I.ctypes.data, rows, cols, int(channels), _TERM[fstTerm.upper()], _TERM[sndTerm.upper()], ctypes.addressof(prm),
                  None if us is None else us.ctypes.data, None if vs is None else vs.ctypes.data, U.ctypes.data, V.ctypes.data)
    finally:
        capi.set_mode(old)
    return U, V


def capi_DispEminND_llin_2D(Il, Ir, fstTerm="rgb", sndTerm="none", mode=capi.MODE_EXACT_ORDER, Us=None, **param):
    """pdeip_disp_nd_llin on MATLAB-shaped numpy arrays."""
    import ctypes
    L, R = _f_single(Il), _f_single(Ir)
    rows, cols, C = L.shape
    U = np.zeros((rows, cols), np.float32, order="F")
    us = _f_double(Us)
    prm = _c_params(param)
    old = capi.get_mode()
    capi.set_mode(mode)
    try:
        capi.call("pdeip_disp_nd_llin", L.ctypes.data, R.ctypes.data, rows, cols, C, _TERM[fstTerm.upper()], _TERM[sndTerm.upper()], ctypes.addressof(prm),
                  None if us is None else us.ctypes.data, U.ctypes.data)
    finally:
        capi.set_mode(old)
    return U


def _c_tv_params(param):
    import ctypes

    class P(ctypes.Structure):
        _fields_ = [(k, ctypes.c_double) for k in ("alpha", "omega", "scl", "scl_factor")] + [(k, ctypes.c_int) for k in ("outer_iter", "inner_iter", "solver")]
    s = P()
    for k, _ in P._fields_:
        setattr(s, k, type(getattr(s, k))(param.get(k, 0) or 0))
    return s


def _capi_tv(entry, I_in, mode, param):
    import ctypes
    I = _f_single(I_in)
    rows, cols, F = I.shape
    out = np.zeros((rows, cols, F), np.float32, order="F")
    prm = _c_tv_params(param)
    old = capi.get_mode()
    capi.set_mode(mode)
    try:
        capi.call(entry, I.ctypes.data, rows, cols, F, ctypes.addressof(prm), out.ctypes.data)
    finally:
        capi.set_mode(old)
    return out if np.asarray(I_in).ndim == 3 else out[:, :, 0]


def capi_TVdenoise8(I_in, mode=capi.MODE_EXACT_ORDER, **param):
    """pdeip_tvdenoise8 on a MATLAB-shaped numpy array: the C++ twin of TVdenoise8 above, same bits."""
    return _capi_tv("pdeip_tvdenoise8", I_in, mode, param)


def capi_TVdenoise4(I_in, mode=capi.MODE_EXACT_ORDER, **param):
    """pdeip_tvdenoise4: the C++ twin of TVdenoise4."""
    return _capi_tv("pdeip_tvdenoise4", I_in, mode, param)


def capi_FlowEminHS_elin_2D_v10(Iin, channels, mode=capi.MODE_EXACT_ORDER, **param):
    """pdeip_flow_hs_elin: the C++ twin of FlowEminHS_elin_2D_v10 above, same bits."""
    import ctypes
    I = _f_single(Iin)
    rows, cols = I.shape[:2]
    U, V = np.zeros((rows, cols), np.float32, order="F"), np.zeros((rows, cols), np.float32, order="F")
    prm = _c_params(param)
    old = capi.get_mode()
    capi.set_mode(mode)
    try:
        capi.call("pdeip_flow_hs_elin", I.ctypes.data, rows, cols, int(channels), ctypes.addressof(prm), U.ctypes.data, V.ctypes.data)
    finally:
        capi.set_mode(old)
    return U, V


def capi_DispEminND_llin_sym_2D(Il, Ir, mode=capi.MODE_EXACT_ORDER, **param):
    """pdeip_disp_nd_llin_sym: the C++ twin of DispEminND_llin_sym_2D above, same bits; -> U [nrows, ncols, 2]."""
    import ctypes

    class P(ctypes.Structure):
        _fields_ = [(k, ctypes.c_double) for k in ("alpha", "beta", "omega", "b1", "b2", "scl_factor")] + \
                   [(k, ctypes.c_int) for k in ("firstLoop", "secondLoop", "iter", "solver")]
    prm = P()
    for k, _ in P._fields_:
        setattr(prm, k, type(getattr(prm, k))(param.get(k, 0) or 0))
    L, R = _f_single(Il), _f_single(Ir)
    rows, cols, C = L.shape
    U = np.zeros((rows, cols, 2), np.float32, order="F")
    old = capi.get_mode()
    capi.set_mode(mode)
    try:
        capi.call("pdeip_disp_nd_llin_sym", L.ctypes.data, R.ctypes.data, rows, cols, C, ctypes.addressof(prm), U.ctypes.data)
    finally:
        capi.set_mode(old)
    return U


def capi_FlowEminAD_llin_2D_v10(Iin, channels, fstTerm="rgb", sndTerm="none", mode=capi.MODE_EXACT_ORDER, Us=None, Vs=None, quantile=0.0, diffusion="image", **param):
    """pdeip_flow_ad_llin: the C++ twin of FlowEminAD_llin_2D_v10 above, same bits."""
    import ctypes
    I = _f_single(Iin)
    rows, cols = I.shape[:2]
    U, V = np.zeros((rows, cols), np.float32, order="F"), np.zeros((rows, cols), np.float32, order="F")
    us, vs = _f_double(Us), _f_double(Vs)
    prm = _c_params(param)
    old = capi.get_mode()
    capi.set_mode(mode)
    try:
        capi.call("pdeip_flow_ad_llin", I.ctypes.data, rows, cols, int(channels), _TERM[fstTerm.upper()], _TERM[sndTerm.upper()], ctypes.addressof(prm),
                  float(quantile), int(str(diffusion).lower() == "flow"), None if us is None else us.ctypes.data, None if vs is None else vs.ctypes.data,
                  U.ctypes.data, V.ctypes.data)
    finally:
        capi.set_mode(old)
    return U, V


def capi_FlowEminNDFASFMG_elin_2D_v10(Iin, channels, mode=capi.MODE_EXACT_ORDER, **param):
    """pdeip_flow_fas_fmg_elin: the C++ twin of FlowEminNDFASFMG_elin_2D_v10 above, same bits."""
    import ctypes

    class P(ctypes.Structure):
        _fields_ = [(k, ctypes.c_double) for k in ("alpha", "omega", "b1", "b2", "scl_factor")] + \
                   [(k, ctypes.c_int) for k in ("firstLoop", "iter", "solver", "cycle_index", "scales")]
    prm = P()
    for k, _ in P._fields_:
        setattr(prm, k, type(getattr(prm, k))(param.get(k, 0) or 0))
    I = _f_single(Iin)
    rows, cols = I.shape[:2]
    U, V = np.zeros((rows, cols), np.float32, order="F"), np.zeros((rows, cols), np.float32, order="F")
    old = capi.get_mode()
    capi.set_mode(mode)
    try:
        capi.call("pdeip_flow_fas_fmg_elin", I.ctypes.data, rows, cols, int(channels), ctypes.addressof(prm), U.ctypes.data, V.ctypes.data)
    finally:
        capi.set_mode(old)
    return U, V
