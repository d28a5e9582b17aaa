"""numpy statement of the MATLAB-side stages of one late-linearisation pyramid level.

TEST INFRASTRUCTURE ONLY (see pdeip_oracle.h): the checker for csrc/pdeip_flow.hpp and flow_level.py.
PARITY UNPINNED: this restates MATLAB array code of matlab/optical_flow/FlowEminND_llin_2D_v10.m
(typing: `single op double -> single`, OPdiffWeights casts to double; expressions left to right); it has
not been compared with MATLAB, which this image does not have.  IPT functions are restated by their
documented meaning: imfilter(...,'replicate') = correlation with edge replication, circshift wraps,
medfilt2(...,'symmetric') with a 3x3 window mirrors the edge pixel, nansum skips NaN and returns 0 for an
all-NaN slice.

Arrays are MATLAB-shaped numpy arrays [nrows, ncols(, C)], float32 unless said otherwise.
"""
import numpy as np

F32 = np.float32


def flow_coords(U, V):
    """single(X+U), single(Y+V) with [X Y] = meshgrid(1:cols, 1:rows)   (:205, :223)"""
    nrows, ncols = U.shape
    X = np.arange(1, ncols + 1, dtype=F32)[None, :] + U.astype(F32)
    Y = np.arange(1, nrows + 1, dtype=F32)[:, None] + V.astype(F32)
    return X.astype(F32), Y.astype(F32)


def _term(It, Ix, Iy, b, alpha, dU, dV):
    It, Ix, Iy = [a if a.ndim == 3 else a[:, :, None] for a in (It, Ix, Iy)]
    du, dv = dU.astype(F32)[:, :, None], dV.astype(F32)[:, :, None]
    r = (It - Ix * du) - Iy * dv                                  # :283 (single)
    opnorm = r * r
    gD = F32(b) / (F32(alpha) * np.sqrt(opnorm + F32(0.00001)))   # :284
    return [(Iy * Ix) * gD, (It * Ix) * gD, (It * Iy) * gD, (Ix * Ix) * gD, (Iy * Iy) * gD]  # :236-240 times gD (:321-325)


def _term_gradmag(Ixt, Iyt, Ixx, Iyy, Ixy, b, alpha, dU, dV):
    """The gradient-magnitude second term (:253-258, :291-293)."""
    Ixt, Iyt, Ixx, Iyy, Ixy = [a if a.ndim == 3 else a[:, :, None] for a in (Ixt, Iyt, Ixx, Iyy, Ixy)]
    du, dv = dU.astype(F32)[:, :, None], dV.astype(F32)[:, :, None]
    r1 = (Ixt - Ixx * du) - Ixy * dv
    r2 = (Iyt - Ixy * du) - Iyy * dv
    gD = F32(b) / (F32(alpha) * np.sqrt((r1 * r1 + r2 * r2) + F32(0.00001)))
    return [(Ixy * (Ixx + Iyy)) * gD, (Ixt * Ixx + Iyt * Ixy) * gD, (Ixt * Ixy + Iyt * Iyy) * gD, (Ixx * Ixx + Ixy * Ixy) * gD,
            (Ixy * Ixy + Iyy * Iyy) * gD]


def rgb2grad(I):
    """rgb2grad (:368-381): frames 2f-1 / 2f = imfilter(I(:,:,f), [1 0 -1] / [1 0 -1]', 'replicate')"""
    I3 = I.astype(F32) if I.ndim == 3 else I.astype(F32)[:, :, None]
    out = np.zeros(I3.shape[:2] + (2 * I3.shape[2],), dtype=F32, order="F")
    Px = np.pad(I3, ((0, 0), (1, 1), (0, 0)), mode="edge")
    Py = np.pad(I3, ((1, 1), (0, 0), (0, 0)), mode="edge")
    out[:, :, 0::2] = Px[:, :-2] - Px[:, 2:]
    out[:, :, 1::2] = Py[:-2] - Py[2:]
    return out


def flow_assemble(term1, term2, dU, dV, alpha):
    """term = (It, Ix, Iy, b), or for the second one None or the gradient-magnitude term (Ixt, Iyt, Ixx, Iyy, Ixy, b).
    Returns MGd, CuGd, CvGd, DuGd, DvGd = nansum over all channels (:321-325)."""
    stacks = _term(*term1[:3], term1[3], alpha, dU, dV)
    if term2 is not None:
        second = _term_gradmag(*term2, alpha, dU, dV) if len(term2) == 6 else _term(*term2[:3], term2[3], alpha, dU, dV)
        stacks = [np.concatenate([a, b], axis=2) for a, b in zip(stacks, second)]
    outs = []
    for s in stacks:
        acc = np.zeros(s.shape[:2], dtype=F32)
        for c in range(s.shape[2]):                               # sequential, in slice order
            v = s[:, :, c]
            acc = np.where(np.isnan(v), acc, (acc + v).astype(F32))
        outs.append(acc)
    return outs


def apriori_slices(Us, U, dU, gammaS, alpha, as_diff, u_double, du_double):
    """[ASCu.*gSu, ASDu.*gSu] as the single slices cat() makes of them (:262-270, :301-318).  Us: float64 constraint field.
    u_double: U is still MATLAB's double array (coarsest scale, first firstLoop); du_double: dU is the double zeros of :272."""
    Us = Us.astype(np.float64)
    asd2 = as_diff * as_diff
    if u_double and du_double:
        asc = Us - U.astype(np.float64)
        t = asc - dU.astype(np.float64)
        gS = gammaS / (alpha * (1.0 + (t * t) / asd2))
        return (asc * gS).astype(F32), gS.astype(F32)
    asc = (Us - U.astype(np.float64)).astype(F32) if u_double else (Us.astype(F32) - U.astype(F32)).astype(F32)
    t = (asc - dU.astype(F32)).astype(F32)
    gS = (F32(gammaS) / (F32(alpha) * (F32(1) + ((t * t).astype(F32) / F32(asd2)).astype(F32)).astype(F32)).astype(F32)).astype(F32)
    return (asc * gS).astype(F32), gS


# exp() for the disparity driver's influence function.  libm's / MATLAB's exp are not reproducible bit for bit between a
# host and a device, so the statement and the kernel (csrc/pdeip_flow.hpp det_exp) share ONE algorithm in IEEE double
# arithmetic without FMA: x = k ln2 + r with |r| <= ln2/2 (two-part ln2), the Taylor polynomial of degree 13 by Horner
# (one multiply, then one add per step), 2^k by ldexp.  Relative error < 3e-16 on [-700, 0]; "parity unpinned" w.r.t.
# MATLAB's own exp like every other MATLAB-side stage.
_EXP_C = [1.0 / f for f in (1, 1, 2, 6, 24, 120, 720, 5040, 40320, 362880, 3628800, 39916800, 479001600, 6227020800)]
_LN2_HI, _LN2_LO, _INV_LN2 = 6.93147180369123816490e-01, 1.90821492927058770002e-10, 1.44269504088896338700e+00


def det_exp(x):
    x = np.maximum(np.asarray(x, dtype=np.float64), -700.0)
    k = np.floor(x * _INV_LN2 + 0.5)
    r = (x - k * _LN2_HI) - k * _LN2_LO
    p = np.full_like(r, _EXP_C[13])
    for c in _EXP_C[12::-1]:
        p = p * r + c
    return np.ldexp(p, np.where(np.isnan(k), 0.0, k).astype(np.int64))  # NaN in, NaN out (p is NaN there)


def disp_apriori_slices(Us, U, dU, gammaS, alpha, as_diff, u_double, du_double):
    """[ASCu.*gS, ASDu.*gS] of DispEminND_llin_2D.m:246-248, :277-284 as the single slices cat() makes of them:
    ASCu = USap - U, ASDu = 1, gS = gammaS/alpha * exp(-(USap - U - dU).^2 / ASdiff^2).  Typing as in apriori_slices."""
    Us = Us.astype(np.float64)
    asd2 = as_diff * as_diff
    k = gammaS / alpha
    if u_double and du_double:
        asc = Us - U.astype(np.float64)
        t = asc - dU.astype(np.float64)
        gS = k * det_exp(-(t * t) / asd2)
        return (asc * gS).astype(F32), gS.astype(F32)
    asc = (Us - U.astype(np.float64)).astype(F32) if u_double else (Us.astype(F32) - U.astype(F32)).astype(F32)
    t = (asc - dU.astype(F32)).astype(F32)
    arg = (-(t * t).astype(F32) / F32(asd2)).astype(F32)
    gS = (F32(k) * det_exp(arg.astype(np.float64)).astype(F32)).astype(F32)   # single exp := the double algorithm, rounded once
    return (asc * gS).astype(F32), gS


def nan_append(acc, v):
    """One more slice of a nansum"""
    return np.where(np.isnan(v), acc, (acc + v).astype(F32))


def _shift(A, di, dj):
    """circshift(A, [di dj]): element (i,j) takes A(i-di, j-dj), wrapping."""
    return np.roll(np.roll(A, di, axis=0), dj, axis=1)


def _ver(A):  # imfilter(A, [0.25 0 -0.25]', 'replicate'): 0.25*north - 0.25*south
    P = np.pad(A, ((1, 1), (0, 0)), mode="edge")
    return 0.25 * P[:-2, :] - 0.25 * P[2:, :]


def _hor(A):  # imfilter(A, [0.25 0 -0.25], 'replicate'): 0.25*west - 0.25*east
    P = np.pad(A, ((0, 0), (1, 1)), mode="edge")
    return 0.25 * P[:, :-2] - 0.25 * P[:, 2:]


def op_diff_weights(U, V, dU, dV):
    """[wW wN wS wE] = OPdiffWeights(U+dU, V+dV) (:389-433), as the single arrays handed to the solver."""
    Uf = (U.astype(F32) + dU.astype(F32)).astype(np.float64)
    Vf = (V.astype(F32) + dV.astype(F32)).astype(np.float64)
    Uver, Vver, Uhor, Vhor = _ver(Uf), _ver(Vf), _hor(Uf), _hor(Vf)

    def w(di, dj, Uc, Vc):
        acc = (_shift(Uf, di, dj) - Uf) ** 2
        acc = acc + (Uc + _shift(Uc, di, dj)) ** 2
        acc = acc + (_shift(Vf, di, dj) - Vf) ** 2
        acc = acc + (Vc + _shift(Vc, di, dj)) ** 2
        return (1.0 / np.sqrt(acc + 0.00001)).astype(F32)

    wW, wE = w(0, 1, Uver, Vver), w(0, -1, Uver, Vver)
    wN, wS = w(1, 0, Uhor, Vhor), w(-1, 0, Uhor, Vhor)
    return wW, wN, wS, wE


def median3_sum(A, B=None):
    """medfilt2(A + B, [3 3], 'symmetric') (:352)."""
    S = A.astype(F32) if B is None else (A.astype(F32) + B.astype(F32)).astype(F32)
    P = np.pad(S, 1, mode="symmetric")
    nrows, ncols = S.shape
    stack = np.stack([P[di:di + nrows, dj:dj + ncols] for dj in range(3) for di in range(3)], axis=0)
    return np.sort(stack, axis=0)[4].astype(F32)


def flow_level(orc, I1t0, I1t1, U, V, param, I2t0=None, I2t1=None, Us=None, Vs=None, as_diff=None, u_double=False):
    """One pyramid level (:208-356, without the pyramid's imresize): firstLoop x [warp, derivatives,
    secondLoop x (assembly, diffusion weights, Oflow_sor_llin4_2d)], median.  `orc` = tests/oracle_lib.
    param: firstLoop, secondLoop, iter, omega, solver, alpha, b1, b2, order (oracle sweep/line order)."""
    U, V = U.astype(F32), V.astype(F32)
    for first in range(param["firstLoop"]):
        X, Y = flow_coords(U, V)
        w1 = orc.BilinInterp_2d(I1t1, X, Y)
        t1 = orc.FstDerivatives5(I1t0, w1) + (param["b1"],)
        t2 = None
        if I2t1 is not None:
            w2 = orc.BilinInterp_2d(I2t1, X, Y)
            snd = param.get("sndTerm", "rgb") == "gradmag"
            t2 = (orc.SndDerivatives5(I2t0, w2) if snd else orc.FstDerivatives5(I2t0, w2)) + (param["b2"],)
        dU, dV = np.zeros_like(U), np.zeros_like(V)
        for k in range(param["secondLoop"]):
            MGd, CuGd, CvGd, DuGd, DvGd = flow_assemble(t1, t2, dU, dV, param["alpha"])
            if Us is not None:
                c, d = apriori_slices(Us, U, dU, param["gammaS"], param["alpha"], as_diff, u_double and first == 0, k == 0)
                CuGd, DuGd = nan_append(CuGd, c), nan_append(DuGd, d)
            if Vs is not None:
                c, d = apriori_slices(Vs, V, dV, param["gammaS"], param["alpha"], as_diff, u_double and first == 0, k == 0)
                CvGd, DvGd = nan_append(CvGd, c), nan_append(DvGd, d)
            wW, wN, wS, wE = op_diff_weights(U, V, dU, dV)
            dU, dV = orc.Oflow_sor_llin4_2d(U, V, dU, dV, MGd, CuGd, CvGd, DuGd, DvGd, wW, wN, wE, wS, param["iter"],
                                            param["omega"], solver=param["solver"], order=param["order"])
        U, V = median3_sum(U, dU), median3_sum(V, dV)
    return U, V


def disp_assemble(term1, term2, dU, alpha):
    """CuGd, DuGd of matlab/disparity/DispEminND_llin_2D.m:258-293; term = (It, Ix, b); plain sum over channels."""
    def one(It, Ix, b):
        It, Ix = [a if a.ndim == 3 else a[:, :, None] for a in (It, Ix)]
        r = It - Ix * dU.astype(F32)[:, :, None]
        gD = F32(b) / (F32(alpha) * np.sqrt(r * r + F32(0.00001)))
        return (It * Ix) * gD, (Ix * Ix) * gD
    def gradmag(Ixt, Iyt, Ixx, Ixy, b):   # :236-238, :271
        Ixt, Iyt, Ixx, Ixy = [a if a.ndim == 3 else a[:, :, None] for a in (Ixt, Iyt, Ixx, Ixy)]
        du_ = dU.astype(F32)[:, :, None]
        r1, r2 = Ixt - Ixx * du_, Iyt - Ixy * du_
        gD = F32(b) / (F32(alpha) * np.sqrt((r1 * r1 + r2 * r2) + F32(0.00001)))
        return (Ixt * Ixx + Iyt * Ixy) * gD, (Ixx * Ixx + Ixy * Ixy) * gD
    cu, du = one(*term1)
    if term2 is not None:
        c2, d2 = gradmag(*term2) if len(term2) == 5 else one(*term2)
        cu, du = np.concatenate([cu, c2], axis=2), np.concatenate([du, d2], axis=2)
    outs = []
    for s in (cu, du):
        acc = s[:, :, 0].astype(F32)
        for c in range(1, s.shape[2]):
            acc = (acc + s[:, :, c]).astype(F32)
        outs.append(acc)
    return outs


def disp_level(orc, I1t0, I1t1, U, param, I2t0=None, I2t1=None, Us=None, as_diff=None, u_double=False):
    """One pyramid level of DispEminND_llin_2D.m (:202-316, without the pyramid's imresize).  Us: the spatial a-priori field of
    the scale (param.Us, float64; gammaS in param), as_diff = 1.75*(1/scl_factor)^-(scl-1), u_double: U is still MATLAB's double
    array (coarsest scale, before its first median)."""
    U = U.astype(F32)
    Z = np.zeros_like(U)
    for first in range(param["firstLoop"]):
        X, Y = flow_coords(U, Z)
        w1 = orc.BilinInterp_2d(I1t1, X, Y)
        d = orc.FstDerivatives5(I1t0, w1)
        t1, t2 = (d[0], d[1], param["b1"]), None
        if I2t1 is not None:
            w2 = orc.BilinInterp_2d(I2t1, X, Y)
            if param.get("sndTerm", "rgb") == "gradmag":
                s2 = orc.SndDerivatives5(I2t0, w2)                     # Ixt, Iyt, Ixx, Iyy, Ixy
                t2 = (s2[0], s2[1], s2[2], s2[4], param["b2"])
            else:
                d2 = orc.FstDerivatives5(I2t0, w2)
                t2 = (d2[0], d2[1], param["b2"])
        dU = np.zeros_like(U)
        for k in range(param["secondLoop"]):
            CuGd, DuGd = disp_assemble(t1, t2, dU, param["alpha"])
            if Us is not None:   # plain sum(): one more slice, NaN propagates (:291-292)
                c, d = disp_apriori_slices(Us, U, dU, param["gammaS"], param["alpha"], as_diff, u_double and first == 0, k == 0)
                CuGd, DuGd = (CuGd + c).astype(F32), (DuGd + d).astype(F32)
            wW, wN, wE, wS = orc.DdiffWeights((U + dU).astype(F32), 0.00001)
            dU = orc.Disp_sor_llin4_2d(U, dU, CuGd, DuGd, wW, wN, wE, wS, param["iter"], param["omega"], solver=param["solver"],
                                       order=param["order"])
        U = median3_sum(U, dU)
    return U


# ---- TV denoising: matlab/denoising/TVdenoise8.m ---------------------------------------------------------------------

def _pad_edge(A):
    return np.pad(A, 1, mode="edge")


def ad_diff_weights(D, quantile=None):
    """[W NW N NE E SE S SW] = ADdiffWeights(D), double; D is [nrows, ncols(, F)] single.
    quantile None: TVdenoise8.m:119-231 (lambda = median, outer rows/columns of the weights zeroed);
    a number: FlowEminAD_llin_2D_v10.m:416-487 (lambda = sorted(round(numel*quantile)), circshift wrap-around kept)."""
    D = D.astype(np.float64)
    if D.ndim == 2:
        D = D[:, :, None]
    nrows, ncols, F = D.shape
    s = 4.0 + np.sqrt(8.0)
    k1, k2 = 1.0 / s, np.sqrt(2.0) / s
    gxs, gys = [], []
    for f in range(F):
        P = _pad_edge(D[:, :, f])
        at = lambda di, dj: P[1 + di:1 + di + nrows, 1 + dj:1 + dj + ncols]
        dx = k1 * at(1, 1)
        dx = dx + (-k1) * at(1, -1)
        dx = dx + k2 * at(0, 1)
        dx = dx + (-k2) * at(0, -1)
        dx = dx + k1 * at(-1, 1)
        dx = dx + (-k1) * at(-1, -1)
        dy = k1 * at(1, 1)
        dy = dy + k2 * at(1, 0)
        dy = dy + k1 * at(1, -1)
        dy = dy + (-k1) * at(-1, 1)
        dy = dy + (-k2) * at(-1, 0)
        dy = dy + (-k1) * at(-1, -1)
        gxs.append(dx); gys.append(dy)
    gx, gy = np.stack(gxs, 2), np.stack(gys, 2)
    nn = gx * gx + gy * gy
    best = np.argmax(nn, axis=2)                       # first maximal frame, like MATLAB's max
    ii, jj = np.meshgrid(np.arange(nrows), np.arange(ncols), indexing="ij")
    mx, my = gx[ii, jj, best], gy[ii, jj, best]
    norm = mx * mx + my * my
    srt = np.sort(norm.ravel())
    srt = srt[srt != 0]
    if quantile is None:
        lam = srt[(srt.size + 1) // 2 - 1] if srt.size else 1.0   # sorted(round(numel*0.5 + eps))
    else:
        lam = srt[max(int(np.floor(srt.size * quantile + 0.5)), 1) - 1] if srt.size else 1.0   # sorted(round(numel*quantile))
    multip = 1.0 / (norm + 2.0 * lam)
    dyy, dxx, dxy = multip * (my * my + lam), multip * (mx * mx + lam), -multip * (mx * my)
    sh = lambda A, di, dj: np.roll(np.roll(A, di, axis=0), dj, axis=1)
    if quantile is not None:
        return [0.5 * (dyy + sh(dyy, 0, 1)), 0.25 * (dxy + sh(dxy, 1, 1)), 0.5 * (dxx + sh(dxx, 1, 0)), -0.25 * (dxy + sh(dxy, 1, -1)),
                0.5 * (dyy + sh(dyy, 0, -1)), 0.25 * (dxy + sh(dxy, -1, -1)), 0.5 * (dxx + sh(dxx, -1, 0)), -0.25 * (dxy + sh(dxy, -1, 1))], lam
    W = 0.5 * (dyy + sh(dyy, 0, 1)); W[:, 0] = 0
    NW = 0.25 * (dxy + sh(dxy, 1, 1)); NW[:, 0] = 0; NW[0, :] = 0
    N = 0.5 * (dxx + sh(dxx, 1, 0)); N[0, :] = 0
    NE = -0.25 * (dxy + sh(dxy, 1, -1)); NE[:, -1] = 0; NE[0, :] = 0
    E = 0.5 * (dyy + sh(dyy, 0, -1)); E[:, -1] = 0
    SE = 0.25 * (dxy + sh(dxy, -1, -1)); SE[:, -1] = 0; SE[-1, :] = 0
    S = 0.5 * (dxx + sh(dxx, -1, 0)); S[-1, :] = 0
    SW = -0.25 * (dxy + sh(dxy, -1, 1)); SW[-1, :] = 0; SW[:, 0] = 0
    return [W, NW, N, NE, E, SE, S, SW], lam


def tv_assemble(Iout, Iin, alpha):
    """TRACE, B and single(alpha*w) of one outer iteration (TVdenoise8.m:82-86); arrays [nrows, ncols(, F)] single."""
    w, _ = ad_diff_weights(Iout)
    shape3 = Iout.shape if Iout.ndim == 3 else Iout.shape + (1,)
    tot = w[0] + w[1]
    for k in range(2, 8):
        tot = tot + w[k]
    diff = (Iout.astype(F32) - Iin.astype(F32)).reshape(shape3)
    psi = F32(1.0) / np.sqrt(diff * diff + F32(2.220446049250313e-16))
    TRACE = (psi + (alpha * tot).astype(F32)[:, :, None]).astype(F32)
    B = (psi * Iin.astype(F32).reshape(shape3)).astype(F32)
    ws = [np.repeat((alpha * a).astype(F32)[:, :, None], shape3[2], axis=2).reshape(Iout.shape) for a in w]
    return TRACE.reshape(Iout.shape), B.reshape(Iout.shape), ws


def tv_level(orc, Iin, Iout, param):
    """The lagged-diffusivity loop of one scale (TVdenoise8.m:78-100)."""
    X = Iout.astype(F32)
    for _ in range(param["outer_iter"] + 1):
        TRACE, B, w = tv_assemble(X, Iin, param["alpha"])
        X = orc.PDEsolver8(X, TRACE, B, *w, param["inner_iter"], param["omega"], solver=param["solver"], order=param["order"])
    return X


# ---- Horn-Schunck, early linearisation: matlab/optical_flow/FlowEminHS_elin_2D_v10.m ------------------------------

HS_PRE = np.array([0.037659, 0.249724, 0.439911, 0.249724, 0.037659], dtype=F32)
HS_D1F = np.array([-0.104550, -0.292315, 0.0, 0.292315, 0.104550], dtype=F32)   # O_dx flipped by 'conv'
HS_D2 = np.array([0.232905, 0.002668, -0.471147, 0.002668, 0.232905], dtype=F32)


def _conv5(A, k, axis):
    """imfilter(A, k, 'replicate', 'conv') for a 5-tap kernel already flipped; single arithmetic, taps left to right."""
    pad = [(0, 0)] * A.ndim
    pad[axis] = (2, 2)
    P = np.pad(A.astype(F32), pad, mode="edge")
    n = A.shape[axis]
    sl = lambda t: tuple(slice(t, t + n) if ax == axis else slice(None) for ax in range(A.ndim))
    s = (k[0] * P[sl(0)]).astype(F32)
    for t in range(1, 5):
        s = (s + (k[t] * P[sl(t)]).astype(F32)).astype(F32)
    return s


def hs_assemble(It0, It1, b1, b2):
    """MGd, CuGd, CvGd, DuGd, DvGd of FlowEminHS_elin_2D_v10.m:133-164; frames [nrows, ncols(, C)] single."""
    It0, It1 = [a.astype(F32) if a.ndim == 3 else a.astype(F32)[:, :, None] for a in (It0, It1)]
    b1, b2 = F32(b1), F32(b2)
    Ist = ((It0 + It1) * F32(0.55)).astype(F32)
    Idt = (It0 - It1).astype(F32)
    vh = lambda A, kv, kh: _conv5(_conv5(A, kv, 0), kh, 1)   # down the columns first, then along the rows
    hv = lambda A, kh, kv: _conv5(_conv5(A, kh, 1), kv, 0)
    Idx, Idy = vh(Ist, HS_PRE, HS_D1F), hv(Ist, HS_PRE, HS_D1F)
    Idxx, Idyy = vh(Ist, HS_PRE, HS_D2), hv(Ist, HS_PRE, HS_D2)
    Idxy = hv(Ist, HS_D1F, HS_D1F)
    Idxt = (vh(It0, HS_PRE, HS_D1F) - vh(It1, HS_PRE, HS_D1F)).astype(F32)
    Idyt = (hv(It0, HS_PRE, HS_D1F) - hv(It1, HS_PRE, HS_D1F)).astype(F32)
    M = (b1 * Idy) * Idx + (b2 * Idxy) * (Idxx + Idyy)
    Cu = (b1 * Idt) * Idx + b2 * (Idxt * Idxx + Idyt * Idxy)
    Cv = (b1 * Idt) * Idy + b2 * (Idxt * Idxy + Idyt * Idyy)
    Du = (b1 * Idx) * Idx + b2 * (Idxx * Idxx + Idxy * Idxy)
    Dv = (b1 * Idy) * Idy + b2 * (Idxy * Idxy + Idyy * Idyy)
    outs = []
    for s in (M, Cu, Cv, Du, Dv):
        acc = s[:, :, 0].astype(F32)
        for c in range(1, s.shape[2]):
            acc = (acc + s[:, :, c]).astype(F32)
        outs.append(acc)
    return outs


def hs_level(orc, It0, It1, U, V, param):
    """One scale of FlowEminHS_elin_2D_v10.m (:119-196, without the pyramid's resize)."""
    M, Cu, Cv, Du, Dv = hs_assemble(It0, It1, param["b1"], param["b2"])
    channels = It0.shape[2] if It0.ndim == 3 else 1
    W = np.full(U.shape, F32(param["alpha"] * channels), dtype=F32, order="F")
    if param["iter"] <= 0:
        return U.astype(F32), V.astype(F32)
    return orc.Oflow_sor_elin4_2d(U, V, M, Cu, Cv, Du, Dv, W, W, W, W, param["iter"], param["omega"], solver=param["solver"],
                                  order=param["order"])


# ---------------------------------------------------------------------------------------------------------
# FAS full multigrid flow with early linearisation (matlab/optical_flow/FlowEminNDFASFMG_elin_2D_v10.m)
# ---------------------------------------------------------------------------------------------------------
FAS_LPF = np.array([1, 4, 6, 4, 1], dtype=np.float64) / 16                                     # lpf (:99)
FAS_D1F_SCL = (np.array([-0.104550, -0.292315, 0.0, 0.292315, 0.104550]) / 255).astype(F32)     # O_dx_scl (:87) flipped
FAS_PLANES = ("Idt", "Idx", "Idy", "Idxx", "Idyy", "Idxy", "Idxt", "Idyt", "M", "Cu", "Cv", "Du", "Dv")


def _c3(A):
    return A.astype(F32) if A.ndim == 3 else A.astype(F32)[:, :, None]


def fas_gaussian5(sigma=1.0):
    """fspecial('gaussian', [5 5], sigma) as single taps (:98)"""
    ax = np.arange(-2, 3, dtype=np.float64)
    g = np.exp(-(ax[:, None] ** 2 + ax[None, :] ** 2) / (2.0 * sigma * sigma))
    g[g < np.finfo(np.float64).eps * g.max()] = 0
    return (g / g.sum()).astype(F32)


def fas_gauss5(I, G):
    """imfilter(I, G, 'replicate', 'conv') for a 5x5 kernel (:104): single arithmetic, taps in column-major order of the
    flipped kernel (our definition of the unspecified order)."""
    I3 = _c3(I)
    P = np.pad(I3, ((2, 2), (2, 2), (0, 0)), mode="edge")
    H, W = I3.shape[:2]
    Gf = G[::-1, ::-1]
    acc = None
    for b in range(5):
        for a in range(5):
            t = (Gf[a, b] * P[a:a + H, b:b + W, :]).astype(F32)
            acc = t if acc is None else (acc + t).astype(F32)
    return acc


def fas_down(I):
    """One pyramid step (:108-111): lpf along the rows of the image, then lpf', then (1:2:end, 1:2:end, :)."""
    k = FAS_LPF.astype(F32)
    return np.ascontiguousarray(_conv5(_conv5(_c3(I), k, 1), k, 0)[::2, ::2, :])


def fas_pyramid(I0, I1, max_scales=None):
    """It0{scl}, It1{scl} (:104-120), finest first; stops after the first scale with a side <= 10."""
    G = fas_gaussian5(1.0)
    P0, P1 = [fas_gauss5(I0, G)], [fas_gauss5(I1, G)]
    while max_scales is None or len(P0) < max_scales:
        P0.append(fas_down(P0[-1]))
        P1.append(fas_down(P1[-1]))
        if P0[-1].shape[0] <= 10 or P0[-1].shape[1] <= 10:
            break
    return P0, P1


def fas_prepare(It0, It1, b1, b2):
    """The per-scale constants (:125-153) as a dict of [nrows, ncols, C] single arrays (FAS_PLANES)."""
    It0, It1 = _c3(It0), _c3(It1)
    b1, b2 = F32(b1), F32(b2)
    Ist = (((It0 + It1) * F32(0.55)).astype(F32) / F32(255)).astype(F32)
    Idt = ((It0 - It1) / F32(255)).astype(F32)
    vh = lambda A, kv, kh: _conv5(_conv5(A, kv, 0), kh, 1)
    hv = lambda A, kh, kv: _conv5(_conv5(A, kh, 1), kv, 0)
    Idx, Idy = vh(Ist, HS_PRE, HS_D1F), hv(Ist, HS_PRE, HS_D1F)
    Idxx, Idyy = vh(Ist, HS_PRE, HS_D2), hv(Ist, HS_PRE, HS_D2)
    Idxy = hv(Ist, HS_D1F, HS_D1F)
    Idxt = (vh(It0, HS_PRE, FAS_D1F_SCL) - vh(It1, HS_PRE, FAS_D1F_SCL)).astype(F32)
    Idyt = (hv(It0, HS_PRE, FAS_D1F_SCL) - hv(It1, HS_PRE, FAS_D1F_SCL)).astype(F32)
    M = (b1 * Idy) * Idx + (b2 * Idxy) * (Idxx + Idyy)
    Cu = (b1 * Idt) * Idx + b2 * (Idxt * Idxx + Idyt * Idxy)
    Cv = (b1 * Idt) * Idy + b2 * (Idxt * Idxy + Idyt * Idyy)
    Du = (b1 * Idx) * Idx + b2 * (Idxx * Idxx + Idxy * Idxy)
    Dv = (b1 * Idy) * Idy + b2 * (Idxy * Idxy + Idyy * Idyy)
    vals = (Idt, Idx, Idy, Idxx, Idyy, Idxy, Idxt, Idyt, M, Cu, Cv, Du, Dv)
    return {k: np.asfortranarray(v.astype(F32)) for k, v in zip(FAS_PLANES, vals)}


def fas_gd(pl, U, V, b1, b2, k):
    """gd = 1./(k*sqrt(OPnorm+0.00001)) with OPnorm of :382-385 (k = channels*alpha in the smoother, alpha for residuals)"""
    u, v = U.astype(F32)[:, :, None], V.astype(F32)[:, :, None]
    r1 = (pl["Idt"] - pl["Idx"] * u) - pl["Idy"] * v
    r2 = (pl["Idxt"] - pl["Idxx"] * u) - pl["Idxy"] * v
    r3 = (pl["Idyt"] - pl["Idxy"] * u) - pl["Idyy"] * v
    opnorm = F32(b1) * (r1 * r1) + F32(b2) * ((r2 * r2) + (r3 * r3))
    return (F32(1) / (F32(k) * np.sqrt(opnorm + F32(0.00001)))).astype(F32)


def _sum3(A):
    acc = A[:, :, 0].astype(F32)
    for c in range(1, A.shape[2]):
        acc = (acc + A[:, :, c]).astype(F32)
    return np.asfortranarray(acc)


def fas_smooth(orc, U, V, pl, Cu, Cv, param, residuals=False):
    """smooth() (:367-464).  Cu, Cv: [nrows, ncols, C] right-hand sides (the scale's own, or fu/fv of the cycle)."""
    channels = pl["Idx"].shape[2]
    U, V = np.asfortranarray(U, dtype=F32), np.asfortranarray(V, dtype=F32)
    zero = np.zeros_like(U)
    for _ in range(param["firstLoop"]):
        gd = fas_gd(pl, U, V, param["b1"], param["b2"], channels * param["alpha"])
        wW, wN, wS, wE = op_diff_weights(U, V, zero, zero)
        MGd, CuGd, CvGd, DuGd, DvGd = [_sum3(a * gd) for a in (pl["M"], Cu, Cv, pl["Du"], pl["Dv"])]
        U, V = orc.Oflow_sor_elin4_2d(U, V, MGd, CuGd, CvGd, DuGd, DvGd, wW, wN, wE, wS, param["iter"], param["omega"],
                                      solver=param["solver"], order=param["order"])
    if not residuals:
        return U, V
    gd = fas_gd(pl, U, V, param["b1"], param["b2"], param["alpha"])
    wW, wN, wS, wE = op_diff_weights(U, V, zero, zero)
    MGd, CuGd, CvGd, DuGd, DvGd = [np.asfortranarray((a * gd).astype(F32)) for a in (pl["M"], Cu, Cv, pl["Du"], pl["Dv"])]
    _, _, RU, RV = orc.Oflow_sor_elin4_2d(U, V, MGd, CuGd, CvGd, DuGd, DvGd, wW, wN, wE, wS, 0, param["omega"],
                                          solver=param["solver"], nargout=4)
    return U, V, RU, RV


def fas_restrict(A, scale):
    """imfilter(A*scale, fw, 'replicate', 'conv')(1:2:end, 1:2:end, :) with fw = [1 2 1; 2 4 2; 1 2 1]/16 (:200, :212-217);
    single arithmetic, taps in column-major order."""
    A3 = (_c3(A) * F32(scale)).astype(F32)
    P = np.pad(A3, ((1, 1), (1, 1), (0, 0)), mode="edge")
    H, W = A3.shape[:2]
    fw = (np.array([[1, 2, 1], [2, 4, 2], [1, 2, 1]], dtype=np.float64) / 16).astype(F32)
    acc = None
    for b in range(3):
        for a in range(3):
            t = (fw[a, b] * P[a:a + H, b:b + W, :]).astype(F32)
            acc = t if acc is None else (acc + t).astype(F32)
    out = acc[::2, ::2, :]
    return np.asfortranarray(out if A.ndim == 3 else out[:, :, 0])


def _bilin_axis(n_in, n_out):
    """imresize 'bilinear' when enlarging (no antialiasing): for output index o the two source indices (clamped)
    and the weight of the second one, pixel centres aligned the MATLAB way."""
    scale = n_out / n_in
    x = (np.arange(n_out, dtype=np.float64) + 0.5) / scale - 0.5
    i0 = np.floor(x)
    f = x - i0
    i0 = i0.astype(np.int64)
    return np.clip(i0, 0, n_in - 1), np.clip(i0 + 1, 0, n_in - 1), f


def fas_prolong_add(U, Uc, Ures, inv_scale):
    """U + imresize((Uc-Ures)*(1/scl_factor), size(U), 'bilinear') (:256-257): rows first, then columns, in double,
    rounded to single once; the sum with U in single."""
    D = ((Uc.astype(F32) - Ures.astype(F32)) * F32(inv_scale)).astype(F32).astype(np.float64)
    r0, r1, fr = _bilin_axis(D.shape[0], U.shape[0])
    c0, c1, fc = _bilin_axis(D.shape[1], U.shape[1])
    T = (1.0 - fr)[:, None] * D[r0, :] + fr[:, None] * D[r1, :]
    R = (1.0 - fc)[None, :] * T[:, c0] + fc[None, :] * T[:, c1]
    return np.asfortranarray((U.astype(F32) + R.astype(F32)).astype(F32))


def fas_cycle(orc, planes, U, V, Cu, Cv, scl, param):
    """FAS_CYCLE (:193-273); planes[scl] = fas_prepare of scale scl (0 = finest), Cu/Cv the right-hand side at this scale."""
    scales = len(planes)
    pl = planes[scl]
    if scl == scales - 1:
        return fas_smooth(orc, U, V, pl, Cu, Cv, param)
    s = param["scl_factor"]
    for _ in range(param["cycle_index"]):
        U, V, RU, RV = fas_smooth(orc, U, V, pl, Cu, Cv, param, residuals=True)
        RUres, RVres = fas_restrict(RU, s), fas_restrict(RV, s)
        Ures, Vres = fas_restrict(U, s), fas_restrict(V, s)
        pc = planes[scl + 1]
        gd = fas_gd(pc, Ures, Vres, param["b1"], param["b2"], param["alpha"])
        zero = np.zeros_like(Ures)
        wW, wN, wS, wE = op_diff_weights(Ures, Vres, zero, zero)
        MGd, DuGd, DvGd = [np.asfortranarray((a * gd).astype(F32)) for a in (pc["M"], pc["Du"], pc["Dv"])]
        Au, Av = orc.Oflow_lhs_elin4_2d(Ures, Vres, MGd, DuGd, DvGd, wW, wN, wE, wS)
        fu = ((_c3(RUres) + _c3(Au)) / gd).astype(F32)
        fv = ((_c3(RVres) + _c3(Av)) / gd).astype(F32)
        Uc, Vc = fas_cycle(orc, planes, Ures, Vres, fu, fv, scl + 1, param)
        U, V = fas_prolong_add(U, Uc, Ures, 1.0 / s), fas_prolong_add(V, Vc, Vres, 1.0 / s)
    return fas_smooth(orc, U, V, pl, Cu, Cv, param)


def _cubic(t):
    t = np.abs(t)
    t2 = t * t
    t3 = t2 * t
    inner = (1.5 * t3 - 2.5 * t2) + 1.0
    outer = ((-0.5 * t3 + 2.5 * t2) - 4.0 * t) + 2.0
    return np.where(t <= 1.0, inner, np.where(t <= 2.0, outer, 0.0))


def _cubic_axis(n_in, n_out):
    scale = n_out / n_in
    x = (np.arange(n_out, dtype=np.float64) + 0.5) / scale - 0.5
    first = np.floor(x).astype(np.int64) - 1
    idx = [np.clip(first + t, 0, n_in - 1) for t in range(4)]
    w = [_cubic(x - (first + t).astype(np.float64)) for t in range(4)]
    total = ((w[0] + w[1]) + w[2]) + w[3]
    return idx, [wt / total for wt in w]


def fas_upscale(U, mul, nrows_out, ncols_out):
    """imresize(U.*mul, [nrows_out ncols_out]) of the outer loop (:177-180): bicubic (imresize's default), enlarging;
    four taps per axis, weights normalised, rows first, then columns, in double, rounded to single once."""
    D = (U.astype(F32) * F32(mul)).astype(F32).astype(np.float64)
    ri, rw = _cubic_axis(D.shape[0], nrows_out)
    ci, cw = _cubic_axis(D.shape[1], ncols_out)
    T = rw[0][:, None] * D[ri[0], :]
    for t in range(1, 4):
        T = T + rw[t][:, None] * D[ri[t], :]
    R = cw[0][None, :] * T[:, ci[0]]
    for t in range(1, 4):
        R = R + cw[t][None, :] * T[:, ci[t]]
    return np.asfortranarray(R.astype(F32))


def fas_fmg(orc, I0, I1, param, max_scales=None):
    """The whole driver (:104-183): pyramid, constants, one cycle per scale coarse to fine, bicubic up-scaling between."""
    P0, P1 = fas_pyramid(I0, I1, max_scales)
    planes = [fas_prepare(a, b, param["b1"], param["b2"]) for a, b in zip(P0, P1)]
    U = V = None
    for scl in range(len(planes) - 1, -1, -1):
        if U is None:
            U = np.zeros(P0[scl].shape[:2], dtype=F32, order="F")
            V = U.copy()
        U, V = fas_cycle(orc, planes, U, V, planes[scl]["Cu"], planes[scl]["Cv"], scl, param)
        if scl > 0:
            r, c = P0[scl - 1].shape[:2]
            U, V = fas_upscale(U, 1.0 / param["scl_factor"], r, c), fas_upscale(V, 1.0 / param["scl_factor"], r, c)
    return U, V


# ---------------------------------------------------------------------------------------------------------
# Anisotropic-diffusion flow with late linearisation (matlab/optical_flow/FlowEminAD_llin_2D_v10.m)
# ---------------------------------------------------------------------------------------------------------
def flow_ad_level(orc, I1t0, I1t1, U, V, param, It0, I2t0=None, I2t1=None, Us=None, Vs=None, as_diff=None, u_double=False):
    """One pyramid level (:198-366, without imresize):
    anisotropic weights from the image It0 ('image', once per level) or from U+dU+V+dV ('flow', every inner iteration),
    robust assembly as in the isotropic driver, Oflow_sor_llin8_2d, median.
    param: firstLoop, secondLoop, iter, omega, solver, alpha, b1, b2, quantile, diffusion, order."""
    U, V = U.astype(F32), V.astype(F32)
    single = lambda ws: [np.asfortranarray(w.astype(F32)) for w in ws]
    if param["diffusion"] == "image":
        w8 = single(ad_diff_weights(It0, param["quantile"])[0])
    for first in range(param["firstLoop"]):
        X, Y = flow_coords(U, V)
        t1 = orc.FstDerivatives5(I1t0, orc.BilinInterp_2d(I1t1, X, Y)) + (param["b1"],)
        t2 = None
        if I2t1 is not None:
            snd = param.get("sndTerm", "rgb") == "gradmag"
            w2 = orc.BilinInterp_2d(I2t1, X, Y)
            t2 = (orc.SndDerivatives5(I2t0, w2) if snd else orc.FstDerivatives5(I2t0, w2)) + (param["b2"],)
        dU, dV = np.zeros_like(U), np.zeros_like(V)
        for k in range(param["secondLoop"]):
            MGd, CuGd, CvGd, DuGd, DvGd = flow_assemble(t1, t2, dU, dV, param["alpha"])
            if Us is not None:
                c, d = apriori_slices(Us, U, dU, param["gammaS"], param["alpha"], as_diff, u_double and first == 0, k == 0)
                CuGd, DuGd = nan_append(CuGd, c), nan_append(DuGd, d)
            if Vs is not None:
                c, d = apriori_slices(Vs, V, dV, param["gammaS"], param["alpha"], as_diff, u_double and first == 0, k == 0)
                CvGd, DvGd = nan_append(CvGd, c), nan_append(DvGd, d)
            if param["diffusion"] == "flow":
                w8 = single(ad_diff_weights((((U + dU).astype(F32) + V).astype(F32) + dV).astype(F32), param["quantile"])[0])
            dU, dV = orc.Oflow_sor_llin8_2d(U, V, dU, dV, MGd, CuGd, CvGd, DuGd, DvGd, *w8, param["iter"], param["omega"],
                                            solver=param["solver"], order=param["order"])
        U, V = median3_sum(U, dU), median3_sum(V, dV)
    return U, V


# ---------------------------------------------------------------------------------------------------------
# TV denoising with 4 neighbours (matlab/denoising/TVdenoise4.m)
# ---------------------------------------------------------------------------------------------------------
def tv4_diff_weights(D):
    """[wW wN wE wS] = DiffWeights(D) (TVdenoise4.m:116-156), single throughout; D [nrows, ncols(, F)].  One plane each
    (the caller's repmat over the frames is left to the caller)."""
    D3 = _c3(D)
    P = np.pad(D3, ((1, 1), (0, 0), (0, 0)), mode="edge")
    ver = (F32(0.25) * P[:-2] - F32(0.25) * P[2:]).astype(F32)          # imfilter(D, [0.25 0 -0.25]', 'replicate')
    P = np.pad(D3, ((0, 0), (1, 1), (0, 0)), mode="edge")
    hor = (F32(0.25) * P[:, :-2] - F32(0.25) * P[:, 2:]).astype(F32)
    sh = lambda A, di, dj: np.roll(np.roll(A, di, axis=0), dj, axis=1)

    def w(di, dj, cross):
        a = (sh(D3, di, dj) - D3).astype(F32)
        b = (cross + sh(cross, di, dj)).astype(F32)
        m = ((a * a).astype(F32) + (b * b).astype(F32)).astype(F32).max(axis=2)
        return (F32(1) / np.sqrt((m + F32(0.00001)).astype(F32))).astype(F32)

    wW, wE, wN, wS = w(0, 1, ver), w(0, -1, ver), w(1, 0, hor), w(-1, 0, hor)
    wW[:, 0] = 0; wE[:, -1] = 0; wN[0, :] = 0; wS[-1, :] = 0
    return wW, wN, wE, wS


def tv4_assemble(Iout, Iin, alpha):
    """TRACE, B and single(alpha*w) of one outer iteration (TVdenoise4.m:84-98); returns TRACE, B, [aW, aN, aE, aS]."""
    shape3 = _c3(Iout).shape
    wW, wN, wE, wS = tv4_diff_weights(Iout)
    tot = (((wW + wN).astype(F32) + wE).astype(F32) + wS).astype(F32)
    diff = (_c3(Iout) - _c3(Iin)).astype(F32)
    psi = (F32(1.0) / np.sqrt(diff * diff + F32(2.220446049250313e-16))).astype(F32)
    TRACE = (psi + (F32(alpha) * tot).astype(F32)[:, :, None]).astype(F32)
    B = (psi * _c3(Iin)).astype(F32)
    ws = [np.asfortranarray(np.repeat((F32(alpha) * a).astype(F32)[:, :, None], shape3[2], axis=2).reshape(Iout.shape)) for a in (wW, wN, wE, wS)]
    return np.asfortranarray(TRACE.reshape(Iout.shape)), np.asfortranarray(B.reshape(Iout.shape)), ws


def tv4_level(orc, Iin, Iout, param):
    """The lagged-diffusivity loop of one scale (TVdenoise4.m:82-103)."""
    X = np.asfortranarray(Iout.astype(F32))
    for _ in range(param["outer_iter"] + 1):
        TRACE, B, (aW, aN, aE, aS) = tv4_assemble(X, Iin, param["alpha"])
        X = orc.PDEsolver4(X, TRACE, B, aW, aN, aE, aS, param["inner_iter"], param["omega"], solver=param["solver"], order=param["order"])
    return X


# ---------------------------------------------------------------------------------------------------------
# Symmetric stereo disparity (matlab/disparity/DispEminND_llin_sym_2D.m)
# ---------------------------------------------------------------------------------------------------------
SYM_PRE = np.array([0.037659, 0.249724, 0.439911, 0.249724, 0.037659], dtype=np.float64)       # prefilter_spa (:71)
SYM_D1F = np.array([-0.104550, -0.292315, 0.0, 0.292315, 0.104550], dtype=np.float64)          # O_dx (:73) flipped by 'conv'


def sym_warp_flow(U, Uq):
    """interp2(X, Y, U, X+Uq, Y) (:140-141), our statement of interp2's default: linear, NaN outside the grid.  The query
    rows are the grid rows, so only x is interpolated: v0*(1-s) + v1*s in double, at xq == cols the last column itself."""
    nrows, ncols = U.shape
    Ud = U.astype(np.float64)
    xq = np.arange(1, ncols + 1, dtype=np.float64)[None, :] + Uq.astype(np.float64)
    ok = (xq >= 1) & (xq <= ncols)                                   # NaN queries fail both
    j0 = np.where(ok, np.floor(np.where(ok, xq, 1.0)), 1.0)
    j0 = np.minimum(j0, ncols - 1)
    s = np.where(ok, xq, 1.0) - j0
    j0 = j0.astype(np.int64) - 1
    ii = np.arange(nrows)[:, None]
    out = Ud[ii, j0] * (1.0 - s) + Ud[ii, np.minimum(j0 + 1, ncols - 1)] * s
    return np.where(ok, out, np.nan)


def _conv5_f64(A, k, axis):
    pad = [(0, 0)] * A.ndim
    pad[axis] = (2, 2)
    P = np.pad(A, pad, mode="edge")
    n = A.shape[axis]
    sl = lambda t: tuple(slice(t, t + n) if ax == axis else slice(None) for ax in range(A.ndim))
    s = k[0] * P[sl(0)]
    for t in range(1, 5):
        s = s + k[t] * P[sl(t)]
    return s


def sym_flow_terms(U, Uw):
    """Udt, Udx, CuS, DuS of one direction (:156-175), double: Udt = (U+Uw)*0.5, Udx = the prefiltered x-derivative of the
    warped other-view disparity, CuS = Udt.*(1+Udx), DuS = 1 + Udx + Udx + Udx.*Udx."""
    Udt = (U.astype(np.float64) + Uw) * 0.5
    Udx = _conv5_f64(_conv5_f64(Uw, SYM_PRE, 0), SYM_D1F, 1)
    CuS = Udt * (1.0 + Udx)
    DuS = ((1.0 + Udx) + Udx) + Udx * Udx
    return Udt, Udx, CuS, DuS


def sym_assemble(d, sym, dU, param, channels, sr_diff, first):
    """CuG, DuG of one view (:189-222).  d = (Idt, Idx, Idxt, Idyt, Idxx, Idxy) single [.., C]; sym = (Udt, Udx, CuS, DuS) double.
    first: dU is still the double zeros of :177-178, so the symmetry weights are evaluated in double; afterwards dU is the
    solver's single output and MATLAB's `single op double -> single` makes them single."""
    Idt, Idx, Idxt, Idyt, Idxx, Idxy = [a if a.ndim == 3 else a[:, :, None] for a in d]
    Udt, Udx, CuS, DuS = sym
    b1, b2, alpha = F32(param["b1"]), F32(param["b2"]), F32(param["alpha"])
    du = dU.astype(F32)[:, :, None]
    r1, r2, r3 = Idt - Idx * du, Idxt - Idxx * du, Idyt - Idxy * du
    opnorm = b1 * (r1 * r1) + b2 * ((r2 * r2) + (r3 * r3))
    gD = (F32(1) / (alpha * np.sqrt(opnorm + F32(0.00001)))).astype(F32)
    CuD = (b1 * Idt) * Idx + b2 * (Idxt * Idxx + Idyt * Idxy)              # :165-168 (constant in the inner loop)
    DuD = (b1 * Idx) * Idx + b2 * (Idxx * Idxx + Idxy * Idxy)
    kS, sr2 = channels * param["beta"] / param["alpha"], sr_diff ** 2
    if first:
        du64 = dU.astype(np.float64)
        sn = (du64 + Udt) + Udx * du64
        gS = kS / (1.0 + (sn * sn) / sr2)
        cS, dS = ((-gS) * CuS).astype(F32), (gS * DuS).astype(F32)
    else:
        du32 = dU.astype(F32)
        sn = ((du32 + Udt.astype(F32)).astype(F32) + (Udx.astype(F32) * du32).astype(F32)).astype(F32)
        gS = (F32(kS) / (F32(1) + ((sn * sn).astype(F32) / F32(sr2)).astype(F32)).astype(F32)).astype(F32)
        cS, dS = ((-gS) * CuS.astype(F32)).astype(F32), (gS * DuS.astype(F32)).astype(F32)
    outs = []
    for data, s_ in (((gD * CuD).astype(F32), cS), ((gD * DuD).astype(F32), dS)):   # sum(cat(3, ...), 3): slices in order, single
        acc = data[:, :, 0]
        for c in range(1, data.shape[2]):
            acc = (acc + data[:, :, c]).astype(F32)
        outs.append(np.asfortranarray((acc + s_).astype(F32)))
    return outs


def disp_sym_level(orc, It0, It1, U0, U1, param, sr_diff):
    """One pyramid level (:116-262, without imresize).  U0 = U(:,:,1), U1 = U(:,:,2); sr_diff = 2*(1/scl_factor)^-(scl-1)."""
    U0, U1 = np.asfortranarray(U0, dtype=F32), np.asfortranarray(U1, dtype=F32)
    channels = It0.shape[2] if It0.ndim == 3 else 1
    Z = np.zeros_like(U0)
    for _ in range(param["firstLoop"]):
        X1, Y = flow_coords(U1, Z)
        X0, _ = flow_coords(U0, Z)
        It0w, It1w = orc.BilinInterp_2d(It0, X1, Y), orc.BilinInterp_2d(It1, X0, Y)      # :134-135
        U0w, U1w = sym_warp_flow(U0, U1), sym_warp_flow(U1, U0)                          # :140-141
        f0, s0 = orc.FstDerivatives5(It0, It1w), orc.SndDerivatives5(It0, It1w)          # Idt0 Idx1 Idy1 / Idxt0 Idyt0 Idxx1 Idyy1 Idxy1
        f1, s1 = orc.FstDerivatives5(It1, It0w), orc.SndDerivatives5(It1, It0w)
        d0 = (f0[0], f0[1], s0[0], s0[1], s0[2], s0[4])
        d1 = (f1[0], f1[1], s1[0], s1[1], s1[2], s1[4])
        sym0, sym1 = sym_flow_terms(U0, U1w), sym_flow_terms(U1, U0w)                    # :156-175
        dU0, dU1 = np.zeros_like(U0), np.zeros_like(U1)
        for k in range(param["secondLoop"]):
            CuG0, DuG0 = sym_assemble(d0, sym0, dU0, param, channels, sr_diff, k == 0)
            CuG1, DuG1 = sym_assemble(d1, sym1, dU1, param, channels, sr_diff, k == 0)
            w0 = orc.DdiffWeights((U0 + dU0).astype(F32), 0.00001)
            w1 = orc.DdiffWeights((U1 + dU1).astype(F32), 0.00001)
            dU0, dU1 = orc.Disp_sor_llin_sym4_2d(U0, dU0, CuG0, DuG0, *w0, U1, dU1, CuG1, DuG1, *w1, param["iter"], param["omega"],
                                                 solver=param["solver"], order=param["order"])
        U0, U1 = median3_sum(U0, dU0), median3_sum(U1, dU1)
    return U0, U1
