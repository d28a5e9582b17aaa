"""Seeded synthetic inputs for every gateway on the hot path (shared by the golden-vector
generator, the CPU tests, the GPU parity tests, smoke() and bench.py).

Shapes follow MATLAB: [nrows, ncols] or [nrows, ncols, nframes], float32, column-major.
The four (eight) weight planes of the parity problems are drawn INDEPENDENTLY on purpose: a kernel that
reads the wrong weight plane or the right plane at a neighbour's position cannot hide behind a symmetry.
The price: wE(i,j) != wW(i,j+1), the operator is not symmetric and SOR at omega = 1.9 does not converge on
these planes -- a few tens of sweeps stay finite, which is all a bit-for-bit comparison needs.  Problems
that must CONVERGE (known-answer tests, bench.py) pass symmetric=True: wE[:, :-1] = wW[:, 1:],
wS[:-1, :] = wN[1:, :], which is what the drivers' OPdiffWeights / DdiffWeights produce.  `nan_frac`
laces the data terms with NaN the way out-of-range warps do in the drivers.
"""
import numpy as np


def _f(a):
    return np.asfortranarray(a.astype(np.float32))


def _plane(rng, shape, lo, hi):
    return _f(rng.uniform(lo, hi, size=shape))


def _weights(rng, shape, n=4):
    return [_plane(rng, shape, 0.5, 5.0) for _ in range(n)]


def symmetrize(wW, wN, wE, wS):
    """Make (wW, wN, wE, wS) the weights of a symmetric operator: the east weight of a pixel is the west weight of
    its east neighbour, the south weight the north weight of its south neighbour (in place on wE, wS)."""
    wE[:, :-1] = wW[:, 1:]
    wS[:-1, :] = wN[1:, :]
    return wW, wN, wE, wS


def _lace(rng, arrays, frac):
    """Put NaN at the same random pixels of every array in `arrays` (in place)."""
    if frac <= 0:
        return
    mask = rng.uniform(size=arrays[0].shape) < frac
    for a in arrays:
        a[mask] = np.nan


def oflow_coeffs(rng, nrows, ncols, nframes=1, nan_frac=0.0, nan_mode="all", amp=1.5):
    """M, Cu, Cv, Du, Dv with the structure of a motion tensor: Du=a^2, Dv=b^2, M=0.9ab; |a|,|b| <= amp (image gradients)."""
    shape = (nrows, ncols) if nframes == 1 else (nrows, ncols, nframes)
    a, b = rng.uniform(-amp, amp, size=shape), rng.uniform(-amp, amp, size=shape)
    c = rng.uniform(-1.0, 1.0, size=shape)
    M, Du, Dv = _f(0.9 * a * b), _f(a * a + 0.05), _f(b * b + 0.05)
    Cu, Cv = _f(-a * c), _f(-b * c)
    if nan_mode == "all":
        _lace(rng, [M, Cu, Cv, Du, Dv], nan_frac)
    elif nan_mode == "C":
        _lace(rng, [Cu], nan_frac)
        _lace(rng, [Cv], nan_frac)
    elif nan_mode == "D":
        _lace(rng, [Du], nan_frac)
        _lace(rng, [Dv], nan_frac)
    return M, Cu, Cv, Du, Dv


def elin4(seed, nrows, ncols, nframes=1, nan_frac=0.0, nan_mode="all", symmetric=False, amp=1.5):
    rng = np.random.default_rng(seed)
    U, V = _plane(rng, (nrows, ncols), -1, 1), _plane(rng, (nrows, ncols), -1, 1)
    M, Cu, Cv, Du, Dv = oflow_coeffs(rng, nrows, ncols, nframes, nan_frac, nan_mode, amp)
    wW, wN, wE, wS = _weights(rng, (nrows, ncols))
    if symmetric:
        symmetrize(wW, wN, wE, wS)
    return dict(U=U, V=V, M=M, Cu=Cu, Cv=Cv, Du=Du, Dv=Dv, wW=wW, wN=wN, wE=wE, wS=wS)


def llin4(seed, nrows, ncols, nframes=1, nan_frac=0.0, nan_mode="all"):
    rng = np.random.default_rng(seed)
    p = elin4(seed + 1000, nrows, ncols, nframes, nan_frac, nan_mode)
    dU, dV = _plane(rng, (nrows, ncols), -0.5, 0.5), _plane(rng, (nrows, ncols), -0.5, 0.5)
    return dict(U=p["U"], V=p["V"], dU=dU, dV=dV, M=p["M"], Cu=p["Cu"], Cv=p["Cv"], Du=p["Du"], Dv=p["Dv"],
                wW=p["wW"], wN=p["wN"], wE=p["wE"], wS=p["wS"])


def llin8(seed, nrows, ncols, nframes=1, nan_frac=0.0):
    rng = np.random.default_rng(seed + 7)
    p = llin4(seed, nrows, ncols, nframes, nan_frac)
    d = _weights(rng, (nrows, ncols))
    return dict(U=p["U"], V=p["V"], dU=p["dU"], dV=p["dV"], M=p["M"], Cu=p["Cu"], Cv=p["Cv"], Du=p["Du"],
                Dv=p["Dv"], wW=p["wW"], wNW=d[0], wN=p["wN"], wNE=d[1], wE=p["wE"], wSE=d[2], wS=p["wS"], wSW=d[3])


def disp4(seed, nrows, ncols, nan_frac=0.0):
    rng = np.random.default_rng(seed)
    U, dU = _plane(rng, (nrows, ncols), -3, 3), _plane(rng, (nrows, ncols), -0.5, 0.5)
    Cu, Du = _plane(rng, (nrows, ncols), -1, 1), _plane(rng, (nrows, ncols), 0.05, 2.0)
    _lace(rng, [Cu, Du], nan_frac)
    wW, wN, wE, wS = _weights(rng, (nrows, ncols))
    return dict(U=U, dU=dU, Cu=Cu, Du=Du, wW=wW, wN=wN, wE=wE, wS=wS)


def dispsym4(seed, nrows, ncols, nan_frac=0.0):
    """Inputs of Disp_sor_llin_sym4_2d: two independent disparity problems (left->right, right->left)."""
    a, b = disp4(seed, nrows, ncols, nan_frac), disp4(seed + 500, nrows, ncols, nan_frac)
    out = {k + "0": v for k, v in a.items()}
    out.update({k + "1": v for k, v in b.items()})
    return out


def pde4(seed, nrows, ncols, nframes=1, nan_frac=0.0):
    rng = np.random.default_rng(seed)
    shape = (nrows, ncols) if nframes == 1 else (nrows, ncols, nframes)
    X, B = _plane(rng, shape, 0, 1), _plane(rng, shape, 0, 1)
    w = _weights(rng, shape)
    TRACE = _f(1.0 + sum(w))
    _lace(rng, [TRACE], nan_frac)
    return dict(X=X, TRACE=TRACE, B=B, wW=w[0], wN=w[1], wE=w[2], wS=w[3])


def pde8(seed, nrows, ncols, nframes=1, nan_frac=0.0):
    rng = np.random.default_rng(seed)
    shape = (nrows, ncols) if nframes == 1 else (nrows, ncols, nframes)
    X, B = _plane(rng, shape, 0, 1), _plane(rng, shape, 0, 1)
    w = _weights(rng, shape, 8)
    TRACE = _f(1.0 + sum(w))
    _lace(rng, [TRACE], nan_frac)
    return dict(X=X, TRACE=TRACE, B=B, wW=w[0], wNW=w[1], wN=w[2], wNE=w[3], wE=w[4], wSE=w[5], wS=w[6], wSW=w[7])


def diffweights(seed, nrows, ncols, nframes=1):
    rng = np.random.default_rng(seed)
    shape = (nrows, ncols) if nframes == 1 else (nrows, ncols, nframes)
    return dict(D=_plane(rng, shape, -4, 4))


def warp(seed, nrows, ncols, nframes=1, max_disp=3.0, special=True):
    """Identity grid plus a random displacement; `special` adds exact-integer, edge, far
    out-of-range, NaN and Inf coordinates."""
    rng = np.random.default_rng(seed)
    shape = (nrows, ncols) if nframes == 1 else (nrows, ncols, nframes)
    Iin = _plane(rng, shape, 0, 1)
    jj, ii = np.meshgrid(np.arange(1, ncols + 1), np.arange(1, nrows + 1))
    X = (jj + rng.uniform(-max_disp, max_disp, size=(nrows, ncols))).astype(np.float32)
    Y = (ii + rng.uniform(-max_disp, max_disp, size=(nrows, ncols))).astype(np.float32)
    if special and nrows >= 6 and ncols >= 6:
        X[0, 0], Y[0, 0] = 1.0, 1.0                        # first pixel exactly
        X[1, 0], Y[1, 0] = ncols, nrows                    # last pixel exactly (clamped +1 taps)
        X[2, 0], Y[2, 0] = ncols + 0.5, 2.0                # just past the last column: still "inside"
        X[3, 0], Y[3, 0] = ncols + 1.0, 2.0                # out
        X[4, 0], Y[4, 0] = 0.999, 3.0                      # floor(X-1) = -1: out
        X[5, 0], Y[5, 0] = -1.0e6, 3.0                     # far out
        X[0, 1], Y[0, 1] = np.nan, 2.0
        X[1, 1], Y[1, 1] = 2.0, np.inf
        X[2, 1], Y[2, 1] = -np.inf, 2.0
        X[3, 1], Y[3, 1] = 2.5, nrows + 0.25               # past the last row, inside the clamp band
        X[4, 1], Y[4, 1] = 3.0e9, 2.0                      # beyond 2^31
    return dict(Iin=Iin, X=_f(X), Y=_f(Y))


def image_pair(seed, nrows, ncols, nframes=1):
    """Two frames of a smooth-ish random texture (inputs of the Simoncelli derivative gateways)."""
    rng = np.random.default_rng(seed)
    shape = (nrows, ncols) if nframes == 1 else (nrows, ncols, nframes)
    It0 = _plane(rng, shape, 0, 1)
    It1 = _f(It0 + rng.uniform(-0.1, 0.1, size=shape))
    return dict(It0=It0, It1=It1)


def bit_equal(a, b):
    """Bitwise equality of float32 arrays, except that any NaN equals any NaN."""
    a, b = np.asarray(a, dtype=np.float32), np.asarray(b, dtype=np.float32)
    if a.shape != b.shape:
        return False
    na, nb = np.isnan(a), np.isnan(b)
    if not np.array_equal(na, nb):
        return False
    ok = ~na
    return bool(np.array_equal(np.ascontiguousarray(a[ok]).view(np.uint32), np.ascontiguousarray(b[ok]).view(np.uint32)))


def describe_mismatch(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    bad = ~((a == b) | (np.isnan(a) & np.isnan(b)))
    n = int(bad.sum())
    if n == 0:
        return "only sign-of-zero / shape differences"
    idx = np.argwhere(bad)
    first = tuple(int(v) for v in idx[0])
    d = np.abs(np.where(bad, a - b, 0.0))
    return "%d of %d differ; first at %s: %r vs %r; max |diff| %.3g; mismatch bbox %s..%s" % (
        n, a.size, first, a[first], b[first], float(np.nanmax(d)), idx.min(axis=0).tolist(), idx.max(axis=0).tolist())
