"""CPU: known-answer checks of the oracle that do not come from the oracle itself.

The reference ships no tests or golden data and cannot be built here, so the oracle's parity with it is
unpinned; these tests pin the oracle's MATHEMATICS instead: the relaxation converges to the solution of
the linear system the reference's formulas define (solved independently with a float64 sparse direct
solver), both sweep orderings reach that solution within the north-star tolerance (1e-4 RMS), the
residual/LHS operators agree with it, and the warp / weights match closed forms.
"""
import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla

import problems as pb

TOL = 1e-4  # north-star tolerance (RMS)


def rms(a, b):
    return float(np.sqrt(np.mean((np.asarray(a, np.float64) - np.asarray(b, np.float64)) ** 2)))


def interior_index(nrows, ncols):
    idx = -np.ones((nrows, ncols), dtype=np.int64)
    idx[1:-1, 1:-1] = np.arange((nrows - 2) * (ncols - 2)).reshape(nrows - 2, ncols - 2)
    return idx


def clamp_to_interior(i, j, nrows, ncols):
    """The border cell (i,j) replicates the nearest interior pixel (rows first, then columns)."""
    return min(max(i, 1), nrows - 2), min(max(j, 1), ncols - 2)


def assemble_scalar(diag, weights, offsets, nrows, ncols):
    """sum_d w_d*(x_c - x_d) style operator:  (diag) x_c - sum_d w_d x_d  with replicate borders."""
    idx = interior_index(nrows, ncols)
    rows, cols, vals = [], [], []
    for i in range(1, nrows - 1):
        for j in range(1, ncols - 1):
            r = idx[i, j]
            rows.append(r); cols.append(r); vals.append(float(diag[i, j]))
            for w, (di, dj) in zip(weights, offsets):
                ci, cj = clamp_to_interior(i + di, j + dj, nrows, ncols)
                rows.append(r); cols.append(idx[ci, cj]); vals.append(-float(w[i, j]))
    n = (nrows - 2) * (ncols - 2)
    return sp.csr_matrix((vals, (rows, cols)), shape=(n, n))


OFF4 = [(0, -1), (-1, 0), (0, 1), (1, 0)]                       # W, N, E, S
OFF8 = [(0, -1), (-1, -1), (-1, 0), (-1, 1), (0, 1), (1, 1), (1, 0), (1, -1)]  # W,NW,N,NE,E,SE,S,SW


def solve_oflow(p, nrows, ncols):
    """Fixed point of GS_SOR_elin4_2d: (sum w + Du) U - sum w_d U_d + M V = Cu, and the V twin."""
    w = [p[k].astype(np.float64) for k in ("wW", "wN", "wE", "wS")]
    s = sum(w)
    Au = assemble_scalar(s + p["Du"], w, OFF4, nrows, ncols)
    Av = assemble_scalar(s + p["Dv"], w, OFF4, nrows, ncols)
    Mi = sp.diags(p["M"][1:-1, 1:-1].astype(np.float64).ravel())
    A = sp.bmat([[Au, Mi], [Mi, Av]]).tocsc()
    b = np.concatenate([p["Cu"][1:-1, 1:-1].ravel(), p["Cv"][1:-1, 1:-1].ravel()]).astype(np.float64)
    x = spla.spsolve(A, b)
    n = (nrows - 2) * (ncols - 2)
    return x[:n].reshape(nrows - 2, ncols - 2), x[n:].reshape(nrows - 2, ncols - 2)


def test_elin4_converges_to_the_linear_system_both_orderings(oracle):
    nrows, ncols = 18, 15
    p = pb.elin4(301, nrows, ncols)
    Ue, Ve = solve_oflow(p, nrows, ncols)
    for order in (oracle.LEX, oracle.COLOUR):
        U, V = oracle.oflow_sor_elin4(*p.values(), 600, 1.5, order)
        assert rms(U[1:-1, 1:-1], Ue) < TOL and rms(V[1:-1, 1:-1], Ve) < TOL
        # border replicate: every border cell equals its nearest interior pixel
        assert np.array_equal(U[0, 1:-1], U[1, 1:-1]) and np.array_equal(U[1:-1, -1], U[1:-1, -2])
        assert U[0, 0] == U[1, 1] and U[-1, -1] == U[-2, -2]
    # and the two orderings agree with each other at convergence
    a = oracle.oflow_sor_elin4(*p.values(), 600, 1.5, oracle.LEX)
    b = oracle.oflow_sor_elin4(*p.values(), 600, 1.5, oracle.COLOUR)
    assert rms(a[0], b[0]) < TOL and rms(a[1], b[1]) < TOL


def test_residual_and_lhs_are_consistent_with_the_solver(oracle):
    nrows, ncols = 18, 15
    p = pb.elin4(302, nrows, ncols)
    U, V = oracle.oflow_sor_elin4(*p.values(), 600, 1.5, oracle.LEX)
    coef = [p[k] for k in ("M", "Cu", "Cv", "Du", "Dv", "wW", "wN", "wE", "wS")]
    RU, RV = oracle.oflow_res_elin4(U, V, *coef)
    assert np.abs(RU).max() < 5e-4 and np.abs(RV).max() < 5e-4
    # r = b - A x  on arbitrary x
    U0, V0 = p["U"], p["V"]
    RU, RV = oracle.oflow_res_elin4(U0, V0, *coef)
    AU, AV = oracle.oflow_lhs_elin4(U0, V0, p["M"], p["Du"], p["Dv"], p["wW"], p["wN"], p["wE"], p["wS"])
    assert np.allclose(RU[1:-1, 1:-1], (p["Cu"] - AU)[1:-1, 1:-1], atol=2e-4)
    assert np.allclose(RV[1:-1, 1:-1], (p["Cv"] - AV)[1:-1, 1:-1], atol=2e-4)


def test_llin4_with_zero_base_flow_is_elin4(oracle):
    """With U=V=0 the late-linearization update is arithmetically the early-linearization one."""
    p = pb.elin4(303, 21, 17, nan_frac=0.05)
    z = np.zeros_like(p["U"])
    coef = [p[k] for k in ("M", "Cu", "Cv", "Du", "Dv", "wW", "wN", "wE", "wS")]
    for order in (oracle.LEX, oracle.COLOUR):
        a = oracle.oflow_sor_elin4(p["U"], p["V"], *coef, 5, 1.9, order)
        b = oracle.oflow_sor_llin4(z, z, p["U"], p["V"], *coef, 5, 1.9, order)
        assert pb.bit_equal(a[0], b[0]) and pb.bit_equal(a[1], b[1])


def test_nan_data_terms_fall_back_to_pure_diffusion(oracle):
    """A NaN Cu/Du pixel is relaxed with the smoothness term only (opticalflowSolvers.c:118-130)."""
    p = pb.elin4(304, 12, 12)
    q = {k: v.copy() for k, v in p.items()}
    for k in ("M", "Cu", "Cv", "Du", "Dv"):
        q[k][:] = np.nan
    U, V = oracle.oflow_sor_elin4(*q.values(), 3, 1.0, oracle.LEX)
    assert np.isfinite(U).all() and np.isfinite(V).all()
    r = {k: v.copy() for k, v in p.items()}
    for k in ("M", "Cu", "Cv", "Du", "Dv"):
        r[k][:] = 0.0
    U2, V2 = oracle.oflow_sor_elin4(*r.values(), 3, 1.0, oracle.LEX)
    assert np.allclose(U, U2, atol=1e-6) and np.allclose(V, V2, atol=1e-6)


def test_disparity_converges_to_its_linear_system(oracle):
    nrows, ncols = 16, 19
    p = pb.disp4(311, nrows, ncols)
    w = [p[k].astype(np.float64) for k in ("wW", "wN", "wE", "wS")]
    A = assemble_scalar(sum(w) + p["Du"], w, OFF4, nrows, ncols)
    # right-hand side: Cu + sum_d w_d (U_d - U_c), U is read as stored (borders included, never replicated)
    U = p["U"].astype(np.float64)
    rhs = p["Cu"].astype(np.float64).copy()
    for wd, (di, dj) in zip(w, OFF4):
        rhs[1:-1, 1:-1] += wd[1:-1, 1:-1] * (U[1 + di:nrows - 1 + di, 1 + dj:ncols - 1 + dj] - U[1:-1, 1:-1])
    x = spla.spsolve(A.tocsc(), rhs[1:-1, 1:-1].ravel()).reshape(nrows - 2, ncols - 2)
    for order in (oracle.LEX, oracle.COLOUR):
        dU = oracle.disp_sor_llin4(*p.values(), 500, 1.5, order)
        assert rms(dU[1:-1, 1:-1], x) < TOL
    RU = oracle.disp_res_llin4(p["U"], oracle.disp_sor_llin4(*p.values(), 500, 1.5, oracle.LEX), *[p[k] for k in ("Cu", "Du", "wW", "wN", "wE", "wS")])
    assert np.abs(RU).max() < 5e-4


def test_pde_solvers_converge_to_their_linear_systems(oracle):
    nrows, ncols = 15, 17
    p4 = pb.pde4(321, nrows, ncols)
    A = assemble_scalar(p4["TRACE"], [p4[k] for k in ("wW", "wN", "wE", "wS")], OFF4, nrows, ncols)
    x = spla.spsolve(A.tocsc(), p4["B"][1:-1, 1:-1].astype(np.float64).ravel()).reshape(nrows - 2, ncols - 2)
    for order in (oracle.LEX, oracle.COLOUR):
        assert rms(oracle.pde_sor4(*p4.values(), 400, 1.3, order)[1:-1, 1:-1], x) < TOL
    p8 = pb.pde8(322, nrows, ncols)
    A = assemble_scalar(p8["TRACE"], [p8[k] for k in ("wW", "wNW", "wN", "wNE", "wE", "wSE", "wS", "wSW")], OFF8, nrows, ncols)
    x = spla.spsolve(A.tocsc(), p8["B"][1:-1, 1:-1].astype(np.float64).ravel()).reshape(nrows - 2, ncols - 2)
    for order in (oracle.LEX, oracle.COLOUR):
        X = oracle.pde_sor8(*p8.values(), 400, 1.3, order)
        assert rms(X[1:-1, 1:-1], x) < TOL
        assert X[0, 0] == X[1, 1] and X[0, -1] == X[1, -2] and np.array_equal(X[-1, 1:-1], X[-2, 1:-1])


def test_multiframe_pde_frames_are_independent(oracle):
    p = pb.pde8(323, 14, 13, nframes=3, nan_frac=0.05)
    X = oracle.pde_sor8(*p.values(), 4, 1.75, oracle.LEX)
    for k in range(3):
        Xk = oracle.pde_sor8(*[v[:, :, k] for v in p.values()], 4, 1.75, oracle.LEX)
        assert pb.bit_equal(X[:, :, k], Xk)


def test_warp_closed_forms(oracle):
    nrows, ncols = 9, 11
    rng = np.random.default_rng(331)
    I = np.asfortranarray(rng.uniform(0, 1, (nrows, ncols, 2)).astype(np.float32))
    jj, ii = np.meshgrid(np.arange(1, ncols + 1, dtype=np.float32), np.arange(1, nrows + 1, dtype=np.float32))
    assert pb.bit_equal(oracle.warp_bilinear(I, jj, ii), I)                       # identity
    out = oracle.warp_bilinear(I, jj + 1, ii)                                     # one column to the right
    assert pb.bit_equal(out[:, :-1], I[:, 1:]) and np.isnan(out[:, -1]).all()     # X = ncols+1 -> floor = ncols: out
    out = oracle.warp_bilinear(I, jj, ii - 1)                                     # one row up: row 0 samples y = -1
    assert np.isnan(out[0]).all() and pb.bit_equal(out[1:], I[:-1])
    out = oracle.warp_bilinear(I, jj + 0.5, ii)                                   # half-way between columns
    want = 0.5 * I[:, :-1].astype(np.float64) + 0.5 * I[:, 1:]
    assert np.allclose(out[:, :-1], want, atol=1e-6)
    assert pb.bit_equal(out[:, -1], I[:, -1])                                     # last column: +1 tap clamps to itself
    bad = jj.copy(); bad[2, 3] = np.nan; bad[4, 5] = -np.inf; bad[6, 7] = 1e30
    out = oracle.warp_bilinear(I, bad, ii)
    assert np.isnan(out[2, 3]).all() and np.isnan(out[4, 5]).all() and np.isnan(out[6, 7]).all()


def test_diffweights_closed_forms(oracle):
    nrows, ncols, eps = 8, 10, 1e-3
    const = np.full((nrows, ncols), 3.0, dtype=np.float32)
    wW, wN, wE, wS = oracle.diffweights6(const, eps)
    k = np.float32(1.0) / np.sqrt(np.float32(eps))
    assert np.allclose(wW[:, 1:], k) and not wW[:, 0].any()       # first column of wW is never written
    assert np.allclose(wN[1:, :], k) and not wN[0, :].any()
    assert np.allclose(wE[:, :-1], k) and not wE[:, -1].any()
    assert np.allclose(wS[:-1, :], k) and not wS[-1, :].any()
    # against a float64 evaluation of the formulas of imageDiffusionWeights.c
    D = pb.diffweights(341, nrows, ncols, nframes=2)["D"]
    wW, wN, wE, wS = oracle.diffweights6(D, eps)
    Dd = D.astype(np.float64)
    tW = np.zeros((nrows, ncols)); tN = np.zeros((nrows, ncols))
    for f in range(2):
        d = Dd[:, :, f]
        up, dn = np.vstack([d[:1], d[:-1]]), np.vstack([d[1:], d[-1:]])
        lf, rt = np.hstack([d[:, :1], d[:, :-1]]), np.hstack([d[:, 1:], d[:, -1:]])
        ver, hor = 0.25 * (up - dn), 0.25 * (lf - rt)
        tw = np.zeros_like(d); tw[:, 1:] = (d[:, 1:] - d[:, :-1]) ** 2 + (ver[:, 1:] + ver[:, :-1]) ** 2
        tn = np.zeros_like(d); tn[1:, :] = (d[1:, :] - d[:-1, :]) ** 2 + (hor[1:, :] + hor[:-1, :]) ** 2
        tW, tN = np.maximum(tW, tw), np.maximum(tN, tn)
    assert np.allclose(wW[:, 1:], 1 / np.sqrt(tW[:, 1:] + eps), rtol=2e-5)
    assert np.allclose(wN[1:, :], 1 / np.sqrt(tN[1:, :] + eps), rtol=2e-5)


def test_simoncelli_derivatives_closed_forms(oracle):
    """Ramp image: first derivatives = slope x filter gains away from the border; second derivatives and
    temporal terms vanish; a constant image has zero spatial derivatives everywhere (replicate border)."""
    nrows, ncols = 11, 14
    jj, ii = np.meshgrid(np.arange(ncols, dtype=np.float64), np.arange(nrows, dtype=np.float64))
    ramp = (2.0 * jj + 3.0 * ii).astype(np.float32)
    S = np.array([0.037659, 0.249724, 0.439911, 0.249724, 0.037659])
    D1 = np.array([-0.104550, -0.292315, 0.0, 0.292315, 0.104550])
    gain = S.sum() * (D1 * np.arange(-2, 3)).sum()
    Idt, Idx, Idy = oracle.FstDerivatives5(ramp, ramp)
    assert not Idt.any()
    assert np.allclose(Idx[2:-2, 2:-2], 2.0 * gain, rtol=1e-5) and np.allclose(Idy[2:-2, 2:-2], 3.0 * gain, rtol=1e-5)
    Idt, _, _ = oracle.FstDerivatives5(ramp, ramp + np.float32(1))
    assert np.allclose(Idt, -0.5)                                   # 0.5*It0 - 0.5*It1
    Idxt, Idyt, Idxx, Idyy, Idxy = oracle.SndDerivatives5(ramp, ramp)
    for a in (Idxt, Idyt):
        assert np.abs(a).max() == 0.0
    assert np.abs(Idxx[2:-2, 2:-2]).max() < 1e-3 and np.abs(Idyy[2:-2, 2:-2]).max() < 1e-3
    assert np.allclose(Idxy[2:-2, 2:-2], 0.0, atol=1e-4)
    const = np.full((nrows, ncols), 0.7, dtype=np.float32)
    _, Idx, Idy = oracle.FstDerivatives5(const, const)
    assert np.abs(Idx).max() < 1e-6 and np.abs(Idy).max() < 1e-6
    # against scipy's correlate1d with replicate borders (float64), the definition in prose
    import scipy.ndimage as ndi
    p = pb.image_pair(351, 16, 13)
    _, Idx, Idy = oracle.FstDerivatives5(p["It0"], p["It1"])
    I1 = p["It1"].astype(np.float64)
    want_x = ndi.correlate1d(ndi.correlate1d(I1, S, axis=0, mode="nearest"), D1, axis=1, mode="nearest")
    want_y = ndi.correlate1d(ndi.correlate1d(I1, S, axis=1, mode="nearest"), D1, axis=0, mode="nearest")
    assert np.allclose(Idx, want_x, atol=2e-6) and np.allclose(Idy, want_y, atol=2e-6)


def test_both_orderings_converge_at_omega_1_9_on_a_symmetric_problem(oracle):
    """What bench.py times: symmetric weights (as the drivers' diffusion weights are), image-gradient sized data terms
    (|Ix|,|Iy| <= 0.5), omega = 1.9, a frame >= 128x128.  Both orderings converge to the same fixed point; with
    independently drawn weights (the parity problems) neither does -- the operator is not symmetric there
    (tests/problems.py).  (The reference updates u and v of a pixel from each other's OLD value, a Jacobi step inside
    the pixel, so at omega = 1.9 it also needs the coupling M small against the diagonal: amp = 0.5.)"""
    nrows, ncols = 136, 150
    p = pb.elin4(311, nrows, ncols, symmetric=True, amp=0.5)
    runs = {}
    for name, order in (("lex", oracle.LEX), ("rb", oracle.COLOUR)):
        U, V = p["U"], p["V"]
        hist = []
        for _ in range(8):  # 8 x 100 sweeps
            prev = U
            U, V = oracle.oflow_sor_elin4(U, V, *[p[k] for k in ("M", "Cu", "Cv", "Du", "Dv", "wW", "wN", "wE", "wS")], 100, 1.9, order)
            hist.append(rms(U, prev))
        assert np.isfinite(U).all() and np.isfinite(V).all()
        assert hist[-1] < 1e-5, (name, hist)  # the iterate has stopped moving
        runs[name] = (U, V)
    assert rms(runs["lex"][0], runs["rb"][0]) < TOL and rms(runs["lex"][1], runs["rb"][1]) < TOL
    # ... and the independently drawn planes of the parity problems do blow up at this omega (finite at 20 sweeps only)
    q = pb.elin4(311, nrows, ncols, amp=0.5)
    with np.errstate(all="ignore"):
        U, V = oracle.oflow_sor_elin4(*q.values(), 400, 1.9, oracle.LEX)
    assert not (np.isfinite(U).all() and np.abs(U).max() < 1e3)


def test_openmp_red_black_equals_the_serial_colour_order(oracle):
    """The all-core CPU comparator of bench.py is the oracle's own colour order, threaded: bit-identical."""
    p = pb.elin4(312, 67, 91, nan_frac=0.02)
    want = oracle.oflow_sor_elin4(*p.values(), 5, 1.9, oracle.COLOUR)
    for threads in (1, 3, 0):
        U, V, used = oracle.oflow_sor_elin4_rb_omp(*p.values(), 5, 1.9, threads)
        assert pb.bit_equal(U, want[0]) and pb.bit_equal(V, want[1])
        assert used >= 1
