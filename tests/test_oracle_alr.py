"""CPU: known-answer checks of the oracle's alternating line relaxation (solver = 2).

Same idea as test_oracle_math.py: the reference cannot be built here, so what is pinned is the
mathematics.  The line solvers treat EVERY pixel as an unknown and drop the neighbours that fall
outside the image (Neumann); their fixed point must be the float64 sparse direct solution of exactly
that system, in the reference's line order and in the zebra order alike.
"""
import numpy as np
import pytest
import scipy.sparse as sp
import scipy.sparse.linalg as spla

import problems as pb

TOL = 1e-4
OFF = {"W": (0, -1), "N": (-1, 0), "E": (0, 1), "S": (1, 0), "NW": (-1, -1), "NE": (-1, 1), "SE": (1, 1), "SW": (1, -1)}


def rms(a, b):
    return float(np.sqrt(np.mean((np.asarray(a, np.float64) - np.asarray(b, np.float64)) ** 2)))


def neumann_operator(weights, nrows, ncols, extra_diag=None, fixed_diag=None, unknown=None):
    """A = diag - sum_{d present} w_d * shift_d over the pixels flagged `unknown` (default: all).

    diag = sum of the PRESENT weights (+ extra_diag), or `fixed_diag` where given (TRACE).
    Returns (A, idx, boundary) where boundary[(r)] collects known-neighbour contributions as
    (row, weight, (i,j)) triples for pixels outside `unknown`.
    """
    if unknown is None:
        unknown = np.ones((nrows, ncols), dtype=bool)
    idx = -np.ones((nrows, ncols), dtype=np.int64)
    idx[unknown] = np.arange(int(unknown.sum()))
    rows, cols, vals, known = [], [], [], []
    for i in range(nrows):
        for j in range(ncols):
            if not unknown[i, j]:
                continue
            r = idx[i, j]
            diag = 0.0
            for name, w in weights.items():
                di, dj = OFF[name]
                ii, jj = i + di, j + dj
                if not (0 <= ii < nrows and 0 <= jj < ncols):
                    continue
                diag += float(w[i, j])
                if unknown[ii, jj]:
                    rows.append(r); cols.append(idx[ii, jj]); vals.append(-float(w[i, j]))
                else:
                    known.append((r, float(w[i, j]), (ii, jj)))
            if fixed_diag is not None:
                diag = float(fixed_diag[i, j])
            elif extra_diag is not None:
                diag += float(extra_diag[i, j])
            rows.append(r); cols.append(r); vals.append(diag)
    n = int(unknown.sum())
    return sp.csr_matrix((vals, (rows, cols)), shape=(n, n)), idx, known


def llin_rhs(C, U, weights, nrows, ncols):
    """C + sum_{d present} w_d (U_d - U_c)."""
    rhs = C.astype(np.float64).copy()
    U = U.astype(np.float64)
    for name, w in weights.items():
        di, dj = OFF[name]
        for i in range(nrows):
            for j in range(ncols):
                ii, jj = i + di, j + dj
                if 0 <= ii < nrows and 0 <= jj < ncols:
                    rhs[i, j] += float(w[i, j]) * (U[ii, jj] - U[i, j])
    return rhs


def solve_coupled(Au, Av, M, bu, bv, shape):
    Mi = sp.diags(M.astype(np.float64).ravel())
    A = sp.bmat([[Au, Mi], [Mi, Av]]).tocsc()
    x = spla.spsolve(A, np.concatenate([bu.ravel(), bv.ravel()]))
    n = shape[0] * shape[1]
    return x[:n].reshape(shape), x[n:].reshape(shape)


W4 = ("wW", "wN", "wE", "wS")
W8 = ("wW", "wNW", "wN", "wNE", "wE", "wSE", "wS", "wSW")


def wdict(p, names):
    return {k[1:]: p[k] for k in names}


@pytest.mark.parametrize("shape", [(14, 11), (9, 16)])
def test_alr_elin4_fixed_point_is_the_neumann_system(oracle, shape):
    nrows, ncols = shape
    p = pb.elin4(401, nrows, ncols)
    w = wdict(p, W4)
    Au, _, _ = neumann_operator(w, nrows, ncols, extra_diag=p["Du"])
    Av, _, _ = neumann_operator(w, nrows, ncols, extra_diag=p["Dv"])
    Ue, Ve = solve_coupled(Au, Av, p["M"], p["Cu"].astype(np.float64), p["Cv"].astype(np.float64), shape)
    for order in (oracle.LEX, oracle.COLOUR):
        U, V = oracle.oflow_alr_elin4(*p.values(), 300, 1.0, order)
        assert rms(U, Ue) < TOL and rms(V, Ve) < TOL
    U, V = oracle.oflow_alr_elin4(*p.values(), 300, 1.3, oracle.LEX)  # over-relaxed: same fixed point
    assert rms(U, Ue) < TOL and rms(V, Ve) < TOL


def test_alr_llin4_and_disparity_fixed_points(oracle):
    nrows, ncols = 13, 12
    p = pb.llin4(402, nrows, ncols)
    w = wdict(p, W4)
    Au, _, _ = neumann_operator(w, nrows, ncols, extra_diag=p["Du"])
    Av, _, _ = neumann_operator(w, nrows, ncols, extra_diag=p["Dv"])
    bu, bv = llin_rhs(p["Cu"], p["U"], w, nrows, ncols), llin_rhs(p["Cv"], p["V"], w, nrows, ncols)
    dUe, dVe = solve_coupled(Au, Av, p["M"], bu, bv, (nrows, ncols))
    for order in (oracle.LEX, oracle.COLOUR):
        dU, dV = oracle.oflow_alr_llin4(*p.values(), 300, 1.0, order)
        assert rms(dU, dUe) < TOL and rms(dV, dVe) < TOL

    q = pb.disp4(403, nrows, ncols)
    w = wdict(q, W4)
    A, _, _ = neumann_operator(w, nrows, ncols, extra_diag=q["Du"])
    x = spla.spsolve(A.tocsc(), llin_rhs(q["Cu"], q["U"], w, nrows, ncols).ravel()).reshape(nrows, ncols)
    for order in (oracle.LEX, oracle.COLOUR):
        assert rms(oracle.disp_alr_llin4(*q.values(), 300, 1.0, order), x) < TOL


def test_alr_llin8_uses_the_diagonal_weights(oracle):
    nrows, ncols = 12, 13
    p = pb.llin8(404, nrows, ncols)
    w = wdict(p, W8)
    Au, _, _ = neumann_operator(w, nrows, ncols, extra_diag=p["Du"])
    Av, _, _ = neumann_operator(w, nrows, ncols, extra_diag=p["Dv"])
    bu, bv = llin_rhs(p["Cu"], p["U"], w, nrows, ncols), llin_rhs(p["Cv"], p["V"], w, nrows, ncols)
    dUe, dVe = solve_coupled(Au, Av, p["M"], bu, bv, (nrows, ncols))
    for order in (oracle.LEX, oracle.COLOUR):
        dU, dV = oracle.oflow_alr_llin8(*p.values(), 400, 1.0, order)
        assert rms(dU, dUe) < TOL and rms(dV, dVe) < TOL
    # and it is NOT the 4-neighbour answer: the diagonals matter
    d4 = oracle.oflow_alr_llin4(*[p[k] for k in ("U", "V", "dU", "dV", "M", "Cu", "Cv", "Du", "Dv", "wW", "wN", "wE", "wS")],
                                400, 1.0, oracle.LEX)
    assert rms(d4[0], dUe) > 10 * TOL


def test_alr_pde4_fixed_point_multiframe(oracle):
    nrows, ncols, F = 11, 12, 2
    p = pb.pde4(405, nrows, ncols, F)
    for order in (oracle.LEX, oracle.COLOUR):
        X = oracle.pde_alr4(*p.values(), 200, 1.0, order)
        for k in range(F):
            w = {n[1:]: p[n][:, :, k] for n in W4}
            A, _, _ = neumann_operator(w, nrows, ncols, fixed_diag=p["TRACE"][:, :, k])
            x = spla.spsolve(A.tocsc(), p["B"][:, :, k].astype(np.float64).ravel()).reshape(nrows, ncols)
            assert rms(X[:, :, k], x) < TOL


def test_alr_pde8_is_one_iteration_of_interior_lines(oracle):
    nrows, ncols = 10, 12
    p = pb.pde8(406, nrows, ncols)
    one = oracle.pde_alr8(*p.values(), 1, 1.0)
    assert pb.bit_equal(one, oracle.pde_alr8(*p.values(), 7, 1.0))   # `iter` is ignored (pdeSolvers.c:362)
    assert pb.bit_equal(one, oracle.pde_alr8(*p.values(), 0, 1.0))
    # corners are on no interior line: never touched
    for c in ((0, 0), (0, -1), (-1, 0), (-1, -1)):
        assert one[c] == p["X"][c]
    # repeated application converges to the system with the four corners as data
    X = p["X"]
    q = dict(p)
    for _ in range(300):
        q["X"] = X
        X = oracle.pde_alr8(*q.values(), 1, 1.0)
    unknown = np.ones((nrows, ncols), dtype=bool)
    unknown[0, 0] = unknown[0, -1] = unknown[-1, 0] = unknown[-1, -1] = False
    A, idx, known = neumann_operator(wdict(p, W8), nrows, ncols, fixed_diag=p["TRACE"], unknown=unknown)
    rhs = p["B"].astype(np.float64)[unknown].copy()
    for r, wgt, (ii, jj) in known:
        rhs[r] += wgt * float(p["X"][ii, jj])
    x = spla.spsolve(A.tocsc(), rhs)
    assert rms(X[unknown], x) < TOL


def test_alr_column_pass_is_an_exact_tridiagonal_solve(oracle):
    """With no horizontal coupling and omega = 1, one iteration solves every column exactly."""
    nrows, ncols = 40, 6
    p = pb.disp4(407, nrows, ncols)
    p["wW"][:] = 0.0
    p["wE"][:] = 0.0
    out = oracle.disp_alr_llin4(*p.values(), 1, 1.0)
    w = wdict(p, W4)
    A, _, _ = neumann_operator(w, nrows, ncols, extra_diag=p["Du"])
    x = spla.spsolve(A.tocsc(), llin_rhs(p["Cu"], p["U"], w, nrows, ncols).ravel()).reshape(nrows, ncols)
    assert rms(out, x) < 2e-6


def test_alr_nan_data_term_drops_the_data_row(oracle):
    """isnan(Cu) -> no Du on the diagonal, no Cu and no M*V on the right-hand side (opticalflowSolvers.c:1921)."""
    nrows, ncols = 12, 10
    p = pb.elin4(408, nrows, ncols)
    q = {k: v.copy() for k, v in p.items()}
    r = {k: v.copy() for k, v in p.items()}
    mask = np.zeros((nrows, ncols), dtype=bool)
    mask[3:6, 2:7] = True
    for k in ("Cu", "Cv"):
        q[k][mask] = np.nan
    for k in ("M", "Cu", "Cv", "Du", "Dv"):
        r[k][mask] = 0.0
    a = oracle.oflow_alr_elin4(*q.values(), 3, 1.2)
    b = oracle.oflow_alr_elin4(*r.values(), 3, 1.2)
    assert np.isfinite(a[0]).all() and np.allclose(a[0], b[0], atol=1e-5) and np.allclose(a[1], b[1], atol=1e-5)
    # NaN in Du alone is NOT special-cased by the line solvers: it poisons the line
    s = {k: v.copy() for k, v in p.items()}
    s["Du"][4, 4] = np.nan
    assert np.isnan(oracle.oflow_alr_elin4(*s.values(), 1, 1.0)[0]).any()


def test_alr_zero_iterations_and_gateway_semantics(oracle):
    p = pb.elin4(409, 9, 8)
    U, V = oracle.oflow_alr_elin4(*p.values(), 0, 1.5)
    assert pb.bit_equal(U, p["U"]) and pb.bit_equal(V, p["V"])
    g = oracle.Oflow_sor_elin4_2d(*p.values(), 0, 1.5, solver=2)
    assert not g[0].any() and not g[1].any()                     # iter <= 0: zero outputs (Oflow_sor_elin4_2d.c:341)
    g = oracle.Oflow_sor_elin4_2d(*p.values(), 2, 1.5, solver=2, nargout=4)
    ref = oracle.oflow_alr_elin4(*p.values(), 2, 1.5)
    assert pb.bit_equal(g[0], ref[0]) and pb.bit_equal(g[1], ref[1])
    RU, RV = oracle.oflow_res_elin4(*p.values())
    assert pb.bit_equal(g[2], RU) and pb.bit_equal(g[3], RV)   # residuals of the INPUT iterate


def test_alr_orders_differ_at_finite_iter_but_zebra_lines_are_independent(oracle):
    p = pb.elin4(410, 16, 15)
    a = oracle.oflow_alr_elin4(*p.values(), 2, 1.4, oracle.LEX)
    b = oracle.oflow_alr_elin4(*p.values(), 2, 1.4, oracle.COLOUR)
    assert not pb.bit_equal(a[0], b[0])
    assert rms(a[0], b[0]) < 0.5
