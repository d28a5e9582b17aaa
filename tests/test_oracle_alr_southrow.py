"""CPU: the south row of the late-linearisation line solvers DIVIDES in its forward elimination.

Every line function of the reference multiplies the middle elements by div = 1/(b - cp'a), except
southRow_llin4 (opticalflowSolvers.c:3059-3060), southRow_llin8 (:3871-3872) and southRow4
(disparitySolvers.c:1986-1987), which compute cp = c/den and dp = (d - dp'a)/den.  The two forms differ by
an ulp now and then, and the difference spreads upward with the next iterations.

The check is a second, independent statement of one whole disparity iteration (GS_ALR_SOR_llin4_2d,
disparitySolvers.c:154-211, with {west,middle,east}Column4 :1376-1703 and {north,middle,south}Row4
:1705-2029) as plain float32 scalar Python, written from the reference's lines with a switch for the
south row's form: the oracle must equal the dividing form bit for bit and must differ from the
multiplying form (= what round 1's oracle and kernels did).
"""
import numpy as np
import pytest

import oracle_lib as orc
import problems as pb

F = np.float32


def _line(p, dU, fixed, vertical, omega, south_divides):
    """One Thomas solve + lagged SOR blend along column `fixed` (vertical) or row `fixed`."""
    U, Cu, Du, wW, wN, wE, wS = p["U"], p["Cu"], p["Du"], p["wW"], p["wN"], p["wE"], p["wS"]
    nrows, ncols = U.shape
    n = nrows if vertical else ncols
    cp, dp = [F(0)] * n, [F(0)] * n
    divide = south_divides and (not vertical) and fixed == nrows - 1

    def at(k):
        return (k, fixed) if vertical else (fixed, k)

    for k in range(n):
        i, j = at(k)
        hasN, hasS, hasW, hasE = i > 0, i < nrows - 1, j > 0, j < ncols - 1
        # b = wN + wS + wE + wW with the missing neighbours skipped (disparitySolvers.c:1536, :1865)
        b = None
        for has, w in ((hasN, wN), (hasS, wS), (hasE, wE), (hasW, wW)):
            if has:
                b = w[i, j] if b is None else F(b + w[i, j])
        # d = W, E, S, N; a neighbour that is not on the line carries its increment (:1538-1541, :1867-1870)
        d = None
        for has, w, (ii, jj), off_line in ((hasW, wW, (i, j - 1), vertical), (hasE, wE, (i, j + 1), vertical),
                                            (hasS, wS, (i + 1, j), not vertical), (hasN, wN, (i - 1, j), not vertical)):
            if not has:
                continue
            g = F(U[ii, jj] - U[i, j])
            if off_line:
                g = F(g + dU[ii, jj])
            t = F(w[i, j] * g)
            d = t if d is None else F(d + t)
        if not np.isnan(Cu[i, j]):
            b = F(b + Du[i, j])
            d = F(d + Cu[i, j])
        if vertical:
            a, c = (F(-wN[i, j]) if hasN else F(0)), (F(-wS[i, j]) if hasS else F(0))
        else:
            a, c = (F(-wW[i, j]) if hasW else F(0)), (F(-wE[i, j]) if hasE else F(0))
        if k == 0:
            cp[k], dp[k] = F(c / b), F(d / b)
        elif k == n - 1:
            dp[k] = F(F(d - F(dp[k - 1] * a)) / F(b - F(cp[k - 1] * a)))
        elif divide:
            den = F(b - F(cp[k - 1] * a))
            cp[k] = F(c / den)
            dp[k] = F(F(d - F(dp[k - 1] * a)) / den)
        else:
            div = F(F(1) / F(b - F(cp[k - 1] * a)))
            cp[k] = F(c * div)
            dp[k] = F(F(d - F(dp[k - 1] * a)) * div)
    # back-substitution with the blend applied one element late (:1575-1587)
    om, om1 = F(omega), F(F(1) - F(omega))
    i, j = at(n - 1)
    temp1 = dU[i, j]
    dU[i, j] = dp[n - 1]
    for k in range(n - 2, -1, -1):
        i, j = at(k)
        i1, j1 = at(k + 1)
        temp2 = dU[i, j]
        dU[i, j] = F(dp[k] - F(cp[k] * dU[i1, j1]))
        dU[i1, j1] = F(F(om * dU[i1, j1]) + F(om1 * temp1))
        temp1 = temp2
    i, j = at(0)
    dU[i, j] = F(F(om * dU[i, j]) + F(om1 * temp1))


def disp_alr_python(p, iters, omega, south_divides):
    dU = p["dU"].copy()
    nrows, ncols = dU.shape
    with np.errstate(all="ignore"):
        for _ in range(iters):
            for j in range(ncols):
                _line(p, dU, j, True, omega, south_divides)
            for i in range(nrows):
                _line(p, dU, i, False, omega, south_divides)
    return dU


@pytest.mark.parametrize("seed,nrows,ncols,iters", [(301, 5, 300, 1), (302, 9, 64, 2), (303, 6, 41, 3)])
def test_south_row_divides(seed, nrows, ncols, iters):
    p = pb.disp4(seed, nrows, ncols, nan_frac=0.02)
    got = orc.disp_alr_llin4(*p.values(), iters, 1.4)
    want = disp_alr_python(p, iters, 1.4, south_divides=True)
    assert pb.bit_equal(got, want), pb.describe_mismatch(got, want)
    old = disp_alr_python(p, iters, 1.4, south_divides=False)
    # the multiplying form is what the reference does NOT do on that row: it must show (first in the last image row)
    assert not pb.bit_equal(got, old)
    if iters == 1:
        rows = np.nonzero((got != old).any(axis=1))[0]
        assert rows.tolist() == [nrows - 1]


def test_south_row_divides_flow_and_llin8():
    """The flow solvers share the switch: with zero coupling and identical fields they reduce to the disparity lines."""
    p = pb.disp4(311, 7, 50)
    z = np.zeros_like(p["U"])
    want = disp_alr_python(p, 2, 1.3, south_divides=True)
    dU, dV = orc.oflow_alr_llin4(p["U"], p["U"], p["dU"], p["dU"], z, p["Cu"], p["Cu"], p["Du"], p["Du"], p["wW"], p["wN"], p["wE"],
                                 p["wS"], 2, 1.3)
    assert pb.bit_equal(dU, want) and pb.bit_equal(dV, want)
    dU, dV = orc.oflow_alr_llin8(p["U"], p["U"], p["dU"], p["dU"], z, p["Cu"], p["Cu"], p["Du"], p["Du"], p["wW"], z, p["wN"], z, p["wE"], z,
                                 p["wS"], z, 2, 1.3)
    # llin8 with zero diagonal weights: same tridiagonal rows up to the order in which zeros are added
    assert np.allclose(dU, want, rtol=0, atol=1e-5) and np.allclose(dV, want, rtol=0, atol=1e-5)
