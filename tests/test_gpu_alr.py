"""GPU parity of the alternating line relaxation (solver = 2, the MATLAB drivers' default) through
the C-ABI against the CPU oracle.  Bar: bit-exact.

EXACT_ORDER mode = the reference's line order (serial walk, k_alr_lex); RED_BLACK mode = zebra order
(k_alr_zebra) against the oracle's zebra order.
"""
import numpy as np
import pytest

import problems as pb

pytestmark = pytest.mark.gpu

SIZES = [(32, 48), (97, 131), (3, 3), (5, 300), (260, 7), (131, 70), (531, 777), (226, 450)]  # incl. partial tiles / rounds of the zebra kernel
MODES = [(0, 0), (1, 1)]
TWO = np.float32(2)


def check(got, want, what):
    got = got if isinstance(got, tuple) else (got,)
    want = want if isinstance(want, tuple) else (want,)
    assert len(got) == len(want)
    for k, (g, w) in enumerate(zip(got, want)):
        assert pb.bit_equal(g, w), "%s output %d: %s" % (what, k, pb.describe_mismatch(g, w))


@pytest.mark.parametrize("mode,order", MODES)
@pytest.mark.parametrize("shape", SIZES)
def test_alr_elin4(pdeip, oracle, mode, order, shape):
    pdeip.mex_api.set_mode(mode)
    for it, nan_frac, nan_mode in ((1, 0.0, "all"), (3, 0.05, "all"), (2, 0.05, "C")):
        p = pb.elin4(511, *shape, nan_frac=nan_frac, nan_mode=nan_mode)
        got = pdeip.mex_api.Oflow_sor_elin4_2d(*p.values(), np.float32(it), np.float32(1.5), TWO, nargout=4)
        want = oracle.Oflow_sor_elin4_2d(*p.values(), it, 1.5, solver=2, nargout=4, order=order)
        check(got, want, "alr elin4 %s it=%d mode=%d" % (shape, it, mode))
    p = pb.elin4(512, *shape)
    got = pdeip.mex_api.Oflow_sor_elin4_2d(*p.values(), np.float32(0), np.float32(1.5), TWO)
    assert not got[0].any() and not got[1].any()  # iter <= 0: zero outputs


@pytest.mark.parametrize("mode,order", MODES)
@pytest.mark.parametrize("shape", SIZES)
def test_alr_llin4_llin8_disparity(pdeip, oracle, mode, order, shape):
    pdeip.mex_api.set_mode(mode)
    for it, nan_frac in ((1, 0.0), (3, 0.04)):
        p = pb.llin4(521, *shape, nan_frac=nan_frac)
        got = pdeip.mex_api.Oflow_sor_llin4_2d(*p.values(), np.float32(it), np.float32(1.4), TWO, nargout=4)
        want = oracle.Oflow_sor_llin4_2d(*p.values(), it, 1.4, solver=2, nargout=4, order=order)
        check(got, want, "alr llin4 %s it=%d mode=%d" % (shape, it, mode))
        p = pb.llin8(522, *shape, nan_frac=nan_frac)
        got = pdeip.mex_api.Oflow_sor_llin8_2d(*p.values(), np.float32(it), np.float32(1.4), TWO, nargout=4)
        want = oracle.Oflow_sor_llin8_2d(*p.values(), it, 1.4, solver=2, nargout=4, order=order)
        check(got, want, "alr llin8 %s it=%d mode=%d" % (shape, it, mode))
        p = pb.disp4(523, *shape, nan_frac=nan_frac)
        got = pdeip.mex_api.Disp_sor_llin4_2d(*p.values(), np.float32(it), np.float32(1.4), TWO, nargout=2)
        want = oracle.Disp_sor_llin4_2d(*p.values(), it, 1.4, solver=2, nargout=2, order=order)
        check(got, want, "alr disp4 %s it=%d mode=%d" % (shape, it, mode))


@pytest.mark.parametrize("mode,order", MODES)
@pytest.mark.parametrize("shape,F", [((32, 48), 1), ((97, 131), 3), ((3, 3), 2), ((5, 300), 1), ((260, 7), 2)])
def test_alr_pde(pdeip, oracle, mode, order, shape, F):
    pdeip.mex_api.set_mode(mode)
    for it, nan_frac in ((1, 0.0), (3, 0.05)):
        p = pb.pde4(531, *shape, nframes=F, nan_frac=nan_frac)
        got = pdeip.mex_api.PDEsolver4(*p.values(), np.float32(it), np.float32(1.3), TWO)
        check(got, oracle.PDEsolver4(*p.values(), it, 1.3, solver=2, order=order), "alr pde4 %s F=%d it=%d mode=%d" % (shape, F, it, mode))
        p = pb.pde8(532, *shape, nframes=F, nan_frac=nan_frac)
        got = pdeip.mex_api.PDEsolver8(*p.values(), np.float32(it), np.float32(1.3), TWO)
        check(got, oracle.PDEsolver8(*p.values(), it, 1.3, solver=2, order=order), "alr pde8 %s F=%d it=%d mode=%d" % (shape, F, it, mode))
    # the 8-neighbour line solver runs exactly one iteration, even for iter = 0 (pdeSolvers.c:362)
    p = pb.pde8(533, *shape, nframes=F)
    a = pdeip.mex_api.PDEsolver8(*p.values(), np.float32(0), np.float32(1.3), TWO)
    b = pdeip.mex_api.PDEsolver8(*p.values(), np.float32(5), np.float32(1.3), TWO)
    check(a, b, "alr pde8 ignores iter")
    # and PDEsolver4 with iter = 0 is a copy
    q = pb.pde4(534, *shape, nframes=F)
    check(pdeip.mex_api.PDEsolver4(*q.values(), np.float32(0), np.float32(1.3), TWO), q["X"], "alr pde4 iter=0")


def test_alr_c1_size_exact_and_zebra(pdeip, oracle):
    """BASELINE config C1's frame size (584x388 image = 388 rows x 584 columns), the H&S driver's iter."""
    p = pb.elin4(541, 388, 584)
    for mode, order in MODES:
        pdeip.mex_api.set_mode(mode)
        got = pdeip.mex_api.Oflow_sor_elin4_2d(*p.values(), np.float32(4), np.float32(1.9), TWO)
        check(got, oracle.Oflow_sor_elin4_2d(*p.values(), 4, 1.9, solver=2, order=order), "alr C1 mode=%d" % mode)
    pdeip.mex_api.set_mode(0)


@pytest.mark.parametrize("mode,order", MODES)
def test_alr_long_lines(pdeip, oracle, mode, order):
    """Lines longer than 5120 pixels: the exact-order walker holds one chain at a time (two would not fit in LDS)."""
    pdeip.mex_api.set_mode(mode)
    for shape in ((5300, 6), (7, 5200)):
        p = pb.elin4(551, *shape, nan_frac=0.01)
        got = pdeip.mex_api.Oflow_sor_elin4_2d(*p.values(), np.float32(2), np.float32(1.5), TWO)
        check(got, oracle.Oflow_sor_elin4_2d(*p.values(), 2, 1.5, solver=2, order=order), "alr long lines %s mode=%d" % (shape, mode))
    pdeip.mex_api.set_mode(0)


def test_alr_lines_beyond_lds(pdeip, oracle):
    """Exact-order lines of more than 10 240 pixels do not fit the 160 KB of LDS: the line buffer then lives in global memory
    (k_alr_lex<..., GL>).  The reference has no such limit; same bits as the oracle's line order, both directions, a coupled and
    a scalar model."""
    pdeip.mex_api.set_mode(0)
    for shape in ((10300, 5), (6, 10290)):
        p = pb.elin4(561, *shape, nan_frac=0.01)
        got = pdeip.mex_api.Oflow_sor_elin4_2d(*p.values(), np.float32(1), np.float32(1.5), TWO)
        check(got, oracle.Oflow_sor_elin4_2d(*p.values(), 1, 1.5, solver=2, order=0), "alr elin4 beyond LDS %s" % (shape,))
        q = pb.pde4(562, *shape, nframes=2, nan_frac=0.01)
        got = pdeip.mex_api.PDEsolver4(*q.values(), np.float32(1), np.float32(1.3), TWO)
        check(got, oracle.PDEsolver4(*q.values(), 1, 1.3, solver=2, order=0), "alr pde4 beyond LDS %s" % (shape,))
