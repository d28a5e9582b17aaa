"""The fused red-black pipeline (k_sor_rbp) marches alternate strips from their last column to their first, so that neighbouring
strips read the halo columns they share at the same time.  A half-sweep does not depend on the order its pixels are visited
in, so both directions must give the oracle's colour-ordered bits: every 5-point model, one and two launches per call (the second
reads the divisor planes the first stored), NaN-laced data terms, frames whose last strip is ragged."""
import os

import numpy as np
import pytest

import problems as pb

pytestmark = pytest.mark.gpu
f = np.float32


@pytest.fixture(params=["0", "2"], ids=["forward", "mirrored"])
def direction(request):
    old = {k: os.environ.get(k) for k in ("PDEIP_RBP_SERPENTINE", "PDEIP_RB_SMALL")}
    os.environ["PDEIP_RBP_SERPENTINE"] = request.param  # 0 (default): no strip mirrored, 2: every strip mirrored (1: the odd ones)
    os.environ["PDEIP_RB_SMALL"] = "0"                   # small frames through the pipeline as well
    yield request.param
    for k, v in old.items():
        if v is None:
            os.environ.pop(k, None)
        else:
            os.environ[k] = v


@pytest.mark.parametrize("shape", [(24, 40), (64, 96), (100, 48), (68, 121), (4, 5), (256, 300), (388, 584)])
@pytest.mark.parametrize("it", [4, 8])
def test_coupled_models_both_directions(pdeip, oracle, direction, shape, it):
    api = pdeip.mex_api
    api.set_mode(1)
    try:
        p = pb.elin4(1201, *shape, nan_frac=0.02)
        for g, w in zip(api.Oflow_sor_elin4_2d(*p.values(), f(it), f(1.9), f(1), nargout=4),
                        oracle.Oflow_sor_elin4_2d(*p.values(), it, 1.9, nargout=4, order=oracle.COLOUR)):
            assert pb.bit_equal(g, w), "elin4 %s it=%d %s: %s" % (shape, it, direction, pb.describe_mismatch(g, w))
        q = pb.llin4(1202, *shape, nan_frac=0.02)
        for g, w in zip(api.Oflow_sor_llin4_2d(*q.values(), f(it), f(1.9), f(1)), oracle.Oflow_sor_llin4_2d(*q.values(), it, 1.9, order=oracle.COLOUR)):
            assert pb.bit_equal(g, w), "llin4 %s it=%d %s: %s" % (shape, it, direction, pb.describe_mismatch(g, w))
    finally:
        api.set_mode(0)


@pytest.mark.parametrize("it", [4, 8])
def test_single_field_models_both_directions(pdeip, oracle, direction, it):
    """One wave per sweep: the pipeline takes these models from 2^21 pixels on."""
    api = pdeip.mex_api
    api.set_mode(1)
    try:
        d = pb.disp4(1203, 1200, 1800, nan_frac=0.01)
        g, w = api.Disp_sor_llin4_2d(*d.values(), f(it), f(1.9), f(1)), oracle.Disp_sor_llin4_2d(*d.values(), it, 1.9, order=oracle.COLOUR)
        assert pb.bit_equal(g, w), "disp4 it=%d %s: %s" % (it, direction, pb.describe_mismatch(g, w))
        e = pb.pde4(1204, 1200, 1801, nframes=1, nan_frac=0.01)
        g, w = api.PDEsolver4(*e.values(), f(it), f(1.75), f(1)), oracle.PDEsolver4(*e.values(), it, 1.75, order=oracle.COLOUR)
        assert pb.bit_equal(g, w), "pde4 it=%d %s: %s" % (it, direction, pb.describe_mismatch(g, w))
        y = pb.dispsym4(1205, 1200, 1800, nan_frac=0.01)
        for g, w in zip(api.Disp_sor_llin_sym4_2d(*y.values(), f(it), f(1.9), f(1)), oracle.Disp_sor_llin_sym4_2d(*y.values(), it, 1.9, order=oracle.COLOUR)):
            assert pb.bit_equal(g, w), "dispsym4 it=%d %s" % (it, direction)
    finally:
        api.set_mode(0)


def test_alternating_strips_against_the_oracle_and_as_a_continuation(pdeip, oracle):
    """PDEIP_RBP_SERPENTINE=1 (odd strips mirrored, a ragged last strip): the oracle's bits, and iter = 4 twice equals iter = 8 in one call."""
    api = pdeip.mex_api
    api.set_mode(1)
    old, old_s = os.environ.get("PDEIP_RB_SMALL"), os.environ.get("PDEIP_RBP_SERPENTINE")
    os.environ["PDEIP_RB_SMALL"] = "0"
    os.environ["PDEIP_RBP_SERPENTINE"] = "1"
    try:
        p = pb.elin4(1206, 512, 700)
        coef = [p[k] for k in ("M", "Cu", "Cv", "Du", "Dv", "wW", "wN", "wE", "wS")]
        one = api.Oflow_sor_elin4_2d(p["U"], p["V"], *coef, f(8), f(1.9), f(1))
        half = api.Oflow_sor_elin4_2d(p["U"], p["V"], *coef, f(4), f(1.9), f(1))
        two = api.Oflow_sor_elin4_2d(half[0], half[1], *coef, f(4), f(1.9), f(1))
        for g, w in zip(two, one):
            assert pb.bit_equal(g, w)
        for g, w in zip(one, oracle.Oflow_sor_elin4_2d(*p.values(), 8, 1.9, order=oracle.COLOUR)):
            assert pb.bit_equal(g, w), pb.describe_mismatch(g, w)
    finally:
        api.set_mode(0)
        for k, v in (("PDEIP_RB_SMALL", old), ("PDEIP_RBP_SERPENTINE", old_s)):
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
