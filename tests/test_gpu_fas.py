"""GPU: the stages of the FAS full-multigrid flow driver (csrc/pdeip_fas.hpp) and the resident cycle (fas.py) against
their numpy statement (oracle/matlab_side.py fas_*), bit for bit.  The statement restates
matlab/optical_flow/FlowEminNDFASFMG_elin_2D_v10.m; parity is with that statement, not with MATLAB ("parity unpinned")."""
import numpy as np
import pytest

import problems as pb
from test_gpu_flow_level import frames, matlab_side, same, sub

pytestmark = pytest.mark.gpu


def frames255(seed, nrows, ncols, C):
    I0, I1 = frames(seed, nrows, ncols, C)
    to = lambda I: np.asfortranarray(((I + 1.5) * np.float32(80.0)).astype(np.float32))
    return to(I0), to(I1)


@pytest.mark.parametrize("shape,C", [((37, 53), 1), ((64, 80), 3), ((11, 300), 2), ((131, 12), 1)])
def test_stages(pdeip, shape, C):
    import torch
    ms, dev, fas = matlab_side(), sub("device"), sub("fas")
    nrows, ncols = shape
    rng = np.random.default_rng(nrows + C)
    I0, I1 = frames255(5 + C, nrows, ncols, C)
    d0, d1 = dev.to_device(I0), dev.to_device(I1)
    # 5x5 Gaussian and one pyramid step
    G = fas.gaussian5(1.0)
    assert pb.bit_equal(G, ms.fas_gaussian5(1.0))
    g0 = dev.fas_gauss5(d0, G)
    w0 = ms.fas_gauss5(I0, G)
    same(dev.to_matlab(g0), w0, "gauss5 %s C=%d" % (shape, C))
    down = dev.fas_down(g0)
    wdown = ms.fas_down(w0)
    assert tuple(down.shape) == (C, (ncols + 1) // 2, (nrows + 1) // 2)
    same(dev.to_matlab(down), wdown, "down %s" % (shape,))
    # per-scale constants
    planes = dev.fas_prepare(d0, d1, 0.03, 0.97)
    want = ms.fas_prepare(I0, I1, 0.03, 0.97)
    for k, name in enumerate(ms.FAS_PLANES):
        same(dev.to_matlab(planes[k]), want[name], "prepare %s %s C=%d" % (name, shape, C))
    # data weights at a flow, both forms
    U = np.asfortranarray(rng.uniform(-2, 2, shape).astype(np.float32))
    V = np.asfortranarray(rng.uniform(-2, 2, shape).astype(np.float32))
    dU, dV = dev.to_device(U), dev.to_device(V)
    one = [torch.empty_like(dU) for _ in range(5)]
    dev.fas_assemble(planes, planes[fas.CU], planes[fas.CV], dU, dV, 0.03, 0.97, C * 0.035, False, *one)
    gd = ms.fas_gd(want, U, V, 0.03, 0.97, C * 0.035)
    for g, name in zip(one, ("M", "Cu", "Cv", "Du", "Dv")):
        same(dev.to_matlab(g), ms._sum3(want[name] * gd), "summed %s.*gd" % name)
    per = [torch.empty_like(planes[0]) for _ in range(6)]
    dev.fas_assemble(planes, planes[fas.CU], planes[fas.CV], dU, dV, 0.03, 0.97, 0.035, True, *per)
    gd = ms.fas_gd(want, U, V, 0.03, 0.97, 0.035)
    for g, name in zip(per[:5], ("M", "Cu", "Cv", "Du", "Dv")):
        same(dev.to_matlab(g), (want[name] * gd).astype(np.float32), "per-frame %s.*gd" % name)
    same(dev.to_matlab(per[5]), gd, "gd")
    # restriction (one plane and a stack), right-hand side, prolongation
    same(dev.to_matlab(dev.fas_restrict(dU, 0.5)), ms.fas_restrict(U, 0.5), "restrict plane")
    same(dev.to_matlab(dev.fas_restrict(planes[fas.M], 0.5)), ms.fas_restrict(want["M"], 0.5), "restrict stack")
    A = np.asfortranarray(rng.uniform(-1, 1, want["M"].shape).astype(np.float32))
    same(dev.to_matlab(dev.fas_rhs(planes[fas.DU], dev.to_device(A), per[5])), ((want["Du"] + A) / gd).astype(np.float32), "rhs")
    cs = ((nrows + 1) // 2, (ncols + 1) // 2)
    Uc = np.asfortranarray(rng.uniform(-1, 1, cs).astype(np.float32))
    Ur = np.asfortranarray(rng.uniform(-1, 1, cs).astype(np.float32))
    got = dU.clone()
    dev.fas_prolong_add(got, dev.to_device(Uc), dev.to_device(Ur), 2.0)
    same(dev.to_matlab(got), ms.fas_prolong_add(U, Uc, Ur, 2.0), "prolongation %s" % (shape,))
    for out_shape in ((2 * cs[0], 2 * cs[1]), (2 * cs[0] - 1, 2 * cs[1] - 1), (cs[0] + 3, 3 * cs[1])):
        up = dev.fas_upscale(dev.to_device(Uc), 2.0, *out_shape)
        same(dev.to_matlab(up), ms.fas_upscale(Uc, 2.0, *out_shape), "bicubic upscale %s -> %s" % (cs, out_shape))


def statement_fmg(ms, py, oracle, I0, I1, param, max_scales=None):
    return ms.fas_fmg(oracle, I0, I1, param, max_scales)


@pytest.mark.parametrize("solver,mode,order,omega,cycle_index,C",
                         [(2, 0, 0, 1.9, 1, 1), (2, 0, 0, 1.9, 2, 3), (2, 1, 1, 1.9, 1, 1), (1, 0, 0, 1.0, 1, 2), (1, 1, 1, 1.0, 2, 1)])
def test_resident_cycle_and_driver(pdeip, oracle, solver, mode, order, omega, cycle_index, C):
    ms, dev, fas, py = matlab_side(), sub("device"), sub("fas"), sub("pyramid")
    I0, I1 = frames255(77 + C, 70, 90, C)
    param = dict(alpha=0.035, omega=omega, firstLoop=2, iter=3, b1=0.03, b2=0.97, scl_factor=0.5, solver=solver, cycle_index=cycle_index,
                 order=order)
    # one cycle from a non-trivial start at the finest of three scales
    P0, P1 = ms.fas_pyramid(I0, I1, 3)
    planes = [ms.fas_prepare(a, b, param["b1"], param["b2"]) for a, b in zip(P0, P1)]
    rng = np.random.default_rng(9)
    U0 = np.asfortranarray(rng.uniform(-0.3, 0.3, (70, 90)).astype(np.float32))
    wU, wV = ms.fas_cycle(oracle, planes, U0, U0.copy(), planes[0]["Cu"], planes[0]["Cv"], 0, param)
    drv = fas.FasFmgFlow(dict(param, scales=3), mode=mode)
    drv.prepare(dev.to_device(I0), dev.to_device(I1))
    assert len(drv.planes) == 3
    gU, gV = drv.cycle(0, dev.to_device(U0), dev.to_device(U0))
    same(dev.to_matlab(gU), wU, "cycle U (solver %d mode %d index %d)" % (solver, mode, cycle_index))
    same(dev.to_matlab(gV), wV, "cycle V (solver %d mode %d index %d)" % (solver, mode, cycle_index))
    assert np.isfinite(wU).all() and np.abs(wU).max() < 50
    # the whole driver, all scales
    wU, wV = statement_fmg(ms, py, oracle, I0, I1, param)
    gU, gV = fas.FasFmgFlow(param, mode=mode).run(dev.to_device(I0), dev.to_device(I1))
    same(dev.to_matlab(gU), wU, "driver U")
    same(dev.to_matlab(gV), wV, "driver V")


@pytest.mark.parametrize("shape,C,solver", [((12, 14), 1, 2), ((11, 23), 2, 1), ((21, 10), 1, 2), ((5, 7), 1, 2)])
def test_driver_on_tiny_frames(pdeip, oracle, shape, C, solver):
    """Two-scale pyramids with 3..7-pixel coarse sides: every halving, restriction and prolongation index at its edge."""
    ms, dev, fas, py = matlab_side(), sub("device"), sub("fas"), sub("pyramid")
    I0, I1 = frames255(90 + shape[0], shape[0], shape[1], C)
    param = dict(alpha=0.035, omega=1.9 if solver == 2 else 1.0, firstLoop=2, iter=2, b1=0.03, b2=0.97, scl_factor=0.5, solver=solver,
                 cycle_index=2, order=0)
    wU, wV = statement_fmg(ms, py, oracle, I0, I1, param)
    drv = fas.FasFmgFlow(param, mode=0)
    gU, gV = drv.run(dev.to_device(I0), dev.to_device(I1))
    assert len(drv.planes) == 2 and tuple(drv.planes[1].shape[-2:]) == ((shape[1] + 1) // 2, (shape[0] + 1) // 2)
    same(dev.to_matlab(gU), wU, "tiny driver U %s" % (shape,))
    same(dev.to_matlab(gV), wV, "tiny driver V %s" % (shape,))
