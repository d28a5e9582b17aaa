"""FAS full-multigrid optical flow with early linearisation, resident in HBM.

Mirror of matlab/optical_flow/FlowEminNDFASFMG_elin_2D_v10.m: image pyramid by halving (:104-120), per-scale
derivative planes and constants (:125-153), and per scale, coarse to fine (:161-183), one FAS V- or W-cycle
(FAS_CYCLE, :193-273) whose smoother (:367-464) is `firstLoop` x [robust data weights, OPdiffWeights,
Oflow_sor_elin4_2d]; residuals through the solver's residual operator, coarse right-hand side through
Oflow_lhs_elin4_2d, full-weighting restriction, bilinear prolongation of the correction.

Every stage is a kernel of libpdeip.so (csrc/pdeip_fas.hpp + the solver / residual / LHS kernels); the recursion
is host control flow only and nothing crosses PCIe inside `run`, the bicubic up-scaling of the flow between two
scales of the outer loop (imresize's default method, :177-180) included.  The IPT calls (imfilter, imresize) are
restated by their documented meaning with our own summation order, see the kernels.

Planes are torch float32 CUDA tensors [C, ncols, nrows] / [ncols, nrows] (device.to_device).
"""
import numpy as np
import torch

from . import capi, device as dev, graphs

IDT, IDX, IDY, IDXX, IDYY, IDXY, IDXT, IDYT, M, CU, CV, DU, DV = range(13)   # plane order of fas_prepare

DEFAULTS = dict(alpha=0.035, omega=1.9, firstLoop=4, iter=4, b1=0.03, b2=0.97, scl_factor=0.5, solver=2, cycle_index=1,
                scales=None)   # :53-69 (the Yosemite settings)


def gaussian5(sigma=1.0):
    """fspecial('gaussian', [5 5], sigma) (:98)"""
    ax = np.arange(-2, 3, dtype=np.float64)
    g = np.exp(-(ax[:, None] ** 2 + ax[None, :] ** 2) / (2.0 * sigma * sigma))
    g[g < np.finfo(np.float64).eps * g.max()] = 0
    return (g / g.sum()).astype(np.float32)


class FasFmgFlow:
    def __init__(self, param=None, mode=capi.MODE_EXACT_ORDER):
        self.p = dict(DEFAULTS, **(param or {}))
        self.mode = mode
        self.planes = []
        self._graphs = {}

    # ---- set-up (:104-153) -------------------------------------------------------------------------
    def prepare(self, I0, I1):
        """I0, I1: [C, ncols, nrows] frames in 0..255.  Builds self.planes[scl] (0 = finest)."""
        p = self.p
        G = gaussian5(1.0)
        P0, P1 = [dev.fas_gauss5(I0, G)], [dev.fas_gauss5(I1, G)]
        while p["scales"] is None or len(P0) < p["scales"]:
            P0.append(dev.fas_down(P0[-1]))
            P1.append(dev.fas_down(P1[-1]))
            if P0[-1].shape[-1] <= 10 or P0[-1].shape[-2] <= 10:
                break
        self.planes = [dev.fas_prepare(a, b, p["b1"], p["b2"]) for a, b in zip(P0, P1)]
        return self.planes

    # ---- smoother (:367-464) -----------------------------------------------------------------------
    def _solve(self, U, V, coef):
        fn = dev.oflow_sor_elin4 if int(self.p["solver"]) == 1 else dev.oflow_alr_elin4
        fn(U, V, *coef, int(self.p["iter"]), float(self.p["omega"]), self.mode)

    def smooth(self, U, V, pl, Cu, Cv, residuals=False):
        """In place on U, V.  Returns (RU, RV) [C, ncols, nrows] when asked for."""
        p = self.p
        C = pl.shape[1]
        new = lambda: torch.empty_like(U)
        coef = [new() for _ in range(9)]   # MGd CuGd CvGd DuGd DvGd wW wN wE wS
        for _ in range(int(p["firstLoop"])):
            # robust data weights (:377-397) and OPdiffWeights(U, V) (:392) of the same iterate: one launch (wW wN wS wE order)
            dev.fas_assemble_weights(pl, Cu, Cv, U, V, p["b1"], p["b2"], C * p["alpha"], *coef[:5], coef[5], coef[6], coef[8], coef[7])
            self._solve(U, V, coef)
        if not residuals:
            return None
        f = [torch.empty_like(pl[0]) for _ in range(5)]
        dev.fas_assemble(pl, Cu, Cv, U, V, p["b1"], p["b2"], p["alpha"], True, *f)
        dev.flow_opdiffweights(U, V, None, None, coef[5], coef[6], coef[8], coef[7])
        RU, RV = torch.empty_like(pl[0]), torch.empty_like(pl[0])
        dev.oflow_res_elin4(RU, RV, U, V, *f, *coef[5:])
        return RU, RV

    # ---- FAS_CYCLE (:193-273) ----------------------------------------------------------------------
    def cycle(self, scl, U, V, Cu=None, Cv=None):
        """One V (cycle_index 1) or W (2) cycle at scale scl on U, V (in place); Cu/Cv default to the scale's own."""
        p, pl = self.p, self.planes[scl]
        Cu = pl[CU] if Cu is None else Cu
        Cv = pl[CV] if Cv is None else Cv
        if scl == len(self.planes) - 1:
            self.smooth(U, V, pl, Cu, Cv)
            return U, V
        s, pc = float(p["scl_factor"]), self.planes[scl + 1]
        for _ in range(int(p["cycle_index"])):
            RU, RV = self.smooth(U, V, pl, Cu, Cv, residuals=True)
            RUres, RVres = dev.fas_restrict(RU, s), dev.fas_restrict(RV, s)
            Ures, Vres = dev.fas_restrict(U, s), dev.fas_restrict(V, s)
            MGd, DuGd, DvGd, gd = [torch.empty_like(pc[0]) for _ in range(4)]
            dev.fas_assemble(pc, None, None, Ures, Vres, p["b1"], p["b2"], p["alpha"], True, MGd, None, None, DuGd, DvGd, gd)
            w = [torch.empty_like(Ures) for _ in range(4)]   # wW wN wE wS
            dev.flow_opdiffweights(Ures, Vres, None, None, w[0], w[1], w[3], w[2])
            Au, Av = torch.empty_like(pc[0]), torch.empty_like(pc[0])
            dev.oflow_lhs_elin4(Au, Av, Ures, Vres, MGd, DuGd, DvGd, *w)
            fu, fv = dev.fas_rhs(RUres, Au, gd), dev.fas_rhs(RVres, Av, gd)
            Uc, Vc = self.cycle(scl + 1, Ures.clone(), Vres.clone(), fu, fv)
            dev.fas_prolong_add(U, Uc, Ures, 1.0 / s)
            dev.fas_prolong_add(V, Vc, Vres, 1.0 / s)
        self.smooth(U, V, pl, Cu, Cv)
        return U, V

    def run_graph(self, I0, I1):
        """run() replayed as a HIP graph (captured on the first call per frame shape; graphs.py).  The returned planes are the
        graph's output buffers: valid until the next call.  Every ordering (round 3: the exact-order walkers' schedule table
        is built on the call's stream, so those calls are capturable too)."""
        key = (tuple(I0.shape), tuple(I1.shape))
        if key not in self._graphs:
            self._graphs[key] = graphs.GraphedRun(self.run)
        return self._graphs[key](I0, I1)

    # ---- the driver's outer loop (:161-183) --------------------------------------------------------
    def run(self, I0, I1):
        """Flow [ncols, nrows] x 2 at the finest scale."""
        self.prepare(I0, I1)
        U = V = None
        for scl in range(len(self.planes) - 1, -1, -1):
            ncols, nrows = self.planes[scl].shape[-2:]
            if U is None:
                U = torch.zeros((ncols, nrows), dtype=torch.float32, device=I0.device)
                V = torch.zeros_like(U)
            U, V = self.cycle(scl, U, V)
            if scl > 0:
                nc, nr = self.planes[scl - 1].shape[-2:]
                U, V = [dev.fas_upscale(t, 1.0 / self.p["scl_factor"], nr, nc) for t in (U, V)]
        return U, V
