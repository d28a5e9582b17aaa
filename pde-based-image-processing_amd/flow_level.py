"""One pyramid level of the late-linearisation optical flow / stereo disparity, resident in HBM.

Mirror of the body of the coarse-to-fine loop of matlab/optical_flow/FlowEminND_llin_2D_v10.m:208-356
(firstLoop x [warp, derivatives, secondLoop x (robust assembly, diffusion weights, Oflow_sor_llin4_2d)],
median) on device planes: every stage is a kernel of libpdeip.so, nothing crosses PCIe between them and
no host synchronisation happens inside a level.  The pyramid itself (imresize, Gaussian pre-smoothing)
is not here: IPT's imresize has no reference to check against in this image.

Planes are torch float32 CUDA tensors laid out [ncols, nrows] / [C, ncols, nrows] (device.to_device).
"""
import torch

from . import capi, device as dev


class FlowLlinLevel:
    """param: firstLoop, secondLoop, iter, omega, solver (1 point SOR, 2 line relaxation), alpha, b1, b2."""

    def __init__(self, param, mode=capi.MODE_EXACT_ORDER):
        self.p, self.mode = dict(param), mode

    def _solve(self, U, V, dU, dV, coef):
        fn = dev.oflow_sor_llin4 if int(self.p["solver"]) == 1 else dev.oflow_alr_llin4
        fn(U, V, dU, dV, *coef, int(self.p["iter"]), float(self.p["omega"]), self.mode)

    def run(self, I1t0, I1t1, U, V, I2t0=None, I2t1=None, Us=None, Vs=None, as_diff=None, u_double=False):
        """I*: [C, ncols, nrows] images of the two frames (first / optional second constancy term);
        U, V: [ncols, nrows] flow entering the level.  Returns the flow leaving it (new tensors).
        Us, Vs: optional spatial a-priori fields of this scale (float64, param.Us / param.Vs with gammaS in param) with
        as_diff = 2*(1/scl_factor)^-(scl-1); u_double: this is the coarsest scale (U is still a MATLAB double in its first firstLoop)."""
        p = self.p
        new = lambda like: torch.empty_like(like)
        w1, d1 = new(I1t1), [new(I1t1) for _ in range(3)]
        gradmag = str(p.get("sndTerm", "rgb")).lower() == "gradmag"   # second term through SndDerivatives5 (:253-258)
        w2, d2 = (new(I2t1), [new(I2t1) for _ in range(5 if gradmag else 3)]) if I2t1 is not None else (None, None)
        coef = [new(U) for _ in range(9)]  # MGd, CuGd, CvGd, DuGd, DvGd, wW, wN, wE, wS
        U, V = U.clone(), V.clone()
        Un, Vn = new(U), new(U)
        for first in range(int(p["firstLoop"])):
            dev.flow_warp(U, V, I1t1, w1, I2t1, w2)                   # both constancy images at single(X+U), single(Y+V): one launch
            dev.fst_derivatives5(I1t0, w1, *d1)                      # Idt, Idx, Idy
            t1, t2 = (d1[0], d1[1], d1[2], p["b1"]), None
            if I2t1 is not None:
                if gradmag:
                    dev.snd_derivatives5(I2t0, w2, *d2)               # Ixt, Iyt, Ixx, Iyy, Ixy
                    t2 = (*d2, p["b2"])
                else:
                    dev.fst_derivatives5(I2t0, w2, *d2)
                    t2 = (d2[0], d2[1], d2[2], p["b2"])
            dU, dV = torch.zeros((2,) + tuple(U.shape), dtype=U.dtype, device=U.device)   # one fill for both increments
            for k in range(int(p["secondLoop"])):
                # robust assembly and OPdiffWeights(U+dU, V+dV) of the same iterate in one launch (weights in wW wN wS wE order)
                dev.flow_assemble_weights(t1, t2, U, V, dU, dV, p["alpha"], *coef[:5], coef[5], coef[6], coef[8], coef[7])
                if Us is not None:
                    dev.flow_apriori(Us, U, dU, p["gammaS"], p["alpha"], as_diff, u_double and first == 0, k == 0, coef[1], coef[3])
                if Vs is not None:
                    dev.flow_apriori(Vs, V, dV, p["gammaS"], p["alpha"], as_diff, u_double and first == 0, k == 0, coef[2], coef[4])
                self._solve(U, V, dU, dV, coef)
            dev.median3_pair(U, dU, Un, V, dV, Vn)
            U, Un = Un, U
            V, Vn = Vn, V
        return U, V


class FlowAdLevel:
    """The anisotropic-diffusion twin (matlab/optical_flow/FlowEminAD_llin_2D_v10.m:198-366): eight ADdiffWeights from
    the image (`diffusion` 'image', once per level) or from U+dU+V+dV ('flow', every inner iteration), and
    Oflow_sor_llin8_2d.
    param: firstLoop, secondLoop, iter, omega, solver, alpha, b1, b2, quantile, diffusion."""

    def __init__(self, param, mode=capi.MODE_EXACT_ORDER):
        self.p, self.mode = dict(param), mode

    def run(self, I1t0, I1t1, U, V, It0, I2t0=None, I2t1=None, Us=None, Vs=None, as_diff=None, u_double=False):
        """Us, Vs, as_diff, u_double: the spatial a-priori fields as in FlowLlinLevel.run."""
        p = self.p
        new = lambda like: torch.empty_like(like)
        X, Y, S = new(U), new(U), new(U)
        w1, d1 = new(I1t1), [new(I1t1) for _ in range(3)]
        gradmag = str(p.get("sndTerm", "rgb")).lower() == "gradmag"
        w2, d2 = (new(I2t1), [new(I2t1) for _ in range(5 if gradmag else 3)]) if I2t1 is not None else (None, None)
        coef = [new(U) for _ in range(5)]   # MGd, CuGd, CvGd, DuGd, DvGd
        w8 = [new(U) for _ in range(8)]     # wW, wNW, wN, wNE, wE, wSE, wS, wSW
        if int(p["solver"]) == 1:   # the 8-neighbour point solver runs the 4-neighbour arithmetic (opticalflowSolvers.c:1487): W, N, E, S only
            solve = lambda U, V, dU, dV, M, Cu, Cv, Du, Dv, wW, wNW, wN, wNE, wE, wSE, wS, wSW, it, om, mode: dev.oflow_sor_llin4(
                U, V, dU, dV, M, Cu, Cv, Du, Dv, wW, wN, wE, wS, it, om, mode)
        else:
            solve = dev.oflow_alr_llin8
        if p["diffusion"] == "image":
            dev.ad_weights(It0, p["quantile"], w8)
        U, V = U.clone(), V.clone()
        Un, Vn = new(U), new(U)
        for first in range(int(p["firstLoop"])):
            dev.flow_coords(U, V, X, Y)
            dev.warp_bilinear(I1t1, X, Y, w1)
            dev.fst_derivatives5(I1t0, w1, *d1)
            t1, t2 = (d1[0], d1[1], d1[2], p["b1"]), None
            if I2t1 is not None:
                dev.warp_bilinear(I2t1, X, Y, w2)
                if gradmag:
                    dev.snd_derivatives5(I2t0, w2, *d2)               # Ixt, Iyt, Ixx, Iyy, Ixy
                    t2 = (*d2, p["b2"])
                else:
                    dev.fst_derivatives5(I2t0, w2, *d2)
                    t2 = (d2[0], d2[1], d2[2], p["b2"])
            dU, dV = torch.zeros_like(U), torch.zeros_like(V)
            for k in range(int(p["secondLoop"])):
                dev.flow_assemble(t1, t2, dU, dV, p["alpha"], *coef)
                if Us is not None:
                    dev.flow_apriori(Us, U, dU, p["gammaS"], p["alpha"], as_diff, u_double and first == 0, k == 0, coef[1], coef[3])
                if Vs is not None:
                    dev.flow_apriori(Vs, V, dV, p["gammaS"], p["alpha"], as_diff, u_double and first == 0, k == 0, coef[2], coef[4])
                if p["diffusion"] == "flow":
                    dev.add(U, dU, S); dev.add(S, V, S); dev.add(S, dV, S)      # U+dU+V+dV, left to right
                    dev.ad_weights(S, p["quantile"], w8)
                solve(U, V, dU, dV, *coef, *w8, int(p["iter"]), float(p["omega"]), self.mode)
            dev.median3(U, dU, Un)
            dev.median3(V, dV, Vn)
            U, Un = Un, U
            V, Vn = Vn, V
        return U, V


class DispLlinLevel:
    """The same for stereo disparity: body of the coarse-to-fine loop of matlab/disparity/DispEminND_llin_2D.m:202-316
    (warp along x only, one unknown, DdiffWeights, Disp_sor_llin4_2d).
    param: firstLoop, secondLoop, iter, omega, solver, alpha, b1, b2."""

    def __init__(self, param, mode=capi.MODE_EXACT_ORDER):
        self.p, self.mode = dict(param), mode

    def run(self, I1t0, I1t1, U, I2t0=None, I2t1=None, Us=None, as_diff=None, u_double=False):
        """Us: optional spatial a-priori disparity of this scale (float64, param.Us with gammaS in param; DispEminND_llin_2D.m:
        246-248, :277-292), as_diff = 1.75*(1/scl_factor)^-(scl-1), u_double: U is still MATLAB's double array (coarsest scale)."""
        p = self.p
        new = lambda like: torch.empty_like(like)
        w1, d1 = new(I1t1), [new(I1t1) for _ in range(3)]
        gradmag = str(p.get("sndTerm", "rgb")).lower() == "gradmag"   # DispEminND_llin_2D.m:236-238
        w2, d2 = (new(I2t1), [new(I2t1) for _ in range(5 if gradmag else 3)]) if I2t1 is not None else (None, None)
        CuGd, DuGd, S = new(U), new(U), new(U)
        wts = [new(U) for _ in range(4)]  # wW, wN, wE, wS
        U, Un = U.clone(), new(U)
        solve = dev.disp_sor_llin4 if int(p["solver"]) == 1 else dev.disp_alr_llin4
        for first in range(int(p["firstLoop"])):
            dev.flow_warp(U, None, I1t1, w1, I2t1, w2)                # both constancy images at single(X+U), Y: one launch
            dev.fst_derivatives5(I1t0, w1, *d1)
            t1, t2 = (d1[0], d1[1], p["b1"]), None
            if I2t1 is not None:
                if gradmag:
                    dev.snd_derivatives5(I2t0, w2, *d2)
                    t2 = (d2[0], d2[1], d2[2], d2[4], p["b2"])        # Ixt, Iyt, Ixx, Ixy
                else:
                    dev.fst_derivatives5(I2t0, w2, *d2)
                    t2 = (d2[0], d2[1], p["b2"])
            dU = torch.zeros_like(U)
            for k in range(int(p["secondLoop"])):
                dev.disp_assemble(t1, t2, dU, p["alpha"], CuGd, DuGd)
                if Us is not None:
                    dev.disp_apriori(Us, U, dU, p["gammaS"], p["alpha"], as_diff, u_double and first == 0, k == 0, CuGd, DuGd)
                dev.add(U, dU, S)
                dev.diffweights6(S, 0.00001, *wts)
                solve(U, dU, CuGd, DuGd, *wts, int(p["iter"]), float(p["omega"]), self.mode)
            dev.median3(U, dU, Un)
            U, Un = Un, U
        return U


class DispSymLevel:
    """Symmetric stereo (matlab/disparity/DispEminND_llin_sym_2D.m:116-262): both views' disparities at once, each warped into
    the other (interp2) for the symmetry term, brightness + gradient constancy data terms, Disp_sor_llin_sym4_2d.
    param: firstLoop, secondLoop, iter, omega, solver, alpha, beta, b1, b2."""

    def __init__(self, param, mode=capi.MODE_EXACT_ORDER):
        self.p, self.mode = dict(param), mode

    def run(self, It0, It1, U0, U1, sr_diff):
        """It*: [C, ncols, nrows]; U0 = U(:,:,1), U1 = U(:,:,2); sr_diff = 2*(1/scl_factor)^-(scl-1) of the scale."""
        p = self.p
        new = lambda like: torch.empty_like(like)
        C = It0.shape[0] if It0.dim() == 3 else 1
        kS, sr2 = C * float(p["beta"]) / float(p["alpha"]), float(sr_diff) ** 2
        zero, X, Y = torch.zeros_like(U0), new(U0), new(U0)
        warped = [new(It0), new(It0)]
        der = [[new(It0) for _ in range(8)] for _ in range(2)]    # Idt Idx Idy | Idxt Idyt Idxx Idyy Idxy
        CuG, DuG, S = [new(U0), new(U0)], [new(U0), new(U0)], new(U0)
        w = [[new(U0) for _ in range(4)] for _ in range(2)]
        U, Un = [U0.clone(), U1.clone()], [new(U0), new(U0)]
        I = [It0, It1]
        for _ in range(int(p["firstLoop"])):
            d, sym = [], []
            for v in range(2):                                     # view v: own image It{v}, the other view warped by U{v}
                dev.flow_coords(U[v], zero, X, Y)
                dev.warp_bilinear(I[1 - v], X, Y, warped[v])       # It1w = It1 at X+U(:,:,1); It0w = It0 at X+U(:,:,2)
            Uw = [dev.sym_warp_flow(U[0], U[1]), dev.sym_warp_flow(U[1], U[0])]   # U0w, U1w (:140-141)
            for v in range(2):
                dev.fst_derivatives5(I[v], warped[v], *der[v][:3])
                dev.snd_derivatives5(I[v], warped[v], *der[v][3:])
                d.append((der[v][0], der[v][1], der[v][3], der[v][4], der[v][5], der[v][7]))
                sym.append(dev.sym_flow_terms(U[v], Uw[1 - v]))     # Udt0 from U1w, Udt1 from U0w
            dU = [torch.zeros_like(U0), torch.zeros_like(U0)]
            for k in range(int(p["secondLoop"])):
                for v in range(2):
                    dev.sym_assemble(d[v], sym[v], dU[v], p["b1"], p["b2"], p["alpha"], kS, sr2, k == 0, CuG[v], DuG[v])
                    dev.add(U[v], dU[v], S)
                    dev.diffweights6(S, 0.00001, *w[v])
                dev.disp_sor_llin_sym4(U[0], dU[0], CuG[0], DuG[0], w[0], U[1], dU[1], CuG[1], DuG[1], w[1], int(p["iter"]), float(p["omega"]),
                                       int(p["solver"]), self.mode)
            for v in range(2):
                dev.median3(U[v], dU[v], Un[v])
            U, Un = Un, U
        return U[0], U[1]


class TvLevel:
    """One scale of the TV denoiser: the lagged-diffusivity loop of matlab/denoising/TVdenoise8.m:78-100
    (outer_iter + 1 times: ADdiffWeights, PsiData/TRACE/B, PDEsolver8) on device planes [F, ncols, nrows] or [ncols, nrows].
    param: alpha, omega, outer_iter, inner_iter, solver (1 point SOR, 2 line relaxation)."""

    def __init__(self, param, mode=capi.MODE_EXACT_ORDER):
        self.p, self.mode = dict(param), mode

    def run(self, Iin, Iout):
        p = self.p
        X = Iout.clone()
        TRACE, B = torch.empty_like(X), torch.empty_like(X)
        w8 = [torch.empty_like(X) for _ in range(8)]  # aW, aNW, aN, aNE, aE, aSE, aS, aSW
        solve = dev.pde_sor8 if int(p["solver"]) == 1 else dev.pde_alr8
        for _ in range(int(p["outer_iter"]) + 1):      # for iter=0:param.outer_iter
            dev.tv_assemble(X, Iin, p["alpha"], TRACE, B, w8)
            solve(X, TRACE, B, *w8, int(p["inner_iter"]), float(p["omega"]), self.mode)
        return X


class Tv4Level:
    """The 4-neighbour denoiser's loop (matlab/denoising/TVdenoise4.m:82-103): outer_iter + 1 times DiffWeights, PsiData/TRACE/B,
    PDEsolver4.  param: alpha, omega, outer_iter, inner_iter, solver."""

    def __init__(self, param, mode=capi.MODE_EXACT_ORDER):
        self.p, self.mode = dict(param), mode

    def run(self, Iin, Iout):
        p = self.p
        X = Iout.clone()
        TRACE, B = torch.empty_like(X), torch.empty_like(X)
        w4 = [torch.empty_like(X) for _ in range(4)]  # aW, aN, aE, aS
        solve = dev.pde_sor4 if int(p["solver"]) == 1 else dev.pde_alr4
        for _ in range(int(p["outer_iter"]) + 1):
            dev.tv4_assemble(X, Iin, p["alpha"], TRACE, B, w4)
            solve(X, TRACE, B, w4[0], w4[1], w4[2], w4[3], int(p["inner_iter"]), float(p["omega"]), self.mode)
        return X


class FlowHsLevel:
    """One scale of Horn-Schunck with early linearisation (matlab/optical_flow/FlowEminHS_elin_2D_v10.m:119-196):
    data terms from the unwarped frames, constant diffusion weight alpha*channels, one Oflow_sor_elin4_2d call.
    param: alpha, b1, b2, iter, omega, solver."""

    def __init__(self, param, mode=capi.MODE_EXACT_ORDER):
        self.p, self.mode = dict(param), mode

    def run(self, It0, It1, U, V):
        p = self.p
        channels = It0.shape[0] if It0.dim() == 3 else 1
        coef = [torch.empty_like(U) for _ in range(5)]
        dev.hs_assemble(It0, It1, p["b1"], p["b2"], *coef)
        W = torch.full_like(U, float(p["alpha"]) * channels)   # W = param.alpha*channels*ones(rows, cols) (:121)
        U, V = U.clone(), V.clone()
        if int(p["iter"]) > 0:
            fn = dev.oflow_sor_elin4 if int(p["solver"]) == 1 else dev.oflow_alr_elin4
            fn(U, V, *coef, W, W, W, W, int(p["iter"]), float(p["omega"]), self.mode)
        return U, V
