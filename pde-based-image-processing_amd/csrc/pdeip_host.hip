// pdeip_host.hip -- libpdeip.so: the host-pointer drop-in entry points (gateway semantics): staging, the *_dev calls, copies back.
//
// Build (build.py): hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -c, one object per translation unit.
// -ffp-contract=off is part of the parity contract: the reference is plain C built without FMA.
#include "pdeip_ctx.hpp"


using namespace pdeip;

namespace {

// ------------------------------------------------------------------------------------------------
// host staging: one arena per call, carved sequentially
// ------------------------------------------------------------------------------------------------
struct Arena {
    float *base = nullptr;
    size_t cap = 0, used = 0;
    int init(size_t nfloats)
    {
        // every plane starts 16-byte aligned so the vector path stays available
        cap = nfloats;
        used = 0;
        return ws_get(WS_ARENA, nfloats * sizeof(float), &base);
    }
    float *take(size_t nfloats)
    {
        float *p = base + used;
        used += (nfloats + 3) & ~(size_t)3;
        return p;
    }
};
int upload(float *dst, const float *src, size_t nfloats)
{
    HIPCHK(hipMemcpy(dst, src, nfloats * sizeof(float), hipMemcpyHostToDevice));
    return PDEIP_OK;
}
int download(float *dst, const float *src, size_t nfloats)
{
    HIPCHK(hipMemcpy(dst, src, nfloats * sizeof(float), hipMemcpyDeviceToHost));
    return PDEIP_OK;
}

} // namespace

// ------------------------------------------------------------------------------------------------
// host-pointer drop-in entry points (gateway semantics)
// ------------------------------------------------------------------------------------------------

// Shared body of Oflow_sor_elin4_2d / Oflow_sor_llin4_2d / Oflow_sor_llin8_2d.
static int oflow_sor_host(const char *who, bool llin, bool fill_residuals, const float *U, const float *V,
                          const float *dU, const float *dV, const float *M, const float *Cu, const float *Cv,
                          const float *Du, const float *Dv, const float *wW, const float *wN, const float *wE,
                          const float *wS, int nrows, int ncols, int F, int iter, float omega, int solver,
                          float *o0, float *o1, float *RU, float *RV, const float *const *diag = nullptr)
{
    NONNULL(who, U); NONNULL(who, V); NONNULL(who, M); NONNULL(who, Cu); NONNULL(who, Cv); NONNULL(who, Du);
    NONNULL(who, Dv); NONNULL(who, wW); NONNULL(who, wN); NONNULL(who, wE); NONNULL(who, wS);
    NONNULL(who, o0); NONNULL(who, o1);
    if (llin) { NONNULL(who, dU); NONNULL(who, dV); }
    if ((RU == nullptr) != (RV == nullptr))
        return set_err(PDEIP_ERR_ARG, "%s: residual outputs RU and RV must be requested together", who);
    RC(check_dims(who, nrows, ncols, F));
    RC(check_solver(who, solver));
    RC(use_device());
    const size_t n = (size_t)nrows * ncols, nf = n * (size_t)F;

    // several devices (pdeip_set_devices / PDEIP_DEVICES): a red-black point-SOR call is cut into column slabs
    if (g.mode == PDEIP_MODE_RED_BLACK && solver == PDEIP_SOLVER_SOR && iter > 0 && RU == nullptr && g.ngroup > 0) {
        MultiCall mc{};
        mc.kind = llin ? 1 : 0;
        mc.n_it = 2; mc.n_ro = llin ? 2 : 0; mc.n_cf = 9; mc.frames = 1;
        mc.it_in[0] = llin ? dU : U; mc.it_in[1] = llin ? dV : V;
        mc.it_out[0] = o0; mc.it_out[1] = o1;
        mc.ro[0] = U; mc.ro[1] = V;
        const float *cfh[9] = {M, Cu, Cv, Du, Dv, wW, wN, wE, wS}; // the solver reads frame 0 of the coefficient planes
        for (int k = 0; k < 9; k++) mc.cf[k] = cfh[k];
        mc.nrows = nrows; mc.ncols = ncols; mc.iter = iter; mc.omega = omega;
        int handled = 0;
        RC(multi_sor(mc, &handled));
        if (handled) return PDEIP_OK;
    }

    Arena ar;
    RC(ar.init(pad4(n) * 16 + pad4(nf) * 7));
    float *dUin = ar.take(n), *dVin = ar.take(n), *ddU = ar.take(n), *ddV = ar.take(n);
    float *ddiag[4] = {nullptr, nullptr, nullptr, nullptr}; // wNW, wNE, wSE, wSW: only the line solvers read them
    if (diag && solver == PDEIP_SOLVER_ALR)
        for (int k = 0; k < 4; k++) {
            ddiag[k] = ar.take(n);
            RC(upload(ddiag[k], diag[k], n));
        }
    else
        diag = nullptr;
    float *dM = ar.take(nf), *dCu = ar.take(nf), *dCv = ar.take(nf), *dDu = ar.take(nf), *dDv = ar.take(nf);
    float *dwW = ar.take(n), *dwN = ar.take(n), *dwE = ar.take(n), *dwS = ar.take(n);
    float *do0 = ar.take(n), *do1 = ar.take(n), *dRU = ar.take(nf), *dRV = ar.take(nf);
    (void)ar.take(n); (void)ar.take(n);
    RC(upload(dUin, U, n)); RC(upload(dVin, V, n));
    if (llin) { RC(upload(ddU, dU, n)); RC(upload(ddV, dV, n)); }
    RC(upload(dM, M, nf)); RC(upload(dCu, Cu, nf)); RC(upload(dCv, Cv, nf)); RC(upload(dDu, Du, nf)); RC(upload(dDv, Dv, nf));
    RC(upload(dwW, wW, n)); RC(upload(dwN, wN, n)); RC(upload(dwE, wE, n)); RC(upload(dwS, wS, n));

    if (iter > 0) { // copy the iterate in, relax it in place (Oflow_sor_elin4_2d.c:341-346); the point solvers relax input -> output
        if (solver == PDEIP_SOLVER_ALR) {
            RC(copy_d2d(0, do0, llin ? ddU : dUin, n));
            RC(copy_d2d(0, do1, llin ? ddV : dVin, n));
        }
        if (solver == PDEIP_SOLVER_ALR && diag)
            RC(pdeip_oflow_alr_llin8_dev(nullptr, dUin, dVin, do0, do1, dM, dCu, dCv, dDu, dDv, dwW, ddiag[0], dwN, ddiag[1], dwE, ddiag[2], dwS, ddiag[3], nrows, ncols, iter, omega, g.mode));
        else if (solver == PDEIP_SOLVER_ALR && llin)
            RC(pdeip_oflow_alr_llin4_dev(nullptr, dUin, dVin, do0, do1, dM, dCu, dCv, dDu, dDv, dwW, dwN, dwE, dwS, nrows, ncols, iter, omega, g.mode));
        else if (solver == PDEIP_SOLVER_ALR)
            RC(pdeip_oflow_alr_elin4_dev(nullptr, do0, do1, dM, dCu, dCv, dDu, dDv, dwW, dwN, dwE, dwS, nrows, ncols, iter, omega, g.mode));
        else if (llin)
            RC(pdeip_oflow_sor_llin4_dev_to(nullptr, dUin, dVin, ddU, ddV, do0, do1, dM, dCu, dCv, dDu, dDv, dwW, dwN, dwE, dwS, nrows, ncols, iter, omega, g.mode, 0));
        else
            RC(pdeip_oflow_sor_elin4_dev_to(nullptr, dUin, dVin, do0, do1, dM, dCu, dCv, dDu, dDv, dwW, dwN, dwE, dwS, nrows, ncols, iter, omega, g.mode, 0));
        RC(download(o0, do0, n));
        RC(download(o1, do1, n));
        RC(pdeip_persist_error());
    } else { // outputs stay as mxCreateNumericArray made them: zero
        memset(o0, 0, n * sizeof(float));
        memset(o1, 0, n * sizeof(float));
    }
    if (RU) { // residuals of the INPUT iterate (:349-350)
        if (!fill_residuals) {
            memset(RU, 0, nf * sizeof(float));
            memset(RV, 0, nf * sizeof(float));
        } else {
            if (llin)
                RC(pdeip_oflow_res_llin4_dev(nullptr, dRU, dRV, dUin, dVin, ddU, ddV, dM, dCu, dCv, dDu, dDv, dwW, dwN, dwE, dwS, nrows, ncols, F));
            else
                RC(pdeip_oflow_res_elin4_dev(nullptr, dRU, dRV, dUin, dVin, dM, dCu, dCv, dDu, dDv, dwW, dwN, dwE, dwS, nrows, ncols, F));
            RC(download(RU, dRU, nf));
            RC(download(RV, dRV, nf));
        }
    }
    HIPCHK(hipDeviceSynchronize());
    return PDEIP_OK;
}

extern "C" int pdeip_oflow_sor_elin4(const float *U, const float *V, const float *M, const float *Cu,
                                     const float *Cv, const float *Du, const float *Dv, const float *wW,
                                     const float *wN, const float *wE, const float *wS, int nrows, int ncols,
                                     int nframes_coef, int iter, float omega, int solver, float *U_out,
                                     float *V_out, float *RU, float *RV)
{
    return oflow_sor_host("Oflow_sor_elin4_2d", false, true, U, V, nullptr, nullptr, M, Cu, Cv, Du, Dv, wW, wN, wE,
                          wS, nrows, ncols, nframes_coef, iter, omega, solver, U_out, V_out, RU, RV);
}

extern "C" int pdeip_oflow_sor_llin4(const float *U, const float *V, const float *dU, const float *dV,
                                     const float *M, const float *Cu, const float *Cv, const float *Du,
                                     const float *Dv, const float *wW, const float *wN, const float *wE,
                                     const float *wS, int nrows, int ncols, int nframes_coef, int iter,
                                     float omega, int solver, float *dU_out, float *dV_out, float *RU, float *RV)
{
    return oflow_sor_host("Oflow_sor_llin4_2d", true, true, U, V, dU, dV, M, Cu, Cv, Du, Dv, wW, wN, wE, wS, nrows,
                          ncols, nframes_coef, iter, omega, solver, dU_out, dV_out, RU, RV);
}

extern "C" int pdeip_oflow_sor_llin8(const float *U, const float *V, const float *dU, const float *dV,
                                     const float *M, const float *Cu, const float *Cv, const float *Du,
                                     const float *Dv, const float *wW, const float *wNW, const float *wN,
                                     const float *wNE, const float *wE, const float *wSE, const float *wS,
                                     const float *wSW, int nrows, int ncols, int nframes_coef, int iter,
                                     float omega, int solver, float *dU_out, float *dV_out, float *RU, float *RV)
{
    const char *who = "Oflow_sor_llin8_2d";
    NONNULL(who, wNW); NONNULL(who, wNE); NONNULL(who, wSE); NONNULL(who, wSW);
    // GS_SOR_llin8_2d never reads the diagonal weights (opticalflowSolvers.c:1550-1591) -- only the line
    // solvers do -- and the gateway leaves RU,RV unfilled (Oflow_sor_llin8_2d.c:466-488).
    const float *diag[4] = {wNW, wNE, wSE, wSW};
    return oflow_sor_host(who, true, false, U, V, dU, dV, M, Cu, Cv, Du, Dv, wW, wN, wE, wS, nrows, ncols,
                          nframes_coef, iter, omega, solver, dU_out, dV_out, RU, RV, diag);
}

static int oflow_lhs_host(const char *who, bool llin, const float *U, const float *V, const float *dU,
                          const float *dV, const float *M, const float *Du, const float *Dv, const float *wW,
                          const float *wN, const float *wE, const float *wS, int nrows, int ncols, int F,
                          float *AU, float *AV)
{
    NONNULL(who, U); NONNULL(who, V); NONNULL(who, M); NONNULL(who, Du); NONNULL(who, Dv); NONNULL(who, wW);
    NONNULL(who, wN); NONNULL(who, wE); NONNULL(who, wS); NONNULL(who, AU); NONNULL(who, AV);
    if (llin) { NONNULL(who, dU); NONNULL(who, dV); }
    RC(check_dims(who, nrows, ncols, F));
    RC(use_device());
    const size_t n = (size_t)nrows * ncols, nf = n * (size_t)F;
    Arena ar;
    RC(ar.init(pad4(n) * 8 + pad4(nf) * 5));
    float *dUin = ar.take(n), *dVin = ar.take(n), *ddU = ar.take(n), *ddV = ar.take(n);
    float *dM = ar.take(nf), *dDu = ar.take(nf), *dDv = ar.take(nf);
    float *dwW = ar.take(n), *dwN = ar.take(n), *dwE = ar.take(n), *dwS = ar.take(n);
    float *dAU = ar.take(nf), *dAV = ar.take(nf);
    RC(upload(dUin, U, n)); RC(upload(dVin, V, n));
    if (llin) { RC(upload(ddU, dU, n)); RC(upload(ddV, dV, n)); }
    RC(upload(dM, M, nf)); RC(upload(dDu, Du, nf)); RC(upload(dDv, Dv, nf));
    RC(upload(dwW, wW, n)); RC(upload(dwN, wN, n)); RC(upload(dwE, wE, n)); RC(upload(dwS, wS, n));
    if (llin) RC(pdeip_oflow_lhs_llin4_dev(nullptr, dAU, dAV, dUin, dVin, ddU, ddV, dM, dDu, dDv, dwW, dwN, dwE, dwS, nrows, ncols, F));
    else RC(pdeip_oflow_lhs_elin4_dev(nullptr, dAU, dAV, dUin, dVin, dM, dDu, dDv, dwW, dwN, dwE, dwS, nrows, ncols, F));
    RC(download(AU, dAU, nf));
    RC(download(AV, dAV, nf));
    return PDEIP_OK;
}

extern "C" int pdeip_oflow_lhs_elin4(const float *U, const float *V, const float *M, const float *Du,
                                     const float *Dv, const float *wW, const float *wN, const float *wE,
                                     const float *wS, int nrows, int ncols, int nframes_coef, float *AU, float *AV)
{
    return oflow_lhs_host("Oflow_lhs_elin4_2d", false, U, V, nullptr, nullptr, M, Du, Dv, wW, wN, wE, wS, nrows,
                          ncols, nframes_coef, AU, AV);
}

extern "C" int pdeip_oflow_lhs_llin4(const float *U, const float *V, const float *dU, const float *dV,
                                     const float *M, const float *Du, const float *Dv, const float *wW,
                                     const float *wN, const float *wE, const float *wS, int nrows, int ncols,
                                     int nframes_coef, float *AU, float *AV)
{
    return oflow_lhs_host("Oflow_lhs_llin4_2d", true, U, V, dU, dV, M, Du, Dv, wW, wN, wE, wS, nrows, ncols,
                          nframes_coef, AU, AV);
}

extern "C" int pdeip_disp_sor_llin4(const float *U, const float *dU, const float *Cu, const float *Du,
                                    const float *wW, const float *wN, const float *wE, const float *wS,
                                    int nrows, int ncols, int iter, float omega, int solver, float *dU_out, float *RU)
{
    const char *who = "Disp_sor_llin4_2d";
    NONNULL(who, U); NONNULL(who, dU); NONNULL(who, Cu); NONNULL(who, Du); NONNULL(who, wW); NONNULL(who, wN);
    NONNULL(who, wE); NONNULL(who, wS); NONNULL(who, dU_out);
    RC(check_dims(who, nrows, ncols, 1));
    RC(check_solver(who, solver));
    RC(use_device());
    const size_t n = (size_t)nrows * ncols;
    if (RU) memset(RU, 0, n * sizeof(float)); // allocated, never computed (Disp_sor_llin4_2d.c:251-281)
    if (iter <= 0) { // output stays zero (:276-280)
        memset(dU_out, 0, n * sizeof(float));
        return PDEIP_OK;
    }
    if (g.mode == PDEIP_MODE_RED_BLACK && solver == PDEIP_SOLVER_SOR) {
        MultiCall mc{};
        mc.kind = 2;
        mc.n_it = 1; mc.n_ro = 1; mc.n_cf = 6; mc.frames = 1;
        mc.it_in[0] = dU; mc.it_out[0] = dU_out; mc.ro[0] = U;
        const float *cfh[6] = {Cu, Du, wW, wN, wE, wS};
        for (int k = 0; k < 6; k++) mc.cf[k] = cfh[k];
        mc.nrows = nrows; mc.ncols = ncols; mc.iter = iter; mc.omega = omega;
        int handled = 0;
        RC(multi_sor(mc, &handled));
        if (handled) return PDEIP_OK;
    }
    Arena ar;
    RC(ar.init(pad4(n) * 8));
    float *dUin = ar.take(n), *ddU = ar.take(n), *dCu = ar.take(n), *dDu = ar.take(n);
    float *dwW = ar.take(n), *dwN = ar.take(n), *dwE = ar.take(n), *dwS = ar.take(n);
    RC(upload(dUin, U, n)); RC(upload(ddU, dU, n)); RC(upload(dCu, Cu, n)); RC(upload(dDu, Du, n));
    RC(upload(dwW, wW, n)); RC(upload(dwN, wN, n)); RC(upload(dwE, wE, n)); RC(upload(dwS, wS, n));
    if (solver == PDEIP_SOLVER_ALR)
        RC(pdeip_disp_alr_llin4_dev(nullptr, dUin, ddU, dCu, dDu, dwW, dwN, dwE, dwS, nrows, ncols, iter, omega, g.mode));
    else
        RC(pdeip_disp_sor_llin4_dev(nullptr, dUin, ddU, dCu, dDu, dwW, dwN, dwE, dwS, nrows, ncols, iter, omega, g.mode, 0));
    RC(download(dU_out, ddU, n));
    RC(pdeip_persist_error());
    return PDEIP_OK;
}

// [dU0 dU1] = Disp_sor_llin_sym4_2d(U0,dU0,Cu0,Du0,wW0,wN0,wE0,wS0, U1,dU1,Cu1,Du1,wW1,wN1,wE1,wS1, iter,omega,solver)
// The gateway copies both increments in and solves unconditionally (Disp_sor_llin_sym4_2d.c:418-440): iter <= 0 returns copies.
extern "C" int pdeip_disp_sor_llin_sym4(const float *U0, const float *dU0, const float *Cu0, const float *Du0, const float *wW0,
                                        const float *wN0, const float *wE0, const float *wS0, const float *U1, const float *dU1,
                                        const float *Cu1, const float *Du1, const float *wW1, const float *wN1, const float *wE1,
                                        const float *wS1, int nrows, int ncols, int iter, float omega, int solver,
                                        float *dU_out0, float *dU_out1)
{
    const char *who = "Disp_sor_llin_sym4_2d";
    const float *in[16] = {U0, dU0, Cu0, Du0, wW0, wN0, wE0, wS0, U1, dU1, Cu1, Du1, wW1, wN1, wE1, wS1};
    for (int k = 0; k < 16; k++)
        if (!in[k]) return set_err(PDEIP_ERR_ARG, "%s: input %d is NULL", who, k + 1);
    NONNULL(who, dU_out0); NONNULL(who, dU_out1);
    RC(check_dims(who, nrows, ncols, 1));
    RC(check_solver(who, solver));
    RC(use_device());
    const size_t n = (size_t)nrows * ncols;
    if (iter <= 0) {
        memcpy(dU_out0, dU0, n * sizeof(float));
        memcpy(dU_out1, dU1, n * sizeof(float));
        return PDEIP_OK;
    }
    Arena ar;
    RC(ar.init(pad4(n) * 16));
    float *d[16];
    for (int k = 0; k < 16; k++) {
        d[k] = ar.take(n);
        RC(upload(d[k], in[k], n));
    }
    RC(pdeip_disp_sor_llin_sym4_dev(nullptr, d[0], d[1], d[2], d[3], d[4], d[5], d[6], d[7], d[8], d[9], d[10], d[11], d[12], d[13],
                                    d[14], d[15], nrows, ncols, iter, omega, solver, g.mode, 0));
    RC(download(dU_out0, d[1], n));
    RC(download(dU_out1, d[9], n));
    RC(pdeip_persist_error());
    return PDEIP_OK;
}

// The PDE gateways accept solver 3 and then call an unbound function pointer (PDEsolver4.c:228);
// that is rejected here like any other unknown solver.
extern "C" int pdeip_pde_sor4(const float *X, const float *TRACE, const float *B, const float *wW, const float *wN,
                              const float *wE, const float *wS, int nrows, int ncols, int nframes, int iter,
                              float omega, int solver, float *X_out)
{
    const char *who = "PDEsolver4";
    NONNULL(who, X); NONNULL(who, TRACE); NONNULL(who, B); NONNULL(who, wW); NONNULL(who, wN); NONNULL(who, wE);
    NONNULL(who, wS); NONNULL(who, X_out);
    RC(check_dims(who, nrows, ncols, nframes));
    RC(check_solver(who, solver));
    RC(use_device());
    const size_t nf = (size_t)nrows * ncols * nframes;
    if (iter <= 0) { // copy-in, zero sweeps (PDEsolver4.c:239-240)
        memcpy(X_out, X, nf * sizeof(float));
        return PDEIP_OK;
    }
    if (g.mode == PDEIP_MODE_RED_BLACK && solver == PDEIP_SOLVER_SOR) {
        MultiCall mc{};
        mc.kind = 3;
        mc.n_it = 1; mc.n_ro = 0; mc.n_cf = 6; mc.frames = nframes;
        mc.it_in[0] = X; mc.it_out[0] = X_out;
        const float *cfh[6] = {TRACE, B, wW, wN, wE, wS};
        for (int k = 0; k < 6; k++) mc.cf[k] = cfh[k];
        mc.nrows = nrows; mc.ncols = ncols; mc.iter = iter; mc.omega = omega;
        int handled = 0;
        RC(multi_sor(mc, &handled));
        if (handled) return PDEIP_OK;
    }
    Arena ar;
    RC(ar.init(pad4(nf) * 7));
    float *dX = ar.take(nf), *dT = ar.take(nf), *dB = ar.take(nf);
    float *dwW = ar.take(nf), *dwN = ar.take(nf), *dwE = ar.take(nf), *dwS = ar.take(nf);
    RC(upload(dX, X, nf)); RC(upload(dT, TRACE, nf)); RC(upload(dB, B, nf));
    RC(upload(dwW, wW, nf)); RC(upload(dwN, wN, nf)); RC(upload(dwE, wE, nf)); RC(upload(dwS, wS, nf));
    if (solver == PDEIP_SOLVER_ALR)
        RC(pdeip_pde_alr4_dev(nullptr, dX, dT, dB, dwW, dwN, dwE, dwS, nrows, ncols, nframes, iter, omega, g.mode));
    else
        RC(pdeip_pde_sor4_dev(nullptr, dX, dT, dB, dwW, dwN, dwE, dwS, nrows, ncols, nframes, iter, omega, g.mode, 0));
    RC(download(X_out, dX, nf));
    RC(pdeip_persist_error());
    return PDEIP_OK;
}

extern "C" int pdeip_pde_sor8(const float *X, const float *TRACE, const float *B, const float *wW, const float *wNW,
                              const float *wN, const float *wNE, const float *wE, const float *wSE, const float *wS,
                              const float *wSW, int nrows, int ncols, int nframes, int iter, float omega,
                              int solver, float *X_out)
{
    const char *who = "PDEsolver8";
    NONNULL(who, X); NONNULL(who, TRACE); NONNULL(who, B); NONNULL(who, wW); NONNULL(who, wNW); NONNULL(who, wN);
    NONNULL(who, wNE); NONNULL(who, wE); NONNULL(who, wSE); NONNULL(who, wS); NONNULL(who, wSW); NONNULL(who, X_out);
    RC(check_dims(who, nrows, ncols, nframes));
    RC(check_solver(who, solver));
    RC(use_device());
    const size_t nf = (size_t)nrows * ncols * nframes;
    if (iter <= 0 && solver != PDEIP_SOLVER_ALR) { // GS_ALR_SOR_8_2d runs its one iteration regardless (pdeSolvers.c:362)
        memcpy(X_out, X, nf * sizeof(float));
        return PDEIP_OK;
    }
    if (g.mode == PDEIP_MODE_RED_BLACK && solver == PDEIP_SOLVER_SOR && iter > 0) {
        MultiCall mc{};
        mc.kind = 4;
        mc.n_it = 1; mc.n_ro = 0; mc.n_cf = 10; mc.frames = nframes;
        mc.it_in[0] = X; mc.it_out[0] = X_out;
        const float *cfh[10] = {TRACE, B, wW, wNW, wN, wNE, wE, wSE, wS, wSW};
        for (int k = 0; k < 10; k++) mc.cf[k] = cfh[k];
        mc.nrows = nrows; mc.ncols = ncols; mc.iter = iter; mc.omega = omega;
        int handled = 0;
        RC(multi_sor(mc, &handled));
        if (handled) return PDEIP_OK;
    }
    Arena ar;
    RC(ar.init(pad4(nf) * 11));
    float *dX = ar.take(nf), *dT = ar.take(nf), *dB = ar.take(nf);
    const float *hw[8] = {wW, wNW, wN, wNE, wE, wSE, wS, wSW};
    float *dw[8];
    RC(upload(dX, X, nf)); RC(upload(dT, TRACE, nf)); RC(upload(dB, B, nf));
    for (int k = 0; k < 8; k++) {
        dw[k] = ar.take(nf);
        RC(upload(dw[k], hw[k], nf));
    }
    if (solver == PDEIP_SOLVER_ALR)
        RC(pdeip_pde_alr8_dev(nullptr, dX, dT, dB, dw[0], dw[1], dw[2], dw[3], dw[4], dw[5], dw[6], dw[7], nrows, ncols,
                              nframes, iter, omega, g.mode));
    else
        RC(pdeip_pde_sor8_dev(nullptr, dX, dT, dB, dw[0], dw[1], dw[2], dw[3], dw[4], dw[5], dw[6], dw[7], nrows, ncols,
                              nframes, iter, omega, g.mode, 0));
    RC(download(X_out, dX, nf));
    return PDEIP_OK;
}

extern "C" int pdeip_diffweights6(const float *D, int nrows, int ncols, int nframes, float eps, float *wW,
                                  float *wN, float *wE, float *wS)
{
    const char *who = "DdiffWeights";
    NONNULL(who, D); NONNULL(who, wW); NONNULL(who, wN); NONNULL(who, wE); NONNULL(who, wS);
    RC(check_dims(who, nrows, ncols, nframes));
    RC(use_device());
    const size_t n = (size_t)nrows * ncols, nf = n * (size_t)nframes;
    Arena ar;
    RC(ar.init(pad4(nf) + pad4(n) * 4));
    float *dD = ar.take(nf), *d0 = ar.take(n), *d1 = ar.take(n), *d2 = ar.take(n), *d3 = ar.take(n);
    RC(upload(dD, D, nf));
    RC(pdeip_diffweights6_dev(nullptr, dD, nrows, ncols, nframes, eps, d0, d1, d2, d3));
    float *outs[4] = {wW, wN, wE, wS};
    float *dev[4] = {d0, d1, d2, d3};
    for (int k = 0; k < 4; k++) {
        RC(download(outs[k], dev[k], n));
        // outputs carry D's dimensions; only frame 0 is written (DdiffWeights.c:97-138)
        if (nframes > 1) memset(outs[k] + n, 0, (nf - n) * sizeof(float));
    }
    return PDEIP_OK;
}

extern "C" int pdeip_warp_bilinear(const float *Iin, const float *X, const float *Y, int nrows, int ncols,
                                   int nframes, float *Iout)
{
    const char *who = "BilinInterp_2d";
    NONNULL(who, Iin); NONNULL(who, X); NONNULL(who, Y); NONNULL(who, Iout);
    RC(check_dims(who, nrows, ncols, nframes));
    RC(use_device());
    const size_t n = (size_t)nrows * ncols, nf = n * (size_t)nframes;
    Arena ar;
    RC(ar.init(pad4(nf) * 2 + pad4(n) * 2));
    float *dI = ar.take(nf), *dX = ar.take(n), *dY = ar.take(n), *dO = ar.take(nf);
    RC(upload(dI, Iin, nf)); RC(upload(dX, X, n)); RC(upload(dY, Y, n));
    RC(pdeip_warp_bilinear_dev(nullptr, dI, dX, dY, nrows, ncols, nframes, dO));
    RC(download(Iout, dO, nf));
    return PDEIP_OK;
}

extern "C" int pdeip_fst_derivatives5(const float *It0, const float *It1, int nrows, int ncols, int nframes,
                                      float *Idt, float *Idx, float *Idy)
{
    const char *who = "FstDerivatives5";
    NONNULL(who, It0); NONNULL(who, It1); NONNULL(who, Idt); NONNULL(who, Idx); NONNULL(who, Idy);
    RC(check_deriv_dims(who, nrows, ncols, nframes));
    RC(use_device());
    const size_t nf = (size_t)nrows * ncols * nframes;
    Arena ar;
    RC(ar.init(pad4(nf) * 5));
    float *d0 = ar.take(nf), *d1 = ar.take(nf), *o0 = ar.take(nf), *o1 = ar.take(nf), *o2 = ar.take(nf);
    RC(upload(d0, It0, nf)); RC(upload(d1, It1, nf));
    RC(pdeip_fst_derivatives5_dev(nullptr, d0, d1, nrows, ncols, nframes, o0, o1, o2));
    RC(download(Idt, o0, nf)); RC(download(Idx, o1, nf)); RC(download(Idy, o2, nf));
    return PDEIP_OK;
}

extern "C" int pdeip_snd_derivatives5(const float *It0, const float *It1, int nrows, int ncols, int nframes,
                                      float *Idxt, float *Idyt, float *Idxx, float *Idyy, float *Idxy)
{
    const char *who = "SndDerivatives5";
    NONNULL(who, It0); NONNULL(who, It1); NONNULL(who, Idxt); NONNULL(who, Idyt); NONNULL(who, Idxx); NONNULL(who, Idyy);
    NONNULL(who, Idxy);
    RC(check_deriv_dims(who, nrows, ncols, nframes));
    RC(use_device());
    const size_t nf = (size_t)nrows * ncols * nframes;
    Arena ar;
    RC(ar.init(pad4(nf) * 7));
    float *d0 = ar.take(nf), *d1 = ar.take(nf);
    float *o[5];
    for (int k = 0; k < 5; k++) o[k] = ar.take(nf);
    RC(upload(d0, It0, nf)); RC(upload(d1, It1, nf));
    RC(pdeip_snd_derivatives5_dev(nullptr, d0, d1, nrows, ncols, nframes, o[0], o[1], o[2], o[3], o[4]));
    float *h[5] = {Idxt, Idyt, Idxx, Idyy, Idxy};
    for (int k = 0; k < 5; k++) RC(download(h[k], o[k], nf));
    return PDEIP_OK;
}
