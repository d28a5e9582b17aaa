// pdeip_stages.hip -- libpdeip.so: pointwise operators and the MATLAB-side stages (flow / FAS / TV / pyramid / symmetric stereo) as *_dev entry points.
//
// Build (build.py): hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -c, one object per translation unit.
// -ffp-contract=off is part of the parity contract: the reference is plain C built without FMA.
#include "pdeip_ctx.hpp"

#include "pdeip_models.hpp"
#include "pdeip_pointwise.hpp"
#include "pdeip_flow.hpp"
#include "pdeip_fas.hpp"
#include "pdeip_sym.hpp"
#include "pdeip_pyr.hpp"
#include "pdeip_tv.hpp"

using namespace pdeip;

extern "C" int pdeip_oflow_res_elin4_dev(void *stream, float *RU, float *RV, const float *U, const float *V,
                                         const float *M, const float *Cu, const float *Cv, const float *Du,
                                         const float *Dv, const float *wW, const float *wN, const float *wE,
                                         const float *wS, int nrows, int ncols, int nframes_coef)
{
    RC(check_dims("pdeip_oflow_res_elin4_dev", nrows, ncols, nframes_coef));
    hipLaunchKernelGGL((k_oflow_operator<false, false>), pixel_grid(nrows, ncols, nframes_coef), dim3(256), 0,
                       static_cast<hipStream_t>(stream), RU, RV, U, V, nullptr, nullptr, M, Cu, Cv, Du, Dv, wW,
                       wN, wE, wS, nrows, ncols, (size_t)nrows * ncols);
    HIPCHK(hipGetLastError());
    return PDEIP_OK;
}

extern "C" int pdeip_oflow_lhs_elin4_dev(void *stream, float *AU, float *AV, const float *U, const float *V,
                                         const float *M, const float *Du, const float *Dv, const float *wW,
                                         const float *wN, const float *wE, const float *wS, int nrows,
                                         int ncols, int nframes_coef)
{
    RC(check_dims("pdeip_oflow_lhs_elin4_dev", nrows, ncols, nframes_coef));
    hipLaunchKernelGGL((k_oflow_operator<false, true>), pixel_grid(nrows, ncols, nframes_coef), dim3(256), 0,
                       static_cast<hipStream_t>(stream), AU, AV, U, V, nullptr, nullptr, M, nullptr, nullptr, Du, Dv,
                       wW, wN, wE, wS, nrows, ncols, (size_t)nrows * ncols);
    HIPCHK(hipGetLastError());
    return PDEIP_OK;
}

extern "C" int pdeip_oflow_res_llin4_dev(void *stream, float *RU, float *RV, const float *U, const float *V,
                                         const float *dU, const float *dV, const float *M, const float *Cu,
                                         const float *Cv, const float *Du, const float *Dv, const float *wW,
                                         const float *wN, const float *wE, const float *wS, int nrows,
                                         int ncols, int nframes_coef)
{
    RC(check_dims("pdeip_oflow_res_llin4_dev", nrows, ncols, nframes_coef));
    hipLaunchKernelGGL((k_oflow_operator<true, false>), pixel_grid(nrows, ncols, nframes_coef), dim3(256), 0,
                       static_cast<hipStream_t>(stream), RU, RV, U, V, dU, dV, M, Cu, Cv, Du, Dv, wW, wN, wE, wS,
                       nrows, ncols, (size_t)nrows * ncols);
    HIPCHK(hipGetLastError());
    return PDEIP_OK;
}

extern "C" int pdeip_oflow_lhs_llin4_dev(void *stream, float *AU, float *AV, const float *U, const float *V,
                                         const float *dU, const float *dV, const float *M, const float *Du,
                                         const float *Dv, const float *wW, const float *wN, const float *wE,
                                         const float *wS, int nrows, int ncols, int nframes_coef)
{
    RC(check_dims("pdeip_oflow_lhs_llin4_dev", nrows, ncols, nframes_coef));
    hipLaunchKernelGGL((k_oflow_operator<true, true>), pixel_grid(nrows, ncols, nframes_coef), dim3(256), 0,
                       static_cast<hipStream_t>(stream), AU, AV, U, V, dU, dV, M, nullptr, nullptr, Du, Dv, wW, wN,
                       wE, wS, nrows, ncols, (size_t)nrows * ncols);
    HIPCHK(hipGetLastError());
    return PDEIP_OK;
}

extern "C" int pdeip_diffweights6_dev(void *stream, const float *D, int nrows, int ncols, int nframes,
                                      float eps, float *wW, float *wN, float *wE, float *wS)
{
    RC(check_dims("pdeip_diffweights6_dev", nrows, ncols, nframes));
    hipLaunchKernelGGL(k_diffweights6, pixel_grid(nrows, ncols, 1), dim3(256), 0, static_cast<hipStream_t>(stream),
                       wW, wN, wE, wS, D, nrows, ncols, nframes, eps);
    HIPCHK(hipGetLastError());
    return PDEIP_OK;
}

extern "C" int pdeip_warp_bilinear_dev(void *stream, const float *Iin, const float *X, const float *Y,
                                       int nrows, int ncols, int nframes, float *Iout)
{
    RC(check_dims("pdeip_warp_bilinear_dev", nrows, ncols, nframes));
    hipLaunchKernelGGL(k_warp_bilinear, pixel_grid(nrows, ncols, 1), dim3(256), 0, static_cast<hipStream_t>(stream),
                       Iout, Iin, X, Y, nrows, ncols, nframes);
    HIPCHK(hipGetLastError());
    return PDEIP_OK;
}

extern "C" int pdeip_fst_derivatives5_dev(void *stream, const float *It0, const float *It1, int nrows, int ncols,
                                          int nframes, float *Idt, float *Idx, float *Idy)
{
    RC(check_deriv_dims("pdeip_fst_derivatives5_dev", nrows, ncols, nframes));
    hipLaunchKernelGGL(k_derivatives5_tiled<false>, dim3((nrows + D5_TR - 1) / D5_TR, (ncols + D5_TC - 1) / D5_TC, nframes), dim3(D5_TR, D5_TC), 0,
                       static_cast<hipStream_t>(stream), Idt, Idx, Idy, nullptr, nullptr, It0, It1, nrows, ncols, (size_t)nrows * ncols);
    HIPCHK(hipGetLastError());
    return PDEIP_OK;
}

extern "C" int pdeip_snd_derivatives5_dev(void *stream, const float *It0, const float *It1, int nrows, int ncols,
                                          int nframes, float *Idxt, float *Idyt, float *Idxx, float *Idyy, float *Idxy)
{
    RC(check_deriv_dims("pdeip_snd_derivatives5_dev", nrows, ncols, nframes));
    hipLaunchKernelGGL(k_derivatives5_tiled<true>, dim3((nrows + D5_TR - 1) / D5_TR, (ncols + D5_TC - 1) / D5_TC, nframes), dim3(D5_TR, D5_TC), 0,
                       static_cast<hipStream_t>(stream), Idxt, Idyt, Idxx, Idyy, Idxy, It0, It1, nrows, ncols, (size_t)nrows * ncols);
    HIPCHK(hipGetLastError());
    return PDEIP_OK;
}

// ------------------------------------------------------------------------------------------------
// MATLAB-side stages of one late-linearisation pyramid level (pdeip_flow.hpp), device-resident only
// ------------------------------------------------------------------------------------------------
extern "C" int pdeip_flow_coords_dev(void *stream, const float *U, const float *V, int nrows, int ncols, float *X, float *Y)
{
    RC(check_dims("pdeip_flow_coords_dev", nrows, ncols, 1));
    hipLaunchKernelGGL(k_flow_coords, pixel_grid(nrows, ncols, 1), dim3(256), 0, static_cast<hipStream_t>(stream), X, Y, U, V, nrows, ncols);
    HIPCHK(hipGetLastError());
    return PDEIP_OK;
}

extern "C" int pdeip_flow_warp_dev(void *stream, const float *U, const float *V, const float *I1, int C1, const float *I2, int C2, int nrows,
                                   int ncols, float *W1, float *W2)
{
    const char *who = "pdeip_flow_warp_dev";
    RC(check_dims(who, nrows, ncols, C1));
    if (!U || !I1 || !W1 || C2 < 0 || (C2 > 0 && (!I2 || !W2))) return set_err(PDEIP_ERR_ARG, "%s: missing plane", who);
    hipLaunchKernelGGL(k_flow_warp, pixel_grid(nrows, ncols, 1), dim3(256), 0, static_cast<hipStream_t>(stream), W1, I1, C1, W2, I2, C2, U, V,
                       nrows, ncols);
    HIPCHK(hipGetLastError());
    return PDEIP_OK;
}

extern "C" int pdeip_flow_assemble_dev(void *stream, const float *It1, const float *Ix1, const float *Iy1, int C1, float b1,
                                       const float *It2, const float *Ix2, const float *Iy2, int C2, float b2, const float *dU,
                                       const float *dV, float alpha, int nrows, int ncols, float *MGd, float *CuGd, float *CvGd,
                                       float *DuGd, float *DvGd)
{
    const char *who = "pdeip_flow_assemble_dev";
    RC(check_dims(who, nrows, ncols, C1));
    if (C2 < 0 || (C2 > 0 && (!It2 || !Ix2 || !Iy2))) return set_err(PDEIP_ERR_ARG, "%s: second data term needs its three derivative arrays", who);
    const FlowTerm t1{It1, Ix1, Iy1, C1, b1}, t2{It2, Ix2, Iy2, C2, b2};
    hipLaunchKernelGGL(k_flow_assemble, pixel_grid(nrows, ncols, 1), dim3(256), 0, static_cast<hipStream_t>(stream), MGd, CuGd, CvGd, DuGd,
                       DvGd, t1, t2, dU, dV, alpha, nrows, ncols, FlowWeightsOut{});
    HIPCHK(hipGetLastError());
    return PDEIP_OK;
}

extern "C" int pdeip_flow_assemble_weights_dev(void *stream, const float *It1, const float *Ix1, const float *Iy1, int C1, float b1,
                                               const float *A2, const float *B2, const float *C2p, const float *Iyy, const float *Ixy, int C2,
                                               float b2, const float *U, const float *V, const float *dU, const float *dV, float alpha, int nrows,
                                               int ncols, float *MGd, float *CuGd, float *CvGd, float *DuGd, float *DvGd, float *wW, float *wN,
                                               float *wS, float *wE)
{
    const char *who = "pdeip_flow_assemble_weights_dev";
    RC(check_dims(who, nrows, ncols, C1));
    if (C2 < 0 || (C2 > 0 && (!A2 || !B2 || !C2p)) || ((Iyy != nullptr) != (Ixy != nullptr)))
        return set_err(PDEIP_ERR_ARG, "%s: second data term needs its three (or, gradient magnitude, five) derivative arrays", who);
    if (!U || !V || !dU || !dV || !wW || !wN || !wS || !wE) return set_err(PDEIP_ERR_ARG, "%s: missing plane", who);
    const FlowTerm t1{It1, Ix1, Iy1, C1, b1}, t2{A2, B2, C2p, C2, b2, Iyy, Ixy};
    FlowWeightsOut W;
    W.wW = wW; W.wN = wN; W.wS = wS; W.wE = wE; W.U = U; W.V = V;
    hipLaunchKernelGGL(k_flow_assemble, pixel_grid(nrows, ncols, 1), dim3(256), 0, static_cast<hipStream_t>(stream), MGd, CuGd, CvGd, DuGd,
                       DvGd, t1, t2, dU, dV, alpha, nrows, ncols, W);
    HIPCHK(hipGetLastError());
    return PDEIP_OK;
}

extern "C" int pdeip_disp_assemble_dev(void *stream, const float *It1, const float *Ix1, int C1, float b1, const float *It2,
                                       const float *Ix2, int C2, float b2, const float *dU, float alpha, int nrows, int ncols,
                                       float *CuGd, float *DuGd)
{
    const char *who = "pdeip_disp_assemble_dev";
    RC(check_dims(who, nrows, ncols, C1));
    if (C2 < 0 || (C2 > 0 && (!It2 || !Ix2))) return set_err(PDEIP_ERR_ARG, "%s: second data term needs its derivative arrays", who);
    const FlowTerm t1{It1, Ix1, nullptr, C1, b1}, t2{It2, Ix2, nullptr, C2, b2};
    hipLaunchKernelGGL(k_disp_assemble, pixel_grid(nrows, ncols, 1), dim3(256), 0, static_cast<hipStream_t>(stream), CuGd, DuGd, t1, t2, dU,
                       alpha, nrows, ncols);
    HIPCHK(hipGetLastError());
    return PDEIP_OK;
}

extern "C" int pdeip_flow_assemble_gradmag_dev(void *stream, const float *It1, const float *Ix1, const float *Iy1, int C1, float b1,
                                               const float *Ixt, const float *Iyt, const float *Ixx, const float *Iyy, const float *Ixy, int C2,
                                               float b2, const float *dU, const float *dV, float alpha, int nrows, int ncols, float *MGd,
                                               float *CuGd, float *CvGd, float *DuGd, float *DvGd)
{
    const char *who = "pdeip_flow_assemble_gradmag_dev";
    RC(check_dims(who, nrows, ncols, C1));
    if (C2 < 1 || !Ixt || !Iyt || !Ixx || !Iyy || !Ixy) return set_err(PDEIP_ERR_ARG, "%s: the gradient-magnitude term needs its five derivative arrays", who);
    const FlowTerm t1{It1, Ix1, Iy1, C1, b1}, t2{Ixt, Iyt, Ixx, C2, b2, Iyy, Ixy};
    hipLaunchKernelGGL(k_flow_assemble, pixel_grid(nrows, ncols, 1), dim3(256), 0, static_cast<hipStream_t>(stream), MGd, CuGd, CvGd, DuGd,
                       DvGd, t1, t2, dU, dV, alpha, nrows, ncols, FlowWeightsOut{});
    HIPCHK(hipGetLastError());
    return PDEIP_OK;
}

extern "C" int pdeip_disp_assemble_gradmag_dev(void *stream, const float *It1, const float *Ix1, int C1, float b1, const float *Ixt,
                                               const float *Iyt, const float *Ixx, const float *Ixy, int C2, float b2, const float *dU, float alpha,
                                               int nrows, int ncols, float *CuGd, float *DuGd)
{
    const char *who = "pdeip_disp_assemble_gradmag_dev";
    RC(check_dims(who, nrows, ncols, C1));
    if (C2 < 1 || !Ixt || !Iyt || !Ixx || !Ixy) return set_err(PDEIP_ERR_ARG, "%s: the gradient-magnitude term needs its four derivative arrays", who);
    const FlowTerm t1{It1, Ix1, nullptr, C1, b1}, t2{Ixt, Iyt, Ixx, C2, b2, nullptr, Ixy};
    hipLaunchKernelGGL(k_disp_assemble, pixel_grid(nrows, ncols, 1), dim3(256), 0, static_cast<hipStream_t>(stream), CuGd, DuGd, t1, t2, dU,
                       alpha, nrows, ncols);
    HIPCHK(hipGetLastError());
    return PDEIP_OK;
}

extern "C" int pdeip_flow_apriori_dev(void *stream, const double *Us, const float *U, const float *dU, double gammaS, double alpha,
                                      double as_diff, int u_double, int du_double, int nrows, int ncols, float *CGd, float *DGd)
{
    RC(check_dims("pdeip_flow_apriori_dev", nrows, ncols, 1));
    if (!Us || !U || !dU || !CGd || !DGd) return set_err(PDEIP_ERR_ARG, "pdeip_flow_apriori_dev: null plane");
    hipLaunchKernelGGL(k_flow_apriori, pixel_grid(nrows, ncols, 1), dim3(256), 0, static_cast<hipStream_t>(stream), CGd, DGd, Us, U, dU, gammaS,
                       alpha, as_diff * as_diff, u_double, du_double, nrows, ncols);
    HIPCHK(hipGetLastError());
    return PDEIP_OK;
}

extern "C" int pdeip_disp_apriori_dev(void *stream, const double *Us, const float *U, const float *dU, double gammaS, double alpha,
                                      double as_diff, int u_double, int du_double, int nrows, int ncols, float *CGd, float *DGd)
{
    RC(check_dims("pdeip_disp_apriori_dev", nrows, ncols, 1));
    if (!Us || !U || !dU || !CGd || !DGd) return set_err(PDEIP_ERR_ARG, "pdeip_disp_apriori_dev: null plane");
    hipLaunchKernelGGL(k_disp_apriori, pixel_grid(nrows, ncols, 1), dim3(256), 0, static_cast<hipStream_t>(stream), CGd, DGd, Us, U, dU, gammaS,
                       alpha, as_diff * as_diff, u_double, du_double, nrows, ncols);
    HIPCHK(hipGetLastError());
    return PDEIP_OK;
}

extern "C" int pdeip_rgb2grad_dev(void *stream, const float *in, int nrows, int ncols, int nframes, float *out)
{
    RC(check_dims("pdeip_rgb2grad_dev", nrows, ncols, nframes));
    if (in == out) return set_err(PDEIP_ERR_ARG, "pdeip_rgb2grad_dev: output must not alias the input");
    hipLaunchKernelGGL(k_rgb2grad, pixel_grid(nrows, ncols, nframes), dim3(256), 0, static_cast<hipStream_t>(stream), out, in, nrows, ncols);
    HIPCHK(hipGetLastError());
    return PDEIP_OK;
}

extern "C" int pdeip_add_dev(void *stream, const float *A, const float *B, int nrows, int ncols, float *out)
{
    RC(check_dims("pdeip_add_dev", nrows, ncols, 1));
    hipLaunchKernelGGL(k_add2, pixel_grid(nrows, ncols, 1), dim3(256), 0, static_cast<hipStream_t>(stream), out, A, B, nrows, ncols);
    HIPCHK(hipGetLastError());
    return PDEIP_OK;
}

extern "C" int pdeip_hs_assemble_dev(void *stream, const float *It0, const float *It1, int C, float b1, float b2, int nrows, int ncols,
                                     float *MGd, float *CuGd, float *CvGd, float *DuGd, float *DvGd)
{
    RC(check_dims("pdeip_hs_assemble_dev", nrows, ncols, C));
    hipLaunchKernelGGL(k_hs_assemble, pixel_grid(nrows, ncols, 1), dim3(256), 0, static_cast<hipStream_t>(stream), MGd, CuGd, CvGd, DuGd, DvGd,
                       It0, It1, C, b1, b2, nrows, ncols);
    HIPCHK(hipGetLastError());
    return PDEIP_OK;
}

// ---- FAS full-multigrid driver stages (pdeip_fas.hpp) ----------------------------------------------
extern "C" int pdeip_fas_gauss5_dev(void *stream, const float *in, int nrows, int ncols, int frames, const float *g25, float *out)
{
    RC(check_dims("pdeip_fas_gauss5_dev", nrows, ncols, frames));
    if (!g25 || in == out) return set_err(PDEIP_ERR_ARG, "pdeip_fas_gauss5_dev: kernel missing or output aliases the input");
    FasTaps25 T;
    for (int b = 0; b < 5; ++b)
        for (int a = 0; a < 5; ++a) T.g[b * 5 + a] = g25[(4 - b) * 5 + (4 - a)]; // 'conv' flips the kernel
    hipLaunchKernelGGL(k_fas_gauss5, pixel_grid(nrows, ncols, frames), dim3(256), 0, static_cast<hipStream_t>(stream), out, in, T, nrows, ncols);
    HIPCHK(hipGetLastError());
    return PDEIP_OK;
}

extern "C" int pdeip_fas_down_dev(void *stream, const float *in, int nrows, int ncols, int frames, float *out)
{
    RC(check_dims("pdeip_fas_down_dev", nrows, ncols, frames));
    const int nr = (nrows + 1) / 2, nc = (ncols + 1) / 2;
    hipLaunchKernelGGL(k_fas_down, pixel_grid(nr, nc, frames), dim3(256), 0, static_cast<hipStream_t>(stream), out, in, nrows, ncols, nr, nc);
    HIPCHK(hipGetLastError());
    return PDEIP_OK;
}

extern "C" int pdeip_fas_prepare_dev(void *stream, const float *It0, const float *It1, int nrows, int ncols, int frames, float b1, float b2,
                                     float *planes)
{
    RC(check_dims("pdeip_fas_prepare_dev", nrows, ncols, frames));
    hipLaunchKernelGGL(k_fas_prepare, dim3((nrows + FP_TR - 1) / FP_TR, (ncols + FP_TC - 1) / FP_TC, frames), dim3(FP_TR, FP_TC), 0, static_cast<hipStream_t>(stream), planes, It0, It1, frames,
                       b1, b2, nrows, ncols);
    HIPCHK(hipGetLastError());
    return PDEIP_OK;
}

extern "C" int pdeip_fas_assemble_dev(void *stream, const float *planes, const float *Cu, const float *Cv, const float *U, const float *V,
                                      int nrows, int ncols, int frames, float b1, float b2, float k, int per_frame, float *MGd, float *CuGd,
                                      float *CvGd, float *DuGd, float *DvGd, float *gd)
{
    RC(check_dims("pdeip_fas_assemble_dev", nrows, ncols, frames));
    if (!MGd || !DuGd || !DvGd || (Cu && !CuGd) || (Cv && !CvGd))
        return set_err(PDEIP_ERR_ARG, "pdeip_fas_assemble_dev: missing output plane");
    const auto s = static_cast<hipStream_t>(stream);
    if (per_frame)
        hipLaunchKernelGGL(k_fas_assemble<true>, pixel_grid(nrows, ncols, 1), dim3(256), 0, s, MGd, CuGd, CvGd, DuGd, DvGd, gd, planes, Cu, Cv, U,
                           V, frames, b1, b2, k, nrows, ncols, FlowWeightsOut{});
    else
        hipLaunchKernelGGL(k_fas_assemble<false>, pixel_grid(nrows, ncols, 1), dim3(256), 0, s, MGd, CuGd, CvGd, DuGd, DvGd, gd, planes, Cu, Cv, U,
                           V, frames, b1, b2, k, nrows, ncols, FlowWeightsOut{});
    HIPCHK(hipGetLastError());
    return PDEIP_OK;
}

extern "C" int pdeip_fas_assemble_weights_dev(void *stream, const float *planes, const float *Cu, const float *Cv, const float *U, const float *V,
                                              int nrows, int ncols, int frames, float b1, float b2, float k, float *MGd, float *CuGd, float *CvGd,
                                              float *DuGd, float *DvGd, float *wW, float *wN, float *wS, float *wE)
{
    RC(check_dims("pdeip_fas_assemble_weights_dev", nrows, ncols, frames));
    if (!MGd || !DuGd || !DvGd || (Cu && !CuGd) || (Cv && !CvGd) || !wW || !wN || !wS || !wE)
        return set_err(PDEIP_ERR_ARG, "pdeip_fas_assemble_weights_dev: missing output plane");
    FlowWeightsOut W;
    W.wW = wW; W.wN = wN; W.wS = wS; W.wE = wE;
    hipLaunchKernelGGL(k_fas_assemble<false>, pixel_grid(nrows, ncols, 1), dim3(256), 0, static_cast<hipStream_t>(stream), MGd, CuGd, CvGd, DuGd,
                       DvGd, nullptr, planes, Cu, Cv, U, V, frames, b1, b2, k, nrows, ncols, W);
    HIPCHK(hipGetLastError());
    return PDEIP_OK;
}

extern "C" int pdeip_fas_restrict_dev(void *stream, const float *in, int nrows, int ncols, int frames, float scale, float *out)
{
    RC(check_dims("pdeip_fas_restrict_dev", nrows, ncols, frames));
    const int nr = (nrows + 1) / 2, nc = (ncols + 1) / 2;
    hipLaunchKernelGGL(k_fas_restrict, pixel_grid(nr, nc, frames), dim3(256), 0, static_cast<hipStream_t>(stream), out, in, scale, nrows, ncols,
                       nr, nc);
    HIPCHK(hipGetLastError());
    return PDEIP_OK;
}

extern "C" int pdeip_fas_rhs_dev(void *stream, const float *R, const float *A, const float *gd, int nrows, int ncols, int frames, float *out)
{
    RC(check_dims("pdeip_fas_rhs_dev", nrows, ncols, frames));
    hipLaunchKernelGGL(k_fas_rhs, pixel_grid(nrows, ncols, frames), dim3(256), 0, static_cast<hipStream_t>(stream), out, R, A, gd, nrows, ncols);
    HIPCHK(hipGetLastError());
    return PDEIP_OK;
}

extern "C" int pdeip_fas_prolong_add_dev(void *stream, float *U, int nrows, int ncols, const float *Uc, const float *Ures, int nrows_c,
                                         int ncols_c, float inv_scale)
{
    RC(check_dims("pdeip_fas_prolong_add_dev", nrows, ncols, 1));
    RC(check_dims("pdeip_fas_prolong_add_dev", nrows_c, ncols_c, 1));
    hipLaunchKernelGGL(k_fas_prolong_add, pixel_grid(nrows, ncols, 1), dim3(256), 0, static_cast<hipStream_t>(stream), U, Uc, Ures, inv_scale,
                       nrows_c, ncols_c, nrows, ncols);
    HIPCHK(hipGetLastError());
    return PDEIP_OK;
}

extern "C" int pdeip_fas_upscale_dev(void *stream, const float *in, int nrows, int ncols, float mul, int nrows_out, int ncols_out, float *out)
{
    RC(check_dims("pdeip_fas_upscale_dev", nrows, ncols, 1));
    RC(check_dims("pdeip_fas_upscale_dev", nrows_out, ncols_out, 1));
    if (nrows_out < nrows || ncols_out < ncols) return set_err(PDEIP_ERR_ARG, "pdeip_fas_upscale_dev: enlarging only");
    hipLaunchKernelGGL(k_fas_upscale, pixel_grid(nrows_out, ncols_out, 1), dim3(256), 0, static_cast<hipStream_t>(stream), out, in, mul, nrows,
                       ncols, nrows_out, ncols_out);
    HIPCHK(hipGetLastError());
    return PDEIP_OK;
}

extern "C" int pdeip_flow_opdiffweights_dev(void *stream, const float *U, const float *V, const float *dU, const float *dV, int nrows,
                                            int ncols, float *wW, float *wN, float *wS, float *wE)
{
    RC(check_dims("pdeip_flow_opdiffweights_dev", nrows, ncols, 1));
    hipLaunchKernelGGL(k_flow_opdiffweights, pixel_grid(nrows, ncols, 1), dim3(256), 0, static_cast<hipStream_t>(stream), wW, wN, wS, wE, U,
                       V, dU, dV, nrows, ncols);
    HIPCHK(hipGetLastError());
    return PDEIP_OK;
}

// Self-check of single_inv_sqrt (pdeip_flow.hpp): n pseudo-random doubles over 1e-5 .. 1e7 plus, for every eighth of them, an
// argument aimed at a single-precision rounding boundary (x = 1/m^2 for a midpoint m between two adjacent singles, nudged by
// -3 .. +3 ulp), each compared with the exact sequence single(1.0 / sqrt(x)).
static __global__ void k_selftest_inv_sqrt(unsigned long long n, unsigned seed, unsigned *mismatches)
{
    const unsigned long long k = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    unsigned long long h = (k + 1) * 0x9E3779B97F4A7C15ull + seed;
    h ^= h >> 29; h *= 0xBF58476D1CE4E5B9ull; h ^= h >> 32; h *= 0x94D049BB133111EBull; h ^= h >> 29;
    const double u = (double)(h >> 11) * (1.0 / 9007199254740992.0);   // [0, 1)
    double x = exp(-11.5 + 27.6 * u);                                   // 1e-5 .. 1e7, log-uniform
    if ((k & 7) == 0) {
        const float f = (float)(1.0 / sqrt(x));
        const double m = 0.5 * ((double)f + (double)__uint_as_float(__float_as_uint(f) + 1u)); // midpoint: a rounding boundary
        x = 1.0 / (m * m);
        const long long nudge = (long long)((h >> 3) % 7) - 3;
        x = __longlong_as_double(__double_as_longlong(x) + nudge);
    }
    if (single_inv_sqrt(x) != (float)(1.0 / sqrt(x))) atomicAdd(mismatches, 1u);
}

extern "C" int pdeip_selftest_inv_sqrt(int n, unsigned seed)
{
    if (n <= 0) return 0;
    unsigned *d = nullptr, h = 0;
    if (hipMalloc(&d, sizeof(unsigned)) != hipSuccess) return -1;
    if (hipMemset(d, 0, sizeof(unsigned)) != hipSuccess) { (void)hipFree(d); return -1; }
    hipLaunchKernelGGL(k_selftest_inv_sqrt, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, nullptr, (unsigned long long)n, seed, d);
    const bool ok = hipMemcpy(&h, d, sizeof(unsigned), hipMemcpyDeviceToHost) == hipSuccess;
    (void)hipFree(d);
    return ok ? (int)h : -1;
}

extern "C" int pdeip_median3_pair_dev(void *stream, const float *A0, const float *B0, const float *A1, const float *B1, int nrows, int ncols,
                                      float *out0, float *out1)
{
    RC(check_dims("pdeip_median3_pair_dev", nrows, ncols, 1));
    if (!A0 || !A1 || !out0 || !out1 || out0 == A0 || out0 == B0 || out1 == A1 || out1 == B1 || out0 == A1 || out0 == B1 || out1 == A0 || out1 == B0)
        return set_err(PDEIP_ERR_ARG, "pdeip_median3_pair_dev: outputs must not alias an input");
    hipLaunchKernelGGL(k_median3_sum, pixel_grid(nrows, ncols, 2), dim3(256), 0, static_cast<hipStream_t>(stream), out0, A0, B0, nrows, ncols, out1,
                       A1, B1);
    HIPCHK(hipGetLastError());
    return PDEIP_OK;
}

extern "C" int pdeip_median3_dev(void *stream, const float *A, const float *B, int nrows, int ncols, float *out)
{
    RC(check_dims("pdeip_median3_dev", nrows, ncols, 1));
    if (out == A || out == B) return set_err(PDEIP_ERR_ARG, "pdeip_median3_dev: output must not alias an input");
    hipLaunchKernelGGL(k_median3_sum, pixel_grid(nrows, ncols, 1), dim3(256), 0, static_cast<hipStream_t>(stream), out, A, B, nrows, ncols, nullptr, nullptr, nullptr);
    HIPCHK(hipGetLastError());
    return PDEIP_OK;
}

// lambda of the squared gradient norms by radix selection (pdeip_tv.hpp): six histogram + scan pairs on the stream
static int tv_select_lambda(hipStream_t s, TvSelectState *st, double *lambda, const double *nrm, size_t n, double quantile)
{
    HIPCHK(hipMemsetAsync(st, 0, sizeof(TvSelectState), s));
    const dim3 fat((unsigned)((n + 256 * 64 - 1) / (256 * 64))), thin((unsigned)((n + 256 * 16 - 1) / (256 * 16))), block(256);
    hipLaunchKernelGGL((k_tvsel_hist<53, 11, 64, 64>), fat, block, 0, s, st, nrm, n);
    hipLaunchKernelGGL((k_tvsel_scan<53, 11, true, false>), dim3(1), block, 0, s, st, lambda, n, quantile);
    hipLaunchKernelGGL((k_tvsel_hist<42, 11, 53, 64>), fat, block, 0, s, st, nrm, n);
    hipLaunchKernelGGL((k_tvsel_scan<42, 11, false, false>), dim3(1), block, 0, s, st, lambda, n, quantile);
    hipLaunchKernelGGL((k_tvsel_hist<31, 11, 42, 16>), thin, block, 0, s, st, nrm, n);
    hipLaunchKernelGGL((k_tvsel_scan<31, 11, false, false>), dim3(1), block, 0, s, st, lambda, n, quantile);
    hipLaunchKernelGGL((k_tvsel_hist<20, 11, 31, 16>), thin, block, 0, s, st, nrm, n);
    hipLaunchKernelGGL((k_tvsel_scan<20, 11, false, false>), dim3(1), block, 0, s, st, lambda, n, quantile);
    hipLaunchKernelGGL((k_tvsel_hist<9, 11, 20, 16>), thin, block, 0, s, st, nrm, n);
    hipLaunchKernelGGL((k_tvsel_scan<9, 11, false, false>), dim3(1), block, 0, s, st, lambda, n, quantile);
    hipLaunchKernelGGL((k_tvsel_hist<0, 9, 9, 16>), thin, block, 0, s, st, nrm, n);
    hipLaunchKernelGGL((k_tvsel_scan<0, 9, false, true>), dim3(1), block, 0, s, st, lambda, n, quantile);
    HIPCHK(hipGetLastError());
    return PDEIP_OK;
}

// One lagged-diffusivity iteration's MATLAB-side work of TVdenoise8 (pdeip_tv.hpp): ADdiffWeights(Iout) incl. the
// quantile lambda, PsiData, TRACE, B and the alpha-scaled weights, ready for pdeip_pde_sor8_dev / pdeip_pde_alr8_dev.
extern "C" int pdeip_tv_assemble_dev(void *stream, const float *Iout, const float *Iin, int nrows, int ncols, int nframes,
                                     float alpha, float *TRACE, float *B, float *aW, float *aNW, float *aN, float *aNE,
                                     float *aE, float *aSE, float *aS, float *aSW)
{
    const char *who = "pdeip_tv_assemble_dev";
    RC(check_dims(who, nrows, ncols, nframes));
    hipStream_t s = static_cast<hipStream_t>(stream);
    const size_t n = (size_t)nrows * ncols;
    const size_t doubles = 3 * n + 2 + (sizeof(TvSelectState) + 7) / 8; // gx, gy, norm, lambda, selection state
    float *basef;
    RC(ws_get(WS_TV, doubles * sizeof(double), &basef));
    double *gx = reinterpret_cast<double *>(basef), *gy = gx + n, *nrm = gy + n, *lambda = nrm + n;
    TvSelectState *st = reinterpret_cast<TvSelectState *>(lambda + 2);
    hipLaunchKernelGGL(k_tv_gradient, pixel_grid(nrows, ncols, 1), dim3(256), 0, s, gx, gy, nrm, Iout, nrows, ncols, nframes);
    RC(tv_select_lambda(s, st, lambda, nrm, n, -1.0));
    hipLaunchKernelGGL(k_tv_assemble, dim3((nrows + TT_R - 1) / TT_R, (ncols + TT_C - 1) / TT_C), dim3(TT_R, TT_C), 0, s, TRACE, B, aW, aNW, aN, aNE, aE, aSE, aS, aSW, gx, gy,
                       nrm, lambda, Iout, Iin, alpha, nrows, ncols, nframes);
    HIPCHK(hipGetLastError());
    return PDEIP_OK;
}

// ---- image pyramid (pdeip_pyr.hpp) ---------------------------------------------------------------------
static int pyr_axis(const char *who, int n_in, int n_out, int cubic, PyrAxis *A)
{
    A->scale = (double)n_out / (double)n_in;
    A->stretch = A->scale >= 1.0 ? 1.0 : 1.0 / A->scale;
    A->width = (cubic ? 2.0 : 1.0) * A->stretch;
    A->T = (int)ceil(2.0 * A->width) + 2;
    A->cubic = cubic;
    if (A->T > PYR_TMAX) return set_err(PDEIP_ERR_UNSUPPORTED, "%s: shrinking %d -> %d needs %d taps (at most %d)", who, n_in, n_out, A->T, PYR_TMAX);
    return PDEIP_OK;
}

extern "C" int pdeip_pyr_resize_dev(void *stream, const float *in, int nrows, int ncols, int nframes, int nrows_out, int ncols_out, int cubic,
                                    float *out)
{
    const char *who = "pdeip_pyr_resize_dev";
    RC(check_dims(who, nrows, ncols, nframes));
    RC(check_dims(who, nrows_out, ncols_out, nframes));
    if (in == out) return set_err(PDEIP_ERR_ARG, "%s: output must not alias the input", who);
    PyrAxis R, C;
    RC(pyr_axis(who, nrows, nrows_out, cubic != 0, &R));
    RC(pyr_axis(who, ncols, ncols_out, cubic != 0, &C));
    hipLaunchKernelGGL(k_pyr_resize, pixel_grid(nrows_out, ncols_out, nframes), dim3(256), 0, static_cast<hipStream_t>(stream), out, in, R, C, nrows,
                       ncols, nrows_out, ncols_out);
    HIPCHK(hipGetLastError());
    return PDEIP_OK;
}

extern "C" int pdeip_pyr_smooth_dev(void *stream, const float *in, int nrows, int ncols, int nframes, const double *G, int size, float *out)
{
    const char *who = "pdeip_pyr_smooth_dev";
    RC(check_dims(who, nrows, ncols, nframes));
    if (!G || size < 1 || size > 7 || size % 2 == 0 || in == out) return set_err(PDEIP_ERR_ARG, "%s: odd mask of at most 7x7, output distinct from input", who);
    PyrMask M;
    for (int k = 0; k < size * size; k++) M.g[k] = G[k];
    M.size = size;
    hipLaunchKernelGGL(k_pyr_smooth, pixel_grid(nrows, ncols, nframes), dim3(256), 0, static_cast<hipStream_t>(stream), out, in, M, nrows, ncols);
    HIPCHK(hipGetLastError());
    return PDEIP_OK;
}

// ---- symmetric stereo driver stages (pdeip_sym.hpp) ----------------------------------------------------
extern "C" int pdeip_sym_warp_flow_dev(void *stream, const float *U, const float *Uq, int nrows, int ncols, double *out)
{
    RC(check_dims("pdeip_sym_warp_flow_dev", nrows, ncols, 1));
    if (ncols < 2) return set_err(PDEIP_ERR_ARG, "pdeip_sym_warp_flow_dev: needs at least two columns");
    hipLaunchKernelGGL(k_sym_warp_flow, pixel_grid(nrows, ncols, 1), dim3(256), 0, static_cast<hipStream_t>(stream), out, U, Uq, nrows, ncols);
    HIPCHK(hipGetLastError());
    return PDEIP_OK;
}

extern "C" int pdeip_sym_flow_terms_dev(void *stream, const float *U, const double *Uw, int nrows, int ncols, double *Udt, double *Udx,
                                        double *CuS, double *DuS)
{
    RC(check_dims("pdeip_sym_flow_terms_dev", nrows, ncols, 1));
    hipLaunchKernelGGL(k_sym_flow_terms, pixel_grid(nrows, ncols, 1), dim3(256), 0, static_cast<hipStream_t>(stream), Udt, Udx, CuS, DuS, U, Uw,
                       nrows, ncols);
    HIPCHK(hipGetLastError());
    return PDEIP_OK;
}

extern "C" int pdeip_sym_assemble_dev(void *stream, const float *Idt, const float *Idx, const float *Idxt, const float *Idyt, const float *Idxx,
                                      const float *Idxy, int C, const double *Udt, const double *Udx, const double *CuS, const double *DuS,
                                      const float *dU, float b1, float b2, float alpha, double kS, double sr2, int first, int nrows, int ncols,
                                      float *CuG, float *DuG)
{
    RC(check_dims("pdeip_sym_assemble_dev", nrows, ncols, C));
    const SymData d{Idt, Idx, Idxt, Idyt, Idxx, Idxy, C};
    hipLaunchKernelGGL(k_sym_assemble, pixel_grid(nrows, ncols, 1), dim3(256), 0, static_cast<hipStream_t>(stream), CuG, DuG, d, Udt, Udx, CuS, DuS,
                       dU, b1, b2, alpha, kS, sr2, first, nrows, ncols);
    HIPCHK(hipGetLastError());
    return PDEIP_OK;
}

extern "C" int pdeip_tv4_assemble_dev(void *stream, const float *Iout, const float *Iin, int nrows, int ncols, int nframes, float alpha,
                                      float *TRACE, float *B, float *aW, float *aN, float *aE, float *aS)
{
    RC(check_dims("pdeip_tv4_assemble_dev", nrows, ncols, nframes));
    hipLaunchKernelGGL(k_tv4_assemble, pixel_grid(nrows, ncols, 1), dim3(256), 0, static_cast<hipStream_t>(stream), TRACE, B, aW, aN, aE, aS, Iout,
                       Iin, alpha, nrows, ncols, nframes);
    HIPCHK(hipGetLastError());
    return PDEIP_OK;
}

extern "C" int pdeip_ad_weights_dev(void *stream, const float *D, int nrows, int ncols, int nframes, double quantile, float *wW, float *wNW,
                                    float *wN, float *wNE, float *wE, float *wSE, float *wS, float *wSW)
{
    const char *who = "pdeip_ad_weights_dev";
    RC(check_dims(who, nrows, ncols, nframes));
    if (!(quantile > 0.0 && quantile <= 1.0)) return set_err(PDEIP_ERR_ARG, "%s: quantile must be in (0, 1]", who);
    hipStream_t s = static_cast<hipStream_t>(stream);
    const size_t n = (size_t)nrows * ncols;
    const size_t doubles = 3 * n + 2 + (sizeof(TvSelectState) + 7) / 8;
    float *basef;
    RC(ws_get(WS_TV, doubles * sizeof(double), &basef));
    double *gx = reinterpret_cast<double *>(basef), *gy = gx + n, *nrm = gy + n, *lambda = nrm + n;
    TvSelectState *st = reinterpret_cast<TvSelectState *>(lambda + 2);
    hipLaunchKernelGGL(k_tv_gradient, pixel_grid(nrows, ncols, 1), dim3(256), 0, s, gx, gy, nrm, D, nrows, ncols, nframes);
    RC(tv_select_lambda(s, st, lambda, nrm, n, quantile));
    hipLaunchKernelGGL(k_ad_weights, dim3((nrows + TT_R - 1) / TT_R, (ncols + TT_C - 1) / TT_C), dim3(TT_R, TT_C), 0, s, wW, wNW, wN, wNE, wE, wSE, wS, wSW, gx, gy, nrm, lambda, nrows,
                       ncols);
    HIPCHK(hipGetLastError());
    return PDEIP_OK;
}

