/*
 * pdeip_oracle.h -- CPU oracle for the MEX-side stencil hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it, and
 * only as the checker / reported CPU baseline.
 *
 * PARITY UNPINNED.  The reference library (the .c files under mex/source/library) includes MATLAB's
 * mex.h / matrix.h, which this image does not have, and the reference ships no
 * tests, golden vectors or recorded outputs.  The functions below are therefore a
 * line-by-line restatement of the reference's arithmetic (same float32 operation
 * order, no FMA contraction), each citing the reference file:line it follows, but
 * they have not been compared against a build of the reference itself.
 *
 * Conventions (SURVEY.md "Conventions"): all arrays are MATLAB column-major float32;
 * element (row i, col j, frame k) is at k*nrows*ncols + j*nrows + i.  West/East is
 * pos -/+ nrows, North/South is pos -/+ 1.
 *
 * Orderings:
 *   *_lex : the reference's lexicographic (column-major) Gauss-Seidel order.
 *   *_rb  : red-black order (colour = (i+j)&1, colour 0 first) with the SAME per-pixel
 *           arithmetic; not in the reference -- it defines the product's RED_BLACK mode.
 *   *_4c  : four-colour order (colour = (i&1)|((j&1)<<1), 0..3) for the 9-point stencil.
 */
#ifndef PDEIP_ORACLE_H
#define PDEIP_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_ORDER_LEX 0
#define ORC_ORDER_COLOUR 1
/* `order` argument: bit 0 selects the ordering; bit 1, used with ORC_ORDER_COLOUR only, is the
 * parity of the global column index of local column 0 when the arrays are one column slab of a
 * larger frame (colour = (i + j_global) & 1). */
#define ORC_ORDER_COLOUR_ODD_COL0 3

/* opticalflowSolvers.c:41-186 (GS_SOR_elin4_2d); U,V updated in place. */
void orc_oflow_sor_elin4(float *U, float *V, const float *M, const float *Cu, const float *Cv,
                         const float *Du, const float *Dv, const float *wW, const float *wN,
                         const float *wE, const float *wS, int nrows, int ncols, int iter,
                         float omega, int order);

/* The colour order of orc_oflow_sor_elin4 on `nthreads` host threads (OpenMP; <= 0: the runtime's default), bit-identical
 * to it.  Not in the reference; the all-core CPU comparator of bench.py.  Returns the threads used. */
int orc_oflow_sor_elin4_rb_omp(float *U, float *V, const float *M, const float *Cu, const float *Cv, const float *Du,
                               const float *Dv, const float *wW, const float *wN, const float *wE, const float *wS,
                               int nrows, int ncols, int iter, float omega, int nthreads);

void orc_plane_copy_omp(float *dst, const float *src, int nrows, int ncols, int nthreads);

/* opticalflowSolvers.c:504-680 (GS_SOR_llin4_2d) == :1487-1667 (GS_SOR_llin8_2d); dU,dV in place. */
void orc_oflow_sor_llin4(const float *U, const float *V, float *dU, float *dV, const float *M,
                         const float *Cu, const float *Cv, const float *Du, const float *Dv,
                         const float *wW, const float *wN, const float *wE, const float *wS,
                         int nrows, int ncols, int iter, float omega, int order);

/* opticalflowSolvers.c:269-380 (Residuals_elin4_2d). RU,RV are [nrows x ncols x nframes]. */
void orc_oflow_res_elin4(float *RU, float *RV, const float *U, const float *V, const float *M,
                         const float *Cu, const float *Cv, const float *Du, const float *Dv,
                         const float *wW, const float *wN, const float *wE, const float *wS,
                         int nrows, int ncols, int nframes);

/* opticalflowSolvers.c:387-496 (LHS_elin4_2d). */
void orc_oflow_lhs_elin4(float *AU, float *AV, const float *U, const float *V, const float *M,
                         const float *Du, const float *Dv, const float *wW, const float *wN,
                         const float *wE, const float *wS, int nrows, int ncols, int nframes);

/* opticalflowSolvers.c:766-916 (Residuals_llin4_2d), including the :912 border quirk. */
void orc_oflow_res_llin4(float *RU, float *RV, const float *U, const float *V, const float *dU,
                         const float *dV, const float *M, const float *Cu, const float *Cv,
                         const float *Du, const float *Dv, const float *wW, const float *wN,
                         const float *wE, const float *wS, int nrows, int ncols, int nframes);

/* opticalflowSolvers.c:923-1070 (LHS_llin4_2d), including the :1056 border quirk.
 * AU,AV must be zero-filled on entry (the gateway hands over mxCreateNumericArray memory). */
void orc_oflow_lhs_llin4(float *AU, float *AV, const float *U, const float *V, const float *dU,
                         const float *dV, const float *M, const float *Du, const float *Dv,
                         const float *wW, const float *wN, const float *wE, const float *wS,
                         int nrows, int ncols, int nframes);

/* disparitySolvers.c:41-144 (GS_SOR_llin4_2d); dU in place. */
void orc_disp_sor_llin4(const float *U, float *dU, const float *Cu, const float *Du,
                        const float *wW, const float *wN, const float *wE, const float *wS,
                        int nrows, int ncols, int iter, float omega, int order);

/* disparitySolvers.c:301-460 (GS_SOR_llinsym4_2d): two disparity fields side by side; dU0,dU1 in place. */
void orc_disp_sor_llinsym4(const float *U0, float *dU0, const float *Cu0, const float *Du0, const float *wW0,
                           const float *wN0, const float *wE0, const float *wS0, const float *U1, float *dU1,
                           const float *Cu1, const float *Du1, const float *wW1, const float *wN1, const float *wE1,
                           const float *wS1, int nrows, int ncols, int iter, float omega, int order);

/* disparitySolvers.c:218-293 (Residuals_llin4_2d). */
void orc_disp_res_llin4(float *RU, const float *U, const float *dU, const float *Cu,
                        const float *Du, const float *wW, const float *wN, const float *wE,
                        const float *wS, int nrows, int ncols);

/* pdeSolvers.c:44-146 (GS_SOR_4_2d); X in place, all planes [nrows x ncols x nframes]. */
void orc_pde_sor4(float *X, const float *TRACE, const float *B, const float *wW, const float *wN,
                  const float *wE, const float *wS, int nrows, int ncols, int nframes, int iter,
                  float omega, int order);

/* pdeSolvers.c:153-268 (GS_SOR_8_2d). */
void orc_pde_sor8(float *X, const float *TRACE, const float *B, const float *wW, const float *wNW,
                  const float *wN, const float *wNE, const float *wE, const float *wSE,
                  const float *wS, const float *wSW, int nrows, int ncols, int nframes, int iter,
                  float omega, int order);

/* imageDiffusionWeights.c:341-378 (diffWeights6_2D_c) + helpers :32-334.
 * Outputs are [nrows x ncols] planes (frame 0 of the gateway's outputs), fully written
 * here (the cells the reference leaves at their zero initialisation are set to 0). */
void orc_diffweights6(float *wW, float *wN, float *wE, float *wS, const float *D, int nrows,
                      int ncols, int nframes, float eps);

/* imageInterpolation.c:44-140 (bilinInterp2), fill value = NaN (the intended value; the
 * reference's own 4-argument call leaves it undefined, BilinInterp_2d.c:120-123). */
void orc_warp_bilinear(float *Iout, const float *Iin, const float *X, const float *Y, int nrows,
                       int ncols, int nframes);

/* imageDerivatives.c:309-388 (fstSimoncelli_c, size 5) with the filters of FstDerivatives5.c:59-61.
 * All planes [nrows x ncols x nframes]; frames are independent. */
void orc_fst_derivatives5(float *Idt, float *Idx, float *Idy, const float *It0, const float *It1,
                          int nrows, int ncols, int nframes);

/* imageDerivatives.c:391-482 (sndSimoncelli_c) with the filters of SndDerivatives5.c:65-68. */
void orc_snd_derivatives5(float *Idxt, float *Idyt, float *Idxx, float *Idyy, float *Idxy,
                          const float *It0, const float *It1, int nrows, int ncols, int nframes);

/* ---- alternating line relaxation, solver = 2 of the gateways (pdeip_oracle_alr.c) ---------------------
 * `order`: ORC_ORDER_LEX = the reference's line order; ORC_ORDER_COLOUR = zebra (even lines, then odd
 * lines) with the same per-line arithmetic. */

/* opticalflowSolvers.c:196-262 (GS_ALR_SOR_elin4_2d) + line solvers :1763-2410; U,V in place. */
void orc_oflow_alr_elin4(float *U, float *V, const float *M, const float *Cu, const float *Cv, const float *Du,
                         const float *Dv, const float *wW, const float *wN, const float *wE, const float *wS,
                         int nrows, int ncols, int iter, float omega, int order);

/* opticalflowSolvers.c:690-759 (GS_ALR_SOR_llin4_2d) + :2415-3100; dU,dV in place. */
void orc_oflow_alr_llin4(const float *U, const float *V, float *dU, float *dV, const float *M, const float *Cu,
                         const float *Cv, const float *Du, const float *Dv, const float *wW, const float *wN,
                         const float *wE, const float *wS, int nrows, int ncols, int iter, float omega, int order);

/* opticalflowSolvers.c:1677-1750 (GS_ALR_SOR_llin8_2d) + :3104-3914: the only place the diagonal weights act. */
void orc_oflow_alr_llin8(const float *U, const float *V, float *dU, float *dV, const float *M, const float *Cu,
                         const float *Cv, const float *Du, const float *Dv, const float *wW, const float *wNW,
                         const float *wN, const float *wNE, const float *wE, const float *wSE, const float *wS,
                         const float *wSW, int nrows, int ncols, int iter, float omega, int order);

/* disparitySolvers.c:154-211 (GS_ALR_SOR_llin4_2d) + :1376-2029. */
void orc_disp_alr_llin4(const float *U, float *dU, const float *Cu, const float *Du, const float *wW,
                        const float *wN, const float *wE, const float *wS, int nrows, int ncols, int iter,
                        float omega, int order);

/* pdeSolvers.c:277-335 (GS_ALR_SOR_4_2d) + :409-1131. */
void orc_pde_alr4(float *X, const float *TRACE, const float *B, const float *wW, const float *wN, const float *wE,
                  const float *wS, int nrows, int ncols, int nframes, int iter, float omega, int order);

/* pdeSolvers.c:344-402 (GS_ALR_SOR_8_2d) + :1132-1393: ONE iteration whatever `iter` says, interior lines only. */
void orc_pde_alr8(float *X, const float *TRACE, const float *B, const float *wW, const float *wNW, const float *wN,
                  const float *wNE, const float *wE, const float *wSE, const float *wS, const float *wSW, int nrows,
                  int ncols, int nframes, int iter, float omega, int order);

#ifdef __cplusplus
}
#endif
#endif
