/*
 * pdeip.h -- C-ABI of libpdeip.so: the MI355X (gfx950) implementation of the MEX-side
 * stencil hot path of JediZ/PDE-based-image-processing.
 *
 * Boundary.  The reference's FFI for this path is MATLAB's MEX gateway,
 *     void mexFunction(int nlhs, mxArray *plhs[], int nrhs, const mxArray *prhs[]),
 * one shared object per gateway source in mex/source/ (mex/buildAll.m:5-25).  Each
 * `pdeip_<name>` host entry point below is what the body of one gateway forwards to
 * after unpacking its mxArrays; it reproduces the gateway's semantics (copy-in,
 * iter<=0 handling, residuals from the INPUT iterate, zero-initialised outputs) so the
 * MEX stub is pure unpacking.  INTEGRATION.md shows the stubs.
 *
 * Data.  Every array is MATLAB column-major float32: element (row i, col j, frame k) at
 * k*nrows*ncols + j*nrows + i.  Scalars that MATLAB passes as 1x1 singles (iter, omega,
 * solver, eps) are plain int/float here; the stub does the cast the gateway did
 * (e.g. Oflow_sor_elin4_2d.c:263-283).
 *
 * Two families of entry points:
 *   pdeip_<name>(...)      host pointers, synchronous, stateless: H2D, solve, D2H.
 *   pdeip_<name>_dev(...)  device pointers (resident in HBM), asynchronous on `stream`
 *                          (a hipStream_t passed as void*; NULL = default stream), in
 *                          place where the reference solves in place.
 *
 * All functions return PDEIP_OK or an error code; pdeip_last_error() gives the message
 * (the text a stub hands to mexErrMsgTxt).  No exceptions cross the boundary.  The
 * library keeps one lazily created context per process (device workspace cache); it is
 * thread-compatible: one call at a time.
 *
 * Sweep ordering (pdeip_set_mode):
 *   PDEIP_MODE_EXACT_ORDER  the reference's lexicographic Gauss-Seidel order, evaluated
 *                           as a pipelined tile wavefront; results are bit-identical to
 *                           the CPU restatement of the reference (default).
 *   PDEIP_MODE_RED_BLACK    red-black (5-point) / four-colour (9-point) ordering with the
 *                           same per-pixel arithmetic: the throughput and multi-GPU mode.
 *                           Converges to the same fixed point; differs at finite `iter`.
 */
#ifndef PDEIP_H
#define PDEIP_H

#ifdef __cplusplus
extern "C" {
#endif

#define PDEIP_OK 0
#define PDEIP_ERR_ARG 1         /* null pointer, nrows/ncols < 3, nframes < 1 */
#define PDEIP_ERR_SOLVER 2      /* "no such solver" (gateway default: branch) */
#define PDEIP_ERR_UNSUPPORTED 3 /* valid request outside what the device path covers */
#define PDEIP_ERR_DEVICE 4      /* HIP runtime error / no gfx950 device */
#define PDEIP_ERR_NOMEM 5

#define PDEIP_MODE_EXACT_ORDER 0
#define PDEIP_MODE_RED_BLACK 1

/* `solver` argument of the solver gateways (e.g. Oflow_sor_elin4_2d.c:328-338).  Both are device paths:
 *   1 -> GS_SOR_*      point Gauss-Seidel SOR
 *   2 -> GS_ALR_SOR_*  alternating line relaxation (the MATLAB drivers' default); in PDEIP_MODE_EXACT_ORDER the
 *        reference's line order (bit-identical, serial by construction), in PDEIP_MODE_RED_BLACK zebra order. */
#define PDEIP_SOLVER_SOR 1
#define PDEIP_SOLVER_ALR 2

/* ---- library state ------------------------------------------------------------------ */
const char *pdeip_version(void);
const char *pdeip_last_error(void);
int pdeip_set_mode(int mode);
int pdeip_get_mode(void);
/* Select the HIP device used by the host-pointer entry points (default 0) = pdeip_set_devices(1, &device_id). */
int pdeip_set_device(int device_id);
/* Device group of the host-pointer entry points.  With n > 1, red-black (PDEIP_MODE_RED_BLACK) point-SOR solver calls
 * are cut into n slabs of consecutive MATLAB columns, one per device.  Every device uploads its slab plus a halo of
 * 2 x iter columns per cut side straight from the caller's host planes and relaxes it for the whole call: nothing is
 * exchanged between devices (for a host-pointer call the upload is the halo refresh; csrc/pdeip_multi.hip).  Results are
 * bit-identical to the single-device red-black call.  Exact-order calls and line relaxation do not decompose (their
 * dependency front crosses the frame) and run on ids[0].  Workspace is cached per device. */
int pdeip_set_devices(int n, const int *ids);
/* Writes up to `capacity` ids of the current group to ids (may be NULL) and returns the group size. */
int pdeip_get_devices(int *ids, int capacity);
/* Environment knobs, read once before the first call that needs them, so that an unchanged MATLAB session can opt in
 * without touching a signature (explicit pdeip_set_mode / pdeip_set_device(s) calls made before that win):
 *   PDEIP_MODE     exact | red_black        sweep ordering of the host entry points (default exact)
 *   PDEIP_DEVICE   n                        = pdeip_set_device(n)
 *   PDEIP_DEVICES  a,b,c,...                = pdeip_set_devices */
/* Release the cached device workspace of every device (optional; the process exit also releases it). */
int pdeip_release(void);
/* Number of kernel launches the last *_dev solver call enqueued (diagnostic). */
int pdeip_last_launch_count(void);
/* Changes whenever the library frees or regrows a cached workspace buffer (a larger frame, pdeip_release, pdeip_set_device).
 * A caller that captured *_dev calls into a HIP graph must re-capture when the value differs from the one at capture time:
 * the graph's kernel arguments point into those buffers. */
int pdeip_workspace_generation(void);
/* Waits for the device and reports PDEIP_ERR_DEVICE if a bounded dependency wait of the persistent
 * exact-order kernel (PDEIP_EXACT_PERSIST=1) timed out during the preceding calls. */
int pdeip_persist_error(void);
/* Diagnostic: sets the sticky abort word as a timed-out dependency wait of an exact-order walker would (calls made while it is set
 * drain at once, their results are invalid), to exercise the reporting path: pdeip_persist_error() must return PDEIP_ERR_DEVICE
 * once and clear it. */
int pdeip_debug_raise_abort(void);
/* Diagnostic: the schedule table the exact-order walkers would use for B strips x T sweeps (affine: the XCD-affine lists of
 * PDEIP_PERSIST_XCD=1), built on the device as a call builds it and copied to `table` (16 + B*T ints: list offsets 0..8, items
 * b | t << 16 from int 16 on). */
int pdeip_debug_persist_order(int B, int T, int affine, int *table);
/* Diagnostic: compares the fused pipeline's fast reciprocal (v_rcp_f32 + one Newton step, taken by the divisor planes of
 * opticalflowSolvers.c:111-127 when every denominator is a normal number with a normal reciprocal) with the IEEE quotient
 * 1.0f / d for EVERY such float (exponent field 1..252, both signs); counts[0] = inputs compared, counts[1] = results that differ
 * in any bit (must be 0). */
int pdeip_debug_rcp_check(unsigned long long *counts);
/* Sweep-kernel timing for bench.py's roofline figure.  While enabled, every *_dev solver call
 * brackets its back-to-back sweep launches (not its prologue) with a pair of HIP events on the
 * call's stream.  pdeip_profile_read() waits for the recorded events, returns the summed elapsed
 * milliseconds and the number of sweep launches they cover, and clears the record. */
int pdeip_profile_enable(int on);
int pdeip_profile_read(double *elapsed_ms, int *sweep_launches);

/* ---- host-pointer drop-in entry points -------------------------------------------------
 * Output pointers marked "optional" may be NULL (the corresponding MATLAB output was not
 * requested, nlhs too small). */

/* [U,V(,RU,RV)] = Oflow_sor_elin4_2d(U,V,M,Cu,Cv,Du,Dv,wW,wN,wE,wS,iter,omega,solver)
 * replaces mexFunction of mex/source/Oflow_sor_elin4_2d.c:64-352 -> GS_SOR_elin4_2d
 * (library/opticalflowSolvers.c:41) + Residuals_elin4_2d (:269).
 * M,Cu,Cv,Du,Dv are [nrows x ncols x nframes_coef]; the solver reads frame 0, the
 * residuals every frame; RU,RV (optional, both or neither) are [.. x nframes_coef].
 * iter<=0: U_out,V_out are all zero (Oflow_sor_elin4_2d.c:341-346). */
int pdeip_oflow_sor_elin4(const float *U, const float *V, const float *M, const float *Cu,
                          const float *Cv, const float *Du, const float *Dv, const float *wW,
                          const float *wN, const float *wE, const float *wS, int nrows, int ncols,
                          int nframes_coef, int iter, float omega, int solver, float *U_out,
                          float *V_out, float *RU, float *RV);

/* [dU,dV(,RU,RV)] = Oflow_sor_llin4_2d(U,V,dU,dV,M,Cu,Cv,Du,Dv,wW,wN,wE,wS,iter,omega,solver)
 * replaces mex/source/Oflow_sor_llin4_2d.c:66-386 -> GS_SOR_llin4_2d
 * (opticalflowSolvers.c:504) + Residuals_llin4_2d (:766). */
int pdeip_oflow_sor_llin4(const float *U, const float *V, const float *dU, const float *dV,
                          const float *M, const float *Cu, const float *Cv, const float *Du,
                          const float *Dv, const float *wW, const float *wN, const float *wE,
                          const float *wS, int nrows, int ncols, int nframes_coef, int iter,
                          float omega, int solver, float *dU_out, float *dV_out, float *RU,
                          float *RV);

/* [dU,dV(,RU,RV)] = Oflow_sor_llin8_2d(U,V,dU,dV,M,Cu,Cv,Du,Dv,wW,wNW,wN,wNE,wE,wSE,wS,wSW,
 *                                      iter,omega,solver)
 * replaces mex/source/Oflow_sor_llin8_2d.c:71-489 -> GS_SOR_llin8_2d (opticalflowSolvers.c:1487),
 * whose point solver never reads the diagonal weights (:1550-1591); they are accepted and
 * ignored here too.  The gateway allocates RU,RV but never fills them (:466-488): zeros. */
int pdeip_oflow_sor_llin8(const float *U, const float *V, const float *dU, const float *dV,
                          const float *M, const float *Cu, const float *Cv, const float *Du,
                          const float *Dv, const float *wW, const float *wNW, const float *wN,
                          const float *wNE, const float *wE, const float *wSE, const float *wS,
                          const float *wSW, int nrows, int ncols, int nframes_coef, int iter,
                          float omega, int solver, float *dU_out, float *dV_out, float *RU,
                          float *RV);

/* [AU,AV] = Oflow_lhs_elin4_2d(U,V,M,Du,Dv,wW,wN,wE,wS)
 * replaces mex/source/Oflow_lhs_elin4_2d.c:56-231 -> LHS_elin4_2d (opticalflowSolvers.c:387). */
int pdeip_oflow_lhs_elin4(const float *U, const float *V, const float *M, const float *Du,
                          const float *Dv, const float *wW, const float *wN, const float *wE,
                          const float *wS, int nrows, int ncols, int nframes_coef, float *AU,
                          float *AV);

/* [AU,AV] = Oflow_lhs_llin4_2d(U,V,dU,dV,M,Du,Dv,wW,wN,wE,wS)
 * replaces mex/source/Oflow_lhs_llin4_2d.c:59-260 -> LHS_llin4_2d (opticalflowSolvers.c:923),
 * including its top-border quirk (:1056). */
int pdeip_oflow_lhs_llin4(const float *U, const float *V, const float *dU, const float *dV,
                          const float *M, const float *Du, const float *Dv, const float *wW,
                          const float *wN, const float *wE, const float *wS, int nrows, int ncols,
                          int nframes_coef, float *AU, float *AV);

/* [dU(,RU)] = Disp_sor_llin4_2d(U,dU,Cu,Du,wW,wN,wE,wS,iter,omega,solver)
 * replaces mex/source/Disp_sor_llin4_2d.c:59-282 -> GS_SOR_llin4_2d (disparitySolvers.c:41).
 * The gateway allocates RU but never computes it (:251-281): zeros. */
int pdeip_disp_sor_llin4(const float *U, const float *dU, const float *Cu, const float *Du,
                         const float *wW, const float *wN, const float *wE, const float *wS,
                         int nrows, int ncols, int iter, float omega, int solver, float *dU_out,
                         float *RU);

/* [dU0 dU1] = Disp_sor_llin_sym4_2d(U0,dU0,Cu0,Du0,wW0,wN0,wE0,wS0, U1,dU1,Cu1,Du1,wW1,wN1,wE1,wS1, iter,omega,solver)
 * replaces mex/source/Disp_sor_llin_sym4_2d.c:82-440 -> GS_SOR_llinsym4_2d / GS_ALR_SOR_llinsym4_2d
 * (disparitySolvers.c:301,462): two disparity fields relaxed side by side that never read each other (the
 * symmetry constraint lives in the MATLAB driver).  The gateway solves unconditionally: iter<=0 returns copies. */
int pdeip_disp_sor_llin_sym4(const float *U0, const float *dU0, const float *Cu0, const float *Du0, const float *wW0,
                             const float *wN0, const float *wE0, const float *wS0, const float *U1, const float *dU1,
                             const float *Cu1, const float *Du1, const float *wW1, const float *wN1, const float *wE1,
                             const float *wS1, int nrows, int ncols, int iter, float omega, int solver,
                             float *dU_out0, float *dU_out1);

/* X = PDEsolver4(X,TRACE,B,wW,wN,wE,wS,iter,omega,solver)
 * replaces mex/source/PDEsolver4.c:54-249 -> GS_SOR_4_2d (pdeSolvers.c:44).  Every plane is
 * [nrows x ncols x nframes].  iter<=0 returns a copy (PDEsolver4.c:239).  solver 3 (unbound
 * function pointer in the reference, PDEsolver4.c:228) is rejected with PDEIP_ERR_SOLVER. */
int pdeip_pde_sor4(const float *X, const float *TRACE, const float *B, const float *wW,
                   const float *wN, const float *wE, const float *wS, int nrows, int ncols,
                   int nframes, int iter, float omega, int solver, float *X_out);

/* X = PDEsolver8(X,TRACE,B,wW,wNW,wN,wNE,wE,wSE,wS,wSW,iter,omega,solver)
 * replaces mex/source/PDEsolver8.c:54-309 -> GS_SOR_8_2d (pdeSolvers.c:153). */
int pdeip_pde_sor8(const float *X, const float *TRACE, const float *B, const float *wW,
                   const float *wNW, const float *wN, const float *wNE, const float *wE,
                   const float *wSE, const float *wS, const float *wSW, int nrows, int ncols,
                   int nframes, int iter, float omega, int solver, float *X_out);

/* [wW,wN,wE,wS] = DdiffWeights(D,eps)
 * replaces mex/source/DdiffWeights.c:50-140 -> diffWeights6_2D_c (imageDiffusionWeights.c:341).
 * D and the four outputs are [nrows x ncols x nframes]; frame 0 of each output holds the
 * weights (max over frames), frames >= 1 stay zero as in the gateway. */
int pdeip_diffweights6(const float *D, int nrows, int ncols, int nframes, float eps, float *wW,
                       float *wN, float *wE, float *wS);

/* Iout = BilinInterp_2d(Iin,X,Y)
 * replaces mex/source/BilinInterp_2d.c:41-124 -> bilinInterp2 (imageInterpolation.c:44).
 * Iin,Iout are [nrows x ncols x nframes]; X,Y are [nrows x ncols] 1-based coordinates
 * (X = column, Y = row).  Out-of-range samples are NaN. */
int pdeip_warp_bilinear(const float *Iin, const float *X, const float *Y, int nrows, int ncols,
                        int nframes, float *Iout);

/* [Idt,Idx,Idy] = FstDerivatives5(It0,It1)
 * replaces mex/source/FstDerivatives5.c:50-145 -> fstSimoncelli_c (library/imageDerivatives.c:309).
 * Every plane is [nrows x ncols x nframes] (frames independent); nrows, ncols >= 4. */
int pdeip_fst_derivatives5(const float *It0, const float *It1, int nrows, int ncols, int nframes,
                           float *Idt, float *Idx, float *Idy);

/* [Idxt,Idyt,Idxx,Idyy,Idxy] = SndDerivatives5(It0,It1)
 * replaces mex/source/SndDerivatives5.c:51-174 -> sndSimoncelli_c (imageDerivatives.c:391). */
int pdeip_snd_derivatives5(const float *It0, const float *It1, int nrows, int ncols, int nframes,
                           float *Idxt, float *Idyt, float *Idxx, float *Idyy, float *Idxy);

/* ---- device-pointer entry points ---------------------------------------------------------
 * Same arithmetic on buffers already resident in HBM; asynchronous on `stream`.  Solvers work
 * in place on the iterate and take the ordering `mode` explicitly.  iter<=0 is a no-op here
 * (the gateway's zero/copy semantics belong to the host entry points).  `col0` is the
 * global column index of local column 0 when the buffers are one slab of a column-slab
 * decomposition (only its parity matters, for the colour of a pixel); 0 for a whole image. */
int pdeip_oflow_sor_elin4_dev(void *stream, float *U, float *V, const float *M, const float *Cu,
                              const float *Cv, const float *Du, const float *Dv, const float *wW,
                              const float *wN, const float *wE, const float *wS, int nrows,
                              int ncols, int iter, float omega, int mode, int col0);
int pdeip_oflow_sor_llin4_dev(void *stream, const float *U, const float *V, float *dU, float *dV,
                              const float *M, const float *Cu, const float *Cv, const float *Du,
                              const float *Dv, const float *wW, const float *wN, const float *wE,
                              const float *wS, int nrows, int ncols, int iter, float omega,
                              int mode, int col0);
int pdeip_disp_sor_llin4_dev(void *stream, const float *U, float *dU, const float *Cu,
                             const float *Du, const float *wW, const float *wN, const float *wE,
                             const float *wS, int nrows, int ncols, int iter, float omega,
                             int mode, int col0);
int pdeip_disp_sor_llin_sym4_dev(void *stream, const float *U0, float *dU0, const float *Cu0, const float *Du0,
                                 const float *wW0, const float *wN0, const float *wE0, const float *wS0,
                                 const float *U1, float *dU1, const float *Cu1, const float *Du1,
                                 const float *wW1, const float *wN1, const float *wE1, const float *wS1,
                                 int nrows, int ncols, int iter, float omega, int solver, int mode, int col0);
int pdeip_pde_sor4_dev(void *stream, float *X, const float *TRACE, const float *B, const float *wW,
                       const float *wN, const float *wE, const float *wS, int nrows, int ncols,
                       int nframes, int iter, float omega, int mode, int col0);
/* Out-of-place forms of the four 5-point point solvers: the iterate planes are only READ and the relaxed iterate is written to
 * the `_out` planes (iter <= 0: a copy) -- the shape of the gateways themselves (copy the input in, solve on the output:
 * Oflow_sor_elin4_2d.c:341-346).  The red-black launches ping-pong between buffers, so this form never needs the
 * device-to-device copy that an in-place call with an odd number of launches ends with; `_out` == the inputs is the in-place
 * call.  Callers that relax the same planes again and again alternate between two sets. */
int pdeip_oflow_sor_elin4_dev_to(void *stream, const float *U, const float *V, float *U_out, float *V_out, const float *M,
                                 const float *Cu, const float *Cv, const float *Du, const float *Dv, const float *wW,
                                 const float *wN, const float *wE, const float *wS, int nrows, int ncols, int iter,
                                 float omega, int mode, int col0);
int pdeip_oflow_sor_llin4_dev_to(void *stream, const float *U, const float *V, const float *dU, const float *dV,
                                 float *dU_out, float *dV_out, const float *M, const float *Cu, const float *Cv,
                                 const float *Du, const float *Dv, const float *wW, const float *wN, const float *wE,
                                 const float *wS, int nrows, int ncols, int iter, float omega, int mode, int col0);
int pdeip_disp_sor_llin4_dev_to(void *stream, const float *U, const float *dU, float *dU_out, const float *Cu,
                                const float *Du, const float *wW, const float *wN, const float *wE, const float *wS,
                                int nrows, int ncols, int iter, float omega, int mode, int col0);
int pdeip_pde_sor4_dev_to(void *stream, const float *X, float *X_out, const float *TRACE, const float *B, const float *wW,
                          const float *wN, const float *wE, const float *wS, int nrows, int ncols, int nframes, int iter,
                          float omega, int mode, int col0);
int pdeip_pde_sor8_dev(void *stream, float *X, const float *TRACE, const float *B, const float *wW,
                       const float *wNW, const float *wN, const float *wNE, const float *wE,
                       const float *wSE, const float *wS, const float *wSW, int nrows, int ncols,
                       int nframes, int iter, float omega, int mode, int col0);
/* Alternating line relaxation, solver = 2 of the gateways (GS_ALR_SOR_*: opticalflowSolvers.c:196,690,1677;
 * disparitySolvers.c:154; pdeSolvers.c:277,344).  Iterate planes in place.
 *   mode PDEIP_MODE_EXACT_ORDER: the reference's line order, bit-identical, inherently serial (one
 *        workgroup per frame; a line of more than 10240 pixels is held in global memory instead of LDS: slow).
 *   mode PDEIP_MODE_RED_BLACK:   zebra order (even lines, then odd lines), lines solved concurrently.
 * pdeip_pde_alr8_dev runs ONE iteration whatever `iter` is, like the reference (pdeSolvers.c:362). */
int pdeip_oflow_alr_elin4_dev(void *stream, float *U, float *V, const float *M, const float *Cu,
                              const float *Cv, const float *Du, const float *Dv, const float *wW,
                              const float *wN, const float *wE, const float *wS, int nrows, int ncols,
                              int iter, float omega, int mode);
int pdeip_oflow_alr_llin4_dev(void *stream, const float *U, const float *V, float *dU, float *dV,
                              const float *M, const float *Cu, const float *Cv, const float *Du,
                              const float *Dv, const float *wW, const float *wN, const float *wE,
                              const float *wS, int nrows, int ncols, int iter, float omega, int mode);
int pdeip_oflow_alr_llin8_dev(void *stream, const float *U, const float *V, float *dU, float *dV,
                              const float *M, const float *Cu, const float *Cv, const float *Du,
                              const float *Dv, const float *wW, const float *wNW, const float *wN,
                              const float *wNE, const float *wE, const float *wSE, const float *wS,
                              const float *wSW, int nrows, int ncols, int iter, float omega, int mode);
int pdeip_disp_alr_llin4_dev(void *stream, const float *U, float *dU, const float *Cu, const float *Du,
                             const float *wW, const float *wN, const float *wE, const float *wS,
                             int nrows, int ncols, int iter, float omega, int mode);
int pdeip_pde_alr4_dev(void *stream, float *X, const float *TRACE, const float *B, const float *wW,
                       const float *wN, const float *wE, const float *wS, int nrows, int ncols,
                       int nframes, int iter, float omega, int mode);
int pdeip_pde_alr8_dev(void *stream, float *X, const float *TRACE, const float *B, const float *wW,
                       const float *wNW, const float *wN, const float *wNE, const float *wE,
                       const float *wSE, const float *wS, const float *wSW, int nrows, int ncols,
                       int nframes, int iter, float omega, int mode);
int pdeip_oflow_res_elin4_dev(void *stream, float *RU, float *RV, const float *U, const float *V,
                              const float *M, const float *Cu, const float *Cv, const float *Du,
                              const float *Dv, const float *wW, const float *wN, const float *wE,
                              const float *wS, int nrows, int ncols, int nframes_coef);
int pdeip_oflow_lhs_elin4_dev(void *stream, float *AU, float *AV, const float *U, const float *V,
                              const float *M, const float *Du, const float *Dv, const float *wW,
                              const float *wN, const float *wE, const float *wS, int nrows,
                              int ncols, int nframes_coef);
int pdeip_oflow_res_llin4_dev(void *stream, float *RU, float *RV, const float *U, const float *V,
                              const float *dU, const float *dV, const float *M, const float *Cu,
                              const float *Cv, const float *Du, const float *Dv, const float *wW,
                              const float *wN, const float *wE, const float *wS, int nrows,
                              int ncols, int nframes_coef);
int pdeip_oflow_lhs_llin4_dev(void *stream, float *AU, float *AV, const float *U, const float *V,
                              const float *dU, const float *dV, const float *M, const float *Du,
                              const float *Dv, const float *wW, const float *wN, const float *wE,
                              const float *wS, int nrows, int ncols, int nframes_coef);
int pdeip_diffweights6_dev(void *stream, const float *D, int nrows, int ncols, int nframes,
                           float eps, float *wW, float *wN, float *wE, float *wS);
int pdeip_warp_bilinear_dev(void *stream, const float *Iin, const float *X, const float *Y,
                            int nrows, int ncols, int nframes, float *Iout);
int pdeip_fst_derivatives5_dev(void *stream, const float *It0, const float *It1, int nrows, int ncols,
                               int nframes, float *Idt, float *Idx, float *Idy);
int pdeip_snd_derivatives5_dev(void *stream, const float *It0, const float *It1, int nrows, int ncols,
                               int nframes, float *Idxt, float *Idyt, float *Idxx, float *Idyy,
                               float *Idxy);

/* ---- MATLAB-side stages of one late-linearisation pyramid level, device-resident ("next row" f2) --------
 * What matlab/optical_flow/FlowEminND_llin_2D_v10.m runs between its MEX calls inside firstLoop/secondLoop
 * (:208-356), so that a level can stay in HBM (pde-based-image-processing_amd/flow_level.py drives them).
 * Restatements of MATLAB array code: single/double typing and expression order as written there;
 * checked against oracle/matlab_side.py, not against MATLAB ("parity unpinned"). */
/* X = single((1:cols) + U), Y = single((1:rows)' + V): the warp coordinates (:223) */
int pdeip_flow_coords_dev(void *stream, const float *U, const float *V, int nrows, int ncols, float *X, float *Y);
/* The warp step of one firstLoop iteration in one launch (:223-231): W1 = bilinInterp2(I1, X+U, Y+V) and, with C2 > 0,
 * W2 = bilinInterp2(I2, X+U, Y+V), X,Y = meshgrid(1:cols,1:rows), the sums rounded to single as pdeip_flow_coords_dev stores
 * them.  V may be NULL (warp along x only). */
int pdeip_flow_warp_dev(void *stream, const float *U, const float *V, const float *I1, int C1, const float *I2, int C2, int nrows,
                        int ncols, float *W1, float *W2);
/* robust data-term assembly (:283-327): gD = b./(alpha*sqrt((It - Ix.*dU - Iy.*dV).^2 + 1e-5)) per channel of one or
 * two data terms ([nrows x ncols x C] derivative arrays; C2 = 0: no second term), then nansum over all channels of
 * (Iy.*Ix).*gD -> MGd, (It.*Ix).*gD -> CuGd, (It.*Iy).*gD -> CvGd, (Ix.*Ix).*gD -> DuGd, (Iy.*Iy).*gD -> DvGd. */
int pdeip_flow_assemble_dev(void *stream, const float *It1, const float *Ix1, const float *Iy1, int C1, float b1,
                            const float *It2, const float *Ix2, const float *Iy2, int C2, float b2, const float *dU,
                            const float *dV, float alpha, int nrows, int ncols, float *MGd, float *CuGd, float *CvGd,
                            float *DuGd, float *DvGd);
/* disparity twin (matlab/disparity/DispEminND_llin_2D.m:258-293): CuGd = sum_c (It.*Ix).*gD, DuGd = sum_c (Ix.*Ix).*gD with
 * gD = b./(alpha.*realsqrt((It - Ix.*dU).^2 + 1e-5)); plain sum: NaN propagates to the solver's isnan(Cu) test */
int pdeip_disp_assemble_dev(void *stream, const float *It1, const float *Ix1, int C1, float b1, const float *It2,
                            const float *Ix2, int C2, float b2, const float *dU, float alpha, int nrows, int ncols,
                            float *CuGd, float *DuGd);
/* The same with a gradient-magnitude second term (sndTerm 'gradmag', :253-258, :291-293 -- what runme.m configures): the five
 * planes of SndDerivatives5(I2t0, I2t1w) [.. x C2] take the place of the first-order derivatives. */
int pdeip_flow_assemble_gradmag_dev(void *stream, const float *It1, const float *Ix1, const float *Iy1, int C1, float b1,
                                    const float *Ixt, const float *Iyt, const float *Ixx, const float *Iyy, const float *Ixy, int C2,
                                    float b2, const float *dU, const float *dV, float alpha, int nrows, int ncols, float *MGd,
                                    float *CuGd, float *CvGd, float *DuGd, float *DvGd);
/* One inner iteration's nine coefficient planes in one pass: the assembly above (second term: three first-order planes with
 * Iyy = Ixy = NULL, or the five gradient-magnitude planes Ixt, Iyt, Ixx, Iyy, Ixy) and OPdiffWeights(U+dU, V+dV) (:389-433,
 * pdeip_flow_opdiffweights_dev) of the same iterate -- both only read dU, dV, so the drivers' two calls fuse into one launch. */
int pdeip_flow_assemble_weights_dev(void *stream, const float *It1, const float *Ix1, const float *Iy1, int C1, float b1,
                                    const float *A2, const float *B2, const float *C2p, const float *Iyy, const float *Ixy, int C2,
                                    float b2, const float *U, const float *V, const float *dU, const float *dV, float alpha, int nrows,
                                    int ncols, float *MGd, float *CuGd, float *CvGd, float *DuGd, float *DvGd, float *wW, float *wN,
                                    float *wS, float *wE);
/* disparity twin (matlab/disparity/DispEminND_llin_2D.m:236-238, :271) */
int pdeip_disp_assemble_gradmag_dev(void *stream, const float *It1, const float *Ix1, int C1, float b1, const float *Ixt,
                                    const float *Iyt, const float *Ixx, const float *Ixy, int C2, float b2, const float *dU, float alpha,
                                    int nrows, int ncols, float *CuGd, float *DuGd);
/* The spatial a-priori slice of that assembly (:262-270, :301-318; param.Us / param.Vs, gammaS): appends ASCu.*gSu to CGd and
 * 1.*gSu to DGd (nansum).  Us: the constraint field of the scale, a MATLAB double array; as_diff = 2*(1/scl_factor)^-(scl-1);
 * u_double: U is still the double array of the coarsest scale's first firstLoop; du_double: first inner iteration (dU = zeros). */
int pdeip_flow_apriori_dev(void *stream, const double *Us, const float *U, const float *dU, double gammaS, double alpha,
                           double as_diff, int u_double, int du_double, int nrows, int ncols, float *CGd, float *DGd);
/* The disparity driver's variant (DispEminND_llin_2D.m:246-248, :277-284, :291-292; param.Us, gammaS): ASCu = Us - U, ASDu = 1,
 * gS = gammaS/alpha * exp(-(Us - U - dU)^2 / as_diff^2) with as_diff = 1.75*(1/scl_factor)^-(scl-1); the slices are added to
 * CGd / DGd by a plain sum (NaN propagates).  exp() is the library's own fixed double algorithm (csrc/pdeip_flow.hpp det_exp),
 * shared with the numpy statement of the driver, so results are reproducible bit for bit; not MATLAB's exp to the last ulp. */
int pdeip_disp_apriori_dev(void *stream, const double *Us, const float *U, const float *dU, double gammaS, double alpha,
                           double as_diff, int u_double, int du_double, int nrows, int ncols, float *CGd, float *DGd);
/* rgb2grad (FlowEminND_llin_2D_v10.m:368-381; fstTerm 'grad'): out [.. x 2*nframes], frames 2f-1 / 2f (1-based) = the [1 0 -1]
 * differences of input frame f along x / y, replicate borders */
int pdeip_rgb2grad_dev(void *stream, const float *in, int nrows, int ncols, int nframes, float *out);
/* out = A + B (single); e.g. the argument of DdiffWeights(single(U+dU), eps) (:283) */
int pdeip_add_dev(void *stream, const float *A, const float *B, int nrows, int ncols, float *out);
/* Horn-Schunck, early linearisation: the data terms of one scale (matlab/optical_flow/FlowEminHS_elin_2D_v10.m:133-164) from the
 * two frames [nrows x ncols x C]: separable 5-tap derivative filters and the b1/b2-weighted motion tensor, summed over channels. */
int pdeip_hs_assemble_dev(void *stream, const float *It0, const float *It1, int C, float b1, float b2, int nrows, int ncols,
                          float *MGd, float *CuGd, float *CvGd, float *DuGd, float *DvGd);
/* [wW wN wS wE] = OPdiffWeights(U+dU, V+dV) (:389-433), evaluated in double, returned as single */
int pdeip_flow_opdiffweights_dev(void *stream, const float *U, const float *V, const float *dU, const float *dV, int nrows,
                                 int ncols, float *wW, float *wN, float *wS, float *wE);
/* dU, dV may both be NULL: OPdiffWeights(U, V) of the early-linearisation drivers (FlowEminNDFASFMG_elin_2D_v10.m:392). */
/* Diagnostic: the weights' single(1 ./ sqrt(x)) takes a short sequence that is proven against the exact one (double sqrt,
 * double divide, one rounding) value by value; this runs n arguments (random ones and ones aimed at rounding boundaries)
 * through both and returns how many differ (0 expected), or -1 on a device error. */
int pdeip_selftest_inv_sqrt(int n, unsigned seed);

/* ---- the drivers' image pyramid.  IPT semantics have nothing to be checked against here: pyramid.py states our definition
 * (tap lists at MATLAB's pixel-centre convention, antialiased when shrinking, replicate borders, double accumulation) and
 * these entry points compute exactly that. ---- */
/* imresize(in, [nrows_out ncols_out], 'bilinear' (cubic = 0) or 'bicubic' (1)) */
int pdeip_pyr_resize_dev(void *stream, const float *in, int nrows, int ncols, int nframes, int nrows_out, int ncols_out, int cubic,
                         float *out);
/* imfilter(in, G, 'replicate'), G an odd size x size mask (row-major doubles in host memory, size <= 7) */
int pdeip_pyr_smooth_dev(void *stream, const float *in, int nrows, int ncols, int nframes, const double *G, int size, float *out);

/* ---- symmetric stereo (matlab/disparity/DispEminND_llin_sym_2D.m): the stages the other drivers do not have.  Planes marked
 * double are MATLAB doubles there (U, the warped disparities and what is derived from them). ---- */
/* out = interp2(X, Y, U, X+Uq, Y) (:140-141): linear along x, NaN outside the grid */
int pdeip_sym_warp_flow_dev(void *stream, const float *U, const float *Uq, int nrows, int ncols, double *out);
/* Udt = (U+Uw)*0.5, Udx = prefiltered x-derivative of Uw, CuS = Udt.*(1+Udx), DuS = 1+Udx+Udx+Udx.*Udx (:156-175) */
int pdeip_sym_flow_terms_dev(void *stream, const float *U, const double *Uw, int nrows, int ncols, double *Udt, double *Udx,
                             double *CuS, double *DuS);
/* CuG, DuG of one view (:189-222): robust data term over the C channels plus the symmetry term with
 * gSYM = kS./(1 + Snorm/sr2), kS = channels*beta/alpha, sr2 = srDiff^2; first != 0 in the first inner iteration (dU still
 * MATLAB's double zeros: symmetry weights in double), 0 afterwards (single) */
int pdeip_sym_assemble_dev(void *stream, const float *Idt, const float *Idx, const float *Idxt, const float *Idyt, const float *Idxx,
                           const float *Idxy, int C, const double *Udt, const double *Udx, const double *CuS, const double *DuS,
                           const float *dU, float b1, float b2, float alpha, double kS, double sr2, int first, int nrows, int ncols,
                           float *CuG, float *DuG);

/* TVdenoise4's work between two PDEsolver4 calls (matlab/denoising/TVdenoise4.m:84-98 with DiffWeights :116-156), all single:
 * the four weights (maximum over the frames, outer column/row zeroed) scaled by alpha, PsiData, TRACE, B; [.. x nframes] each. */
int pdeip_tv4_assemble_dev(void *stream, const float *Iout, const float *Iin, int nrows, int ncols, int nframes, float alpha,
                           float *TRACE, float *B, float *aW, float *aN, float *aE, float *aS);
/* [W NW N NE E SE S SW] = ADdiffWeights(D, quantile) of the anisotropic flow driver (matlab/optical_flow/
 * FlowEminAD_llin_2D_v10.m:416-487): Alvarez derivative in double, strongest frame per pixel, lambda = the quantile of the
 * non-zero squared gradient norms, tensor weights with circshift wrap-around; returned as single (the solver's arguments). */
int pdeip_ad_weights_dev(void *stream, const float *D, int nrows, int ncols, int nframes, double quantile, float *wW, float *wNW,
                         float *wN, float *wNE, float *wE, float *wSE, float *wS, float *wSW);

/* ---- FAS full-multigrid flow (matlab/optical_flow/FlowEminNDFASFMG_elin_2D_v10.m): the stages between its MEX calls ----
 * Planes are [nrows x ncols x frames], column-major, as everywhere.  Output dimensions of the two halving stages are
 * ceil(nrows/2) x ceil(ncols/2) (MATLAB's 1:2:end). */
/* imfilter(I, G, 'replicate', 'conv') with a 5x5 kernel (:104-105); g25 = the kernel G itself, column-major (host memory) */
int pdeip_fas_gauss5_dev(void *stream, const float *in, int nrows, int ncols, int frames, const float *g25, float *out);
/* one pyramid step (:108-111): [1 4 6 4 1]/16 along both axes, then (1:2:end, 1:2:end, :) */
int pdeip_fas_down_dev(void *stream, const float *in, int nrows, int ncols, int frames, float *out);
/* the per-scale constants (:125-153) from the frames (0..255 range): planes = [13][frames][ncols][nrows] in the order
 * Idt, Idx, Idy, Idxx, Idyy, Idxy, Idxt, Idyt, M, Cu, Cv, Du, Dv */
int pdeip_fas_prepare_dev(void *stream, const float *It0, const float *It1, int nrows, int ncols, int frames, float b1, float b2,
                          float *planes);
/* gd = 1./(k*sqrt(OPnorm+0.00001)) at (U,V) and the solver's planes (:377-397 summed over the frames when per_frame = 0,
 * one plane each; :425-445 / :228-237 per frame when per_frame = 1, [.. x frames] each, plus gd).  Cu/Cv: the right-hand
 * side [.. x frames] (the scale's own or the cycle's fu/fv); they, their outputs and gd may be NULL. */
int pdeip_fas_assemble_dev(void *stream, const float *planes, const float *Cu, const float *Cv, const float *U, const float *V,
                           int nrows, int ncols, int frames, float b1, float b2, float k, int per_frame, float *MGd, float *CuGd,
                           float *CvGd, float *DuGd, float *DvGd, float *gd);
/* The smoother's two calls of one firstLoop iteration in one launch: pdeip_fas_assemble_dev with per_frame = 0 and
 * OPdiffWeights(U, V) (:392). */
int pdeip_fas_assemble_weights_dev(void *stream, const float *planes, const float *Cu, const float *Cv, const float *U, const float *V,
                                   int nrows, int ncols, int frames, float b1, float b2, float k, float *MGd, float *CuGd, float *CvGd,
                                   float *DuGd, float *DvGd, float *wW, float *wN, float *wS, float *wE);
/* imfilter(in*scale, [1 2 1;2 4 2;1 2 1]/16, 'replicate', 'conv')(1:2:end, 1:2:end, :) (:200, :212-217) */
int pdeip_fas_restrict_dev(void *stream, const float *in, int nrows, int ncols, int frames, float scale, float *out);
/* out = (R + A)./gd (:250-251) */
int pdeip_fas_rhs_dev(void *stream, const float *R, const float *A, const float *gd, int nrows, int ncols, int frames, float *out);
/* U = U + imresize((Uc-Ures)*inv_scale, size(U), 'bilinear') (:256-257); Uc, Ures are [nrows_c x ncols_c] */
int pdeip_fas_prolong_add_dev(void *stream, float *U, int nrows, int ncols, const float *Uc, const float *Ures, int nrows_c,
                              int ncols_c, float inv_scale);
/* out = imresize(in.*mul, [nrows_out ncols_out]) with imresize's default bicubic kernel, enlarging (:177-180) */
int pdeip_fas_upscale_dev(void *stream, const float *in, int nrows, int ncols, float mul, int nrows_out, int ncols_out, float *out);
/* TVdenoise8's work between two PDEsolver8 calls (matlab/denoising/TVdenoise8.m:80-86 with ADdiffWeights :119-231):
 * the anisotropic weights of Iout (double; Alvarez derivative, strongest frame per pixel, lambda = median of the
 * non-zero squared gradient norms), then TRACE = PsiData + alpha*sum(w), B = PsiData.*Iin with
 * PsiData = 1./sqrt((Iout-Iin).^2 + eps), and single(alpha*w) for the eight weights; all [nrows x ncols x nframes]. */
int pdeip_tv_assemble_dev(void *stream, const float *Iout, const float *Iin, int nrows, int ncols, int nframes,
                          float alpha, float *TRACE, float *B, float *aW, float *aNW, float *aN, float *aNE,
                          float *aE, float *aSE, float *aS, float *aSW);
/* out = medfilt2(A + B, [3 3], 'symmetric') (:352); B may be NULL (out = medfilt2(A)); out must not alias A or B */
int pdeip_median3_dev(void *stream, const float *A, const float *B, int nrows, int ncols, float *out);
/* Two fields in one launch: out0 = medfilt2(A0 + B0), out1 = medfilt2(A1 + B1) (:352-353 filters U+dU and V+dV). */
int pdeip_median3_pair_dev(void *stream, const float *A0, const float *B0, const float *A1, const float *B1, int nrows, int ncols,
                           float *out0, float *out1);

/* ---- whole drivers, resident on the device (csrc/pdeip_drivers.hip) --------------------------------------------------------
 * What `runme.m` calls -- its eight drivers, FlowEminND_llin_2D_v10 (runme.m:44) and DispEminND_llin_2D (runme.m:20) first -- as ONE
 * host-pointer call each: the frames go up once, the coarse-to-fine loop (pyramid, warps, derivatives, robust assembly, diffusion weights,
 * solver calls, medians, up-scaling) runs on device planes, the result comes down once.  A MATLAB session reaches them through
 * the stubs mex/FlowEminND_llin_2D_v10_gpu.c and mex/DispEminND_llin_2D_gpu.c.  The pyramid's IPT calls
 * (imresize, imfilter, fspecial) are OUR definitions of them (pyramid.py); everything between them is the arithmetic the
 * per-stage entry points above are tested for. */
#define PDEIP_TERM_NONE 0
#define PDEIP_TERM_RGB 1
#define PDEIP_TERM_GRAD 2     /* first term only: rgb2grad */
#define PDEIP_TERM_GRADMAG 3  /* second term only: gradient magnitude through SndDerivatives5 */
/* param struct of both drivers (FlowEminND_llin_2D_v10.m:52-67, DispEminND_llin_2D.m:51-66); a member that is <= 0 (or NaN)
 * keeps the driver's own default; `scales` limits the number of pyramid scales (param.scales).  NULL: all defaults. */
typedef struct pdeip_driver_params {
    double alpha, omega, gammaS, b1, b2, scl_factor;
    int firstLoop, secondLoop, iter, solver, scales;
} pdeip_driver_params;
/* [U V] = FlowEminAD_llin_2D_v10(Iin, channels, fstTerm, sndTerm, param) (matlab/optical_flow/FlowEminAD_llin_2D_v10.m,
 * runme.m:54,64): the anisotropic-diffusion flow driver; arguments as pdeip_flow_nd_llin plus param.quantile (<= 0: 0.9) and
 * param.diffusion (flow_diffusion 0: 'image', the default -- eight weights from frame 0 of a scale, once per scale; 1: 'flow' --
 * from U+dU+V+dV in every inner iteration). */
int pdeip_flow_ad_llin(const float *Iin, int nrows, int ncols, int channels, int fst_term, int snd_term, const pdeip_driver_params *prm,
                       double quantile, int flow_diffusion, const double *Us, const double *Vs, float *U, float *V);
/* [U V] = FlowEminNDFASFMG_elin_2D_v10(Iin, channels, param) (matlab/optical_flow/FlowEminNDFASFMG_elin_2D_v10.m, runme.m:90): the FAS
 * full-multigrid flow driver in one call; Iin as for pdeip_flow_nd_llin (0..255, not rescaled by this driver).  A member that is
 * <= 0 (or NaN) keeps the driver's default (alpha 0.035, omega 1.9, firstLoop 4, iter 4, b1 0.03, b2 0.97, scl_factor 0.5,
 * solver 2, cycle_index 1 = V-cycle, scales = until a side is <= 10 pixels); NULL: all defaults. */
typedef struct pdeip_fas_params {
    double alpha, omega, b1, b2, scl_factor;
    int firstLoop, iter, solver, cycle_index, scales;
} pdeip_fas_params;
int pdeip_flow_fas_fmg_elin(const float *Iin, int nrows, int ncols, int channels, const pdeip_fas_params *prm, float *U, float *V);
/* [U V] = FlowEminHS_elin_2D_v10(Iin, channels, param) (matlab/optical_flow/FlowEminHS_elin_2D_v10.m, runme.m:74): Horn-Schunck
 * with early linearisation, the whole coarse-to-fine run in one call; Iin as for pdeip_flow_nd_llin.  Of the parameter struct
 * alpha (0.2), omega (1.9), iter (20), b1 (0.25), b2 (0.75), scl_factor (0.75) and solver (2) are this driver's. */
int pdeip_flow_hs_elin(const float *Iin, int nrows, int ncols, int channels, const pdeip_driver_params *prm, float *U, float *V);
/* U = DispEminND_llin_sym_2D(Il, Ir, param) (matlab/disparity/DispEminND_llin_sym_2D.m, runme.m:28): symmetric stereo, both
 * views' disparities; Il, Ir single [nrows x ncols x channels] (this driver does not divide by 255); U: [nrows x ncols x 2].
 * A member that is <= 0 (or NaN) keeps the driver's default (alpha 0.035, beta 0.4, omega 1.9, firstLoop 3, secondLoop 4, iter 4,
 * b1 0.25, b2 0.72, scl_factor 0.75, solver 2); NULL: all defaults. */
typedef struct pdeip_sym_params {
    double alpha, beta, omega, b1, b2, scl_factor;
    int firstLoop, secondLoop, iter, solver;
} pdeip_sym_params;
int pdeip_disp_nd_llin_sym(const float *Il, const float *Ir, int nrows, int ncols, int channels, const pdeip_sym_params *prm, float *U);
/* Iout = TVdenoise8(I_in, param) (matlab/denoising/TVdenoise8.m, runme.m:144) and Iout = TVdenoise4(I_in, param)
 * (TVdenoise4.m, runme.m:143) as one host-pointer call each: I_in single, 0..1, [nrows x ncols x frames] column-major; the short
 * pyramid, the lagged-diffusivity loop (weights, PsiData / TRACE / B, PDEsolver8 | PDEsolver4) and the up-scaling stay on the
 * device.  A member of the parameter struct that is <= 0 (or NaN) keeps the driver's own default; NULL: all defaults
 * (TVdenoise8: alpha 500, omega 1.75, outer_iter 20, inner_iter 4, solver 2, scl 0.75, scl_factor 0.75; TVdenoise4: alpha 5,
 * omega 1.75, outer_iter 10, inner_iter 5, solver 2, scl 0.5, scl_factor 0.75). */
typedef struct pdeip_tv_params {
    double alpha, omega, scl, scl_factor;
    int outer_iter, inner_iter, solver;
} pdeip_tv_params;
int pdeip_tvdenoise8(const float *Iin, int nrows, int ncols, int frames, const pdeip_tv_params *prm, float *Iout);
int pdeip_tvdenoise4(const float *Iin, int nrows, int ncols, int frames, const pdeip_tv_params *prm, float *Iout);
/* [U V] = FlowEminND_llin_2D_v10(Iin, channels, fstTerm, sndTerm, param): Iin = cat(3, frame0, frame1), single, 0..255,
 * [nrows x ncols x 2*channels] column-major; fst_term PDEIP_TERM_RGB | _GRAD, snd_term _NONE | _RGB | _GRADMAG; Us, Vs:
 * param.Us / param.Vs, double [nrows x ncols] or NULL; U, V: [nrows x ncols].  Ordering: pdeip_set_mode / PDEIP_MODE. */
int pdeip_flow_nd_llin(const float *Iin, int nrows, int ncols, int channels, int fst_term, int snd_term, const pdeip_driver_params *prm,
                       const double *Us, const double *Vs, float *U, float *V);
/* U = DispEminND_llin_2D(Il, Ir, fstTerm, sndTerm, param): Il, Ir single 0..255 [nrows x ncols x channels]; Us: param.Us or NULL. */
int pdeip_disp_nd_llin(const float *Il, const float *Ir, int nrows, int ncols, int channels, int fst_term, int snd_term,
                       const pdeip_driver_params *prm, const double *Us, float *U);

#ifdef __cplusplus
}
#endif
#endif /* PDEIP_H */
