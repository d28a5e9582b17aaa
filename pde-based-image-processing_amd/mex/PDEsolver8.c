/* X = PDEsolver8(X,TRACE,B,wW,wNW,wN,wNE,wE,wSE,wS,wSW,iter,omega,solver)
 * Drop-in for mex/source/PDEsolver8.c (reference gateway :54-309). */
#include "pdeip_mex_util.h"

void mexFunction(int nlhs, mxArray *plhs[], int nrhs, const mxArray *prhs[])
{
    static const char *who = "PDEsolver8";
    static const char *names[11] = {"X", "TRACE", "B", "wW", "wNW", "wN", "wNE", "wE", "wSE", "wS", "wSW"};
    const float *p[11];
    float *Xo;
    int k;
    if (nrhs != 14) mexErrMsgTxt("error: wrong number of input parameters!");
    for (k = 0; k < 11; k++) p[k] = pdeip_single(prhs[k], who, names[k]);
    if (nlhs < 1) mexErrMsgTxt("error insufficient number of outputs.");
    Xo = pdeip_out_like(&plhs[0], prhs[0]);
    pdeip_check(pdeip_pde_sor8(p[0], p[1], p[2], p[3], p[4], p[5], p[6], p[7], p[8], p[9], p[10], pdeip_rows(prhs[0]),
                               pdeip_cols(prhs[0]), pdeip_frames(prhs[0]), (int)pdeip_scalar(prhs[11], who, "iter"),
                               pdeip_scalar(prhs[12], who, "omega"), (int)pdeip_scalar(prhs[13], who, "solver"), Xo));
}
