/* X = PDEsolver4(X,TRACE,B,wW,wN,wE,wS,iter,omega,solver)
 * Drop-in for mex/source/PDEsolver4.c (reference gateway :54-249). */
#include "pdeip_mex_util.h"

void mexFunction(int nlhs, mxArray *plhs[], int nrhs, const mxArray *prhs[])
{
    static const char *who = "PDEsolver4";
    static const char *names[7] = {"X", "TRACE", "B", "wW", "wN", "wE", "wS"};
    const float *p[7];
    float *Xo;
    int k;
    if (nrhs != 10) mexErrMsgTxt("error: wrong number of input parameters!");
    for (k = 0; k < 7; k++) p[k] = pdeip_single(prhs[k], who, names[k]);
    if (nlhs < 1) mexErrMsgTxt("error insufficient number of outputs.");
    Xo = pdeip_out_like(&plhs[0], prhs[0]);
    pdeip_check(pdeip_pde_sor4(p[0], p[1], p[2], p[3], p[4], p[5], p[6], pdeip_rows(prhs[0]), pdeip_cols(prhs[0]),
                               pdeip_frames(prhs[0]), (int)pdeip_scalar(prhs[7], who, "iter"),
                               pdeip_scalar(prhs[8], who, "omega"), (int)pdeip_scalar(prhs[9], who, "solver"), Xo));
}
