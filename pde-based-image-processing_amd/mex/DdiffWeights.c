/* [wW,wN,wE,wS] = DdiffWeights(D,eps)
 * Drop-in for mex/source/DdiffWeights.c (reference gateway :50-140). */
#include "pdeip_mex_util.h"

void mexFunction(int nlhs, mxArray *plhs[], int nrhs, const mxArray *prhs[])
{
    static const char *who = "DdiffWeights";
    const float *D;
    float *w[4];
    int k;
    if (nrhs != 2) mexErrMsgTxt("diffusion6_2d error: wrong number of input parameters!");
    D = pdeip_single(prhs[0], who, "D");
    if (nlhs < 4)
        mexErrMsgTxt("diffusion6_2d error insufficient number of outputs. Outputs from this function are 'wW', 'wN', 'wE' and 'wS'");
    for (k = 0; k < 4; k++) w[k] = pdeip_out_like(&plhs[k], prhs[0]);
    pdeip_check(pdeip_diffweights6(D, pdeip_rows(prhs[0]), pdeip_cols(prhs[0]), pdeip_frames(prhs[0]),
                                   pdeip_scalar(prhs[1], who, "eps"), w[0], w[1], w[2], w[3]));
}
