/* Iout = BilinInterp_2d(Iin,X,Y)
 * Drop-in for mex/source/BilinInterp_2d.c (reference gateway :41-124). */
#include "pdeip_mex_util.h"

void mexFunction(int nlhs, mxArray *plhs[], int nrhs, const mxArray *prhs[])
{
    static const char *who = "BilinInterp_2d";
    const float *I, *X, *Y;
    float *out;
    if (nrhs != 3 || nlhs > 3) mexErrMsgTxt("proper function call is 'bilinInterp2( Iin, X, Y)'");
    I = pdeip_single(prhs[0], who, "Iin");
    X = pdeip_single(prhs[1], who, "X");
    Y = pdeip_single(prhs[2], who, "Y");
    if (nlhs < 1) mexErrMsgTxt("insufficient number of outputs. Outputs from this function is 'Iout'");
    out = pdeip_out_like(&plhs[0], prhs[0]);
    pdeip_check(pdeip_warp_bilinear(I, X, Y, pdeip_rows(prhs[0]), pdeip_cols(prhs[0]), pdeip_frames(prhs[0]), out));
}
