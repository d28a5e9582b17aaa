/* [Idxt,Idyt,Idxx,Idyy,Idxy] = SndDerivatives5(It0,It1)
 * Drop-in for mex/source/SndDerivatives5.c (reference gateway :51-174). */
#include "pdeip_mex_util.h"

void mexFunction(int nlhs, mxArray *plhs[], int nrhs, const mxArray *prhs[])
{
    static const char *who = "sndDerivatives";
    const float *a, *b;
    float *o[5];
    int k;
    if (nrhs != 2) mexErrMsgTxt("sndDerivatives: wrong number of input parameters!");
    a = pdeip_single(prhs[0], who, "It0");
    b = pdeip_single(prhs[1], who, "It1");
    if (nlhs < 5) mexErrMsgTxt("sndDerivatives: insufficient number of outputs.");
    for (k = 0; k < 5; k++) o[k] = pdeip_out_like(&plhs[k], prhs[0]);
    pdeip_check(pdeip_snd_derivatives5(a, b, pdeip_rows(prhs[0]), pdeip_cols(prhs[0]), pdeip_frames(prhs[0]), o[0], o[1], o[2],
                                       o[3], o[4]));
}
