/* [Idt,Idx,Idy] = FstDerivatives5(It0,It1)
 * Drop-in for mex/source/FstDerivatives5.c (reference gateway :50-145). */
#include "pdeip_mex_util.h"

void mexFunction(int nlhs, mxArray *plhs[], int nrhs, const mxArray *prhs[])
{
    static const char *who = "fstDerivatives";
    const float *a, *b;
    float *o[3];
    int k;
    if (nrhs != 2) mexErrMsgTxt("fstDerivatives: wrong number of input parameters!");
    a = pdeip_single(prhs[0], who, "It0");
    b = pdeip_single(prhs[1], who, "It1");
    if (nlhs < 3) mexErrMsgTxt("fstDerivatives: insufficient number of outputs...outputs from this function are 'Idt', 'Idx' and 'Idy'.");
    for (k = 0; k < 3; k++) o[k] = pdeip_out_like(&plhs[k], prhs[0]);
    pdeip_check(pdeip_fst_derivatives5(a, b, pdeip_rows(prhs[0]), pdeip_cols(prhs[0]), pdeip_frames(prhs[0]), o[0], o[1], o[2]));
}
