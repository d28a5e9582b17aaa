/* [AU,AV] = Oflow_lhs_llin4_2d(U,V,dU,dV,M,Du,Dv,wW,wN,wE,wS)
 * Drop-in for mex/source/Oflow_lhs_llin4_2d.c (reference gateway :59-260). */
#include "pdeip_mex_util.h"

void mexFunction(int nlhs, mxArray *plhs[], int nrhs, const mxArray *prhs[])
{
    static const char *who = "Oflow_lhs_llin4_2d";
    static const char *names[11] = {"U_in", "V_in", "dU_in", "dV_in", "M", "Du", "Dv", "wW", "wN", "wE", "wS"};
    const float *p[11];
    float *AU, *AV;
    int k;
    if (nrhs != 11) mexErrMsgTxt("Oflow_lhs_llin4_2d parameter error: wrong number of input parameters!");
    for (k = 0; k < 11; k++) p[k] = pdeip_single(prhs[k], who, names[k]);
    if (nlhs < 2) mexErrMsgTxt("Oflow_lhs_llin4_2d insufficient number of outputs. Outputs from this function are 'AU' and 'AV'");
    AU = pdeip_out_like(&plhs[0], prhs[4]);
    AV = pdeip_out_like(&plhs[1], prhs[4]);
    pdeip_check(pdeip_oflow_lhs_llin4(p[0], p[1], p[2], p[3], p[4], p[5], p[6], p[7], p[8], p[9], p[10],
                                      pdeip_rows(prhs[0]), pdeip_cols(prhs[0]), pdeip_frames(prhs[4]), AU, AV));
}
