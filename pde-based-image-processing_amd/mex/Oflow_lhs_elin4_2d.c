/* [AU,AV] = Oflow_lhs_elin4_2d(U,V,M,Du,Dv,wW,wN,wE,wS)
 * Drop-in for mex/source/Oflow_lhs_elin4_2d.c (reference gateway :56-231). */
#include "pdeip_mex_util.h"

void mexFunction(int nlhs, mxArray *plhs[], int nrhs, const mxArray *prhs[])
{
    static const char *who = "Oflow_lhs_elin4_2d";
    static const char *names[9] = {"U_in", "V_in", "M", "Du", "Dv", "wW", "wN", "wE", "wS"};
    const float *p[9];
    float *AU, *AV;
    int k;
    if (nrhs != 9) mexErrMsgTxt("Oflow_lhs_elin4_2d parameter error: wrong number of input parameters!");
    for (k = 0; k < 9; k++) p[k] = pdeip_single(prhs[k], who, names[k]);
    if (nlhs < 2) mexErrMsgTxt("Oflow_lhs_elin4_2d insufficient number of outputs. Outputs from this function are 'AU' and 'AV'");
    AU = pdeip_out_like(&plhs[0], prhs[2]);
    AV = pdeip_out_like(&plhs[1], prhs[2]);
    pdeip_check(pdeip_oflow_lhs_elin4(p[0], p[1], p[2], p[3], p[4], p[5], p[6], p[7], p[8], pdeip_rows(prhs[0]),
                                      pdeip_cols(prhs[0]), pdeip_frames(prhs[2]), AU, AV));
}
