/* [dU(,RU)] = Disp_sor_llin4_2d(U,dU,Cu,Du,wW,wN,wE,wS,iter,omega,solver)
 * Drop-in for mex/source/Disp_sor_llin4_2d.c (reference gateway :59-282). */
#include "pdeip_mex_util.h"

void mexFunction(int nlhs, mxArray *plhs[], int nrhs, const mxArray *prhs[])
{
    static const char *who = "Disp_sor_llin4_2d";
    static const char *names[8] = {"U_in", "dU_in", "Cu", "Du", "wW", "wN", "wE", "wS"};
    const float *p[8];
    float *dUo, *RU = NULL;
    int k;
    if (nrhs != 11) mexErrMsgTxt("Disp_sor_llin4_2d parameter error: wrong number of input parameters!");
    for (k = 0; k < 8; k++) p[k] = pdeip_single(prhs[k], who, names[k]);
    if (nlhs < 1) mexErrMsgTxt("Disp_sor_llin4_2d insufficient number of outputs. Outputs from this function is 'dU'");
    dUo = pdeip_out_like(&plhs[0], prhs[1]);
    if (nlhs >= 2) RU = pdeip_out_like(&plhs[1], prhs[0]);
    pdeip_check(pdeip_disp_sor_llin4(p[0], p[1], p[2], p[3], p[4], p[5], p[6], p[7], pdeip_rows(prhs[0]), pdeip_cols(prhs[0]),
                                     (int)pdeip_scalar(prhs[8], who, "iter"), pdeip_scalar(prhs[9], who, "omega"),
                                     (int)pdeip_scalar(prhs[10], who, "solver"), dUo, RU));
}
