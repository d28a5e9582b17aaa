/* [dU0 dU1] = Disp_sor_llin_sym4_2d(U0,dU0,Cu0,Du0,wW0,wN0,wE0,wS0, U1,dU1,Cu1,Du1,wW1,wN1,wE1,wS1, iter,omega,solver)
 * Drop-in for mex/source/Disp_sor_llin_sym4_2d.c (reference gateway :82-440). */
#include "pdeip_mex_util.h"

void mexFunction(int nlhs, mxArray *plhs[], int nrhs, const mxArray *prhs[])
{
    static const char *who = "Disp_sor_llin_sym4_2d";
    static const char *names[16] = {"U_in0", "dU_in0", "Cu0", "Du0", "wW0", "wN0", "wE0", "wS0",
                                    "U_in1", "dU_in1", "Cu1", "Du1", "wW1", "wN1", "wE1", "wS1"};
    const float *p[16];
    float *o0, *o1;
    int k;
    if (nrhs != 19) mexErrMsgTxt("Disp_sor_llin_sym4_2d parameter error: wrong number of input parameters!");
    for (k = 0; k < 16; k++) p[k] = pdeip_single(prhs[k], who, names[k]);
    if (nlhs < 2) mexErrMsgTxt("Disp_sor_llin_sym4_2d insufficient number of outputs. Outputs from this function are 'dU0' and 'dU1'");
    o0 = pdeip_out_like(&plhs[0], prhs[1]);
    o1 = pdeip_out_like(&plhs[1], prhs[9]);
    pdeip_check(pdeip_disp_sor_llin_sym4(p[0], p[1], p[2], p[3], p[4], p[5], p[6], p[7], p[8], p[9], p[10], p[11], p[12], p[13],
                                         p[14], p[15], pdeip_rows(prhs[0]), pdeip_cols(prhs[0]),
                                         (int)pdeip_scalar(prhs[16], who, "iter"), pdeip_scalar(prhs[17], who, "omega"),
                                         (int)pdeip_scalar(prhs[18], who, "solver"), o0, o1));
}
