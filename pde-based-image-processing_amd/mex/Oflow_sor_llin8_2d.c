/* [dU,dV(,RU,RV)] = Oflow_sor_llin8_2d(U,V,dU,dV,M,Cu,Cv,Du,Dv,wW,wNW,wN,wNE,wE,wSE,wS,wSW,iter,omega,solver)
 * Drop-in for mex/source/Oflow_sor_llin8_2d.c (reference gateway :71-489). */
#include "pdeip_mex_util.h"

void mexFunction(int nlhs, mxArray *plhs[], int nrhs, const mxArray *prhs[])
{
    static const char *who = "Oflow_sor_llin8_2d";
    static const char *names[17] = {"U_in", "V_in", "dU_in", "dV_in", "M", "Cu", "Cv", "Du", "Dv",
                                    "wW", "wNW", "wN", "wNE", "wE", "wSE", "wS", "wSW"};
    const float *p[17];
    float *o0, *o1, *RU = NULL, *RV = NULL;
    int k;
    if (nrhs != 20) mexErrMsgTxt("Oflow_sor_llin8_2d parameter error: wrong number of input parameters!");
    for (k = 0; k < 17; k++) p[k] = pdeip_single(prhs[k], who, names[k]);
    if (nlhs < 2) mexErrMsgTxt("Oflow_sor_llin8_2d insufficient number of outputs. Outputs from this function are 'dU' and 'dV'");
    o0 = pdeip_out_like(&plhs[0], prhs[2]);
    o1 = pdeip_out_like(&plhs[1], prhs[3]);
    if (nlhs >= 4) {
        RU = pdeip_out_like(&plhs[2], prhs[4]);
        RV = pdeip_out_like(&plhs[3], prhs[4]);
    }
    pdeip_check(pdeip_oflow_sor_llin8(p[0], p[1], p[2], p[3], p[4], p[5], p[6], p[7], p[8], p[9], p[10], p[11], p[12],
                                      p[13], p[14], p[15], p[16], pdeip_rows(prhs[0]), pdeip_cols(prhs[0]),
                                      pdeip_frames(prhs[4]), (int)pdeip_scalar(prhs[17], who, "iter"),
                                      pdeip_scalar(prhs[18], who, "omega"), (int)pdeip_scalar(prhs[19], who, "solver"),
                                      o0, o1, RU, RV));
}
