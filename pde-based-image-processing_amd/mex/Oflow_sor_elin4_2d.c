/* [U,V(,RU,RV)] = Oflow_sor_elin4_2d(U,V,M,Cu,Cv,Du,Dv,wW,wN,wE,wS,iter,omega,solver)
 * Drop-in for mex/source/Oflow_sor_elin4_2d.c (reference gateway :64-352). */
#include "pdeip_mex_util.h"

void mexFunction(int nlhs, mxArray *plhs[], int nrhs, const mxArray *prhs[])
{
    static const char *who = "Oflow_sor_elin4_2d";
    static const char *names[11] = {"U_in", "V_in", "M", "Cu", "Cv", "Du", "Dv", "wW", "wN", "wE", "wS"};
    const float *p[11];
    float *Uo, *Vo, *RU = NULL, *RV = NULL;
    int k;
    if (nrhs != 14) mexErrMsgTxt("Oflow_sor_elin4_2d parameter error: wrong number of input parameters!");
    for (k = 0; k < 11; k++) p[k] = pdeip_single(prhs[k], who, names[k]);
    if (nlhs < 2) mexErrMsgTxt("Oflow_sor_elin4_2d insufficient number of outputs. Outputs from this function are 'U' and 'V'");
    Uo = pdeip_out_like(&plhs[0], prhs[0]);
    Vo = pdeip_out_like(&plhs[1], prhs[1]);
    if (nlhs >= 4) { /* residual outputs take M's dimensions (:309-325) */
        RU = pdeip_out_like(&plhs[2], prhs[2]);
        RV = pdeip_out_like(&plhs[3], prhs[2]);
    }
    pdeip_check(pdeip_oflow_sor_elin4(p[0], p[1], p[2], p[3], p[4], p[5], p[6], p[7], p[8], p[9], p[10],
                                      pdeip_rows(prhs[0]), pdeip_cols(prhs[0]), pdeip_frames(prhs[2]),
                                      (int)pdeip_scalar(prhs[11], who, "iter"), pdeip_scalar(prhs[12], who, "omega"),
                                      (int)pdeip_scalar(prhs[13], who, "solver"), Uo, Vo, RU, RV));
}
