/* [U V] = FlowEminAD_llin_2D_v10_gpu(Iin, channels, fstTerm, sndTerm, params, Us, Vs)
 * The whole anisotropic-diffusion flow driver (matlab/optical_flow/FlowEminAD_llin_2D_v10.m:52-383) in one call, resident on the
 * device (pdeip_flow_ad_llin, csrc/pdeip_drivers.hip).  Numeric arguments only; the wrapper matlab/FlowEminAD_llin_2D_v10_gpu.m
 * keeps the reference driver's argument list and calls this:
 *   Iin      single [rows x cols x 2*channels] = cat(3, frame0, frame1), 0..255
 *   channels, fstTerm (1 rgb, 2 grad), sndTerm (0 none, 1 rgb, 3 gradmag)     single scalars
 *   params   single vector [alpha omega gammaS firstLoop secondLoop iter b1 b2 scl_factor solver scales quantile diffusion]
 *            (<= 0: the default; diffusion 0 'image', 1 'flow')
 *   Us, Vs   param.Us / param.Vs: double [rows x cols], or omitted / empty */
#include "pdeip_mex_util.h"

void mexFunction(int nlhs, mxArray *plhs[], int nrhs, const mxArray *prhs[])
{
    static const char *who = "FlowEminAD_llin_2D_v10_gpu";
    pdeip_driver_params p;
    const double *Us = NULL, *Vs = NULL;
    const float *I, *pv;
    mwSize dims[2];
    int rows, cols, C;
    if (nrhs < 5 || nrhs > 7) mexErrMsgTxt("FlowEminAD_llin_2D_v10_gpu parameter error: wrong number of input parameters!");
    if (nlhs < 2) mexErrMsgTxt("FlowEminAD_llin_2D_v10_gpu insufficient number of outputs. Outputs from this function are 'U' and 'V'");
    I = pdeip_single(prhs[0], who, "Iin");
    rows = pdeip_rows(prhs[0]);
    cols = pdeip_cols(prhs[0]);
    C = (int)pdeip_scalar(prhs[1], who, "channels");
    if (C < 1 || pdeip_frames(prhs[0]) != 2 * C) mexErrMsgTxt("FlowEminAD_llin_2D_v10_gpu: Iin must have 2*channels frames");
    pv = pdeip_single(prhs[4], who, "params");
    if (mxGetNumberOfElements(prhs[4]) != 13) mexErrMsgTxt("FlowEminAD_llin_2D_v10_gpu: 'params' must have 13 elements");
    p.alpha = pv[0]; p.omega = pv[1]; p.gammaS = pv[2]; p.firstLoop = (int)pv[3]; p.secondLoop = (int)pv[4]; p.iter = (int)pv[5];
    p.b1 = pv[6]; p.b2 = pv[7]; p.scl_factor = pv[8]; p.solver = (int)pv[9]; p.scales = (int)pv[10];
    if (nrhs > 5) Us = pdeip_double_plane(prhs[5], rows, cols, who, "Us");
    if (nrhs > 6) Vs = pdeip_double_plane(prhs[6], rows, cols, who, "Vs");
    dims[0] = (mwSize)rows;
    dims[1] = (mwSize)cols;
    plhs[0] = mxCreateNumericArray(2, dims, mxSINGLE_CLASS, mxREAL);
    plhs[1] = mxCreateNumericArray(2, dims, mxSINGLE_CLASS, mxREAL);
    pdeip_check(pdeip_flow_ad_llin(I, rows, cols, C, (int)pdeip_scalar(prhs[2], who, "fstTerm"), (int)pdeip_scalar(prhs[3], who, "sndTerm"), &p, (double)pv[11],
                                   pv[12] > 0.5f ? 1 : 0, Us, Vs, (float *)mxGetData(plhs[0]), (float *)mxGetData(plhs[1])));
}
