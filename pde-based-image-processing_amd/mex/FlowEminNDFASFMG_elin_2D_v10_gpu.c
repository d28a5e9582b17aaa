/* [U V] = FlowEminNDFASFMG_elin_2D_v10_gpu(Iin, channels, params)
 * The whole FAS full-multigrid flow driver (matlab/optical_flow/FlowEminNDFASFMG_elin_2D_v10.m:53-273) in one call, resident on the
 * device (pdeip_flow_fas_fmg_elin, csrc/pdeip_drivers.hip).  Numeric arguments only; the wrapper
 * matlab/FlowEminNDFASFMG_elin_2D_v10_gpu.m keeps the reference driver's argument list and calls this:
 *   Iin      single [rows x cols x 2*channels] = cat(3, frame0, frame1), 0..255
 *   channels single scalar
 *   params   single vector [alpha omega firstLoop iter b1 b2 scl_factor solver cycle_index scales], <= 0: the driver's default */
#include "pdeip_mex_util.h"

void mexFunction(int nlhs, mxArray *plhs[], int nrhs, const mxArray *prhs[])
{
    static const char *who = "FlowEminNDFASFMG_elin_2D_v10_gpu";
    pdeip_fas_params p;
    const float *I, *pv;
    mwSize dims[2];
    int rows, cols, C;
    if (nrhs != 3) mexErrMsgTxt("FlowEminNDFASFMG_elin_2D_v10_gpu parameter error: wrong number of input parameters!");
    if (nlhs < 2) mexErrMsgTxt("FlowEminNDFASFMG_elin_2D_v10_gpu insufficient number of outputs. Outputs from this function are 'U' and 'V'");
    I = pdeip_single(prhs[0], who, "Iin");
    rows = pdeip_rows(prhs[0]);
    cols = pdeip_cols(prhs[0]);
    C = (int)pdeip_scalar(prhs[1], who, "channels");
    if (C < 1 || pdeip_frames(prhs[0]) != 2 * C) mexErrMsgTxt("FlowEminNDFASFMG_elin_2D_v10_gpu: Iin must have 2*channels frames");
    pv = pdeip_single(prhs[2], who, "params");
    if (mxGetNumberOfElements(prhs[2]) != 10) mexErrMsgTxt("FlowEminNDFASFMG_elin_2D_v10_gpu: 'params' must have 10 elements");
    p.alpha = pv[0]; p.omega = pv[1]; p.firstLoop = (int)pv[2]; p.iter = (int)pv[3]; p.b1 = pv[4]; p.b2 = pv[5]; p.scl_factor = pv[6];
    p.solver = (int)pv[7]; p.cycle_index = (int)pv[8]; p.scales = (int)pv[9];
    dims[0] = (mwSize)rows;
    dims[1] = (mwSize)cols;
    plhs[0] = mxCreateNumericArray(2, dims, mxSINGLE_CLASS, mxREAL);
    plhs[1] = mxCreateNumericArray(2, dims, mxSINGLE_CLASS, mxREAL);
    pdeip_check(pdeip_flow_fas_fmg_elin(I, rows, cols, C, &p, (float *)mxGetData(plhs[0]), (float *)mxGetData(plhs[1])));
}
