/* Iout = TVdenoise4_gpu(I_in, params)
 * The whole TV denoising driver (matlab/denoising/TVdenoise4.m) in one call, resident on the device (pdeip_tvdenoise4,
 * csrc/pdeip_drivers.hip).  Numeric arguments only; the wrapper matlab/TVdenoise4_gpu.m keeps the reference driver's argument
 * list and calls this:
 *   I_in     single [rows x cols x frames], 0..1
 *   params   single vector [alpha omega outer_iter inner_iter solver scl scl_factor], <= 0: the driver's default */
#include "pdeip_mex_util.h"

void mexFunction(int nlhs, mxArray *plhs[], int nrhs, const mxArray *prhs[])
{
    static const char *who = "TVdenoise4_gpu";
    pdeip_tv_params p;
    const float *I, *pv;
    mwSize dims[3];
    int rows, cols, F;
    if (nrhs != 2) mexErrMsgTxt("TVdenoise4_gpu parameter error: wrong number of input parameters!");
    if (nlhs < 1) mexErrMsgTxt("TVdenoise4_gpu insufficient number of outputs. Output from this function is 'Iout'");
    I = pdeip_single(prhs[0], who, "I_in");
    rows = pdeip_rows(prhs[0]);
    cols = pdeip_cols(prhs[0]);
    F = pdeip_frames(prhs[0]);
    pv = pdeip_single(prhs[1], who, "params");
    if (mxGetNumberOfElements(prhs[1]) != 7) mexErrMsgTxt("TVdenoise4_gpu: 'params' must have 7 elements");
    p.alpha = pv[0]; p.omega = pv[1]; p.outer_iter = (int)pv[2]; p.inner_iter = (int)pv[3]; p.solver = (int)pv[4]; p.scl = pv[5]; p.scl_factor = pv[6];
    dims[0] = (mwSize)rows;
    dims[1] = (mwSize)cols;
    dims[2] = (mwSize)F;
    plhs[0] = mxCreateNumericArray(F > 1 ? 3 : 2, dims, mxSINGLE_CLASS, mxREAL);
    pdeip_check(pdeip_tvdenoise4(I, rows, cols, F, &p, (float *)mxGetData(plhs[0])));
}
