/* U = DispEminND_llin_sym_2D_gpu(Il, Ir, params)
 * The whole symmetric stereo driver (matlab/disparity/DispEminND_llin_sym_2D.m:51-275) in one call, resident on the device
 * (pdeip_disp_nd_llin_sym, csrc/pdeip_drivers.hip).  Numeric arguments only; the wrapper matlab/DispEminND_llin_sym_2D_gpu.m keeps
 * the reference driver's argument list and calls this:
 *   Il, Ir   single [rows x cols x channels]
 *   params   single vector [alpha beta omega firstLoop secondLoop iter b1 b2 scl_factor solver], <= 0: the driver's default
 * U: single [rows x cols x 2] */
#include "pdeip_mex_util.h"

void mexFunction(int nlhs, mxArray *plhs[], int nrhs, const mxArray *prhs[])
{
    static const char *who = "DispEminND_llin_sym_2D_gpu";
    pdeip_sym_params p;
    const float *Il, *Ir, *pv;
    mwSize dims[3];
    int rows, cols, C;
    if (nrhs != 3) mexErrMsgTxt("DispEminND_llin_sym_2D_gpu parameter error: wrong number of input parameters!");
    if (nlhs < 1) mexErrMsgTxt("DispEminND_llin_sym_2D_gpu insufficient number of outputs. Output from this function is 'U'");
    Il = pdeip_single(prhs[0], who, "Il");
    Ir = pdeip_single(prhs[1], who, "Ir");
    rows = pdeip_rows(prhs[0]);
    cols = pdeip_cols(prhs[0]);
    C = pdeip_frames(prhs[0]);
    if (pdeip_rows(prhs[1]) != rows || pdeip_cols(prhs[1]) != cols || pdeip_frames(prhs[1]) != C) mexErrMsgTxt("DispEminND_llin_sym_2D_gpu: Il and Ir must have the same size");
    pv = pdeip_single(prhs[2], who, "params");
    if (mxGetNumberOfElements(prhs[2]) != 10) mexErrMsgTxt("DispEminND_llin_sym_2D_gpu: 'params' must have 10 elements");
    p.alpha = pv[0]; p.beta = pv[1]; p.omega = pv[2]; p.firstLoop = (int)pv[3]; p.secondLoop = (int)pv[4]; p.iter = (int)pv[5];
    p.b1 = pv[6]; p.b2 = pv[7]; p.scl_factor = pv[8]; p.solver = (int)pv[9];
    dims[0] = (mwSize)rows;
    dims[1] = (mwSize)cols;
    dims[2] = 2;
    plhs[0] = mxCreateNumericArray(3, dims, mxSINGLE_CLASS, mxREAL);
    pdeip_check(pdeip_disp_nd_llin_sym(Il, Ir, rows, cols, C, &p, (float *)mxGetData(plhs[0])));
}
