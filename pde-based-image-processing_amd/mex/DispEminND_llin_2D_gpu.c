/* U = DispEminND_llin_2D_gpu(Il, Ir, fstTerm, sndTerm, params, Us)
 * The whole stereo disparity driver (matlab/disparity/DispEminND_llin_2D.m:51-316) in one call, resident on the device
 * (pdeip_disp_nd_llin, csrc/pdeip_drivers.hip).  Numeric arguments only; the wrapper matlab/DispEminND_llin_2D_gpu.m keeps the
 * reference driver's argument list and calls this:
 *   Il, Ir   single [rows x cols x channels], 0..255
 *   fstTerm (1 rgb, 2 grad), sndTerm (0 none, 1 rgb, 3 gradmag)     single scalars
 *   params   single vector [alpha omega gammaS firstLoop secondLoop iter b1 b2 scl_factor solver scales], <= 0: the default
 *   Us       param.Us: double [rows x cols], or omitted / empty */
#include "pdeip_mex_util.h"

void mexFunction(int nlhs, mxArray *plhs[], int nrhs, const mxArray *prhs[])
{
    static const char *who = "DispEminND_llin_2D_gpu";
    pdeip_driver_params p;
    const double *Us = NULL;
    const float *Il, *Ir, *pv;
    mwSize dims[2];
    int rows, cols, C;
    if (nrhs < 5 || nrhs > 6) mexErrMsgTxt("DispEminND_llin_2D_gpu parameter error: wrong number of input parameters!");
    if (nlhs < 1) mexErrMsgTxt("DispEminND_llin_2D_gpu insufficient number of outputs. Output from this function is 'U'");
    Il = pdeip_single(prhs[0], who, "Il");
    Ir = pdeip_single(prhs[1], who, "Ir");
    rows = pdeip_rows(prhs[0]);
    cols = pdeip_cols(prhs[0]);
    C = pdeip_frames(prhs[0]);
    if (pdeip_rows(prhs[1]) != rows || pdeip_cols(prhs[1]) != cols || pdeip_frames(prhs[1]) != C) mexErrMsgTxt("DispEminND_llin_2D_gpu: Il and Ir must have the same size");
    pv = pdeip_single(prhs[4], who, "params");
    if (mxGetNumberOfElements(prhs[4]) != 11) mexErrMsgTxt("DispEminND_llin_2D_gpu: 'params' must have 11 elements");
    p.alpha = pv[0]; p.omega = pv[1]; p.gammaS = pv[2]; p.firstLoop = (int)pv[3]; p.secondLoop = (int)pv[4]; p.iter = (int)pv[5];
    p.b1 = pv[6]; p.b2 = pv[7]; p.scl_factor = pv[8]; p.solver = (int)pv[9]; p.scales = (int)pv[10];
    if (nrhs > 5) Us = pdeip_double_plane(prhs[5], rows, cols, who, "Us");
    dims[0] = (mwSize)rows;
    dims[1] = (mwSize)cols;
    plhs[0] = mxCreateNumericArray(2, dims, mxSINGLE_CLASS, mxREAL);
    pdeip_check(pdeip_disp_nd_llin(Il, Ir, rows, cols, C, (int)pdeip_scalar(prhs[2], who, "fstTerm"), (int)pdeip_scalar(prhs[3], who, "sndTerm"), &p, Us,
                                   (float *)mxGetData(plhs[0])));
}
