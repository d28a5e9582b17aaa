/*
 * pdeip_mex_util.h -- shared unpacking helpers of the drop-in MEX gateways.
 *
 * Each gateway source in this directory has the NAME and the CALL SIGNATURE of the reference
 * gateway it replaces (mex/source/<name>.c, built by mex/buildAll.m:5-25) and does nothing but
 * unpack mxArrays, allocate outputs and forward to libpdeip.so (include/pdeip.h).  Build, e.g.:
 *     mex -I<repo>/include Oflow_sor_elin4_2d.c -L<repo>/pde-based-image-processing_amd -lpdeip -outdir ./build
 * The checks mirror the reference gateways' (type = single, argument count, number of outputs).
 */
#ifndef PDEIP_MEX_UTIL_H
#define PDEIP_MEX_UTIL_H

#include <string.h>

#include "mex.h"
#include "pdeip.h"

/* mxIsSingle check + data pointer (e.g. Oflow_sor_elin4_2d.c:117-123) */
static const float *pdeip_single(const mxArray *a, const char *who, const char *name)
{
    char msg[256];
    if (!mxIsSingle(a) || mxIsComplex(a)) {
        strcpy(msg, who);
        strcat(msg, ": '");
        strcat(msg, name);
        strcat(msg, "' must be a noncomplex single-valued matrix.");
        mexErrMsgTxt(msg);
    }
    return (const float *)mxGetData(a);
}

/* 1x1 single scalar (Oflow_sor_elin4_2d.c:260-283) */
static float pdeip_scalar(const mxArray *a, const char *who, const char *name)
{
    char msg[256];
    if (!mxIsSingle(a) || mxGetNumberOfElements(a) != 1) {
        strcpy(msg, who);
        strcat(msg, ": '");
        strcat(msg, name);
        strcat(msg, "' must be a noncomplex, single-type scalar");
        mexErrMsgTxt(msg);
    }
    return *(const float *)mxGetData(a);
}

static int pdeip_rows(const mxArray *a) { return (int)mxGetDimensions(a)[0]; }
static int pdeip_cols(const mxArray *a) { return (int)mxGetDimensions(a)[1]; }
static int pdeip_frames(const mxArray *a) { return mxGetNumberOfDimensions(a) > 2 ? (int)mxGetDimensions(a)[2] : 1; }

/* zero-filled single output with the dimensions of `like` (mxCreateNumericArray, :299-307) */
static float *pdeip_out_like(mxArray **slot, const mxArray *like)
{
    *slot = mxCreateNumericArray(mxGetNumberOfDimensions(like), mxGetDimensions(like), mxSINGLE_CLASS, mxREAL);
    return (float *)mxGetData(*slot);
}

/* an optional double [rows x cols] input (param.Us / param.Vs of the drivers): NULL when the argument is empty */
static const double *pdeip_double_plane(const mxArray *a, int rows, int cols, const char *who, const char *name)
{
    char msg[256];
    if (mxGetNumberOfElements(a) == 0) return NULL;
    if (!mxIsDouble(a) || mxIsComplex(a) || pdeip_rows(a) != rows || pdeip_cols(a) != cols) {
        strcpy(msg, who);
        strcat(msg, ": '");
        strcat(msg, name);
        strcat(msg, "' must be a real double matrix of the image's size (or empty).");
        mexErrMsgTxt(msg);
    }
    return (const double *)mxGetData(a);
}

static void pdeip_check(int rc)
{
    if (rc != PDEIP_OK) mexErrMsgTxt(pdeip_last_error());
}

#endif
