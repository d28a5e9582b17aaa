/* Minimal mock MEX runtime: arrays, zero-filled creation, mexErrMsgTxt as a longjmp back to mock_call(). */
#include <setjmp.h>
#include <stdlib.h>
#include <string.h>

#include "mex.h"

struct mxArray_tag {
    mwSize ndim;
    mwSize dims[4];
    mxClassID classid;
    void *data;
};

static jmp_buf g_jmp;
static char g_err[512];

static size_t elsize(mxClassID c) { return c == mxSINGLE_CLASS ? 4 : 8; }

bool mxIsSingle(const mxArray *a) { return a->classid == mxSINGLE_CLASS; }
bool mxIsDouble(const mxArray *a) { return a->classid == mxDOUBLE_CLASS; }
bool mxIsComplex(const mxArray *a) { (void)a; return false; }
void *mxGetData(const mxArray *a) { return a->data; }
size_t mxGetNumberOfElements(const mxArray *a)
{
    size_t n = 1;
    for (mwSize k = 0; k < a->ndim; k++) n *= a->dims[k];
    return n;
}
const mwSize *mxGetDimensions(const mxArray *a) { return a->dims; }
mwSize mxGetNumberOfDimensions(const mxArray *a) { return a->ndim; }
mxArray *mxCreateNumericArray(mwSize ndim, const mwSize *dims, mxClassID classid, mxComplexity flag)
{
    (void)flag;
    mxArray *a = (mxArray *)calloc(1, sizeof *a);
    a->ndim = ndim < 2 ? 2 : ndim;
    a->dims[0] = a->dims[1] = a->dims[2] = a->dims[3] = 1;
    for (mwSize k = 0; k < ndim && k < 4; k++) a->dims[k] = dims[k];
    a->classid = classid;
    a->data = calloc(mxGetNumberOfElements(a) ? mxGetNumberOfElements(a) : 1, elsize(classid)); /* zero-filled */
    return a;
}
void mexErrMsgTxt(const char *msg)
{
    strncpy(g_err, msg, sizeof g_err - 1);
    longjmp(g_jmp, 1);
}

/* ---- driver API used by tests/test_mex_stubs.py ---- */
mxArray *mock_make(int ndim, const long *dims, int classid, const void *data)
{
    mwSize d[4] = {1, 1, 1, 1};
    for (int k = 0; k < ndim && k < 4; k++) d[k] = (mwSize)dims[k];
    mxArray *a = mxCreateNumericArray((mwSize)ndim, d, (mxClassID)classid, mxREAL);
    memcpy(a->data, data, mxGetNumberOfElements(a) * elsize((mxClassID)classid));
    return a;
}
void mock_free(mxArray *a)
{
    if (a) {
        free(a->data);
        free(a);
    }
}
int mock_ndim(const mxArray *a) { return (int)a->ndim; }
long mock_dim(const mxArray *a, int k) { return (long)a->dims[k]; }
void *mock_data(const mxArray *a) { return a->data; }
const char *mock_last_error(void) { return g_err; }
int mock_call(int nlhs, mxArray **plhs, int nrhs, const mxArray **prhs)
{
    g_err[0] = 0;
    if (setjmp(g_jmp)) return 1;
    mexFunction(nlhs, plhs, nrhs, prhs);
    return 0;
}
