// pdeip_pyr.hpp -- the image pyramid of the MATLAB drivers on the device: imresize (triangle / cubic kernel, antialiased when
// shrinking) and imfilter(., fspecial('gaussian', ...), 'replicate').  There is no IPT to compare with: pyramid.py states
// OUR definition of both (tap lists summed left to right in double, rounded to single once) and these kernels walk the same
// lists, bit for bit (tests/test_gpu_drivers.py).
#pragma once
#include <hip/hip_runtime.h>

#include "pdeip_pointwise.hpp"

namespace pdeip {

constexpr int PYR_TMAX = 16; // taps per axis: ceil(2 * support / scale) + 2

struct PyrAxis {
    double scale, stretch, width;
    int T, cubic;
};

__device__ __forceinline__ double pyr_weight(const PyrAxis &A, double x, int idx)
{
    const double a = fabs(((double)idx - x) / A.stretch);
    if (!A.cubic) return fmax(0.0, 1.0 - a);
    const double a2 = a * a, a3 = a2 * a;
    if (a <= 1.0) return (1.5 * a3 - 2.5 * a2) + 1.0;
    if (a <= 2.0) return ((-0.5 * a3 + 2.5 * a2) - 4.0 * a) + 2.0;
    return 0.0;
}

// pyramid.resize: rows first, then columns
__global__ void k_pyr_resize(float *out, const float *in, PyrAxis R, PyrAxis C, int nrows_in, int ncols_in, int nrows, int ncols)
{
    PDEIP_PIXEL_INDEX();
    const float *src = in + (size_t)blockIdx.z * nrows_in * ncols_in;
    const double xr = ((double)i + 0.5) / R.scale - 0.5, xc = ((double)j + 0.5) / C.scale - 0.5;
    const int fr = (int)floor(xr - R.width), fc = (int)floor(xc - C.width);
    double rw[PYR_TMAX], rtot = 0.0, ctot = 0.0;
    for (int k = 0; k < R.T; ++k) {
        rw[k] = pyr_weight(R, xr, fr + k);
        rtot = k ? rtot + rw[k] : rw[k];
    }
    for (int k = 0; k < C.T; ++k) {
        const double w = pyr_weight(C, xc, fc + k);
        ctot = k ? ctot + w : w;
    }
    double acc = 0.0;
    for (int kc = 0; kc < C.T; ++kc) {
        const float *col = src + (size_t)min(max(fc + kc, 0), ncols_in - 1) * nrows_in;
        double t = 0.0;
        for (int kr = 0; kr < R.T; ++kr) {
            const double v = (rw[kr] / rtot) * (double)col[min(max(fr + kr, 0), nrows_in - 1)];
            t = kr ? t + v : v;
        }
        const double v = (pyr_weight(C, xc, fc + kc) / ctot) * t;
        acc = kc ? acc + v : v;
    }
    out[(size_t)blockIdx.z * nrows * ncols + pos] = (float)acc;
}

struct PyrMask {
    double g[49]; // row-major [size][size], size <= 7
    int size;
};

// pyramid.smooth: imfilter(I, G, 'replicate'), double accumulation in mask order
__global__ void k_pyr_smooth(float *out, const float *in, PyrMask M, int nrows, int ncols)
{
    PDEIP_PIXEL_INDEX();
    const size_t fo = (size_t)blockIdx.z * nrows * ncols;
    const int r = M.size / 2;
    double acc = 0.0;
    for (int a = 0; a < M.size; ++a)
        for (int b = 0; b < M.size; ++b) {
            const int ii = min(max(i + a - r, 0), nrows - 1), jj = min(max(j + b - r, 0), ncols - 1);
            acc += M.g[a * M.size + b] * (double)in[fo + (size_t)jj * nrows + ii];
        }
    out[fo + pos] = (float)acc;
}

} // namespace pdeip
