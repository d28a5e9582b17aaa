// pdeip_tv.hpp -- the MATLAB-side stages of one lagged-diffusivity iteration of the TV denoiser
// (matlab/denoising/TVdenoise8.m:80-87 and its ADdiffWeights, :119-231), as device kernels, so that the
// whole outer loop around PDEsolver8 stays in HBM.
//
// Like pdeip_flow.hpp these restate MATLAB array code (no reference build exists to compare with):
// ADdiffWeights works in double on an Alvarez 3x3 derivative ('conv' = true convolution, 'replicate'
// borders), picks per pixel the frame with the largest gradient, takes lambda as the median of the
// non-zero squared gradient norms (sort + index round(numel*quantile + eps)), builds the anisotropic
// tensor and the eight weights with circshift wrap-around and zeroed outer rows/columns.  TVdenoise8 then
// forms PsiData, TRACE and B in single.  oracle/matlab_side.py holds the numpy statement the tests compare
// with, bit for bit.
#pragma once
#include <hip/hip_runtime.h>

#include "pdeip_models.hpp"
#include "pdeip_pointwise.hpp"

namespace pdeip {

// Ddx, Ddy of the frame with the largest squared gradient norm (first frame on ties, like MATLAB's max),
// and that norm.  Kernel elements in MATLAB order of the rotated kernel, zero taps skipped.
__global__ void k_tv_gradient(double *gx, double *gy, double *nrm, const float *D, int nrows, int ncols, int nframes)
{
    PDEIP_PIXEL_INDEX();
    const size_t n = (size_t)nrows * ncols;
    const double s = 4.0 + sqrt(8.0);
    const double k1 = 1.0 / s, k2 = sqrt(2.0) / s; // [1 sqrt(2) 1] ./ (4+sqrt(8))
    auto ci = [&](int v) { return v < 0 ? 0 : (v > nrows - 1 ? nrows - 1 : v); };
    auto cj = [&](int v) { return v < 0 ? 0 : (v > ncols - 1 ? ncols - 1 : v); };
    double bx = 0.0, by = 0.0, bn = -1.0;
    for (int f = 0; f < nframes; ++f) {
        const float *P = D + (size_t)f * n;
        auto at = [&](int ii, int jj) { return (double)P[(size_t)cj(jj) * nrows + ci(ii)]; };
        // conv with O_dx = [1 0 -1; sqrt2 0 -sqrt2; 1 0 -1]/s: sum over kernel (u,v) of K(u,v) * D(i-u, j-v), rows of K first
        double dx = k1 * at(i + 1, j + 1);
        dx = dx + (-k1) * at(i + 1, j - 1);
        dx = dx + k2 * at(i, j + 1);
        dx = dx + (-k2) * at(i, j - 1);
        dx = dx + k1 * at(i - 1, j + 1);
        dx = dx + (-k1) * at(i - 1, j - 1);
        // O_dy = [1 sqrt2 1; 0 0 0; -1 -sqrt2 -1]/s
        double dy = k1 * at(i + 1, j + 1);
        dy = dy + k2 * at(i + 1, j);
        dy = dy + k1 * at(i + 1, j - 1);
        dy = dy + (-k1) * at(i - 1, j + 1);
        dy = dy + (-k2) * at(i - 1, j);
        dy = dy + (-k1) * at(i - 1, j - 1);
        const double nn = dx * dx + dy * dy;
        if (nn > bn) { // strictly greater: the first maximal frame wins
            bn = nn;
            bx = dx;
            by = dy;
        }
    }
    gx[pos] = bx;
    gy[pos] = by;
    nrm[pos] = bx * bx + by * by;
}

// lambda = sorted_nonzero(round(numel * 0.5 + eps)) from the ascending array `sorted` (zeros first); 1 if all are zero.
// quantile >= 0: the flow driver's form sorted_nonzero(round(numel * quantile)) (FlowEminAD_llin_2D_v10.m:461-467).
__global__ void k_tv_lambda(double *lambda, const double *sorted, size_t n, double quantile)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    size_t lo = 0, hi = n; // first index with sorted[idx] > 0 (norms are >= 0; NaN sorts last and counts as non-zero)
    while (lo < hi) {
        const size_t mid = lo + (hi - lo) / 2;
        if (sorted[mid] > 0.0 || sorted[mid] != sorted[mid]) hi = mid;
        else lo = mid + 1;
    }
    const size_t cnt = n - lo;
    if (cnt == 0) {
        *lambda = 1.0;
        return;
    }
    size_t idx = (cnt + 1) / 2; // round(cnt*0.5 + eps), 1-based
    if (quantile >= 0.0) {
        idx = (size_t)floor((double)cnt * quantile + 0.5);
        idx = idx < 1 ? 1 : (idx > cnt ? cnt : idx);
    }
    *lambda = sorted[lo + idx - 1];
}

// The eight weights (times alpha, as single), TRACE and B of one outer iteration (TVdenoise8.m:82-86).
__global__ void k_tv_assemble(float *TRACE, float *B, float *aW, float *aNW, float *aN, float *aNE, float *aE, float *aSE,
                              float *aS, float *aSW, const double *gx, const double *gy, const double *nrm,
                              const double *lambda_p, const float *Iout, const float *Iin, float alpha_f, int nrows, int ncols,
                              int nframes)
{
    PDEIP_PIXEL_INDEX();
    const size_t n = (size_t)nrows * ncols;
    const double lambda = *lambda_p, alpha = (double)alpha_f;
    // tensor entries at (ii,jj) with circshift wrap-around
    auto tens = [&](int ii, int jj, double &dyy, double &dxx, double &dxy) {
        ii = ii < 0 ? nrows - 1 : (ii > nrows - 1 ? 0 : ii);
        jj = jj < 0 ? ncols - 1 : (jj > ncols - 1 ? 0 : jj);
        const size_t p = (size_t)jj * nrows + ii;
        const double x = gx[p], y = gy[p];
        const double multip = 1.0 / (nrm[p] + 2.0 * lambda);
        dyy = multip * (y * y + lambda);
        dxx = multip * (x * x + lambda);
        dxy = -multip * (x * y);
    };
    double dyy, dxx, dxy, a, b, c;
    tens(i, j, dyy, dxx, dxy);
    const bool c0 = j == 0, cE = j == ncols - 1, r0 = i == 0, rE = i == nrows - 1;
    tens(i, j - 1, a, b, c);
    const double W = c0 ? 0.0 : 0.5 * (dyy + a);
    tens(i - 1, j - 1, a, b, c);
    const double NW = (c0 || r0) ? 0.0 : 0.25 * (dxy + c);
    tens(i - 1, j, a, b, c);
    const double N = r0 ? 0.0 : 0.5 * (dxx + b);
    tens(i - 1, j + 1, a, b, c);
    const double NE = (cE || r0) ? 0.0 : -0.25 * (dxy + c);
    tens(i, j + 1, a, b, c);
    const double E = cE ? 0.0 : 0.5 * (dyy + a);
    tens(i + 1, j + 1, a, b, c);
    const double SE = (cE || rE) ? 0.0 : 0.25 * (dxy + c);
    tens(i + 1, j, a, b, c);
    const double S = rE ? 0.0 : 0.5 * (dxx + b);
    tens(i + 1, j - 1, a, b, c);
    const double SW = (rE || c0) ? 0.0 : -0.25 * (dxy + c);
    double sum = W + NW; // wW+wNW+wN+wNE+wE+wSE+wS+wSW, left to right
    sum = sum + N;
    sum = sum + NE;
    sum = sum + E;
    sum = sum + SE;
    sum = sum + S;
    sum = sum + SW;
    const float asum = (float)(alpha * sum);
    const float w8[8] = {(float)(alpha * W), (float)(alpha * NW), (float)(alpha * N), (float)(alpha * NE),
                         (float)(alpha * E), (float)(alpha * SE), (float)(alpha * S), (float)(alpha * SW)};
    float *outs[8] = {aW, aNW, aN, aNE, aE, aSE, aS, aSW};
    for (int f = 0; f < nframes; ++f) { // the weights are repmat'ed over the frames (:222-231); PsiData is per frame
        const size_t p = (size_t)f * n + pos;
        const float diff = Iout[p] - Iin[p];
        const float psi = 1.0f / sqrtf(diff * diff + 2.220446049250313e-16f); // 1./sqrt((Iout-Iin).^2 + eps), single
        TRACE[p] = psi + asum;
        B[p] = psi * Iin[p];
#pragma unroll
        for (int k = 0; k < 8; ++k) outs[k][p] = w8[k];
    }
}

// TVdenoise4.m:84-98 with its DiffWeights (:116-156), all single: per direction the squared difference to the neighbour plus
// the squared sum of the cross derivatives (imfilter [0.25 0 -0.25], replicate; circshift wraps), maximum over the frames,
// 1/sqrt(. + 0.00001), outer column / row zeroed; then PsiData, TRACE, B and the alpha-scaled weights per frame.
__global__ void k_tv4_assemble(float *TRACE, float *B, float *aW, float *aN, float *aE, float *aS, const float *Iout, const float *Iin,
                               float alpha, int nrows, int ncols, int nframes)
{
    PDEIP_PIXEL_INDEX();
    const size_t n = (size_t)nrows * ncols;
    auto wrap_i = [&](int v) { return v < 0 ? nrows - 1 : (v > nrows - 1 ? 0 : v); };
    auto wrap_j = [&](int v) { return v < 0 ? ncols - 1 : (v > ncols - 1 ? 0 : v); };
    float m[4] = {0.0f, 0.0f, 0.0f, 0.0f}; // W, E, N, S
    for (int f = 0; f < nframes; ++f) {
        const float *P = Iout + (size_t)f * n;
        auto at = [&](int ii, int jj) { return P[(size_t)jj * nrows + ii]; };
        auto ver = [&](int ii, int jj) { return 0.25f * at(max(ii - 1, 0), jj) - 0.25f * at(min(ii + 1, nrows - 1), jj); };
        auto hor = [&](int ii, int jj) { return 0.25f * at(ii, max(jj - 1, 0)) - 0.25f * at(ii, min(jj + 1, ncols - 1)); };
        const float c = at(i, j);
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            const int ii = d == 2 ? wrap_i(i - 1) : (d == 3 ? wrap_i(i + 1) : i), jj = d == 0 ? wrap_j(j - 1) : (d == 1 ? wrap_j(j + 1) : j);
            const float a = at(ii, jj) - c;
            const float b = d < 2 ? ver(i, j) + ver(ii, jj) : hor(i, j) + hor(ii, jj);
            const float v = a * a + b * b;
            m[d] = f == 0 ? v : fmaxf(m[d], v);
        }
    }
    float w[4];
#pragma unroll
    for (int d = 0; d < 4; ++d) w[d] = 1.0f / sqrtf(m[d] + 0.00001f);
    if (j == 0) w[0] = 0.0f;
    if (j == ncols - 1) w[1] = 0.0f;
    if (i == 0) w[2] = 0.0f;
    if (i == nrows - 1) w[3] = 0.0f;
    const float tot = ((w[0] + w[2]) + w[1]) + w[3]; // wW+wN+wE+wS
    const float atot = alpha * tot;
    for (int f = 0; f < nframes; ++f) {
        const size_t p = (size_t)f * n + pos;
        const float diff = Iout[p] - Iin[p];
        const float psi = 1.0f / sqrtf(diff * diff + 2.220446049250313e-16f);
        TRACE[p] = psi + atot;
        B[p] = psi * Iin[p];
        aW[p] = alpha * w[0];
        aE[p] = alpha * w[1];
        aN[p] = alpha * w[2];
        aS[p] = alpha * w[3];
    }
}

// The eight weights of the flow driver's ADdiffWeights (FlowEminAD_llin_2D_v10.m:469-487): the same tensor, circshift
// wrap-around kept at the frame edges (nothing zeroed), handed to Oflow_sor_llin8_2d as single.
__global__ void k_ad_weights(float *wW, float *wNW, float *wN, float *wNE, float *wE, float *wSE, float *wS, float *wSW, const double *gx,
                             const double *gy, const double *nrm, const double *lambda_p, int nrows, int ncols)
{
    PDEIP_PIXEL_INDEX();
    const double lambda = *lambda_p;
    auto tens = [&](int ii, int jj, double &dyy, double &dxx, double &dxy) {
        ii = ii < 0 ? nrows - 1 : (ii > nrows - 1 ? 0 : ii);
        jj = jj < 0 ? ncols - 1 : (jj > ncols - 1 ? 0 : jj);
        const size_t p = (size_t)jj * nrows + ii;
        const double x = gx[p], y = gy[p];
        const double multip = 1.0 / (nrm[p] + 2.0 * lambda);
        dyy = multip * (y * y + lambda);
        dxx = multip * (x * x + lambda);
        dxy = -multip * (x * y);
    };
    double dyy, dxx, dxy, a, b, c;
    tens(i, j, dyy, dxx, dxy);
    tens(i, j - 1, a, b, c);
    wW[pos] = (float)(0.5 * (dyy + a));
    tens(i - 1, j - 1, a, b, c);
    wNW[pos] = (float)(0.25 * (dxy + c));
    tens(i - 1, j, a, b, c);
    wN[pos] = (float)(0.5 * (dxx + b));
    tens(i - 1, j + 1, a, b, c);
    wNE[pos] = (float)(-0.25 * (dxy + c));
    tens(i, j + 1, a, b, c);
    wE[pos] = (float)(0.5 * (dyy + a));
    tens(i + 1, j + 1, a, b, c);
    wSE[pos] = (float)(0.25 * (dxy + c));
    tens(i + 1, j, a, b, c);
    wS[pos] = (float)(0.5 * (dxx + b));
    tens(i + 1, j - 1, a, b, c);
    wSW[pos] = (float)(-0.25 * (dxy + c));
}

} // namespace pdeip
