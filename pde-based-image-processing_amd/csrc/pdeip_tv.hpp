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

// lambda = sorted_nonzero(round(numel * 0.5 + eps)) of the squared gradient norms (zeros excluded; 1 if all are zero);
// quantile >= 0: the flow driver's form sorted_nonzero(round(numel * quantile)) (FlowEminAD_llin_2D_v10.m:461-467).
// Round 1 sorted all n norms (rocPRIM radix sort, 0.55 ms at 4K) to read one element.  This is the selection alone: the norms
// are non-negative doubles (NaN counts as non-zero and as the largest value, as in an ascending sort), so their bit patterns
// order like unsigned integers; six histogram passes over the digits 11+11+11+11+11+9 bits, most significant first, each
// narrowing the candidates to those that share the digits found so far, leave exactly the bits of the wanted element.
constexpr int TVSEL_BINS = 2048;
struct TvSelectState {
    unsigned long long prefix;  // digits found so far (in place, lower bits zero)
    unsigned long long rank;    // 0-based rank of the wanted element among the candidates that share the prefix
    unsigned long long zeros;   // number of exact zeros (counted in pass 0)
    unsigned int done;          // all zero: lambda = 1
    unsigned int hist[TVSEL_BINS];
};

__device__ __forceinline__ unsigned long long tvsel_key(double v)
{
    return v != v ? 0x7ff8000000000000ull : (unsigned long long)__double_as_longlong(v);
}

// pass P: histogram of digit P over the candidates.  SHIFT = position of the digit, PREV = position of the digit before it
// RUN norms per thread: the first two passes see (nearly) all elements spread over many bins, where fewer, fatter workgroups
// mean fewer global merges of the 2048 bins (64); the later passes count a handful of candidates and only stream (16).
template <int SHIFT, int BITS, int PREV, int RUN>
__global__ void __launch_bounds__(256) k_tvsel_hist(TvSelectState *st, const double *nrm, size_t n)
{
    __shared__ unsigned int h[TVSEL_BINS];
    __shared__ unsigned int zc;
    for (int b = threadIdx.x; b < TVSEL_BINS; b += 256) h[b] = 0;
    if (threadIdx.x == 0) zc = 0;
    __syncthreads();
    const unsigned long long prefix = PREV < 64 ? st->prefix : 0ull;
    if (PREV < 64 && st->done) return;
    // a workgroup takes 256 x RUN consecutive norms, coalesced (lane = element); a thread merges equal consecutive digits of
    // its share before it touches LDS (the exponent digits cluster)
    const size_t base = (size_t)blockIdx.x * 256 * RUN + threadIdx.x;
    unsigned int cur = 0xffffffffu, cnt = 0, zeros = 0;
#pragma unroll 4
    for (int e = 0; e < RUN; e++) {
        const size_t p = base + (size_t)e * 256;
        if (p >= n) break;
        const unsigned long long k = tvsel_key(nrm[p]);
        if (PREV >= 64) zeros += (k == 0ull);
        if (PREV < 64 && (k >> PREV) != (prefix >> PREV)) continue;
        const unsigned int d = (unsigned int)(k >> SHIFT) & ((1u << BITS) - 1u);
        if (d != cur) {
            if (cnt) atomicAdd(&h[cur], cnt);
            cur = d;
            cnt = 0;
        }
        cnt++;
    }
    if (cnt) atomicAdd(&h[cur], cnt);
    if (PREV >= 64 && zeros) atomicAdd(&zc, zeros);
    __syncthreads();
    for (int b = threadIdx.x; b < (1 << BITS); b += 256)
        if (h[b]) atomicAdd(&st->hist[b], h[b]);
    if (PREV >= 64 && threadIdx.x == 0 && zc) atomicAdd(&st->zeros, (unsigned long long)zc);
}

// one workgroup: find the digit whose bin holds the wanted rank, extend the prefix, clear the histogram for the next pass;
// FIRST: turn (count of non-zeros, quantile) into the rank first; LAST: write lambda.
template <int SHIFT, int BITS, bool FIRST, bool LAST>
__global__ void __launch_bounds__(256) k_tvsel_scan(TvSelectState *st, double *lambda, size_t n, double quantile)
{
    __shared__ unsigned long long part[256];
    __shared__ unsigned long long s_rank;
    __shared__ int s_done;
    const int t = threadIdx.x;
    if (t == 0) {
        s_done = FIRST ? 0 : (int)st->done;
        if (FIRST) {
            const unsigned long long cnt = (unsigned long long)n - st->zeros;
            if (cnt == 0) {
                s_done = 1;
                st->done = 1;
                *lambda = 1.0;
            } else {
                unsigned long long idx = (cnt + 1) / 2; // round(cnt*0.5 + eps), 1-based
                if (quantile >= 0.0) {
                    idx = (unsigned long long)floor((double)cnt * quantile + 0.5);
                    idx = idx < 1 ? 1 : (idx > cnt ? cnt : idx);
                }
                st->rank = st->zeros + idx - 1; // zeros sort first
                st->done = 0;
            }
        }
        s_rank = st->rank;
    }
    __syncthreads();
    constexpr int NB = 1 << BITS, PER = (NB + 255) / 256;
    unsigned long long loc[PER], sum = 0;
#pragma unroll
    for (int e = 0; e < PER; e++) {
        const int b = t * PER + e;
        loc[e] = b < NB ? st->hist[b] : 0;
        sum += loc[e];
    }
    part[t] = sum;
    __syncthreads();
    if (!s_done) {
        unsigned long long before = 0;
        for (int q = 0; q < t; q++) before += part[q]; // 256 x 256 adds: negligible
        const unsigned long long rank = s_rank;
        if (rank >= before && rank < before + sum) {
            unsigned long long acc = before;
#pragma unroll
            for (int e = 0; e < PER; e++) {
                if (rank >= acc && rank < acc + loc[e]) {
                    const unsigned long long pre = st->prefix | ((unsigned long long)(t * PER + e) << SHIFT);
                    st->prefix = FIRST ? ((unsigned long long)(t * PER + e) << SHIFT) : pre;
                    st->rank = rank - acc;
                    if (LAST) *lambda = __longlong_as_double((long long)(FIRST ? ((unsigned long long)(t * PER + e) << SHIFT) : pre));
                }
                acc += loc[e];
            }
        }
    }
    __syncthreads();
    for (int b = t; b < TVSEL_BINS; b += 256) st->hist[b] = 0;
    if (LAST && t == 0) st->zeros = 0;
}

// The eight weights (times alpha, as single), TRACE and B of one outer iteration (TVdenoise8.m:82-86).
// Tile form of the tensor stencils below: a workgroup of 64 x 4 threads owns 64 rows x 4 columns and evaluates the diffusion
// tensor (one double division per pixel) ONCE for its 66 x 6 neighbourhood into LDS -- circshift wrap-around at the frame
// edges as the drivers have it -- instead of nine times per output pixel.  Same expressions, same bits.
constexpr int TT_R = 64, TT_C = 4;
struct TensorTile {
    double dyy[TT_C + 2][TT_R + 2], dxx[TT_C + 2][TT_R + 2], dxy[TT_C + 2][TT_R + 2];
};

__device__ __forceinline__ void tensor_tile_fill(TensorTile &T, const double *gx, const double *gy, const double *nrm, double lambda, int i0,
                                                 int j0, int nrows, int ncols)
{
    const int tid = threadIdx.y * TT_R + threadIdx.x;
    for (int e = tid; e < (TT_C + 2) * (TT_R + 2); e += TT_R * TT_C) {
        const int c = e / (TT_R + 2), r = e - c * (TT_R + 2);
        int ii = i0 - 1 + r, jj = j0 - 1 + c;
        if (ii > nrows || jj > ncols) continue; // beyond the wrap position: no output pixel reads it
        ii = ii < 0 ? nrows - 1 : (ii > nrows - 1 ? 0 : ii);
        jj = jj < 0 ? ncols - 1 : (jj > ncols - 1 ? 0 : jj);
        const size_t p = (size_t)jj * nrows + ii;
        const double x = gx[p], y = gy[p];
        const double multip = 1.0 / (nrm[p] + 2.0 * lambda);
        T.dyy[c][r] = multip * (y * y + lambda);
        T.dxx[c][r] = multip * (x * x + lambda);
        T.dxy[c][r] = -multip * (x * y);
    }
}

__global__ void __launch_bounds__(TT_R *TT_C)
k_tv_assemble(float *TRACE, float *B, float *aW, float *aNW, float *aN, float *aNE, float *aE, float *aSE, float *aS, float *aSW,
              const double *gx, const double *gy, const double *nrm, const double *lambda_p, const float *Iout, const float *Iin,
              float alpha_f, int nrows, int ncols, int nframes)
{
    __shared__ TensorTile T;
    const int i0 = blockIdx.x * TT_R, j0 = blockIdx.y * TT_C;
    const double lambda = *lambda_p, alpha = (double)alpha_f;
    tensor_tile_fill(T, gx, gy, nrm, lambda, i0, j0, nrows, ncols);
    __syncthreads();
    const int i = i0 + threadIdx.x, j = j0 + threadIdx.y;
    if (i >= nrows || j >= ncols) return;
    const size_t n = (size_t)nrows * ncols, pos = (size_t)j * nrows + i;
    const int r = threadIdx.x + 1, c = threadIdx.y + 1;
    const double dyy = T.dyy[c][r], dxx = T.dxx[c][r], dxy = T.dxy[c][r];
    const bool c0 = j == 0, cE = j == ncols - 1, r0 = i == 0, rE = i == nrows - 1;
    const double W = c0 ? 0.0 : 0.5 * (dyy + T.dyy[c - 1][r]);
    const double NW = (c0 || r0) ? 0.0 : 0.25 * (dxy + T.dxy[c - 1][r - 1]);
    const double N = r0 ? 0.0 : 0.5 * (dxx + T.dxx[c][r - 1]);
    const double NE = (cE || r0) ? 0.0 : -0.25 * (dxy + T.dxy[c + 1][r - 1]);
    const double E = cE ? 0.0 : 0.5 * (dyy + T.dyy[c + 1][r]);
    const double SE = (cE || rE) ? 0.0 : 0.25 * (dxy + T.dxy[c + 1][r + 1]);
    const double S = rE ? 0.0 : 0.5 * (dxx + T.dxx[c][r + 1]);
    const double SW = (rE || c0) ? 0.0 : -0.25 * (dxy + T.dxy[c - 1][r + 1]);
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
__global__ void __launch_bounds__(TT_R *TT_C)
k_ad_weights(float *wW, float *wNW, float *wN, float *wNE, float *wE, float *wSE, float *wS, float *wSW, const double *gx, const double *gy,
             const double *nrm, const double *lambda_p, int nrows, int ncols)
{
    __shared__ TensorTile T;
    const int i0 = blockIdx.x * TT_R, j0 = blockIdx.y * TT_C;
    tensor_tile_fill(T, gx, gy, nrm, *lambda_p, i0, j0, nrows, ncols);
    __syncthreads();
    const int i = i0 + threadIdx.x, j = j0 + threadIdx.y;
    if (i >= nrows || j >= ncols) return;
    const size_t pos = (size_t)j * nrows + i;
    const int r = threadIdx.x + 1, c = threadIdx.y + 1;
    const double dyy = T.dyy[c][r], dxx = T.dxx[c][r], dxy = T.dxy[c][r];
    wW[pos] = (float)(0.5 * (dyy + T.dyy[c - 1][r]));
    wNW[pos] = (float)(0.25 * (dxy + T.dxy[c - 1][r - 1]));
    wN[pos] = (float)(0.5 * (dxx + T.dxx[c][r - 1]));
    wNE[pos] = (float)(-0.25 * (dxy + T.dxy[c + 1][r - 1]));
    wE[pos] = (float)(0.5 * (dyy + T.dyy[c + 1][r]));
    wSE[pos] = (float)(0.25 * (dxy + T.dxy[c + 1][r + 1]));
    wS[pos] = (float)(0.5 * (dxx + T.dxx[c][r + 1]));
    wSW[pos] = (float)(-0.25 * (dxy + T.dxy[c - 1][r + 1]));
}

} // namespace pdeip
