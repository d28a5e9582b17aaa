// pdeip_fas.hpp -- the stages of the FAS full-multigrid flow driver (matlab/optical_flow/FlowEminNDFASFMG_elin_2D_v10.m)
// that sit between its Oflow_sor_elin4_2d / Oflow_lhs_elin4_2d calls, as device kernels: image pyramid (:104-120),
// per-scale derivative planes and constants (:125-153), robust data weights of the smoother (:377-392, :425-441),
// full-weighting restriction (:200, :212-217), the coarse right-hand side (:250-251) and the bilinear prolongation of
// the correction (:256-257).  With the solver, residual and LHS kernels a whole V/W cycle stays in HBM.
//
// Like pdeip_flow.hpp these restate MATLAB array code: single op double -> single, expressions left to right; IPT's
// imfilter / imresize are restated by their documented meaning with our own summation order (stated at each kernel).
// oracle/matlab_side.py (fas_*) is the numpy statement the tests compare against bit for bit; parity with MATLAB
// itself is unpinned.
#pragma once
#include <hip/hip_runtime.h>

#include "pdeip_flow.hpp"

namespace pdeip {

__constant__ float FAS_LPF[5] = {0.0625f, 0.25f, 0.375f, 0.25f, 0.0625f};                       // lpf = [1 4 6 4 1]/16 (:99)
__constant__ float FAS_D1F_SCL[5] = {(float)(-0.104550 / 255), (float)(-0.292315 / 255), 0.0f, (float)(0.292315 / 255),
                                     (float)(0.104550 / 255)};                                  // O_dx_scl (:87), flipped

struct FasTaps25 {
    float g[25]; // flipped kernel, column-major: g[b*5+a] multiplies in(i+a-2, j+b-2)
};

// imfilter(I, G, 'replicate', 'conv') with a 5x5 kernel (:104-105): single arithmetic, taps in column-major order.
__global__ void k_fas_gauss5(float *out, const float *in, FasTaps25 T, int nrows, int ncols)
{
    PDEIP_PIXEL_INDEX();
    const size_t fo = (size_t)blockIdx.z * nrows * ncols;
    float s = 0.0f;
#pragma unroll
    for (int b = 0; b < 5; ++b)
#pragma unroll
        for (int a = 0; a < 5; ++a) {
            const int ii = min(max(i + a - 2, 0), nrows - 1), jj = min(max(j + b - 2, 0), ncols - 1);
            const float t = T.g[b * 5 + a] * in[fo + (size_t)jj * nrows + ii];
            s = (a == 0 && b == 0) ? t : s + t;
        }
    out[fo + pos] = s;
}

// One pyramid step (:108-111): lpf along the rows of the image, lpf' down its columns, keep (1:2:end, 1:2:end).
// nrows/ncols are the OUTPUT dimensions (ceil of half the input's).
__global__ void k_fas_down(float *out, const float *in, int nrows_in, int ncols_in, int nrows, int ncols)
{
    PDEIP_PIXEL_INDEX();
    const float *src = in + (size_t)blockIdx.z * nrows_in * ncols_in;
    const HsImage A{src, src, 1, nrows_in, ncols_in};
    out[(size_t)blockIdx.z * nrows * ncols + pos] = hs_hv(A, FAS_LPF, FAS_LPF, 2 * i, 2 * j);
}

// The per-scale constants (:125-153), one thread per pixel and channel (blockIdx.z).  `planes` is one block
// [13][C][ncols][nrows] in the order Idt, Idx, Idy, Idxx, Idyy, Idxy, Idxt, Idyt, M, Cu, Cv, Du, Dv.
enum { FAS_IDT, FAS_IDX, FAS_IDY, FAS_IDXX, FAS_IDYY, FAS_IDXY, FAS_IDXT, FAS_IDYT, FAS_M, FAS_CU, FAS_CV, FAS_DU, FAS_DV, FAS_NPLANES };

// Tiled like k_derivatives5_tiled: a 64 x 4 tile stages its 68 x 8 neighbourhood of both frames (clamped coordinates =
// 'replicate') and of S = ((It0 + It1) * 0.55) / 255 in LDS, evaluates each of the seven first-pass results once, then the nine
// second passes -- instead of nine 25-tap evaluations per pixel.  Products and left-to-right sums as hs_v5 / hs_h5 have them.
constexpr int FP_TR = 64, FP_TC = 4, FP_IR = FP_TR + 4, FP_IC = FP_TC + 4;

__global__ void __launch_bounds__(FP_TR *FP_TC)
k_fas_prepare(float *planes, const float *It0, const float *It1, int C, float b1, float b2, int nrows, int ncols)
{
    __shared__ float inS[FP_IC][FP_IR], in0[FP_IC][FP_IR], in1[FP_IC][FP_IR];
    __shared__ float VS[FP_IC][FP_TR], V0[FP_IC][FP_TR], V1[FP_IC][FP_TR];                       // V5(pre) at own rows, columns j-2..j+2
    __shared__ float HS[FP_TC][FP_IR], HD[FP_TC][FP_IR], H0[FP_TC][FP_IR], H1[FP_TC][FP_IR];    // H5(pre) S, H5(d1f) S, H5(pre) It0 / It1
    const int tr = threadIdx.x, tc = threadIdx.y, tid = tc * FP_TR + tr;
    const int i0 = blockIdx.x * FP_TR, j0 = blockIdx.y * FP_TC;
    const size_t n = (size_t)nrows * ncols, c = blockIdx.z, blk = n * C;
    const float *a = It0 + c * n, *b = It1 + c * n;
    for (int e = tid; e < FP_IC * FP_IR; e += FP_TR * FP_TC) {
        const int cc = e / FP_IR, r = e - cc * FP_IR;
        const int ii = min(max(i0 - 2 + r, 0), nrows - 1), jj = min(max(j0 - 2 + cc, 0), ncols - 1);
        const size_t g = (size_t)jj * nrows + ii;
        const float va = a[g], vb = b[g];
        in0[cc][r] = va;
        in1[cc][r] = vb;
        inS[cc][r] = ((va + vb) * 0.55f) / 255.0f; // HsImage mode 3
    }
    __syncthreads();
    auto five = [](float k0, float x0, float k1, float x1, float k2, float x2, float k3, float x3, float k4, float x4) {
        float s = k0 * x0; // hs_v5 / hs_h5: products summed left to right
        s = s + k1 * x1;
        s = s + k2 * x2;
        s = s + k3 * x3;
        s = s + k4 * x4;
        return s;
    };
    for (int e = tid; e < FP_IC * FP_TR; e += FP_TR * FP_TC) { // vertical first passes
        const int cc = e / FP_TR, r = e - cc * FP_TR;
        VS[cc][r] = five(HS_PRE[0], inS[cc][r], HS_PRE[1], inS[cc][r + 1], HS_PRE[2], inS[cc][r + 2], HS_PRE[3], inS[cc][r + 3], HS_PRE[4], inS[cc][r + 4]);
        V0[cc][r] = five(HS_PRE[0], in0[cc][r], HS_PRE[1], in0[cc][r + 1], HS_PRE[2], in0[cc][r + 2], HS_PRE[3], in0[cc][r + 3], HS_PRE[4], in0[cc][r + 4]);
        V1[cc][r] = five(HS_PRE[0], in1[cc][r], HS_PRE[1], in1[cc][r + 1], HS_PRE[2], in1[cc][r + 2], HS_PRE[3], in1[cc][r + 3], HS_PRE[4], in1[cc][r + 4]);
    }
    for (int e = tid; e < FP_TC * FP_IR; e += FP_TR * FP_TC) { // horizontal first passes
        const int cc = e / FP_IR, r = e - cc * FP_IR;
        HS[cc][r] = five(HS_PRE[0], inS[cc][r], HS_PRE[1], inS[cc + 1][r], HS_PRE[2], inS[cc + 2][r], HS_PRE[3], inS[cc + 3][r], HS_PRE[4], inS[cc + 4][r]);
        HD[cc][r] = five(HS_D1F[0], inS[cc][r], HS_D1F[1], inS[cc + 1][r], HS_D1F[2], inS[cc + 2][r], HS_D1F[3], inS[cc + 3][r], HS_D1F[4], inS[cc + 4][r]);
        H0[cc][r] = five(HS_PRE[0], in0[cc][r], HS_PRE[1], in0[cc + 1][r], HS_PRE[2], in0[cc + 2][r], HS_PRE[3], in0[cc + 3][r], HS_PRE[4], in0[cc + 4][r]);
        H1[cc][r] = five(HS_PRE[0], in1[cc][r], HS_PRE[1], in1[cc + 1][r], HS_PRE[2], in1[cc + 2][r], HS_PRE[3], in1[cc + 3][r], HS_PRE[4], in1[cc + 4][r]);
    }
    __syncthreads();
    const int i = i0 + tr, j = j0 + tc;
    if (i >= nrows || j >= ncols) return;
    const size_t pos = (size_t)j * nrows + i;
    auto h5 = [&](const float (&V)[FP_IC][FP_TR], const float *k) { return five(k[0], V[tc][tr], k[1], V[tc + 1][tr], k[2], V[tc + 2][tr], k[3], V[tc + 3][tr], k[4], V[tc + 4][tr]); };
    auto v5 = [&](const float (&H)[FP_TC][FP_IR], const float *k) { return five(k[0], H[tc][tr], k[1], H[tc][tr + 1], k[2], H[tc][tr + 2], k[3], H[tc][tr + 3], k[4], H[tc][tr + 4]); };
    const float Idt = (in0[tc + 2][tr + 2] - in1[tc + 2][tr + 2]) / 255.0f;
    const float Idx = h5(VS, HS_D1F);
    const float Idy = v5(HS, HS_D1F);
    const float Idxx = h5(VS, HS_D2);
    const float Idyy = v5(HS, HS_D2);
    const float Idxy = v5(HD, HS_D1F);
    const float Idxt = h5(V0, FAS_D1F_SCL) - h5(V1, FAS_D1F_SCL);
    const float Idyt = v5(H0, FAS_D1F_SCL) - v5(H1, FAS_D1F_SCL);
    float *o = planes + c * n + pos;
    o[FAS_IDT * blk] = Idt;
    o[FAS_IDX * blk] = Idx;
    o[FAS_IDY * blk] = Idy;
    o[FAS_IDXX * blk] = Idxx;
    o[FAS_IDYY * blk] = Idyy;
    o[FAS_IDXY * blk] = Idxy;
    o[FAS_IDXT * blk] = Idxt;
    o[FAS_IDYT * blk] = Idyt;
    o[FAS_M * blk] = (b1 * Idy) * Idx + (b2 * Idxy) * (Idxx + Idyy);
    o[FAS_CU * blk] = (b1 * Idt) * Idx + b2 * (Idxt * Idxx + Idyt * Idxy);
    o[FAS_CV * blk] = (b1 * Idt) * Idy + b2 * (Idxt * Idxy + Idyt * Idyy);
    o[FAS_DU * blk] = (b1 * Idx) * Idx + b2 * (Idxx * Idxx + Idxy * Idxy);
    o[FAS_DV * blk] = (b1 * Idy) * Idy + b2 * (Idxy * Idxy + Idyy * Idyy);
}

// gd = 1./(k*sqrt(OPnorm+0.00001)) (:382-386, :430-434, :228-232) and the planes handed to the solver:
//   PER_FRAME = false: sum(M.*gd,3) ... (:393-397), one plane each           (k = channels*alpha)
//   PER_FRAME = true : M.*gd ... per channel (:441-445, :235-237) and gd itself (k = alpha); Cu/Cv and gd_out may be null.
template <bool PER_FRAME>
__global__ void k_fas_assemble(float *MGd, float *CuGd, float *CvGd, float *DuGd, float *DvGd, float *gd_out, const float *planes,
                               const float *Cu, const float *Cv, const float *U, const float *V, int C, float b1, float b2, float k,
                               int nrows, int ncols, FlowWeightsOut W)
{
    PDEIP_PIXEL_INDEX();
    if (W.wW) opdiffweights_pixel(W.wW, W.wN, W.wS, W.wE, U, V, nullptr, nullptr, i, j, pos, nrows, ncols); // OPdiffWeights(U, V) (:392)
    const size_t n = (size_t)nrows * ncols, blk = n * C;
    const float u = U[pos], v = V[pos];
    float m = 0.0f, cu = 0.0f, cv = 0.0f, du = 0.0f, dv = 0.0f;
    for (int c = 0; c < C; ++c) {
        const size_t q = (size_t)c * n + pos;
        const float *p = planes + q;
        const float Idxy = p[FAS_IDXY * blk];
        float r1 = p[FAS_IDT * blk] - p[FAS_IDX * blk] * u;
        r1 = r1 - p[FAS_IDY * blk] * v;
        float r2 = p[FAS_IDXT * blk] - p[FAS_IDXX * blk] * u;
        r2 = r2 - Idxy * v;
        float r3 = p[FAS_IDYT * blk] - Idxy * u;
        r3 = r3 - p[FAS_IDYY * blk] * v;
        const float opnorm = b1 * (r1 * r1) + b2 * ((r2 * r2) + (r3 * r3));
        const float gd = 1.0f / (k * sqrtf(opnorm + 0.00001f));
        const float Mg = p[FAS_M * blk] * gd, Dug = p[FAS_DU * blk] * gd, Dvg = p[FAS_DV * blk] * gd;
        const float Cug = Cu ? Cu[q] * gd : 0.0f, Cvg = Cv ? Cv[q] * gd : 0.0f;
        if (PER_FRAME) {
            MGd[q] = Mg;
            DuGd[q] = Dug;
            DvGd[q] = Dvg;
            if (Cu) CuGd[q] = Cug;
            if (Cv) CvGd[q] = Cvg;
            if (gd_out) gd_out[q] = gd;
        } else {
            m = c ? m + Mg : Mg;
            cu = c ? cu + Cug : Cug;
            cv = c ? cv + Cvg : Cvg;
            du = c ? du + Dug : Dug;
            dv = c ? dv + Dvg : Dvg;
        }
    }
    if (!PER_FRAME) {
        MGd[pos] = m;
        DuGd[pos] = du;
        DvGd[pos] = dv;
        if (Cu) CuGd[pos] = cu;
        if (Cv) CvGd[pos] = cv;
    }
}

// imfilter(A*scale, fw, 'replicate', 'conv')(1:2:end, 1:2:end, :), fw = [1 2 1; 2 4 2; 1 2 1]/16 (:200, :212-217).
// nrows/ncols are the OUTPUT dimensions; taps in column-major order, single arithmetic.
__global__ void k_fas_restrict(float *out, const float *in, float scale, int nrows_in, int ncols_in, int nrows, int ncols)
{
    PDEIP_PIXEL_INDEX();
    const float *src = in + (size_t)blockIdx.z * nrows_in * ncols_in;
    float s = 0.0f;
#pragma unroll
    for (int b = 0; b < 3; ++b)
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const int ii = min(max(2 * i + a - 1, 0), nrows_in - 1), jj = min(max(2 * j + b - 1, 0), ncols_in - 1);
            const float w = (a == 1 ? 2.0f : 1.0f) * (b == 1 ? 2.0f : 1.0f) * 0.0625f;
            const float t = w * (src[(size_t)jj * nrows_in + ii] * scale);
            s = (a == 0 && b == 0) ? t : s + t;
        }
    out[(size_t)blockIdx.z * nrows * ncols + pos] = s;
}

// fu = (RUres + Au)./gd (:250-251), all [nrows x ncols x C]
__global__ void k_fas_rhs(float *out, const float *R, const float *A, const float *gd, int nrows, int ncols)
{
    PDEIP_PIXEL_INDEX();
    const size_t q = (size_t)blockIdx.z * nrows * ncols + pos;
    out[q] = (R[q] + A[q]) / gd[q];
}

// U = U + imresize((Uc-Ures)*(1/scl_factor), size(U), 'bilinear') (:256-257).  Enlarging, so no antialiasing: two taps per
// axis at MATLAB's pixel-centre alignment, indices clamped; rows first, then columns, in double, rounded to single once.
__device__ __forceinline__ void fas_bilin_axis(int o, int n_in, int n_out, int &i0, int &i1, double &f)
{
    const double scale = (double)n_out / (double)n_in;
    const double x = ((double)o + 0.5) / scale - 0.5;
    const double fl = floor(x);
    f = x - fl;
    const int k = (int)fl;
    i0 = min(max(k, 0), n_in - 1);
    i1 = min(max(k + 1, 0), n_in - 1);
}

__global__ void k_fas_prolong_add(float *U, const float *Uc, const float *Ures, float inv_scale, int nrows_c, int ncols_c, int nrows,
                                  int ncols)
{
    PDEIP_PIXEL_INDEX();
    int r0, r1, c0, c1;
    double fr, fc;
    fas_bilin_axis(i, nrows_c, nrows, r0, r1, fr);
    fas_bilin_axis(j, ncols_c, ncols, c0, c1, fc);
    auto D = [&](int ii, int jj) -> double {
        const size_t p = (size_t)jj * nrows_c + ii;
        return (double)((Uc[p] - Ures[p]) * inv_scale);
    };
    const double t0 = (1.0 - fr) * D(r0, c0) + fr * D(r1, c0);
    const double t1 = (1.0 - fr) * D(r0, c1) + fr * D(r1, c1);
    const double r = (1.0 - fc) * t0 + fc * t1;
    U[pos] = U[pos] + (float)r;
}

// U = imresize(U.*(1/scl_factor), size_of_the_finer_scale) of the driver's outer loop (:177-180): imresize's default method,
// bicubic (Keys kernel, a = -0.5), enlarging so not antialiased.  Four taps per axis at MATLAB's pixel-centre alignment,
// indices clamped, weights divided by their sum as imresize does; rows first, then columns, in double, rounded once.
__device__ __forceinline__ double fas_cubic(double t)
{
    t = fabs(t);
    const double t2 = t * t, t3 = t2 * t;
    if (t <= 1.0) return (1.5 * t3 - 2.5 * t2) + 1.0;
    if (t <= 2.0) return ((-0.5 * t3 + 2.5 * t2) - 4.0 * t) + 2.0;
    return 0.0;
}

__device__ __forceinline__ void fas_cubic_axis(int o, int n_in, int n_out, int idx[4], double w[4])
{
    const double scale = (double)n_out / (double)n_in;
    const double x = ((double)o + 0.5) / scale - 0.5;
    const int first = (int)floor(x) - 1;
    double sum = 0.0;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        w[t] = fas_cubic(x - (double)(first + t));
        sum = t ? sum + w[t] : w[t];
        idx[t] = min(max(first + t, 0), n_in - 1);
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) w[t] = w[t] / sum;
}

__global__ void k_fas_upscale(float *out, const float *in, float mul, int nrows_in, int ncols_in, int nrows, int ncols)
{
    PDEIP_PIXEL_INDEX();
    int ri[4], ci[4];
    double rw[4], cw[4];
    fas_cubic_axis(i, nrows_in, nrows, ri, rw);
    fas_cubic_axis(j, ncols_in, ncols, ci, cw);
    double acc = 0.0;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const float *col = in + (size_t)ci[c] * nrows_in;
        double t = rw[0] * (double)(col[ri[0]] * mul);
#pragma unroll
        for (int r = 1; r < 4; ++r) t = t + rw[r] * (double)(col[ri[r]] * mul);
        acc = c ? acc + cw[c] * t : cw[c] * t;
    }
    out[pos] = (float)acc;
}

} // namespace pdeip
