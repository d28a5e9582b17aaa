// pdeip_pointwise.hpp -- the embarrassingly parallel kernels of the path (gfx950):
// divisor prologues, residual / LHS operators, diffusion weights and the bilinear warp.
//
// Thread mapping for all of them: threadIdx.x runs along the image row index i (the
// contiguous direction), so every plane access of a wave is one coalesced segment.
// Residual/LHS outputs include the reference's border replicate by evaluating the operator at
// the clamped interior coordinate, so no second pass (and no inter-workgroup ordering) is needed.
#pragma once
#include "pdeip_models.hpp"

namespace pdeip {

#define PDEIP_PIXEL_INDEX()                                          \
    const int i = blockIdx.x * blockDim.x + threadIdx.x;             \
    const int j = blockIdx.y;                                        \
    if (i >= nrows) return;                                          \
    const size_t pos = (size_t)j * nrows + i

// ---- divisor prologues (what the reference builds during its first sweep) -------------------

// Generic form for the 5-point models: P.cf[D0], P.cf[D1] point at the raw planes; the derived
// planes go to out0/out1.  Used by the exact-order path (the red-black path fuses this into its
// first sweep).  Loads of slots derive() does not read are dead code.
template <class Mdl>
__global__ void k_derive(SweepPlanes<Mdl> P, float *out0, float *out1, int nrows, int ncols, size_t frame_stride)
{
    PDEIP_PIXEL_INDEX();
    const size_t q = pos + (size_t)blockIdx.z * frame_stride;
    float k[Mdl::NCF];
#pragma unroll
    for (int f = 0; f < Mdl::NCF; f++) k[f] = P.cf[f][q];
    Mdl::derive(k);
    out0[q] = k[Mdl::D0];
    out1[q] = k[Mdl::D1];
}

// pdeSolvers.c:217-237
static __global__ void k_pde8_divisors(float *bt, float *inv, const float *TRACE, const float *B,
                                const float *wW, const float *wNW, const float *wN,
                                const float *wNE, const float *wE, const float *wSE,
                                const float *wS, const float *wSW, int nrows, int ncols,
                                size_t frame_stride)
{
    PDEIP_PIXEL_INDEX();
    const size_t p = pos + (size_t)blockIdx.z * frame_stride;
    const float tr = TRACE[p];
    float t = wE[p] + wW[p];
    t += wS[p] + wN[p];
    t += wSW[p] + wNW[p];
    t += wSE[p] + wNE[p];
    const bool ok = !is_nan(tr);
    inv[p] = ok ? 1.0f / tr : 1.0f / t;
    bt[p] = ok ? B[p] : 0.0f;
}

// ---- residual / LHS of the coupled (u,v) systems ---------------------------------------------

__device__ __forceinline__ int clampi(int x, int lo, int hi) { return x < lo ? lo : (x > hi ? hi : x); }

// LLIN: late linearization (iterate = dU,dV on top of U,V); else early linearization.
// LHS : A*x instead of b - A*x.
// opticalflowSolvers.c:269-380, :387-496, :766-916, :923-1070
template <bool LLIN, bool LHS>
__device__ __forceinline__ void oflow_operator(float &outU, float &outV, int ii, int jj, int nrows,
                                               size_t fo, const float *U, const float *V,
                                               const float *dU, const float *dV, const float *M,
                                               const float *Cu, const float *Cv, const float *Du,
                                               const float *Dv, const float *wW, const float *wN,
                                               const float *wE, const float *wS)
{
    const size_t pos = (size_t)jj * nrows + ii, os = pos + fo;
    const float ww = wW[pos], wn = wN[pos], we = wE[pos], ws = wS[pos];
    float nbU, nbV, xu, xv; // neighbourhood sums and the centre unknowns
    if (LLIN) {
        float a = dU[pos - nrows] + U[pos - nrows], b = dU[pos + nrows] + U[pos + nrows];
        float c = dU[pos - 1] + U[pos - 1], d = dU[pos + 1] + U[pos + 1];
        const float uc = U[pos];
        a -= uc; b -= uc; c -= uc; d -= uc;
        a *= ww; b *= we; c *= wn; d *= ws;
        a += b; c += d; a += c;
        nbU = a;
        a = dV[pos - nrows] + V[pos - nrows]; b = dV[pos + nrows] + V[pos + nrows];
        c = dV[pos - 1] + V[pos - 1]; d = dV[pos + 1] + V[pos + 1];
        const float vc = V[pos];
        a -= vc; b -= vc; c -= vc; d -= vc;
        a *= ww; b *= we; c *= wn; d *= ws;
        a += b; c += d; a += c;
        nbV = a;
        xu = dU[pos];
        xv = dV[pos];
    } else {
        nbU = U[pos - nrows] * ww;
        nbU = nbU + U[pos + nrows] * we;
        nbU = nbU + U[pos - 1] * wn;
        nbU = nbU + U[pos + 1] * ws;
        nbV = V[pos - nrows] * ww;
        nbV = nbV + V[pos + nrows] * we;
        nbV = nbV + V[pos - 1] * wn;
        nbV = nbV + V[pos + 1] * ws;
        xu = U[pos];
        xv = V[pos];
    }
    float s = ww + we;
    const float s2 = wn + ws;
    s += s2;
    const float m = M[os];
    if (LHS) {
        const float du = Du[os], dv = Dv[os];
        float t = m * xv - nbU;
        outU = !is_nan(du) ? t + (du + s) * xu : -nbU + s * xu;
        t = m * xu - nbV;
        outV = !is_nan(dv) ? t + (dv + s) * xv : -nbV + s * xv;
    } else {
        const float cu = Cu[os], cv = Cv[os];
        float t = cu - m * xv;
        t = t + nbU;
        outU = !is_nan(cu) ? t - (Du[os] + s) * xu : nbU - s * xu;
        t = cv - m * xu;
        t = t + nbV;
        outV = !is_nan(cv) ? t - (Dv[os] + s) * xv : nbV - s * xv;
    }
}

template <bool LLIN, bool LHS>
__global__ void k_oflow_operator(float *OU, float *OV, const float *U, const float *V,
                                 const float *dU, const float *dV, const float *M, const float *Cu,
                                 const float *Cv, const float *Du, const float *Dv, const float *wW,
                                 const float *wN, const float *wE, const float *wS, int nrows,
                                 int ncols, size_t frame_stride)
{
    PDEIP_PIXEL_INDEX();
    const int k = blockIdx.z;
    const size_t fo = (size_t)k * frame_stride;
    const int ii = clampi(i, 1, nrows - 2);
    int jj = clampi(j, 1, ncols - 2);
    float ou, ov;
    oflow_operator<LLIN, LHS>(ou, ov, ii, jj, nrows, fo, U, V, dU, dV, M, Cu, Cv, Du, Dv, wW, wN, wE, wS);
    if (LLIN && !LHS && j == 0 && k > 0) {
        // Residuals_llin4_2d fills the west border of RV from frame 0 (opticalflowSolvers.c:912)
        float dummy;
        oflow_operator<LLIN, LHS>(dummy, ov, ii, 1, nrows, 0, U, V, dU, dV, M, Cu, Cv, Du, Dv, wW, wN, wE, wS);
    }
    if (LLIN && LHS && i == 0) {
        // LHS_llin4_2d fills the north border of AV from a stale frame-0 cell
        // (opticalflowSolvers.c:1056): zero in frame 0, AV(nrows-2, ncols-2, frame 0) afterwards
        if (k == 0) ov = 0.0f;
        else {
            float dummy;
            oflow_operator<LLIN, LHS>(dummy, ov, nrows - 2, ncols - 2, nrows, 0, U, V, dU, dV, M, Cu, Cv, Du, Dv, wW, wN, wE, wS);
        }
    }
    OU[pos + fo] = ou;
    OV[pos + fo] = ov;
}

// ---- diffusion weights: diffWeights6_2D_c (imageDiffusionWeights.c:341-378) -------------------

__device__ __forceinline__ float dw_ver(const float *d, int i, int j, int nrows)
{ // Dver :44-69: 0.25*D(north) - 0.25*D(south), replicate ends
    const size_t pos = (size_t)j * nrows + i;
    const float up = d[i > 0 ? pos - 1 : pos], dn = d[i < nrows - 1 ? pos + 1 : pos];
    const float A = 0.25f * up, B = -0.25f * dn;
    return A + B;
}
__device__ __forceinline__ float dw_hor(const float *d, int i, int j, int nrows, int ncols)
{ // Dhor :85-108
    const size_t pos = (size_t)j * nrows + i;
    const float lf = d[j > 0 ? pos - nrows : pos], rt = d[j < ncols - 1 ? pos + nrows : pos];
    const float A = 0.25f * lf, B = -0.25f * rt;
    return A + B;
}
__device__ __forceinline__ float dw_inv_sqrt(float t, float eps)
{ // :156  1.0f/(float)sqrt(temp+eps): correctly rounded f32 sqrt, then f32 divide.
  // sqrtf() is the correctly rounded form here (v_sqrt_f32 + fma fix-up); __fsqrt_rn is the bare
  // 1-ulp v_sqrt_f32 on this toolchain.
    const float s = t + eps;
    return 1.0f / sqrtf(s);
}

static __global__ void k_diffweights6(float *wW, float *wN, float *wE, float *wS, const float *D,
                               int nrows, int ncols, int nframes, float eps)
{
    PDEIP_PIXEL_INDEX();
    const size_t n = (size_t)nrows * ncols;
    float tW = 0.f, tN = 0.f, tE = 0.f, tS = 0.f;
    for (int k = 0; k < nframes; k++) {
        const float *d = D + (size_t)k * n;
        const float dc = d[pos];
        const float vc = dw_ver(d, i, j, nrows), hc = dw_hor(d, i, j, nrows, ncols);
        float A, B, t;
        if (j >= 1) { // Calc_wW :123-147
            A = dc - d[pos - nrows];
            B = vc + dw_ver(d, i, j - 1, nrows);
            A = A * A; B = B * B; t = A + B;
            if (k == 0) tW = t; else if (t > tW) tW = t;
        }
        if (i >= 1) { // Calc_wN :177-200
            A = dc - d[pos - 1];
            B = hc + dw_hor(d, i - 1, j, nrows, ncols);
            A = A * A; B = B * B; t = A + B;
            if (k == 0) tN = t; else if (t > tN) tN = t;
        }
        if (j <= ncols - 2) { // Calc_wE :236-258
            A = dc - d[pos + nrows];
            B = vc + dw_ver(d, i, j + 1, nrows);
            A = A * A; B = B * B; t = A + B;
            if (k == 0) tE = t; else if (t > tE) tE = t;
        }
        if (i <= nrows - 2) { // Calc_wS :289-311
            A = dc - d[pos + 1];
            B = hc + dw_hor(d, i + 1, j, nrows, ncols);
            A = A * A; B = B * B; t = A + B;
            if (k == 0) tS = t; else if (t > tS) tS = t;
        }
    }
    // cells the reference never writes keep the gateway's zero initialisation
    wW[pos] = (j >= 1) ? dw_inv_sqrt(tW, eps) : 0.0f;
    wN[pos] = (i >= 1) ? dw_inv_sqrt(tN, eps) : 0.0f;
    wE[pos] = (j <= ncols - 2) ? dw_inv_sqrt(tE, eps) : 0.0f;
    wS[pos] = (i <= nrows - 2) ? dw_inv_sqrt(tS, eps) : 0.0f;
}

// ---- bilinear warp: bilinInterp2 (imageInterpolation.c:44-140) --------------------------------

// One pixel of bilinInterp2 at the 1-based sampling position (Xp, Yp): every frame of up to two image stacks (the drivers warp
// the first and the second constancy image with the same coordinates).
__device__ __forceinline__ void warp_pixel(float Xp, float Yp, size_t pos, int nrows, int ncols, float *Iout, const float *Iin, int nframes,
                                           float *Iout2, const float *Iin2, int nframes2)
{
    const size_t n = (size_t)nrows * ncols;
    // :82-86.  The reference casts floor() to unsigned and lets negatives wrap out of range; HIP's
    // float->unsigned conversion saturates instead, so the range test is explicit and signed.
    const float xm = Xp - 1.0f, ym = Yp - 1.0f;
    const float fx = floorf(xm), fy = floorf(ym);
    const bool inside = (fx >= 0.0f) && (fx < (float)ncols) && (fy >= 0.0f) && (fy < (float)nrows);
    if (inside) {
        const int x = (int)fx, y = (int)fy;
        const float xf = xm - (float)x, yf = ym - (float)y; // :89-90
        const float w00 = (1.0f - xf) * (1.0f - yf);        // :93-96
        const float w10 = xf * (1.0f - yf);
        const float w01 = (1.0f - xf) * yf;
        const float w11 = xf * yf;
        const size_t p00 = (size_t)nrows * x + y;
        const size_t dx = (x < ncols - 1) ? (size_t)nrows : 0, dy = (y < nrows - 1) ? 1 : 0; // :105-110
        const size_t p10 = p00 + dx, p01 = p00 + dy;
        const size_t p11 = (x < ncols - 1 && y < nrows - 1) ? p00 + nrows + 1 : p00;
        for (int k = 0; k < nframes + nframes2; k++) {
            const float *src = k < nframes ? Iin + (size_t)k * n : Iin2 + (size_t)(k - nframes) * n;
            float r = w00 * src[p00]; // :120-123, left to right
            r = r + w10 * src[p10];
            r = r + w01 * src[p01];
            r = r + w11 * src[p11];
            (k < nframes ? Iout + (size_t)k * n : Iout2 + (size_t)(k - nframes) * n)[pos] = r;
        }
    } else {
        const float nanv = __int_as_float(0x7fc00000);
        for (int k = 0; k < nframes; k++) Iout[(size_t)k * n + pos] = nanv; // :129-135
        for (int k = 0; k < nframes2; k++) Iout2[(size_t)k * n + pos] = nanv;
    }
}

static __global__ void k_warp_bilinear(float *Iout, const float *Iin, const float *X, const float *Y,
                                int nrows, int ncols, int nframes)
{
    PDEIP_PIXEL_INDEX();
    warp_pixel(X[pos], Y[pos], pos, nrows, ncols, Iout, Iin, nframes, nullptr, nullptr, 0);
}

// The drivers' warp step in one launch (FlowEminND_llin_2D_v10.m:223-231): X,Y = meshgrid(1:cols,1:rows); both constancy images
// sampled at single(X+U), single(Y+V) (V may be NULL: the disparity drivers warp along x only).
static __global__ void k_flow_warp(float *Iout, const float *Iin, int nframes, float *Iout2, const float *Iin2, int nframes2, const float *U,
                                   const float *V, int nrows, int ncols)
{
    PDEIP_PIXEL_INDEX();
    const float Xp = (float)(j + 1) + U[pos], Yp = V ? (float)(i + 1) + V[pos] : (float)(i + 1);
    warp_pixel(Xp, Yp, pos, nrows, ncols, Iout, Iin, nframes, Iout2, Iin2, nframes2);
}

// ---- Simoncelli derivatives: fstSimoncelli_c / sndSimoncelli_c (imageDerivatives.c:309-482) --------
// The reference runs separable 5-tap correlations through float temporaries (replicate ends, five
// products summed left to right).  Here each output pixel re-evaluates the five temporaries it needs
// from the 5x5 neighbourhood with exactly the same operations, so the results are bit-identical and no
// temporary plane touches HBM (the 25 taps of neighbouring threads overlap in L1/L2).

__constant__ float SIM_SMOOTH[5] = {0.037659f, 0.249724f, 0.439911f, 0.249724f, 0.037659f}; // FstDerivatives5.c:59
__constant__ float SIM_D1[5] = {-0.104550f, -0.292315f, 0.0f, 0.292315f, 0.104550f};         // :60
__constant__ float SIM_D2[5] = {0.232905f, 0.002668f, -0.471147f, 0.002668f, 0.232905f};     // SndDerivatives5.c:67

__device__ __forceinline__ float sum5(const float (&t)[5])
{
    float r = t[0] + t[1];
    r = r + t[2];
    r = r + t[3];
    r = r + t[4];
    return r;
}
// VerticalConvWO5 at (i, j): taps along the rows of column j (imageDerivatives.c:66-119)
__device__ __forceinline__ float v5_at(const float *in, const float *op, int i, int j, int nrows)
{
    const float *c = in + (size_t)j * nrows;
    float t[5];
#pragma unroll
    for (int k = 0; k < 5; k++) t[k] = c[clampi(i - 2 + k, 0, nrows - 1)] * op[k];
    return sum5(t);
}
// HorizontalConvWO5 at (i, j): taps along the columns of row i (:125-211)
__device__ __forceinline__ float h5_at(const float *in, const float *op, int i, int j, int nrows, int ncols)
{
    float t[5];
#pragma unroll
    for (int k = 0; k < 5; k++) t[k] = in[(size_t)clampi(j - 2 + k, 0, ncols - 1) * nrows + i] * op[k];
    return sum5(t);
}
// H5(op2) of the temporary V5(op1)(in), and V5(op2) of the temporary H5(op1)(in)
__device__ __forceinline__ float h5_of_v5(const float *in, const float *op1, const float *op2, int i, int j, int nrows, int ncols)
{
    float t[5];
#pragma unroll
    for (int k = 0; k < 5; k++) t[k] = v5_at(in, op1, i, clampi(j - 2 + k, 0, ncols - 1), nrows) * op2[k];
    return sum5(t);
}
__device__ __forceinline__ float v5_of_h5(const float *in, const float *op1, const float *op2, int i, int j, int nrows, int ncols)
{
    float t[5];
#pragma unroll
    for (int k = 0; k < 5; k++) t[k] = h5_at(in, op1, clampi(i - 2 + k, 0, nrows - 1), j, nrows, ncols) * op2[k];
    return sum5(t);
}

// Tiled form.  A workgroup of 64 x 4 threads owns 64 rows x 4 columns of one frame: it stages the (64+4) x (4+4) input
// neighbourhood of both frames in LDS (replicate ends = clamped coordinates, as the reference's loops read them), evaluates
// every first-stage temporary the tile needs ONCE -- V5(smooth) of both frames on 64 x 8, H5(smooth) of both frames and
// H5(d1) of the second on 68 x 4 -- and then the second-stage taps from LDS.  Per output pixel that is about 70 multiply-adds
// and 4 global loads instead of 7 x 25 of each; products and left-to-right sums are the reference's, so the bits are too.
constexpr int D5_TR = 64, D5_TC = 4, D5_IR = D5_TR + 4, D5_IC = D5_TC + 4;

template <bool SND>
__global__ void __launch_bounds__(D5_TR *D5_TC)
k_derivatives5_tiled(float *o0, float *o1, float *o2, float *o3, float *o4, const float *It0, const float *It1, int nrows, int ncols,
                     size_t frame_stride)
{
    __shared__ float inA[D5_IC][D5_IR], inB[D5_IC][D5_IR];                  // [tile column][tile row]: rows contiguous
    __shared__ float VA[D5_IC][D5_TR], VB[D5_IC][D5_TR];                    // V5(smooth) at own rows, columns j-2..j+2
    __shared__ float HA[D5_TC][D5_IR], HB[D5_TC][D5_IR], HD[D5_TC][D5_IR];  // H5(smooth) a, b; H5(d1) b at rows i-2..i+2, own columns
    const int tr = threadIdx.x, tc = threadIdx.y, tid = tc * D5_TR + tr;
    const int i0 = blockIdx.x * D5_TR, j0 = blockIdx.y * D5_TC;
    const size_t fo = (size_t)blockIdx.z * frame_stride;
    const float *a = It0 + fo, *b = It1 + fo;
    for (int e = tid; e < D5_IC * D5_IR; e += D5_TR * D5_TC) {
        const int c = e / D5_IR, r = e - c * D5_IR;
        const size_t g = (size_t)clampi(j0 - 2 + c, 0, ncols - 1) * nrows + clampi(i0 - 2 + r, 0, nrows - 1);
        if (SND) inA[c][r] = a[g];
        inB[c][r] = b[g];
    }
    __syncthreads();
    for (int e = tid; e < D5_IC * D5_TR; e += D5_TR * D5_TC) { // vertical temporaries: own rows, every staged column
        const int c = e / D5_TR, r = e - c * D5_TR;
        float t[5];
        if (SND) {
#pragma unroll
            for (int k = 0; k < 5; k++) t[k] = inA[c][r + k] * SIM_SMOOTH[k];
            VA[c][r] = sum5(t);
        }
#pragma unroll
        for (int k = 0; k < 5; k++) t[k] = inB[c][r + k] * SIM_SMOOTH[k];
        VB[c][r] = sum5(t);
    }
    for (int e = tid; e < D5_TC * D5_IR; e += D5_TR * D5_TC) { // horizontal temporaries: every staged row, own columns
        const int c = e / D5_IR, r = e - c * D5_IR;
        float t[5];
        if (SND) {
#pragma unroll
            for (int k = 0; k < 5; k++) t[k] = inA[c + k][r] * SIM_SMOOTH[k];
            HA[c][r] = sum5(t);
#pragma unroll
            for (int k = 0; k < 5; k++) t[k] = inB[c + k][r] * SIM_D1[k];
            HD[c][r] = sum5(t);
        }
#pragma unroll
        for (int k = 0; k < 5; k++) t[k] = inB[c + k][r] * SIM_SMOOTH[k];
        HB[c][r] = sum5(t);
    }
    __syncthreads();
    const int i = i0 + tr, j = j0 + tc;
    if (i >= nrows || j >= ncols) return;
    const size_t pos = fo + (size_t)j * nrows + i;
    auto h5 = [&](const float (&V)[D5_IC][D5_TR], const float *op) { // H5(op) of a vertical temporary at (i, j)
        float t[5];
#pragma unroll
        for (int k = 0; k < 5; k++) t[k] = V[tc + k][tr] * op[k];
        return sum5(t);
    };
    auto v5 = [&](const float (&H)[D5_TC][D5_IR], const float *op) { // V5(op) of a horizontal temporary at (i, j)
        float t[5];
#pragma unroll
        for (int k = 0; k < 5; k++) t[k] = H[tc][tr + k] * op[k];
        return sum5(t);
    };
    if (SND) {
        o0[pos] = h5(VA, SIM_D1) * 0.50f + h5(VB, SIM_D1) * -0.50f; // Idxt (:459-463)
        o1[pos] = v5(HA, SIM_D1) * 0.50f + v5(HB, SIM_D1) * -0.50f; // Idyt (:465-469)
        o2[pos] = h5(VB, SIM_D2);                                   // Idxx (:472-473)
        o3[pos] = v5(HB, SIM_D2);                                   // Idyy (:476-477)
        o4[pos] = v5(HD, SIM_D1);                                   // Idxy (:480-481)
    } else {
        o0[pos] = It0[pos] * 0.50f + inB[tc + 2][tr + 2] * -0.50f; // Idt: TemporalConvWO2 (:44-60)
        o1[pos] = h5(VB, SIM_D1);                                   // Idx (:376-378)
        o2[pos] = v5(HB, SIM_D1);                                   // Idy (:380-382)
    }
}

} // namespace pdeip
