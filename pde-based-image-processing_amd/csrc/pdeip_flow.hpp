// pdeip_flow.hpp -- the stages the MATLAB driver runs BETWEEN the MEX calls of one pyramid level of
// the late-linearisation optical flow (matlab/optical_flow/FlowEminND_llin_2D_v10.m:208-356), as device
// kernels, so that a whole firstLoop body (warp -> derivatives -> [robust weights, diffusion weights,
// assembly, solver] x secondLoop -> median) stays in HBM instead of crossing PCIe 13 planes per call.
//
// These are restatements of MATLAB array code, not of C: there is no reference build to compare with.
// What is reproduced is MATLAB's typing (single op double -> single; OPdiffWeights casts to double) and
// the left-to-right order of each expression; oracle/matlab_side.py is the numpy statement of the same
// and the tests compare bit for bit against it.  ("parity unpinned" with respect to MATLAB itself.)
#pragma once
#include <hip/hip_runtime.h>

#include "pdeip_models.hpp"
#include "pdeip_pointwise.hpp"

namespace pdeip {

// single(X+U), single(Y+V) of the warp (FlowEminND_llin_2D_v10.m:223): X,Y = meshgrid(1:cols,1:rows)
__global__ void k_flow_coords(float *X, float *Y, const float *U, const float *V, int nrows, int ncols)
{
    PDEIP_PIXEL_INDEX();
    X[pos] = (float)(j + 1) + U[pos];
    Y[pos] = (float)(i + 1) + V[pos];
}

// One data term of the robust assembly (:283-327): for each channel c
//   OPnorm = (It - Ix.*dU - Iy.*dV).^2;  gD = b./(alpha*sqrt(OPnorm+0.00001))          (all single)
//   M = Iy.*Ix; Cu = It.*Ix; Cv = It.*Iy; Du = Ix.*Ix; Dv = Iy.*Iy                     (:236-240)
// and the five planes accumulate nansum(cat(3, M.*gD, ...), 3): NaN products are skipped, in channel
// order, first term then second term; a pixel whose products are all NaN gets 0 (nansum's empty sum).
// A GRADMAG second term (:253-258, :291-293; sndTerm 'gradmag', what runme.m asks for) carries the five second-order planes
// of SndDerivatives5 instead: OPnorm = (Ixt - Ixx.*dU - Ixy.*dV).^2 + (Iyt - Ixy.*dU - Iyy.*dV).^2,
//   M = Ixy.*(Ixx+Iyy); Cu = Ixt.*Ixx + Iyt.*Ixy; Cv = Ixt.*Ixy + Iyt.*Iyy; Du = Ixx.*Ixx + Ixy.*Ixy; Dv = Ixy.*Ixy + Iyy.*Iyy.
struct FlowTerm {
    const float *It, *Ix, *Iy; // [nrows x ncols x C]; GRADMAG: Ixt, Iyt, Ixx
    int C;
    float b;
    const float *Iyy = nullptr, *Ixy = nullptr; // GRADMAG only (Ixy != nullptr marks the kind)
};

__device__ __forceinline__ void nan_add(float &acc, float v)
{
    if (!is_nan(v)) acc = acc + v;
}

__device__ __forceinline__ void opdiffweights_pixel(float *wW, float *wN, float *wS, float *wE, const float *U, const float *V,
                                                    const float *dU, const float *dV, int i, int j, size_t pos, int nrows, int ncols);

// One inner iteration's coefficient planes in one pass: the robust data terms below and, when the weight planes are given,
// OPdiffWeights(U+dU, V+dV) of the same iterate (both only read dU, dV; the weights also their neighbours).
struct FlowWeightsOut {
    float *wW = nullptr, *wN = nullptr, *wS = nullptr, *wE = nullptr;
    const float *U = nullptr, *V = nullptr;
};

__global__ void k_flow_assemble(float *MGd, float *CuGd, float *CvGd, float *DuGd, float *DvGd, FlowTerm t1, FlowTerm t2,
                                const float *dU, const float *dV, float alpha, int nrows, int ncols, FlowWeightsOut W)
{
    PDEIP_PIXEL_INDEX();
    if (W.wW) opdiffweights_pixel(W.wW, W.wN, W.wS, W.wE, W.U, W.V, dU, dV, i, j, pos, nrows, ncols);
    const size_t n = (size_t)nrows * ncols;
    const float du = dU[pos], dv = dV[pos];
    float m = 0.0f, cu = 0.0f, cv = 0.0f, Du = 0.0f, Dv = 0.0f;
#pragma unroll
    for (int term = 0; term < 2; ++term) {
        const FlowTerm &t = term == 0 ? t1 : t2;
        for (int c = 0; c < t.C; ++c) {
            const size_t p = (size_t)c * n + pos;
            if (t.Ixy) {
                const float Ixt = t.It[p], Iyt = t.Ix[p], Ixx = t.Iy[p], Iyy = t.Iyy[p], Ixy = t.Ixy[p];
                float r1 = Ixt - Ixx * du;
                r1 = r1 - Ixy * dv;
                float r2 = Iyt - Ixy * du;
                r2 = r2 - Iyy * dv;
                const float gD = t.b / (alpha * sqrtf((r1 * r1 + r2 * r2) + 0.00001f));
                nan_add(m, (Ixy * (Ixx + Iyy)) * gD);
                nan_add(cu, (Ixt * Ixx + Iyt * Ixy) * gD);
                nan_add(cv, (Ixt * Ixy + Iyt * Iyy) * gD);
                nan_add(Du, (Ixx * Ixx + Ixy * Ixy) * gD);
                nan_add(Dv, (Ixy * Ixy + Iyy * Iyy) * gD);
                continue;
            }
            const float It = t.It[p], Ix = t.Ix[p], Iy = t.Iy[p];
            float r = It - Ix * du; // (It - Ix.*dU) - Iy.*dV
            r = r - Iy * dv;
            const float opnorm = r * r;
            const float gD = t.b / (alpha * sqrtf(opnorm + 0.00001f));
            nan_add(m, (Iy * Ix) * gD);
            nan_add(cu, (It * Ix) * gD);
            nan_add(cv, (It * Iy) * gD);
            nan_add(Du, (Ix * Ix) * gD);
            nan_add(Dv, (Iy * Iy) * gD);
        }
    }
    MGd[pos] = m;
    CuGd[pos] = cu;
    CvGd[pos] = cv;
    DuGd[pos] = Du;
    DvGd[pos] = Dv;
}

// Spatial a-priori slice of the assembly (:262-270, :301-318, :321-325): with a constraint field Us (a MATLAB double array)
//   ASCu = Us - U;  APUnorm = (Us - U - dU).^2;  gSu = gammaS./(alpha*(1 + APUnorm/ASdiff^2))
// and nansum(cat(3, ..., ASCu.*gSu), 3) / nansum(cat(3, ..., 1.*gSu), 3) append one more slice to CuGd / DuGd.  MATLAB's
// typing decides the arithmetic: U is double only before the first median of the coarsest scale (u_double), dU is the double
// zeros of :272 in the first inner iteration (du_double) and single afterwards; a double meeting a single is rounded first.
__global__ void k_flow_apriori(float *CGd, float *DGd, const double *Us, const float *U, const float *dU, double gammaS, double alpha,
                               double asd2, int u_double, int du_double, int nrows, int ncols)
{
    PDEIP_PIXEL_INDEX();
    float cS, dS;
    if (u_double && du_double) {
        const double asc = Us[pos] - (double)U[pos];
        const double t = asc - (double)dU[pos];
        const double gS = gammaS / (alpha * (1.0 + (t * t) / asd2));
        cS = (float)(asc * gS);
        dS = (float)gS;
    } else {
        const float asc = u_double ? (float)(Us[pos] - (double)U[pos]) : (float)Us[pos] - U[pos];
        const float t = asc - dU[pos];
        const float gS = (float)gammaS / ((float)alpha * (1.0f + (t * t) / (float)asd2));
        cS = asc * gS;
        dS = gS;
    }
    float c = CGd[pos], d = DGd[pos];
    nan_add(c, cS);
    nan_add(d, dS);
    CGd[pos] = c;
    DGd[pos] = d;
}

// exp() of the disparity driver's influence function: the algorithm of oracle/matlab_side.py det_exp, operation for operation
// (IEEE double, no FMA: this file is compiled -ffp-contract=off), so host statement and kernel agree bit for bit.
__device__ __forceinline__ double det_exp(double x)
{
    constexpr double LN2_HI = 6.93147180369123816490e-01, LN2_LO = 1.90821492927058770002e-10, INV_LN2 = 1.44269504088896338700e+00;
    constexpr double C[14] = {1.0 / 1, 1.0 / 1, 1.0 / 2, 1.0 / 6, 1.0 / 24, 1.0 / 120, 1.0 / 720, 1.0 / 5040, 1.0 / 40320, 1.0 / 362880,
                              1.0 / 3628800, 1.0 / 39916800, 1.0 / 479001600, 1.0 / 6227020800.0};
    x = x < -700.0 ? -700.0 : x; // NaN stays NaN (the comparison is false)
    const double k = floor(x * INV_LN2 + 0.5);
    const double r = (x - k * LN2_HI) - k * LN2_LO;
    double p = C[13];
#pragma unroll
    for (int i = 12; i >= 0; --i) p = p * r + C[i];
    return ldexp(p, (int)k);
}

// Spatial a-priori slice of the DISPARITY assembly (DispEminND_llin_2D.m:246-248, :277-284, :291-292):
//   ASCu = USap - U;  ASDu = 1;  gS = gammaS/alpha * exp(-(USap - U - dU).^2 / ASdiff^2)
// appended to CuGd / DuGd by the driver's plain sum() (a NaN propagates).  Typing as in k_flow_apriori; the single-precision
// exp is defined as the double algorithm above rounded once.
__global__ void k_disp_apriori(float *CGd, float *DGd, const double *Us, const float *U, const float *dU, double gammaS, double alpha,
                               double asd2, int u_double, int du_double, int nrows, int ncols)
{
    PDEIP_PIXEL_INDEX();
    const double k = gammaS / alpha;
    float cS, dS;
    if (u_double && du_double) {
        const double asc = Us[pos] - (double)U[pos];
        const double t = asc - (double)dU[pos];
        const double gS = k * det_exp(-(t * t) / asd2);
        cS = (float)(asc * gS);
        dS = (float)gS;
    } else {
        const float asc = u_double ? (float)(Us[pos] - (double)U[pos]) : (float)Us[pos] - U[pos];
        const float t = asc - dU[pos];
        const float arg = -(t * t) / (float)asd2;
        const float gS = (float)k * (float)det_exp((double)arg);
        cS = asc * gS;
        dS = gS;
    }
    CGd[pos] = CGd[pos] + cS;
    DGd[pos] = DGd[pos] + dS;
}

// Disparity twin of the assembly (matlab/disparity/DispEminND_llin_2D.m:258-293): one unknown, and a plain
// sum() over the channels -- a NaN (out-of-range warp) propagates into CuGd/DuGd, where the solver's
// isnan(Cu) test picks it up (disparitySolvers.c:66).
__global__ void k_disp_assemble(float *CuGd, float *DuGd, FlowTerm t1, FlowTerm t2, const float *dU, float alpha, int nrows, int ncols)
{
    PDEIP_PIXEL_INDEX();
    const size_t n = (size_t)nrows * ncols;
    const float du = dU[pos];
    float cu = 0.0f, Du = 0.0f;
    bool first = true;
#pragma unroll
    for (int term = 0; term < 2; ++term) {
        const FlowTerm &t = term == 0 ? t1 : t2;
        for (int c = 0; c < t.C; ++c) {
            const size_t p = (size_t)c * n + pos;
            float a, b;
            if (t.Ixy) { // GRADMAG (DispEminND_llin_2D.m:236-238, :271): Iy holds Ixx
                const float Ixt = t.It[p], Iyt = t.Ix[p], Ixx = t.Iy[p], Ixy = t.Ixy[p];
                const float r1 = Ixt - Ixx * du, r2 = Iyt - Ixy * du;
                const float gD = t.b / (alpha * sqrtf((r1 * r1 + r2 * r2) + 0.00001f));
                a = (Ixt * Ixx + Iyt * Ixy) * gD;
                b = (Ixx * Ixx + Ixy * Ixy) * gD;
            } else {
                const float It = t.It[p], Ix = t.Ix[p];
                const float r = It - Ix * du;
                const float gD = t.b / (alpha * sqrtf(r * r + 0.00001f));
                a = (It * Ix) * gD;
                b = (Ix * Ix) * gD;
            }
            cu = first ? a : cu + a; // sum(cat(3, ...), 3): slices added in order
            Du = first ? b : Du + b;
            first = false;
        }
    }
    CuGd[pos] = cu;
    DuGd[pos] = Du;
}

// rgb2grad (FlowEminND_llin_2D_v10.m:368-381): per input frame f, out(:,:,2f-1) = imfilter(IN, [1 0 -1], 'replicate') (west minus
// east), out(:,:,2f) = imfilter(IN, [1 0 -1]', 'replicate') (north minus south).  blockIdx.z = input frame.
__global__ void k_rgb2grad(float *out, const float *in, int nrows, int ncols)
{
    PDEIP_PIXEL_INDEX();
    const size_t n = (size_t)nrows * ncols;
    const float *P = in + blockIdx.z * n;
    const int jw = max(j - 1, 0), je = min(j + 1, ncols - 1), in_ = max(i - 1, 0), is = min(i + 1, nrows - 1);
    out[(2 * blockIdx.z) * n + pos] = P[(size_t)jw * nrows + i] - P[(size_t)je * nrows + i];
    out[(2 * blockIdx.z + 1) * n + pos] = P[(size_t)j * nrows + in_] - P[(size_t)j * nrows + is];
}

// out = A + B (single): DdiffWeights(single(U+dU), ...) takes the sum as its input (:283)
__global__ void k_add2(float *out, const float *A, const float *B, int nrows, int ncols)
{
    PDEIP_PIXEL_INDEX();
    out[pos] = A[pos] + B[pos];
}

// single(1 ./ sqrt(x)) of a double x, i.e. the correctly rounded double square root, the correctly rounded double quotient,
// then one rounding to single -- without the two long IEEE sequences in the common case.  v_rsq_f64 plus two Newton steps
// is within 2 ulp (double) of 1/sqrt(x), and so is the exact sequence's quotient, so both round to the same single unless a
// single rounding boundary (a double whose low 29 mantissa bits are 0x10000000) lies within a few ulp of the estimate; those
// values (and non-finite, huge or tiny x, where single goes subnormal) take the exact sequence.  Same bits, always.
__device__ __forceinline__ float single_inv_sqrt(double x)
{
    double y = __builtin_amdgcn_rsq(x);
    double e = __builtin_fma(-(x * y), y, 1.0);
    y = __builtin_fma(0.5 * y, e, y);
    e = __builtin_fma(-(x * y), y, 1.0);
    y = __builtin_fma(0.5 * y, e, y);
    const unsigned lo = (unsigned)__double2loint(y) & 0x1fffffffu;
    const bool safe = (lo - 0x10000000u + 16u > 32u) && (x > 1e-30) && (x < 1e60); // false for NaN
    if (safe) return (float)y;
    return (float)(1.0 / sqrt(x));
}

// OPdiffWeights (:389-433) on (U+dU, V+dV): 6-point discretisation, evaluated in double like the MATLAB
// function (it casts its inputs), circshift wrap-around at the frame edges included, cast to single on
// the way into the solver call.  ver = imfilter(.,[0.25 0 -0.25]','replicate') = 0.25*(north) - 0.25*(south),
// hor likewise with west/east.
// The weight between two pixels is one number: wW(i,j) and wE(i,j-1) are the same expression (the difference enters
// squared, the two cross terms are summed in the other order -- both exact symmetries in IEEE arithmetic -- and the
// circshift pairs the same pixels across the frame edge), likewise wN(i,j) and wS(i-1,j).  A thread therefore evaluates the
// east and the south weight of its pixel and also stores them as the west weight of its east neighbour and the north weight
// of its south neighbour: half the double-precision work of evaluating four directions per pixel, every plane entry
// written exactly once.
__device__ __forceinline__ void opdiffweights_pixel(float *wW, float *wN, float *wS, float *wE, const float *U, const float *V,
                                                    const float *dU, const float *dV, int i, int j, size_t pos, int nrows, int ncols)
{
    // value of field f (0: U+dU, 1: V+dV) at (ii,jj), double
    auto F = [&](int f, int ii, int jj) -> double {
        const size_t p = (size_t)jj * nrows + ii;
        if (!dU) return (double)(f == 0 ? U[p] : V[p]);                     // OPdiffWeights(U, V) of the early-linearisation drivers
        const float s = (f == 0 ? U[p] : V[p]) + (f == 0 ? dU[p] : dV[p]); // single(U) + single(dU), then double()
        return (double)s;
    };
    auto clampi = [](int v, int hi) { return v < 0 ? 0 : (v > hi ? hi : v); };
    const int je = j == ncols - 1 ? 0 : j + 1, is = i == nrows - 1 ? 0 : i + 1; // circshift wraps
    const int iu = clampi(i - 1, nrows - 1), id = clampi(i + 1, nrows - 1);     // 'replicate' rows / columns of the cross stencils
    const int jl = clampi(j - 1, ncols - 1), jr = clampi(j + 1, ncols - 1);
    double accE = 0.0, accS = 0.0;
#pragma unroll
    for (int f = 0; f < 2; ++f) { // (U_d - U)^2 + (Uver + Uver_d)^2 + (V_d - V)^2 + (Vver + Vver_d)^2, left to right
        const double c = F(f, i, j), n = F(f, iu, j), s = F(f, id, j), w = F(f, i, jl), e = F(f, i, jr);
        const double ver_c = 0.25 * n - 0.25 * s, hor_c = 0.25 * w - 0.25 * e;
        // east neighbour (i, je): its vertical stencil; south neighbour (is, j): its horizontal stencil
        const double ce = F(f, i, je), cs = F(f, is, j);
        const double ver_e = 0.25 * F(f, iu, je) - 0.25 * F(f, id, je);
        const double hor_s = 0.25 * F(f, is, jl) - 0.25 * F(f, is, jr);
        const double dE = ce - c, dS = cs - c;
        const double xE = ver_c + ver_e, xS = hor_c + hor_s;
        accE = f == 0 ? dE * dE : accE + dE * dE;
        accE = accE + xE * xE;
        accS = f == 0 ? dS * dS : accS + dS * dS;
        accS = accS + xS * xS;
    }
    const float we = single_inv_sqrt(accE + 0.00001), ws = single_inv_sqrt(accS + 0.00001);
    wE[pos] = we;
    wW[(size_t)je * nrows + i] = we;
    wS[pos] = ws;
    wN[(size_t)j * nrows + is] = ws;
}

__global__ void k_flow_opdiffweights(float *wW, float *wN, float *wS, float *wE, const float *U, const float *V,
                                     const float *dU, const float *dV, int nrows, int ncols)
{
    PDEIP_PIXEL_INDEX();
    opdiffweights_pixel(wW, wN, wS, wE, U, V, dU, dV, i, j, pos, nrows, ncols);
}

// medfilt2(A + B, [3 3], 'symmetric') (:352-353): exact selection of the 5th of 9, symmetric padding
// (the edge pixel is mirrored, i.e. index -1 -> 0 and n -> n-1).
__device__ __forceinline__ void cswap(float &a, float &b)
{
    const float lo = fminf(a, b), hi = fmaxf(a, b);
    a = lo;
    b = hi;
}

// blockIdx.z = 1 filters the second field (out1 = medfilt2(A1 + B1)): the drivers filter U+dU and V+dV back to back
__global__ void k_median3_sum(float *out, const float *A, const float *B, int nrows, int ncols, float *out1 = nullptr, const float *A1 = nullptr,
                              const float *B1 = nullptr)
{
    PDEIP_PIXEL_INDEX();
    if (blockIdx.z == 1) {
        out = out1;
        A = A1;
        B = B1;
    }
    float v[9];
#pragma unroll
    for (int dj = -1; dj <= 1; ++dj)
#pragma unroll
        for (int di = -1; di <= 1; ++di) {
            const int ii = min(max(i + di, 0), nrows - 1), jj = min(max(j + dj, 0), ncols - 1);
            const size_t p = (size_t)jj * nrows + ii;
            v[(dj + 1) * 3 + di + 1] = B ? A[p] + B[p] : A[p];
        }
    // 19-exchange median-of-9 network
    cswap(v[1], v[2]); cswap(v[4], v[5]); cswap(v[7], v[8]);
    cswap(v[0], v[1]); cswap(v[3], v[4]); cswap(v[6], v[7]);
    cswap(v[1], v[2]); cswap(v[4], v[5]); cswap(v[7], v[8]);
    cswap(v[0], v[3]); cswap(v[5], v[8]); cswap(v[4], v[7]);
    cswap(v[3], v[6]); cswap(v[1], v[4]); cswap(v[2], v[5]);
    cswap(v[4], v[7]); cswap(v[4], v[2]); cswap(v[6], v[4]);
    cswap(v[4], v[2]);
    out[pos] = v[4];
}

// ------------------------------------------------------------------------------------------------
// Horn-Schunck with early linearisation: the per-scale data terms of FlowEminHS_elin_2D_v10.m:133-164.
// Ist = (It0+It1).*0.55, Idt = It0-It1; separable 5-tap filters (imfilter(...,'replicate','conv'), two passes with
// the first pass stored as single): Idx, Idy, Idxx, Idyy, Idxy of Ist and Idxt, Idyt as differences of the
// filtered frames; then per channel
//   M  = b1*Idy.*Idx + b2*Idxy.*(Idxx+Idyy)        Cu = b1*Idt.*Idx + b2*(Idxt.*Idxx + Idyt.*Idxy)
//   Cv = b1*Idt.*Idy + b2*(Idxt.*Idxy + Idyt.*Idyy)   Du = b1*Idx.*Idx + b2*(Idxx.*Idxx + Idxy.*Idxy)
//   Dv = b1*Idy.*Idy + b2*(Idxy.*Idxy + Idyy.*Idyy)
// summed over the channels.  Taps are summed left to right over the flipped kernel (our definition of imfilter's
// unspecified order); every pixel re-evaluates the first pass where it needs it, so nothing intermediate touches HBM.
// ------------------------------------------------------------------------------------------------
__constant__ float HS_PRE[5] = {0.037659f, 0.249724f, 0.439911f, 0.249724f, 0.037659f};      // prefilter_spa (:77), symmetric
__constant__ float HS_D1F[5] = {-0.104550f, -0.292315f, 0.0f, 0.292315f, 0.104550f};          // O_dx (:78) flipped by 'conv'
__constant__ float HS_D2[5] = {0.232905f, 0.002668f, -0.471147f, 0.002668f, 0.232905f};       // O_dxx (:80), symmetric

struct HsImage { // one channel of one "image" the filters run on: a*P0 (+ b*P1)
    const float *P0, *P1;
    int mode; // 0: (P0+P1)*0.55   1: P0   2: P1   3: (P0+P1)*0.55/255 (the FAS driver's 0..255 frames)
    int nrows, ncols;
    __device__ __forceinline__ float at(int i, int j) const
    {
        i = i < 0 ? 0 : (i > nrows - 1 ? nrows - 1 : i);
        j = j < 0 ? 0 : (j > ncols - 1 ? ncols - 1 : j);
        const size_t p = (size_t)j * nrows + i;
        if (mode == 3) return ((P0[p] + P1[p]) * 0.55f) / 255.0f;
        return mode == 0 ? (P0[p] + P1[p]) * 0.55f : (mode == 1 ? P0[p] : P1[p]);
    }
};

// first pass along the rows index i (a column vector kernel), evaluated at (i, j) with j clamped by the caller
__device__ __forceinline__ float hs_v5(const HsImage &A, const float *k, int i, int j)
{
    float s = k[0] * A.at(i - 2, j);
    s = s + k[1] * A.at(i - 1, j);
    s = s + k[2] * A.at(i, j);
    s = s + k[3] * A.at(i + 1, j);
    s = s + k[4] * A.at(i + 2, j);
    return s;
}
__device__ __forceinline__ float hs_h5(const HsImage &A, const float *k, int i, int j)
{
    float s = k[0] * A.at(i, j - 2);
    s = s + k[1] * A.at(i, j - 1);
    s = s + k[2] * A.at(i, j);
    s = s + k[3] * A.at(i, j + 1);
    s = s + k[4] * A.at(i, j + 2);
    return s;
}
// vertical pass with kv, then horizontal pass with kh over the stored (replicated) first-pass result
__device__ __forceinline__ float hs_vh(const HsImage &A, const float *kv, const float *kh, int i, int j)
{
    auto cj = [&](int v) { return v < 0 ? 0 : (v > A.ncols - 1 ? A.ncols - 1 : v); };
    float s = kh[0] * hs_v5(A, kv, i, cj(j - 2));
    s = s + kh[1] * hs_v5(A, kv, i, cj(j - 1));
    s = s + kh[2] * hs_v5(A, kv, i, j);
    s = s + kh[3] * hs_v5(A, kv, i, cj(j + 1));
    s = s + kh[4] * hs_v5(A, kv, i, cj(j + 2));
    return s;
}
// horizontal pass with kh, then vertical pass with kv
__device__ __forceinline__ float hs_hv(const HsImage &A, const float *kh, const float *kv, int i, int j)
{
    auto ci = [&](int v) { return v < 0 ? 0 : (v > A.nrows - 1 ? A.nrows - 1 : v); };
    float s = kv[0] * hs_h5(A, kh, ci(i - 2), j);
    s = s + kv[1] * hs_h5(A, kh, ci(i - 1), j);
    s = s + kv[2] * hs_h5(A, kh, i, j);
    s = s + kv[3] * hs_h5(A, kh, ci(i + 1), j);
    s = s + kv[4] * hs_h5(A, kh, ci(i + 2), j);
    return s;
}

__global__ void k_hs_assemble(float *MGd, float *CuGd, float *CvGd, float *DuGd, float *DvGd, const float *It0, const float *It1,
                              int C, float b1, float b2, int nrows, int ncols)
{
    PDEIP_PIXEL_INDEX();
    const size_t n = (size_t)nrows * ncols;
    float m = 0.0f, cu = 0.0f, cv = 0.0f, du = 0.0f, dv = 0.0f;
    for (int c = 0; c < C; ++c) {
        const HsImage S{It0 + c * n, It1 + c * n, 0, nrows, ncols}, A0{It0 + c * n, It1 + c * n, 1, nrows, ncols},
            A1{It0 + c * n, It1 + c * n, 2, nrows, ncols};
        const float Idt = It0[c * n + pos] - It1[c * n + pos];
        const float Idx = hs_vh(S, HS_PRE, HS_D1F, i, j);   // smooth down the column, derive along the row
        const float Idy = hs_hv(S, HS_PRE, HS_D1F, i, j);   // smooth along the row, derive down the column
        const float Idxx = hs_vh(S, HS_PRE, HS_D2, i, j);
        const float Idyy = hs_hv(S, HS_PRE, HS_D2, i, j);
        const float Idxy = hs_hv(S, HS_D1F, HS_D1F, i, j);  // imfilter(imfilter(Ist,O_dx),O_dy)
        const float Idxt = hs_vh(A0, HS_PRE, HS_D1F, i, j) - hs_vh(A1, HS_PRE, HS_D1F, i, j);
        const float Idyt = hs_hv(A0, HS_PRE, HS_D1F, i, j) - hs_hv(A1, HS_PRE, HS_D1F, i, j);
        const float Mc = (b1 * Idy) * Idx + (b2 * Idxy) * (Idxx + Idyy);
        const float Cuc = (b1 * Idt) * Idx + b2 * (Idxt * Idxx + Idyt * Idxy);
        const float Cvc = (b1 * Idt) * Idy + b2 * (Idxt * Idxy + Idyt * Idyy);
        const float Duc = (b1 * Idx) * Idx + b2 * (Idxx * Idxx + Idxy * Idxy);
        const float Dvc = (b1 * Idy) * Idy + b2 * (Idxy * Idxy + Idyy * Idyy);
        m = c ? m + Mc : Mc;   // sum(.,3)
        cu = c ? cu + Cuc : Cuc;
        cv = c ? cv + Cvc : Cvc;
        du = c ? du + Duc : Duc;
        dv = c ? dv + Dvc : Dvc;
    }
    MGd[pos] = m;
    CuGd[pos] = cu;
    CvGd[pos] = cv;
    DuGd[pos] = du;
    DvGd[pos] = dv;
}

} // namespace pdeip
