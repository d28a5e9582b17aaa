// pdeip_sym.hpp -- the stages of the symmetric stereo driver (matlab/disparity/DispEminND_llin_sym_2D.m) that the other
// drivers do not have: warping one view's disparity into the other (interp2, :140-141), the symmetry terms built from it
// (:156-175) and the assembly of data + symmetry terms for Disp_sor_llin_sym4_2d (:189-222).
//
// MATLAB typing matters here: U and the warped disparities are double arrays, the image derivatives single, and the
// increment dU is the double zeros of :177-178 in the first inner iteration and the solver's single output afterwards --
// so the symmetry weights are evaluated in double once and in single from then on (`single op double -> single`).
// interp2 is restated by its documented default (linear, NaN outside the grid).  oracle/matlab_side.py (sym_*) is the
// numpy statement the tests compare with bit for bit; parity with MATLAB itself is unpinned.
#pragma once
#include <hip/hip_runtime.h>

#include "pdeip_pointwise.hpp"

namespace pdeip {

__constant__ double SYM_PRE[5] = {0.037659, 0.249724, 0.439911, 0.249724, 0.037659};   // prefilter_spa (:71)
__constant__ double SYM_D1F[5] = {-0.104550, -0.292315, 0.0, 0.292315, 0.104550};      // O_dx (:73) flipped by 'conv'

// out = interp2(X, Y, U, X+Uq, Y): the query rows are the grid rows, so only x is interpolated
__global__ void k_sym_warp_flow(double *out, const float *U, const float *Uq, int nrows, int ncols)
{
    PDEIP_PIXEL_INDEX();
    const double xq = (double)(j + 1) + (double)Uq[pos];
    if (!(xq >= 1.0 && xq <= (double)ncols)) { // also a NaN query
        out[pos] = __longlong_as_double(0x7ff8000000000000LL);
        return;
    }
    double j0 = floor(xq);
    if (j0 > (double)(ncols - 1)) j0 = (double)(ncols - 1); // xq == ncols: the last column itself (s = 1)
    const double s = xq - j0;
    const int c0 = (int)j0 - 1, c1 = min(c0 + 1, ncols - 1);
    out[pos] = (double)U[(size_t)c0 * nrows + i] * (1.0 - s) + (double)U[(size_t)c1 * nrows + i] * s;
}

// Udt = (U + Uw)*0.5; Udx = imfilter(imfilter(Uw, prefilter_spa', 'replicate', 'conv'), O_dx, 'replicate', 'conv');
// CuS = Udt.*(1+Udx); DuS = 1 + Udx + Udx + Udx.*Udx   (all double; the first pass is re-evaluated per tap)
__global__ void k_sym_flow_terms(double *Udt, double *Udx, double *CuS, double *DuS, const float *U, const double *Uw, int nrows, int ncols)
{
    PDEIP_PIXEL_INDEX();
    auto ci = [&](int v) { return v < 0 ? 0 : (v > nrows - 1 ? nrows - 1 : v); };
    auto cj = [&](int v) { return v < 0 ? 0 : (v > ncols - 1 ? ncols - 1 : v); };
    auto v5 = [&](int jj) {
        const double *c = Uw + (size_t)jj * nrows;
        double s = SYM_PRE[0] * c[ci(i - 2)];
        s = s + SYM_PRE[1] * c[ci(i - 1)];
        s = s + SYM_PRE[2] * c[i];
        s = s + SYM_PRE[3] * c[ci(i + 1)];
        s = s + SYM_PRE[4] * c[ci(i + 2)];
        return s;
    };
    double dx = SYM_D1F[0] * v5(cj(j - 2));
    dx = dx + SYM_D1F[1] * v5(cj(j - 1));
    dx = dx + SYM_D1F[2] * v5(j);
    dx = dx + SYM_D1F[3] * v5(cj(j + 1));
    dx = dx + SYM_D1F[4] * v5(cj(j + 2));
    const double dt = ((double)U[pos] + Uw[pos]) * 0.5;
    Udt[pos] = dt;
    Udx[pos] = dx;
    CuS[pos] = dt * (1.0 + dx);
    DuS[pos] = ((1.0 + dx) + dx) + dx * dx;
}

struct SymData { // one view's image derivatives, [nrows x ncols x C] single
    const float *Idt, *Idx, *Idxt, *Idyt, *Idxx, *Idxy;
    int C;
};

// CuG = sum(cat(3, gD.*CuD, -gSYM.*CuS), 3), DuG = sum(cat(3, gD.*DuD, gSYM.*DuS), 3)   (:189-222)
__global__ void k_sym_assemble(float *CuG, float *DuG, SymData d, const double *Udt, const double *Udx, const double *CuS, const double *DuS,
                               const float *dU, float b1, float b2, float alpha, double kS, double sr2, int first, int nrows, int ncols)
{
    PDEIP_PIXEL_INDEX();
    const size_t n = (size_t)nrows * ncols;
    const float du = dU[pos];
    float cu = 0.0f, dd = 0.0f;
    for (int c = 0; c < d.C; ++c) {
        const size_t p = (size_t)c * n + pos;
        const float Idt = d.Idt[p], Idx = d.Idx[p], Idxt = d.Idxt[p], Idyt = d.Idyt[p], Idxx = d.Idxx[p], Idxy = d.Idxy[p];
        const float r1 = Idt - Idx * du, r2 = Idxt - Idxx * du, r3 = Idyt - Idxy * du;
        const float opnorm = b1 * (r1 * r1) + b2 * ((r2 * r2) + (r3 * r3));
        const float gD = 1.0f / (alpha * sqrtf(opnorm + 0.00001f));
        const float CuD = (b1 * Idt) * Idx + b2 * (Idxt * Idxx + Idyt * Idxy);
        const float DuD = (b1 * Idx) * Idx + b2 * (Idxx * Idxx + Idxy * Idxy);
        const float a = gD * CuD, b = gD * DuD;
        cu = c ? cu + a : a;
        dd = c ? dd + b : b;
    }
    float cS, dS;
    if (first) { // dU is still double: everything in double, rounded when cat() meets the single slices
        const double du64 = (double)du;
        const double sn = (du64 + Udt[pos]) + Udx[pos] * du64;
        const double gS = kS / (1.0 + (sn * sn) / sr2);
        cS = (float)((-gS) * CuS[pos]);
        dS = (float)(gS * DuS[pos]);
    } else {
        const float sn = (du + (float)Udt[pos]) + (float)Udx[pos] * du;
        const float gS = (float)kS / (1.0f + (sn * sn) / (float)sr2);
        cS = (-gS) * (float)CuS[pos];
        dS = gS * (float)DuS[pos];
    }
    CuG[pos] = cu + cS;
    DuG[pos] = dd + dS;
}

} // namespace pdeip
