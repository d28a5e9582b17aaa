// pdeip_sor_rb.hpp -- one fused red+black SOR sweep for the 5-point solvers (gfx950).
//
// Layout reminder: the buffers are MATLAB column-major, i.e. row-major [ncols][nrows] seen
// from here; the contiguous direction is the image row index i.
//
// Design (register marching, no LDS):
//   * one wave owns a unit = 248 image rows x TJ image columns.  Lane l holds four
//     consecutive rows r..r+3 of the current column (r = 248*a - 4 + 4*l), so every plane is
//     read with one 16-byte load per lane = 1 KiB per wave instruction, fully coalesced.
//     Lanes 0 and 63 are halo lanes (4 rows each) that are recomputed, never stored.
//   * the wave marches along the columns keeping a window of columns in registers:
//     O(c-1..c+1) old values, R(c-2..c) values after the red half-sweep.  Per step it
//     computes the red update of column c and the black update of column c-1, then stores
//     the finished column c-1.  North/south neighbours come from the lane's own float4 or
//     from the adjacent lane through a wavefront shuffle.
//   * colour of pixel (i,j) is (i + j + col0) & 1; because r is a multiple of 4 the red
//     pixels of a column are elements {p, p+2} of every lane with p = (c + col0) & 1, so a
//     half-sweep is branch-uniform across the wave.
//   * sweeps ping-pong between two iterate buffers (in -> out): no inter-workgroup race,
//     each plane is read once and the iterate written once per sweep (plus the unit halo).
//   * the image border is never relaxed; the unit that finishes column 1 / ncols-2 (row 1 /
//     nrows-2) also writes the replicated border cells, which reproduces the reference's
//     rows-then-columns border fill (opticalflowSolvers.c:161-179).
#pragma once
#include "pdeip_models.hpp"

namespace pdeip {

constexpr int RB_OWN_ROWS = 248; // 62 storing lanes x 4 rows
constexpr int RB_WAVES_PER_BLOCK = 4;
#ifndef PDEIP_RB_NT
#define PDEIP_RB_NT 0 /* measured at 4K: non-temporal coefficient loads are ~5 % slower than default-policy loads */
#endif
constexpr bool RB_NT_COEF = PDEIP_RB_NT != 0;

// Loads rows r..r+3 of column c.  Always issued, never branched on: a bounds check around a load
// makes the compiler wait for it inside the branch, which serialises the thirteen loads of a step.
// Out-of-image lanes/columns read a clamped (valid) address instead; their values are never used
// for a stored result (they only feed border cells, which are not relaxed).
template <bool VEC, bool NT = false>
__device__ __forceinline__ void rb_load4(float (&d)[4], const float *__restrict__ p, int c, int r,
                                         int nrows, int ncols)
{
    const int cc = c < 0 ? 0 : (c > ncols - 1 ? ncols - 1 : c);
    const float *q = p + (size_t)cc * nrows;
    if (VEC) {
        const int rr = r < 0 ? 0 : (r > nrows - 4 ? nrows - 4 : r);
        typedef float v4f __attribute__((ext_vector_type(4)));
        // NT: coefficient planes are streamed exactly once per sweep -- non-temporal loads keep them from
        // displacing the iterate columns that neighbouring units re-read
        const v4f t = NT ? __builtin_nontemporal_load(reinterpret_cast<const v4f *>(q + rr))
                         : *reinterpret_cast<const v4f *>(q + rr);
        d[0] = t.x;
        d[1] = t.y;
        d[2] = t.z;
        d[3] = t.w;
    } else {
#pragma unroll
        for (int e = 0; e < 4; e++) {
            const int i = r + e;
            d[e] = q[i < 0 ? 0 : (i > nrows - 1 ? nrows - 1 : i)];
        }
    }
}

template <bool VEC>
__device__ __forceinline__ void rb_store4(const float (&d)[4], float *__restrict__ p, int c, int r,
                                          int nrows)
{
    float *q = p + (size_t)c * nrows;
    if (VEC) {
        if (r >= 0 && r < nrows) *reinterpret_cast<float4 *>(q + r) = make_float4(d[0], d[1], d[2], d[3]);
    } else {
#pragma unroll
        for (int e = 0; e < 4; e++) {
            const int i = r + e;
            if (i >= 0 && i < nrows) q[i] = d[e];
        }
    }
}

// Relax elements {E0, E0+2} of column C (the pixels of one colour) in place.
template <class Mdl, int E0>
__device__ __forceinline__ void rb_phase(float (&C)[Mdl::NIT][4], const float (&W)[Mdl::NIT][4],
                                         const float (&E)[Mdl::NIT][4],
                                         const float (&rC)[at_least_one<Mdl::NRO>::value][4],
                                         const float (&rW)[at_least_one<Mdl::NRO>::value][4],
                                         const float (&rE)[at_least_one<Mdl::NRO>::value][4],
                                         const float (&cf)[Mdl::NCF][4], int r, int nrows,
                                         float omega, float om1)
{
    constexpr int NIT = Mdl::NIT, NRO = Mdl::NRO, NRO1 = at_least_one<NRO>::value;
    // values held by the neighbouring lane (executed by the whole wave)
    float edge[NIT], redge[NRO1];
#pragma unroll
    for (int f = 0; f < NIT; f++)
        edge[f] = (E0 == 0) ? lane_above(C[f][3]) : lane_below(C[f][0]);
#pragma unroll
    for (int f = 0; f < NRO1; f++)
        redge[f] = (NRO == 0) ? 0.0f : ((E0 == 0) ? lane_above(rC[f][3]) : lane_below(rC[f][0]));

#pragma unroll
    for (int e = E0; e < 4; e += 2) {
        const int i = r + e;
        if (i >= 1 && i <= nrows - 2) {
            float c[NIT], w[NIT], ea[NIT], n[NIT], s[NIT];
            float rc[NRO1], rw[NRO1], re[NRO1], rn[NRO1], rs[NRO1], k[Mdl::NCF];
#pragma unroll
            for (int f = 0; f < NIT; f++) {
                c[f] = C[f][e];
                w[f] = W[f][e];
                ea[f] = E[f][e];
                n[f] = (e == 0) ? edge[f] : C[f][e == 0 ? 0 : e - 1];
                s[f] = (e == 3) ? edge[f] : C[f][e == 3 ? 3 : e + 1];
            }
#pragma unroll
            for (int f = 0; f < NRO1; f++) {
                rc[f] = rC[f][e];
                rw[f] = rW[f][e];
                re[f] = rE[f][e];
                rn[f] = (e == 0) ? redge[f] : rC[f][e == 0 ? 0 : e - 1];
                rs[f] = (e == 3) ? redge[f] : rC[f][e == 3 ? 3 : e + 1];
            }
#pragma unroll
            for (int f = 0; f < Mdl::NCF; f++) k[f] = cf[f][e];
            Mdl::update(c, w, ea, n, s, rc, rw, re, rn, rs, k, omega, om1);
#pragma unroll
            for (int f = 0; f < NIT; f++) C[f][e] = c[f];
        }
    }
}

struct RbGeom {
    int r, j0, j1, nrows, ncols, col0;
    float omega;
    bool store_lane;
};

// The column march of one unit.  DIR=+1 walks the owned columns [j0,j1) left to right, DIR=-1 right
// to left; "previous" below means the column the march just left (c - DIR).  Per step: red update of
// column c, black update of the previous column, store the previous column.
template <class Mdl, bool VEC, bool FIRST, int DIR>
__device__ __forceinline__ void rb_march(const SweepPlanes<Mdl> &P, float *dout0, float *dout1, const RbGeom &gm)
{
    constexpr int NIT = Mdl::NIT, NRO = Mdl::NRO, NRO1 = at_least_one<NRO>::value, NCF = Mdl::NCF;
    const int r = gm.r, j0 = gm.j0, j1 = gm.j1, nrows = gm.nrows, ncols = gm.ncols;
    const float omega = gm.omega, om1 = 1.0f - gm.omega;

    // register windows, in march order: prev-prev, prev, current, next
    float Oprev[NIT][4], Oc[NIT][4], Onext[NIT][4];     // old iterate at c-DIR, c, c+DIR
    float Rpp[NIT][4], Rp[NIT][4];                       // after red, at c-2DIR, c-DIR
    float ROpp[NRO1][4], ROp[NRO1][4], ROc[NRO1][4], ROnext[NRO1][4];
    float CFp[NCF][4], CFc[NCF][4];                      // coefficients at c-DIR, c

    int c = (DIR > 0) ? j0 - 1 : j1;
    const int nsteps = j1 - j0 + 2; // red on both halo columns, black on the owned ones
#pragma unroll
    for (int f = 0; f < NIT; f++) {
        rb_load4<VEC>(Oprev[f], P.it_in[f], c - DIR, r, nrows, ncols);
        rb_load4<VEC>(Oc[f], P.it_in[f], c, r, nrows, ncols);
        rb_load4<VEC>(Onext[f], P.it_in[f], c + DIR, r, nrows, ncols);
#pragma unroll
        for (int e = 0; e < 4; e++) Rpp[f][e] = Rp[f][e] = 0.0f;
    }
#pragma unroll
    for (int f = 0; f < NRO1; f++) {
        if (NRO > 0) {
            rb_load4<VEC>(ROp[f], P.ro[f], c - DIR, r, nrows, ncols);
            rb_load4<VEC>(ROc[f], P.ro[f], c, r, nrows, ncols);
            rb_load4<VEC>(ROnext[f], P.ro[f], c + DIR, r, nrows, ncols);
        }
#pragma unroll
        for (int e = 0; e < 4; e++) {
            ROpp[f][e] = 0.0f;
            if (NRO == 0) ROp[f][e] = ROc[f][e] = ROnext[f][e] = 0.0f;
        }
    }
#pragma unroll
    for (int f = 0; f < NCF; f++) {
        rb_load4<VEC>(CFc[f], P.cf[f], c, r, nrows, ncols);
#pragma unroll
        for (int e = 0; e < 4; e++) CFp[f][e] = 0.0f;
    }

    for (int step = 0; step < nsteps; step++, c += DIR) {
        // prefetch what the next step needs; consumed after the rotation below
        float On[NIT][4], ROn[NRO1][4], CFn[NCF][4];
#pragma unroll
        for (int f = 0; f < NIT; f++) rb_load4<VEC>(On[f], P.it_in[f], c + 2 * DIR, r, nrows, ncols);
#pragma unroll
        for (int f = 0; f < NRO1; f++) {
            if (NRO > 0) rb_load4<VEC>(ROn[f], P.ro[f], c + 2 * DIR, r, nrows, ncols);
            else ROn[f][0] = ROn[f][1] = ROn[f][2] = ROn[f][3] = 0.0f;
        }
#pragma unroll
        for (int f = 0; f < NCF; f++) rb_load4<VEC, RB_NT_COEF>(CFn[f], P.cf[f], c + DIR, r, nrows, ncols);

        const int p = (c + gm.col0) & 1;

        if (FIRST) { // column c becomes current: build its divisors, keep them for the later sweeps
#pragma unroll
            for (int e = 0; e < 4; e++) {
                float k[NCF];
#pragma unroll
                for (int f = 0; f < NCF; f++) k[f] = CFc[f][e];
                Mdl::derive(k);
                CFc[Mdl::D0][e] = k[Mdl::D0];
                CFc[Mdl::D1][e] = k[Mdl::D1];
            }
            if (gm.store_lane && c >= j0 && c < j1) {
                rb_store4<VEC>(CFc[Mdl::D0], dout0, c, r, nrows);
                rb_store4<VEC>(CFc[Mdl::D1], dout1, c, r, nrows);
            }
        }

        // red half-sweep on column c from the old columns on both sides (west = c-1, east = c+1)
        float Rc[NIT][4];
#pragma unroll
        for (int f = 0; f < NIT; f++)
#pragma unroll
            for (int e = 0; e < 4; e++) Rc[f][e] = Oc[f][e];
        if (c >= 1 && c <= ncols - 2) {
            if (DIR > 0) {
                if (p == 0) rb_phase<Mdl, 0>(Rc, Oprev, Onext, ROc, ROp, ROnext, CFc, r, nrows, omega, om1);
                else        rb_phase<Mdl, 1>(Rc, Oprev, Onext, ROc, ROp, ROnext, CFc, r, nrows, omega, om1);
            } else {
                if (p == 0) rb_phase<Mdl, 0>(Rc, Onext, Oprev, ROc, ROnext, ROp, CFc, r, nrows, omega, om1);
                else        rb_phase<Mdl, 1>(Rc, Onext, Oprev, ROc, ROnext, ROp, CFc, r, nrows, omega, om1);
            }
        }

        // black half-sweep on the previous column cb = c-DIR from R(cb-1), R(cb), R(cb+1); then store it.
        // The red pixels of cb are elements {1-p, 3-p}, so its black ones are {p, p+2}: the same
        // element set as the red update of column c above.
        const int cb = c - DIR;
        if (cb >= j0 && cb < j1 && cb >= 1 && cb <= ncols - 2) {
            float F[NIT][4];
#pragma unroll
            for (int f = 0; f < NIT; f++)
#pragma unroll
                for (int e = 0; e < 4; e++) F[f][e] = Rp[f][e];
            if (DIR > 0) {
                if (p == 0) rb_phase<Mdl, 0>(F, Rpp, Rc, ROp, ROpp, ROc, CFp, r, nrows, omega, om1);
                else        rb_phase<Mdl, 1>(F, Rpp, Rc, ROp, ROpp, ROc, CFp, r, nrows, omega, om1);
            } else {
                if (p == 0) rb_phase<Mdl, 0>(F, Rc, Rpp, ROp, ROc, ROpp, CFp, r, nrows, omega, om1);
                else        rb_phase<Mdl, 1>(F, Rc, Rpp, ROp, ROc, ROpp, CFp, r, nrows, omega, om1);
            }
            // replicate into the border rows 0 and nrows-1 (rows first, opticalflowSolvers.c:161-170)
#pragma unroll
            for (int f = 0; f < NIT; f++) {
                const float prev3 = VEC ? 0.0f : lane_above(F[f][3]);
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    const int i = r + e;
                    if (i == 0) F[f][e] = F[f][e == 3 ? 3 : e + 1];
                    if (i == nrows - 1) F[f][e] = (e == 0) ? prev3 : F[f][e == 0 ? 0 : e - 1];
                }
                if (gm.store_lane) {
                    rb_store4<VEC>(F[f], P.it_out[f], cb, r, nrows);
                    // then columns (:172-179): column 0 copies column 1, the last copies ncols-2
                    if (cb == 1) rb_store4<VEC>(F[f], P.it_out[f], 0, r, nrows);
                    if (cb == ncols - 2) rb_store4<VEC>(F[f], P.it_out[f], ncols - 1, r, nrows);
                }
            }
        }

        // rotate the windows
#pragma unroll
        for (int f = 0; f < NIT; f++)
#pragma unroll
            for (int e = 0; e < 4; e++) {
                Rpp[f][e] = Rp[f][e];
                Rp[f][e] = Rc[f][e];
                Oprev[f][e] = Oc[f][e];
                Oc[f][e] = Onext[f][e];
                Onext[f][e] = On[f][e];
            }
#pragma unroll
        for (int f = 0; f < NRO1; f++)
#pragma unroll
            for (int e = 0; e < 4; e++) {
                ROpp[f][e] = ROp[f][e];
                ROp[f][e] = ROc[f][e];
                ROc[f][e] = ROnext[f][e];
                ROnext[f][e] = ROn[f][e];
            }
#pragma unroll
        for (int f = 0; f < NCF; f++)
#pragma unroll
            for (int e = 0; e < 4; e++) {
                CFp[f][e] = CFc[f][e];
                CFc[f][e] = CFn[f][e];
            }
    }
}

// ------------------------------------------------------------------------------------------------
// Two sweeps per launch.
//
// One sweep moves every plane through HBM once; the arithmetic of a sweep keeps the VALUs less than half
// busy.  Two consecutive sweeps share all nine coefficient planes and the iterate never has to leave the
// registers between them, so fusing them nearly halves the traffic per sweep.  The march keeps four
// stages in flight, all on the same element set {p, p+2} of their column:
//     red 1 on column c, black 1 on c-D, red 2 on c-2D, black 2 on c-3D (then stored)
// with windows O (old), A (after red 1), B (after sweep 1), C (after red 2).  A unit therefore loads
// four halo columns per side instead of two and its halo lanes (4 rows above and below) are used up
// exactly: each half-sweep invalidates one more halo row.  Between the sweeps the reference replicates
// the border (rows, then columns; opticalflowSolvers.c:161-179): the rows are replicated in the B column as
// soon as black 1 has produced it, and "column 0 = column 1 after sweep 1" is substituted where sweep 2
// reads it (the west operand of column 1 is B(1); same on the east side).  Results are bit-identical to
// two launches of k_sor_rb.
// ------------------------------------------------------------------------------------------------
template <class Mdl, bool VEC>
__device__ __forceinline__ void rb_replicate_rows(float (&F)[Mdl::NIT][4], int r, int nrows)
{
#pragma unroll
    for (int f = 0; f < Mdl::NIT; f++) {
        const float prev3 = VEC ? 0.0f : lane_above(F[f][3]);
#pragma unroll
        for (int e = 0; e < 4; e++) {
            const int i = r + e;
            if (i == 0) F[f][e] = F[f][e == 3 ? 3 : e + 1];
            if (i == nrows - 1) F[f][e] = (e == 0) ? prev3 : F[f][e == 0 ? 0 : e - 1];
        }
    }
}

template <class Mdl, bool VEC, bool FIRST, int DIR>
__device__ __forceinline__ void rb_march2(const SweepPlanes<Mdl> &P, float *dout0, float *dout1, const RbGeom &gm)
{
    constexpr int NIT = Mdl::NIT, NRO = Mdl::NRO, NRO1 = at_least_one<NRO>::value, NCF = Mdl::NCF;
    const int r = gm.r, j0 = gm.j0, j1 = gm.j1, nrows = gm.nrows, ncols = gm.ncols;
    const float omega = gm.omega, om1 = 1.0f - gm.omega;

    // windows, relative to the column c of stage "red 1" (D = DIR); the newest entry of A, B, C is produced
    // in the step.  (A five-fold unrolled march over register rings removes the window moves below but
    // quintuples the code: it ran 30 % slower -- instruction fetch, not VALU, was the limit.)
    float Om[NIT][4], Oc[NIT][4], Op[NIT][4];   // old iterate at c-D, c, c+D
    float A2[NIT][4], A1[NIT][4];                // after red 1 at c-2D, c-D
    float B3[NIT][4], B2[NIT][4];                // after sweep 1 at c-3D, c-2D
    float C4[NIT][4], C3[NIT][4];                // after red 2 at c-4D, c-3D
    float K0[NCF][4], K1[NCF][4], K2[NCF][4], K3[NCF][4]; // coefficients at c, c-D, c-2D, c-3D
    // read-only neighbour fields (the base flow of the late-linearisation models) at c+D .. c-4D; they do not
    // change between the sweeps, so their border columns are read as stored
    float Rp[NRO1][4], R0[NRO1][4], R1[NRO1][4], R2[NRO1][4], R3[NRO1][4], R4[NRO1][4];

    int c = (DIR > 0) ? j0 - 3 : j1 + 2;
    const int nsteps = j1 - j0 + 6;
#pragma unroll
    for (int f = 0; f < NIT; f++) {
        rb_load4<VEC>(Om[f], P.it_in[f], c - DIR, r, nrows, ncols);
        rb_load4<VEC>(Oc[f], P.it_in[f], c, r, nrows, ncols);
        rb_load4<VEC>(Op[f], P.it_in[f], c + DIR, r, nrows, ncols);
#pragma unroll
        for (int e = 0; e < 4; e++) A2[f][e] = A1[f][e] = B3[f][e] = B2[f][e] = C4[f][e] = C3[f][e] = 0.0f;
    }
#pragma unroll
    for (int f = 0; f < NCF; f++) {
        rb_load4<VEC>(K0[f], P.cf[f], c, r, nrows, ncols);
#pragma unroll
        for (int e = 0; e < 4; e++) K1[f][e] = K2[f][e] = K3[f][e] = 0.0f;
    }
#pragma unroll
    for (int f = 0; f < NRO1; f++) {
#pragma unroll
        for (int e = 0; e < 4; e++) Rp[f][e] = R0[f][e] = R1[f][e] = R2[f][e] = R3[f][e] = R4[f][e] = 0.0f;
        if (NRO > 0) {
            rb_load4<VEC>(R1[f], P.ro[f], c - DIR, r, nrows, ncols);
            rb_load4<VEC>(R0[f], P.ro[f], c, r, nrows, ncols);
            rb_load4<VEC>(Rp[f], P.ro[f], c + DIR, r, nrows, ncols);
        }
    }

#define PDEIP_RB2_PHASE(CEN, PREV, NEXT, RC, RPREV, RNEXT, KK)                                                      \
    do {                                                                                                           \
        if (DIR > 0) {                                                                                             \
            if (p == 0) rb_phase<Mdl, 0>(CEN, PREV, NEXT, RC, RPREV, RNEXT, KK, r, nrows, omega, om1);             \
            else        rb_phase<Mdl, 1>(CEN, PREV, NEXT, RC, RPREV, RNEXT, KK, r, nrows, omega, om1);             \
        } else {                                                                                                   \
            if (p == 0) rb_phase<Mdl, 0>(CEN, NEXT, PREV, RC, RNEXT, RPREV, KK, r, nrows, omega, om1);             \
            else        rb_phase<Mdl, 1>(CEN, NEXT, PREV, RC, RNEXT, RPREV, KK, r, nrows, omega, om1);             \
        }                                                                                                          \
    } while (0)

    auto inner = [&](int col) { return col >= 1 && col <= ncols - 2; };
    for (int step = 0; step < nsteps; step++, c += DIR) {
        float On[NIT][4], Kn[NCF][4], Rn[NRO1][4]; // prefetch for the next step
#pragma unroll
        for (int f = 0; f < NIT; f++) rb_load4<VEC>(On[f], P.it_in[f], c + 2 * DIR, r, nrows, ncols);
#pragma unroll
        for (int f = 0; f < NRO1; f++) {
            if (NRO > 0) rb_load4<VEC>(Rn[f], P.ro[f], c + 2 * DIR, r, nrows, ncols);
            else Rn[f][0] = Rn[f][1] = Rn[f][2] = Rn[f][3] = 0.0f;
        }
#pragma unroll
        for (int f = 0; f < NCF; f++) rb_load4<VEC, RB_NT_COEF>(Kn[f], P.cf[f], c + DIR, r, nrows, ncols);

        const int p = (c + gm.col0) & 1;
        const int c1 = c - DIR, c2 = c - 2 * DIR, c3 = c - 3 * DIR;

        if (FIRST) { // column c becomes current: build its divisors, keep them for the later sweeps
#pragma unroll
            for (int e = 0; e < 4; e++) {
                float k[NCF];
#pragma unroll
                for (int f = 0; f < NCF; f++) k[f] = K0[f][e];
                Mdl::derive(k);
                K0[Mdl::D0][e] = k[Mdl::D0];
                K0[Mdl::D1][e] = k[Mdl::D1];
            }
            if (gm.store_lane && c >= j0 && c < j1) {
                rb_store4<VEC>(K0[Mdl::D0], dout0, c, r, nrows);
                rb_store4<VEC>(K0[Mdl::D1], dout1, c, r, nrows);
            }
        }

        // stage 1: red 1 on column c from O(c-D), O(c+D)
        float A0[NIT][4];
#pragma unroll
        for (int f = 0; f < NIT; f++)
#pragma unroll
            for (int e = 0; e < 4; e++) A0[f][e] = Oc[f][e];
        if (inner(c)) PDEIP_RB2_PHASE(A0, Om, Op, R0, R1, Rp, K0);

        // stage 2: black 1 on column c1 from A(c1-D), A(c1+D); afterwards it is "column c1 after sweep 1"
        float B1[NIT][4];
#pragma unroll
        for (int f = 0; f < NIT; f++)
#pragma unroll
            for (int e = 0; e < 4; e++) B1[f][e] = A1[f][e];
        if (inner(c1)) {
            PDEIP_RB2_PHASE(B1, A2, A0, R1, R2, R0, K1);
            rb_replicate_rows<Mdl, VEC>(B1, r, nrows); // rows first (:161-170); the column replicate is substituted below
        }

        // stage 3: red 2 on column c2 from B(c2-D), B(c2+D); sweep 2 sees a border column as the replicate of
        // its inner neighbour after sweep 1 (:172-179), i.e. as B(c2) itself
        float C2[NIT][4];
#pragma unroll
        for (int f = 0; f < NIT; f++)
#pragma unroll
            for (int e = 0; e < 4; e++) C2[f][e] = B2[f][e];
        if (inner(c2)) {
            const bool prev_is_border = !inner(c2 - DIR), next_is_border = !inner(c2 + DIR);
            if (prev_is_border || next_is_border) { // only the units that touch the first / last inner column
                float Pv[NIT][4], Nx[NIT][4];
#pragma unroll
                for (int f = 0; f < NIT; f++)
#pragma unroll
                    for (int e = 0; e < 4; e++) {
                        Pv[f][e] = prev_is_border ? B2[f][e] : B3[f][e];
                        Nx[f][e] = next_is_border ? B2[f][e] : B1[f][e];
                    }
                PDEIP_RB2_PHASE(C2, Pv, Nx, R2, R3, R1, K2);
            } else {
                PDEIP_RB2_PHASE(C2, B3, B1, R2, R3, R1, K2);
            }
        }

        // stage 4: black 2 on column c3 from C(c3-D), C(c3+D); a border column is B(c3)
        if (c3 >= j0 && c3 < j1 && inner(c3)) {
            float F[NIT][4];
#pragma unroll
            for (int f = 0; f < NIT; f++)
#pragma unroll
                for (int e = 0; e < 4; e++) F[f][e] = C3[f][e];
            const bool prev_is_border = !inner(c3 - DIR), next_is_border = !inner(c3 + DIR);
            if (prev_is_border || next_is_border) {
                float Pv[NIT][4], Nx[NIT][4];
#pragma unroll
                for (int f = 0; f < NIT; f++)
#pragma unroll
                    for (int e = 0; e < 4; e++) {
                        Pv[f][e] = prev_is_border ? B3[f][e] : C4[f][e];
                        Nx[f][e] = next_is_border ? B3[f][e] : C2[f][e];
                    }
                PDEIP_RB2_PHASE(F, Pv, Nx, R3, R4, R2, K3);
            } else {
                PDEIP_RB2_PHASE(F, C4, C2, R3, R4, R2, K3);
            }
            rb_replicate_rows<Mdl, VEC>(F, r, nrows);
            if (gm.store_lane) {
#pragma unroll
                for (int f = 0; f < NIT; f++) {
                    rb_store4<VEC>(F[f], P.it_out[f], c3, r, nrows);
                    if (c3 == 1) rb_store4<VEC>(F[f], P.it_out[f], 0, r, nrows);
                    if (c3 == ncols - 2) rb_store4<VEC>(F[f], P.it_out[f], ncols - 1, r, nrows);
                }
            }
        }

        // advance one column
#pragma unroll
        for (int f = 0; f < NIT; f++)
#pragma unroll
            for (int e = 0; e < 4; e++) {
                C4[f][e] = C3[f][e];
                C3[f][e] = C2[f][e];
                B3[f][e] = B2[f][e];
                B2[f][e] = B1[f][e];
                A2[f][e] = A1[f][e];
                A1[f][e] = A0[f][e];
                Om[f][e] = Oc[f][e];
                Oc[f][e] = Op[f][e];
                Op[f][e] = On[f][e];
            }
#pragma unroll
        for (int f = 0; f < NCF; f++)
#pragma unroll
            for (int e = 0; e < 4; e++) {
                K3[f][e] = K2[f][e];
                K2[f][e] = K1[f][e];
                K1[f][e] = K0[f][e];
                K0[f][e] = Kn[f][e];
            }
#pragma unroll
        for (int f = 0; f < NRO1; f++)
#pragma unroll
            for (int e = 0; e < 4; e++) {
                R4[f][e] = R3[f][e];
                R3[f][e] = R2[f][e];
                R2[f][e] = R1[f][e];
                R1[f][e] = R0[f][e];
                R0[f][e] = Rp[f][e];
                Rp[f][e] = Rn[f][e];
            }
    }
#undef PDEIP_RB2_PHASE
}

// FIRST: this is sweep 0 of a call.  The two derived coefficient planes (divisors) do not exist
// yet: their slots in P.cf point at the raw planes (e.g. Du, Dv), every column is passed through
// Mdl::derive() as it becomes current, and the owning unit stores the derived planes to
// dout0/dout1 for the later sweeps -- what the reference does inside its first sweep
// (opticalflowSolvers.c:111-127), at the cost of two plane writes instead of a separate pass.
template <class Mdl, bool VEC, bool FIRST, bool TWO = false>
__global__ void __launch_bounds__(64 * RB_WAVES_PER_BLOCK)
k_sor_rb(SweepPlanes<Mdl> P, float *dout0, float *dout1, int nrows, int ncols, int TJ, int ntiles_r,
         int nunits, float omega, int col0, size_t frame_stride)
{
    constexpr int NIT = Mdl::NIT, NRO = Mdl::NRO, NRO1 = at_least_one<NRO>::value, NCF = Mdl::NCF;
    const int lane = threadIdx.x & 63;
    // XCD-aware unit order: workgroups are dealt round-robin over the 8 XCDs, so give each XCD a
    // contiguous range of units (neighbouring strips share their halo columns through one L2).
    // Placement only changes speed, never results.
    int bid = blockIdx.x;
    {
        const int nb = gridDim.x, per = nb >> 3;
        if (bid < (per << 3)) bid = (bid & 7) * per + (bid >> 3);
    }
    const int unit = bid * RB_WAVES_PER_BLOCK + (threadIdx.x >> 6);
    if (unit >= nunits) return; // whole wave leaves; no workgroup barrier is used below
    const size_t fo = (size_t)blockIdx.y * frame_stride;
    if (FIRST) {
        dout0 += fo;
        dout1 += fo;
    }
#pragma unroll
    for (int f = 0; f < NIT; f++) {
        P.it_in[f] += fo;
        P.it_out[f] += fo;
    }
#pragma unroll
    for (int f = 0; f < NRO1; f++)
        if (NRO > 0) P.ro[f] += fo;
#pragma unroll
    for (int f = 0; f < NCF; f++) P.cf[f] += fo;

    // strips fastest: the four waves of a workgroup (and neighbouring workgroups of one XCD) relax
    // adjacent strips of the same row tile at the same time
    const int nstrips = nunits / ntiles_r;
    const int a = unit / nstrips, b = unit % nstrips;
    RbGeom gm;
    gm.r = a * RB_OWN_ROWS - 4 + 4 * lane;
    gm.j0 = b * TJ;
    gm.j1 = (gm.j0 + TJ < ncols) ? gm.j0 + TJ : ncols;
    gm.nrows = nrows;
    gm.ncols = ncols;
    gm.col0 = col0;
    gm.omega = omega;
    gm.store_lane = (lane >= 1) && (lane <= 62);
    // Alternate the marching direction from strip to strip: strips 2k and 2k+1 finish at their common
    // boundary together and strips 2k+1 and 2k+2 start at theirs together, so the halo columns both
    // sides need are touched at about the same time and the second toucher hits L1/L2 instead of HBM.
    if (TWO) {
        if (b & 1) rb_march2<Mdl, VEC, FIRST, -1>(P, dout0, dout1, gm);
        else rb_march2<Mdl, VEC, FIRST, +1>(P, dout0, dout1, gm);
    } else {
        if (b & 1) rb_march<Mdl, VEC, FIRST, -1>(P, dout0, dout1, gm);
        else rb_march<Mdl, VEC, FIRST, +1>(P, dout0, dout1, gm);
    }
}

} // namespace pdeip
