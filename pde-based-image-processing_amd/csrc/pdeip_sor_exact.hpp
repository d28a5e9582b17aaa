// pdeip_sor_exact.hpp -- the reference's lexicographic Gauss-Seidel order on the GPU (gfx950).
//
// The reference sweeps j (MATLAB column) outer, i (row) inner, in place
// (opticalflowSolvers.c:76-79): pixel (i,j) of sweep t needs the NEW values of (i-1,j) and
// (i,j-1) and the OLD (sweep t-1) values of (i+1,j), (i,j+1) and of itself.  Those are the
// only dependencies, so every pixel on a hyperplane i + j + 2t = const is independent and
// several sweeps can be in flight at once.  Results are bit-identical to the serial loop.
//
// Tile wavefront.  The (row, column, sweep) space is cut into tiles:
//   * a strip b is 64 adjacent interior columns, one per lane (lane l <-> column 1+64b+l);
//   * a tile (a,b,t) is R=64 steps of strip b in sweep t; at step q lane l relaxes row
//     1 + a*R + q - l, so a wave is a diagonal front: the value lane l-1 produced one step
//     earlier is lane l's west neighbour (wavefront shuffle), the lane's own previous value
//     is its north neighbour, and its south/east neighbours are still sweep t-1 in memory.
//   * tile (a,b,t) depends only on tiles with a smaller m = a + 2b + 3t, and nothing it
//     overwrites is still needed by a tile with the same m (see DESIGN.md "exact order").
//     One launch relaxes every tile of one m; launches are stream-ordered, so there is no
//     inter-workgroup synchronisation inside a launch and nothing that can spin.
//
// Borders.  The reference replicates the border after every sweep (:161-179).  For a 5-point
// stencil an interior pixel only ever reads the border cell next to itself, whose value
// after sweep t-1 is that pixel's own sweep t-1 value; in sweep 0 it is the caller's border
// cell.  So border cells are read from memory in sweep 0, substituted by the pixel's own
// old value in later sweeps, never written during the sweeps, and filled once at the end.
#pragma once
#include "pdeip_models.hpp"

namespace pdeip {

constexpr int EX_R = 64; // steps per tile; the dependency analysis needs EX_R >= 63

template <class Mdl>
__global__ void __launch_bounds__(64)
k_sor_exact(SweepPlanes<Mdl> P, int nrows, int ncols, int A, int B, int T, int m, float omega,
            size_t frame_stride)
{
    constexpr int NIT = Mdl::NIT, NRO = Mdl::NRO, NRO1 = at_least_one<NRO>::value, NCF = Mdl::NCF;
    const int lane = threadIdx.x;
    const int b = blockIdx.x % B, t = blockIdx.x / B;
    const int a = m - 2 * b - 3 * t;
    if (a < 0 || a >= A) return;
    const size_t fo = (size_t)blockIdx.y * frame_stride;
    float *it[NIT];
    const float *ro[NRO1], *cfp[NCF];
#pragma unroll
    for (int f = 0; f < NIT; f++) it[f] = P.it_out[f] + fo;
#pragma unroll
    for (int f = 0; f < NRO1; f++) ro[f] = (NRO > 0) ? P.ro[f] + fo : nullptr;
#pragma unroll
    for (int f = 0; f < NCF; f++) cfp[f] = P.cf[f] + fo;

    const int j = 1 + 64 * b + lane;              // this lane's column
    const bool col_in = j <= ncols - 1;           // inside the buffer (may be the border column)
    const bool col_ok = j <= ncols - 2;           // interior column: relaxed
    const size_t cb = (size_t)j * nrows;          // column base offset
    const float om1 = 1.0f - omega;
    const bool first_sweep = (t == 0);
    const int i0 = 1 + a * EX_R - lane;           // row at step 0

    float prev[NIT], cen[NIT], rcen[NRO1], rnorth[NRO1];
#pragma unroll
    for (int f = 0; f < NIT; f++) {
        prev[f] = 0.0f;
        cen[f] = (col_in && i0 >= 0 && i0 <= nrows - 1) ? it[f][cb + i0] : 0.0f;
    }
#pragma unroll
    for (int f = 0; f < NRO1; f++) {
        rcen[f] = (NRO > 0 && col_in && i0 >= 0 && i0 <= nrows - 1) ? ro[f][cb + i0] : 0.0f;
        rnorth[f] = (NRO > 0 && col_in && i0 - 1 >= 0 && i0 - 1 <= nrows - 1) ? ro[f][cb + i0 - 1] : 0.0f;
    }

    for (int q = 0; q < EX_R; q++) {
        const int i = i0 + q;
        const bool row_ok = (i >= 1) && (i <= nrows - 2);
        const bool active = col_ok && row_ok;
        const bool can_load_s = col_in && (i + 1 >= 0) && (i + 1 <= nrows - 1);

        // south neighbours, raw memory (sweep t-1 for the iterate)
        float sraw[NIT], rsouth[NRO1];
#pragma unroll
        for (int f = 0; f < NIT; f++) sraw[f] = can_load_s ? it[f][cb + i + 1] : 0.0f;
#pragma unroll
        for (int f = 0; f < NRO1; f++) rsouth[f] = (NRO > 0 && can_load_s) ? ro[f][cb + i + 1] : 0.0f;

        // east neighbours: lane l+1 sits one row higher, so its south value is (i, j+1)
        float eraw[NIT], reast[NRO1];
#pragma unroll
        for (int f = 0; f < NIT; f++) {
            eraw[f] = __shfl_down(sraw[f], 1);
            if (lane == 63) eraw[f] = (row_ok && j + 1 <= ncols - 1) ? it[f][cb + nrows + i] : 0.0f;
        }
#pragma unroll
        for (int f = 0; f < NRO1; f++) {
            reast[f] = (NRO > 0) ? __shfl_down(rsouth[f], 1) : 0.0f;
            if (NRO > 0 && lane == 63) reast[f] = (row_ok && j + 1 <= ncols - 1) ? ro[f][cb + nrows + i] : 0.0f;
        }

        // west neighbours: lane l-1 relaxed (i, j-1) one step ago; at step 0 and for lane 0 the
        // value was written by an earlier launch and is read from memory
        float wnew[NIT], rwest[NRO1];
        const bool w_from_mem = (q == 0) || (lane == 0);
#pragma unroll
        for (int f = 0; f < NIT; f++) {
            wnew[f] = __shfl_up(prev[f], 1);
            if (w_from_mem) wnew[f] = (active) ? it[f][cb - nrows + i] : 0.0f;
        }
#pragma unroll
        for (int f = 0; f < NRO1; f++) {
            rwest[f] = (NRO > 0) ? __shfl_up(rnorth[f], 1) : 0.0f; // lane l-1's centre of the previous step
            if (NRO > 0 && w_from_mem) rwest[f] = (active) ? ro[f][cb - nrows + i] : 0.0f;
        }

        if (active) {
            float c[NIT], w[NIT], e[NIT], n[NIT], s[NIT], k[NCF];
#pragma unroll
            for (int f = 0; f < NIT; f++) {
                c[f] = cen[f];
                // north: own previous result; from memory at step 0 (earlier launch); border rule at row 1
                float nv = (q == 0) ? it[f][cb + i - 1] : prev[f];
                if (i == 1) nv = first_sweep ? it[f][cb] : cen[f];
                n[f] = nv;
                s[f] = (i + 1 == nrows - 1 && !first_sweep) ? cen[f] : sraw[f];
                e[f] = (j + 1 == ncols - 1 && !first_sweep) ? cen[f] : eraw[f];
                w[f] = (j - 1 == 0 && !first_sweep) ? cen[f] : wnew[f];
            }
#pragma unroll
            for (int f = 0; f < NCF; f++) k[f] = cfp[f][cb + i];
            Mdl::update(c, w, e, n, s, rcen, rwest, reast, rnorth, rsouth, k, omega, om1);
#pragma unroll
            for (int f = 0; f < NIT; f++) {
                it[f][cb + i] = c[f];
                prev[f] = c[f];
            }
        }
        // advance one row
#pragma unroll
        for (int f = 0; f < NIT; f++) cen[f] = sraw[f];
#pragma unroll
        for (int f = 0; f < NRO1; f++) {
            rnorth[f] = rcen[f];
            rcen[f] = rsouth[f];
        }
    }
}

// Final border replicate of the iterate (rows first, then columns; opticalflowSolvers.c:161-179):
// border cell <- nearest interior pixel.  Reads interior cells only, writes border cells only.
__global__ void k_fill_borders(float *p0, float *p1, int nfields, int nrows, int ncols,
                               size_t frame_stride)
{
    const int n = 2 * ncols + 2 * (nrows - 2);
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n) return;
    int i, j;
    if (idx < ncols) { i = 0; j = idx; }
    else if (idx < 2 * ncols) { i = nrows - 1; j = idx - ncols; }
    else if (idx < 2 * ncols + nrows - 2) { i = 1 + idx - 2 * ncols; j = 0; }
    else { i = 1 + idx - 2 * ncols - (nrows - 2); j = ncols - 1; }
    const int si = i < 1 ? 1 : (i > nrows - 2 ? nrows - 2 : i);
    const int sj = j < 1 ? 1 : (j > ncols - 2 ? ncols - 2 : j);
    const size_t fo = (size_t)blockIdx.y * frame_stride;
    float *p = (blockIdx.z == 0 ? p0 : p1) + fo;
    if ((int)blockIdx.z < nfields) p[(size_t)j * nrows + i] = p[(size_t)sj * nrows + si];
}

} // namespace pdeip
