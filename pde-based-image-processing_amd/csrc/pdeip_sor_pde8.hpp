// pdeip_sor_pde8.hpp -- GS_SOR_8_2d (pdeSolvers.c:153-268): scalar SOR with a 9-point stencil.
//
// Two orderings, same ModelPde8::update():
//
//  * four-colour (RED_BLACK mode): colour = (i&1) | ((j+col0)&1)<<1, passes 0..3.  No two pixels
//    of one colour are neighbours in a 9-point stencil, so a pass is one in-place launch; the
//    border replicate (read by the next sweep's diagonal taps) is a fifth small launch.
//
//  * exact order: the reference's lexicographic order.  Pixel (i,j) needs the new values of its
//    N, NW, W, SW neighbours, so the independent fronts are i + 2j + 4t = const.  Same tile
//    wavefront as pdeip_sor_exact.hpp with a skew of two rows per lane: lane l relaxes row
//    1 + 64a + q - 2l of column 1 + 64b + l; tile (a,b,t) depends only on tiles with a smaller
//    m = a + 3b + 4t.  West values arrive through a 3-deep shift register fed by one wavefront
//    shuffle per step, east values through three shuffles of the neighbour's look-ahead queue.
//    Because diagonal taps read border cells that hold the PREVIOUS sweep's replicate, the
//    border ring of every sweep is kept in a small side array (2*(nrows+ncols) floats per
//    sweep and frame) written by the lane that relaxes the adjacent interior pixel.
#pragma once
#include "pdeip_models.hpp"
#include "pdeip_sor_exact.hpp"
#include "pdeip_sor_rb.hpp"

namespace pdeip {

struct Pde8Planes {
    float *x;
    const float *cf[ModelPde8::NCF];
};

// ---------------------------------------------------------------------------------------------
// four-colour ordering: one fused launch per sweep (register marching, like pdeip_sor_rb.hpp)
// ---------------------------------------------------------------------------------------------
// colour = (i&1) | ((j+col0)&1)<<1.  An even column is finished by colours 0 (even rows) and 1 (odd
// rows), which read only OLD odd columns; an odd column is then finished by colours 2 and 3, which
// read the FINISHED even columns on both sides.  So the march is the same two-stage pipeline as the
// red-black kernel: stage 1 on column c (colours 0,1 if c is even), stage 2 on column c-1 (colours 2,3
// if it is odd), store column c-1.  Lane l holds rows r..r+3 (r a multiple of 4: even rows are
// elements {0,2}); the 9-point taps at the lane edges come from the adjacent lane by shuffle.  Three
// rows of vertical and two columns of horizontal halo are recomputed per unit.

// Relax elements {E0, E0+2} of column C using the columns W and E on either side.
template <int E0>
__device__ __forceinline__ void p8_phase(float (&C)[4], const float (&W)[4], const float (&E)[4],
                                         const float (&cf)[ModelPde8::NCF][4], int r, int nrows, float omega, float om1)
{
    const float ce = (E0 == 0) ? __shfl_up(C[3], 1) : __shfl_down(C[0], 1);
    const float we = (E0 == 0) ? __shfl_up(W[3], 1) : __shfl_down(W[0], 1);
    const float ee = (E0 == 0) ? __shfl_up(E[3], 1) : __shfl_down(E[0], 1);
#pragma unroll
    for (int e = E0; e < 4; e += 2) {
        const int i = r + e;
        if (i >= 1 && i <= nrows - 2) {
            const int em = e == 0 ? 0 : e - 1, ep = e == 3 ? 3 : e + 1;
            const float xN = (e == 0) ? ce : C[em], xS = (e == 3) ? ce : C[ep];
            const float xNW = (e == 0) ? we : W[em], xSW = (e == 3) ? we : W[ep];
            const float xNE = (e == 0) ? ee : E[em], xSE = (e == 3) ? ee : E[ep];
            float k[ModelPde8::NCF];
#pragma unroll
            for (int f = 0; f < ModelPde8::NCF; f++) k[f] = cf[f][e];
            C[e] = ModelPde8::update(C[e], W[e], E[e], xN, xS, xNW, xNE, xSW, xSE, k, omega, om1);
        }
    }
}

// pdeSolvers.c:217-237: the slots of B_temp / INV_TRACE hold B / TRACE before this
__device__ __forceinline__ void p8_derive(float (&k)[ModelPde8::NCF])
{
    const float tr = k[ModelPde8::cInv];
    float t = k[ModelPde8::cWE] + k[ModelPde8::cWW];
    t += k[ModelPde8::cWS] + k[ModelPde8::cWN];
    t += k[ModelPde8::cWSW] + k[ModelPde8::cWNW];
    t += k[ModelPde8::cWSE] + k[ModelPde8::cWNE];
    const bool ok = !is_nan(tr);
    k[ModelPde8::cInv] = ok ? 1.0f / tr : 1.0f / t;
    k[ModelPde8::cB] = ok ? k[ModelPde8::cB] : 0.0f;
}

struct Pde8SweepPlanes {
    const float *x_in;
    float *x_out;
    const float *cf[ModelPde8::NCF];
};

template <bool VEC, bool FIRST>
__global__ void __launch_bounds__(64 * RB_WAVES_PER_BLOCK)
k_pde8_colour(Pde8SweepPlanes P, float *dout0, float *dout1, int nrows, int ncols, int TJ, int ntiles_r,
              int nunits, float omega, int col0, size_t frame_stride)
{
    constexpr int NCF = ModelPde8::NCF;
    const int lane = threadIdx.x & 63;
    int bid = blockIdx.x;
    {
        const int nb = gridDim.x, per = nb >> 3; // XCD-aware unit order (speed only)
        if (bid < (per << 3)) bid = (bid & 7) * per + (bid >> 3);
    }
    const int unit = bid * RB_WAVES_PER_BLOCK + (threadIdx.x >> 6);
    if (unit >= nunits) return;
    const size_t fo = (size_t)blockIdx.y * frame_stride;
    P.x_in += fo;
    P.x_out += fo;
#pragma unroll
    for (int f = 0; f < NCF; f++) P.cf[f] += fo;
    if (FIRST) {
        dout0 += fo;
        dout1 += fo;
    }
    const int nstrips = nunits / ntiles_r;
    const int a = unit / nstrips, b = unit % nstrips;
    const int r = a * RB_OWN_ROWS - 4 + 4 * lane;
    const int j0 = b * TJ, j1 = (j0 + TJ < ncols) ? j0 + TJ : ncols;
    const float om1 = 1.0f - omega;
    const bool store_lane = (lane >= 1) && (lane <= 62);

    float Om[4], Oc[4], Op[4];   // old X at columns c-1, c, c+1
    float Rmm[4], Rm[4];         // after stage 1, columns c-2, c-1
    float CFm[NCF][4], CFc[NCF][4];
    int c = j0 - 1;
    rb_load4<VEC>(Om, P.x_in, c - 1, r, nrows, ncols);
    rb_load4<VEC>(Oc, P.x_in, c, r, nrows, ncols);
    rb_load4<VEC>(Op, P.x_in, c + 1, r, nrows, ncols);
#pragma unroll
    for (int e = 0; e < 4; e++) Rmm[e] = Rm[e] = 0.0f;
#pragma unroll
    for (int f = 0; f < NCF; f++) {
        rb_load4<VEC>(CFc[f], P.cf[f], c, r, nrows, ncols);
#pragma unroll
        for (int e = 0; e < 4; e++) CFm[f][e] = 0.0f;
    }
    for (; c <= j1; c++) {
        float On[4], CFn[NCF][4];
        rb_load4<VEC>(On, P.x_in, c + 2, r, nrows, ncols);
#pragma unroll
        for (int f = 0; f < NCF; f++) rb_load4<VEC>(CFn[f], P.cf[f], c + 1, r, nrows, ncols);

        if (FIRST) {
#pragma unroll
            for (int e = 0; e < 4; e++) {
                float k[NCF];
#pragma unroll
                for (int f = 0; f < NCF; f++) k[f] = CFc[f][e];
                p8_derive(k);
                CFc[ModelPde8::cB][e] = k[ModelPde8::cB];
                CFc[ModelPde8::cInv][e] = k[ModelPde8::cInv];
            }
            if (store_lane && c >= j0 && c < j1) {
                rb_store4<VEC>(CFc[ModelPde8::cB], dout0, c, r, nrows);
                rb_store4<VEC>(CFc[ModelPde8::cInv], dout1, c, r, nrows);
            }
        }
        const bool c_even = ((c + col0) & 1) == 0;
        // stage 1 on column c: colours 0 then 1 (even columns only), from the old columns on both sides
        float Rc[4];
#pragma unroll
        for (int e = 0; e < 4; e++) Rc[e] = Oc[e];
        if (c_even && c >= 1 && c <= ncols - 2) {
            p8_phase<0>(Rc, Om, Op, CFc, r, nrows, omega, om1);
            p8_phase<1>(Rc, Om, Op, CFc, r, nrows, omega, om1);
        }
        // stage 2 on column c-1: colours 2 then 3 (odd columns only), from the finished even columns
        const int cb = c - 1;
        if (cb >= j0 && cb >= 1 && cb <= ncols - 2) {
            float F[4];
#pragma unroll
            for (int e = 0; e < 4; e++) F[e] = Rm[e];
            if (c_even) { // then c-1 is odd
                p8_phase<0>(F, Rmm, Rc, CFm, r, nrows, omega, om1);
                p8_phase<1>(F, Rmm, Rc, CFm, r, nrows, omega, om1);
            }
            const float prev3 = VEC ? 0.0f : __shfl_up(F[3], 1);
#pragma unroll
            for (int e = 0; e < 4; e++) { // border rows replicate (pdeSolvers.c:249-255)
                const int i = r + e;
                if (i == 0) F[e] = F[e == 3 ? 3 : e + 1];
                if (i == nrows - 1) F[e] = (e == 0) ? prev3 : F[e == 0 ? 0 : e - 1];
            }
            if (store_lane) {
                rb_store4<VEC>(F, P.x_out, cb, r, nrows);
                if (cb == 1) rb_store4<VEC>(F, P.x_out, 0, r, nrows);             // then columns (:256-262)
                if (cb == ncols - 2) rb_store4<VEC>(F, P.x_out, ncols - 1, r, nrows);
            }
        }
#pragma unroll
        for (int e = 0; e < 4; e++) {
            Rmm[e] = Rm[e];
            Rm[e] = Rc[e];
            Om[e] = Oc[e];
            Oc[e] = Op[e];
            Op[e] = On[e];
        }
#pragma unroll
        for (int f = 0; f < NCF; f++)
#pragma unroll
            for (int e = 0; e < 4; e++) {
                CFm[f][e] = CFc[f][e];
                CFc[f][e] = CFn[f][e];
            }
    }
}

// ---------------------------------------------------------------------------------------------
// exact (lexicographic) ordering
// ---------------------------------------------------------------------------------------------
constexpr int P8_R = 64;    // steps per tile
constexpr int P8_SKEW = 2;  // rows per lane
constexpr int P8_G = 3;     // m = a + P8_G*b + P8_H*t   (see header comment / DESIGN.md)
constexpr int P8_H = 4;

__host__ __device__ inline size_t pde8_side_stride(int nrows, int ncols) { return 2 * ((size_t)nrows + ncols); }
inline size_t pde8_exact_scratch_floats(int nrows, int ncols, int nframes, int iter)
{
    return pde8_side_stride(nrows, ncols) * (size_t)nframes * (size_t)(iter > 0 ? iter : 1);
}

__global__ void __launch_bounds__(64)
k_pde8_exact(Pde8Planes P, float *side, int nrows, int ncols, int A, int B, int T, int m, float omega,
             size_t frame_stride)
{
    const int lane = threadIdx.x;
    const int b = blockIdx.x % B, t = blockIdx.x / B;
    const int a = m - P8_G * b - P8_H * t;
    if (a < 0 || a >= A) return;
    const size_t fo = (size_t)blockIdx.y * frame_stride;
    float *x = P.x + fo;
    const size_t sstride = pde8_side_stride(nrows, ncols);
    // border ring after sweep t-1 (read) and after sweep t (written): top[ncols] bot[ncols] left[nrows] right[nrows]
    float *ring_w = side + ((size_t)blockIdx.y * T + t) * sstride;
    const float *ring_r = (t > 0) ? ring_w - sstride : nullptr;

    const int j = 1 + 64 * b + lane;
    const bool col_in = j <= ncols - 1, col_ok = j <= ncols - 2;
    const size_t cb = (size_t)j * nrows;
    const float om1 = 1.0f - omega;
    const int i0 = 1 + a * P8_R - P8_SKEW * lane;

    // value of cell (ii,jj) as the reference's sweep t sees a cell that sweep t has not (yet) relaxed:
    // border cells hold the replicate of sweep t-1 (the caller's cell in sweep 0)
    auto old_at = [&](int ii, int jj) -> float {
        if (ii < 0 || ii > nrows - 1 || jj < 0 || jj > ncols - 1) return 0.0f;
        if (ring_r != nullptr) {
            if (ii == 0) return ring_r[jj];
            if (ii == nrows - 1) return ring_r[ncols + jj];
            if (jj == 0) return ring_r[2 * ncols + ii];
            if (jj == ncols - 1) return ring_r[2 * ncols + nrows + ii];
        }
        return x[(size_t)jj * nrows + ii];
    };
    auto is_border = [&](int ii, int jj) -> bool { return ii <= 0 || ii >= nrows - 1 || jj <= 0 || jj >= ncols - 1; };

    // own column look-ahead queue of old values: rows i, i+1, i+2, i+3
    float o0 = col_in ? old_at(i0, j) : 0.0f, o1 = col_in ? old_at(i0 + 1, j) : 0.0f, o2 = col_in ? old_at(i0 + 2, j) : 0.0f;
    // west column shift register (new values): rows i-1, i (i+1 is fetched per step)
    float nw = 0.0f, w = 0.0f, prev = 0.0f;
    if (col_ok) {
        nw = old_at(i0 - 1, j - 1); // memory holds sweep-t values there already, or the border rule applies
        w = old_at(i0, j - 1);
    }

    for (int q = 0; q < P8_R; q++) {
        const int i = i0 + q;
        const bool row_ok = (i >= 1) && (i <= nrows - 2);
        const bool active = col_ok && row_ok;
        const float o3 = col_in ? old_at(i + 3, j) : 0.0f;

        // east column (old): lane l+1 is two rows higher, its queue rows +1,+2,+3 are my rows i-1,i,i+1
        float ne = __shfl_down(o1, 1), e = __shfl_down(o2, 1), se = __shfl_down(o3, 1);
        if (lane == 63) {
            ne = old_at(i - 1, j + 1);
            e = old_at(i, j + 1);
            se = old_at(i + 1, j + 1);
        }
        // south-west (new): lane l-1 relaxed (i+1, j-1) one step ago
        float sw = __shfl_up(prev, 1);
        if (lane == 0 || q == 0 || is_border(i + 1, j - 1)) sw = col_ok ? old_at(i + 1, j - 1) : 0.0f;

        if (active) {
            float n = (q == 0 || i == 1) ? old_at(i - 1, j) : prev;
            float k[ModelPde8::NCF];
#pragma unroll
            for (int f = 0; f < ModelPde8::NCF; f++) k[f] = P.cf[f][fo + cb + i];
            const float v = ModelPde8::update(o0, w, e, n, o1, nw, ne, sw, se, k, omega, om1);
            x[cb + i] = v;
            prev = v;
            // border ring after this sweep: nearest-interior replicate (pdeSolvers.c:249-262)
            float *top = ring_w, *bot = ring_w + ncols, *left = ring_w + 2 * ncols, *right = left + nrows;
            if (i == 1) {
                top[j] = v;
                if (j == 1) { top[0] = v; left[0] = v; }
                if (j == ncols - 2) { top[ncols - 1] = v; right[0] = v; }
            }
            if (i == nrows - 2) {
                bot[j] = v;
                if (j == 1) { bot[0] = v; left[nrows - 1] = v; }
                if (j == ncols - 2) { bot[ncols - 1] = v; right[nrows - 1] = v; }
            }
            if (j == 1) left[i] = v;
            if (j == ncols - 2) right[i] = v;
        }
        // advance one row
        o0 = o1; o1 = o2; o2 = o3;
        nw = w; w = sw;
    }
}

inline int pde8_run_exact(hipStream_t s, Pde8Planes P, float *side, int nrows, int ncols, int nframes,
                          int iter, float omega)
{
    const size_t n = (size_t)nrows * ncols;
    const int A = (nrows - 2 + P8_SKEW * 63 + P8_R - 1) / P8_R;
    const int B = (ncols - 2 + 63) / 64;
    const int last_m = (A - 1) + P8_G * (B - 1) + P8_H * (iter - 1);
    const dim3 grid((unsigned)(B * iter), (unsigned)nframes);
    int launches = 0;
    for (int m = 0; m <= last_m; m++) {
        hipLaunchKernelGGL(k_pde8_exact, grid, dim3(64), 0, s, P, side, nrows, ncols, A, B, iter, m, omega, n);
        launches++;
    }
    const int nb = 2 * ncols + 2 * (nrows - 2);
    hipLaunchKernelGGL(k_fill_borders, dim3((nb + 255) / 256, nframes, 1), dim3(256), 0, s, P.x, P.x, 1, nrows, ncols, n);
    return launches + 1;
}

} // namespace pdeip
