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
    const float ce = (E0 == 0) ? lane_above(C[3]) : lane_below(C[0]);
    const float we = (E0 == 0) ? lane_above(W[3]) : lane_below(W[0]);
    const float ee = (E0 == 0) ? lane_above(E[3]) : lane_below(E[0]);
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
            const float prev3 = VEC ? 0.0f : lane_above(F[3]);
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

// Two four-colour sweeps per launch: the four-stage pipeline of rb_march2 (pdeip_sor_rb.hpp) with the 9-point
// phases.  A column of even class is relaxed in stages 1 and 3 (colours 0,1 of sweeps 1 and 2), a column of odd
// class in stages 2 and 4 (colours 2,3); the other stages pass it through.  One 9-point sweep uses up four halo
// rows, so this kernel has two halo lanes per side (240 owned rows per wave).  The border replicate between
// the sweeps (pdeSolvers.c:249-262) is reproduced as in rb_march2.  Bit-identical to two launches of k_pde8_colour.
constexpr int P8_OWN_ROWS2 = 240; // 60 storing lanes x 4 rows

__device__ __forceinline__ void p8_replicate_rows(float (&F)[4], int r, int nrows, bool vec)
{
    const float prev3 = vec ? 0.0f : lane_above(F[3]);
#pragma unroll
    for (int e = 0; e < 4; e++) {
        const int i = r + e;
        if (i == 0) F[e] = F[e == 3 ? 3 : e + 1];
        if (i == nrows - 1) F[e] = (e == 0) ? prev3 : F[e == 0 ? 0 : e - 1];
    }
}

template <bool VEC, bool FIRST>
__global__ void __launch_bounds__(64 * RB_WAVES_PER_BLOCK)
k_pde8_colour2(Pde8SweepPlanes P, float *dout0, float *dout1, int nrows, int ncols, int TJ, int ntiles_r,
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
    const int r = a * P8_OWN_ROWS2 - 8 + 4 * lane;
    const int j0 = b * TJ, j1 = (j0 + TJ < ncols) ? j0 + TJ : ncols;
    const float om1 = 1.0f - omega;
    const bool store_lane = (lane >= 2) && (lane <= 61);

    // windows relative to the column c of stage 1: O old at c-1,c,c+1; A after stage 1 at c-2,c-1; B after
    // sweep 1 at c-3,c-2; C after stage 3 at c-4,c-3; K coefficients at c..c-3
    float Om[4], Oc[4], Op[4], A2[4], A1[4], B3[4], B2[4], C4[4], C3[4];
    float K0[NCF][4], K1[NCF][4], K2[NCF][4], K3[NCF][4];
    int c = j0 - 3;
    rb_load4<VEC>(Om, P.x_in, c - 1, r, nrows, ncols);
    rb_load4<VEC>(Oc, P.x_in, c, r, nrows, ncols);
    rb_load4<VEC>(Op, P.x_in, c + 1, r, nrows, ncols);
#pragma unroll
    for (int e = 0; e < 4; e++) A2[e] = A1[e] = B3[e] = B2[e] = C4[e] = C3[e] = 0.0f;
#pragma unroll
    for (int f = 0; f < NCF; f++) {
        rb_load4<VEC>(K0[f], P.cf[f], c, r, nrows, ncols);
#pragma unroll
        for (int e = 0; e < 4; e++) K1[f][e] = K2[f][e] = K3[f][e] = 0.0f;
    }
    auto inner = [&](int col) { return col >= 1 && col <= ncols - 2; };
    for (; c <= j1 + 2; c++) {
        float On[4], Kn[NCF][4];
        rb_load4<VEC>(On, P.x_in, c + 2, r, nrows, ncols);
#pragma unroll
        for (int f = 0; f < NCF; f++) rb_load4<VEC>(Kn[f], P.cf[f], c + 1, r, nrows, ncols);

        if (FIRST) {
#pragma unroll
            for (int e = 0; e < 4; e++) {
                float k[NCF];
#pragma unroll
                for (int f = 0; f < NCF; f++) k[f] = K0[f][e];
                p8_derive(k);
                K0[ModelPde8::cB][e] = k[ModelPde8::cB];
                K0[ModelPde8::cInv][e] = k[ModelPde8::cInv];
            }
            if (store_lane && c >= j0 && c < j1) {
                rb_store4<VEC>(K0[ModelPde8::cB], dout0, c, r, nrows);
                rb_store4<VEC>(K0[ModelPde8::cInv], dout1, c, r, nrows);
            }
        }
        const bool c_even = ((c + col0) & 1) == 0; // class of columns c and c-2; c-1 and c-3 are of the other class
        const int c1 = c - 1, c2 = c - 2, c3 = c - 3;

        // stage 1: colours 0,1 of sweep 1 on column c (even class), from the old columns on both sides
        float A0[4];
#pragma unroll
        for (int e = 0; e < 4; e++) A0[e] = Oc[e];
        if (c_even && inner(c)) {
            p8_phase<0>(A0, Om, Op, K0, r, nrows, omega, om1);
            p8_phase<1>(A0, Om, Op, K0, r, nrows, omega, om1);
        }
        // stage 2: colours 2,3 of sweep 1 on column c-1 (odd class), from the finished even columns; then it is
        // "column c-1 after sweep 1" whatever its class: replicate its border rows (pdeSolvers.c:249-255)
        float B1[4];
#pragma unroll
        for (int e = 0; e < 4; e++) B1[e] = A1[e];
        if (inner(c1)) {
            if (c_even) {
                p8_phase<0>(B1, A2, A0, K1, r, nrows, omega, om1);
                p8_phase<1>(B1, A2, A0, K1, r, nrows, omega, om1);
            }
            p8_replicate_rows(B1, r, nrows, VEC);
        }
        // stage 3: colours 0,1 of sweep 2 on column c-2 (even class); a border column is the replicate of its
        // inner neighbour after sweep 1 (:256-262), i.e. B(c-2) itself
        float C2[4];
#pragma unroll
        for (int e = 0; e < 4; e++) C2[e] = B2[e];
        if (c_even && inner(c2)) {
            float Pv[4], Nx[4];
#pragma unroll
            for (int e = 0; e < 4; e++) {
                Pv[e] = inner(c2 - 1) ? B3[e] : B2[e];
                Nx[e] = inner(c2 + 1) ? B1[e] : B2[e];
            }
            p8_phase<0>(C2, Pv, Nx, K2, r, nrows, omega, om1);
            p8_phase<1>(C2, Pv, Nx, K2, r, nrows, omega, om1);
        }
        // stage 4: colours 2,3 of sweep 2 on column c-3 (odd class); store column c-3
        if (c3 >= j0 && c3 < j1 && inner(c3)) {
            float F[4];
#pragma unroll
            for (int e = 0; e < 4; e++) F[e] = C3[e];
            if (c_even) {
                float Pv[4], Nx[4];
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    Pv[e] = inner(c3 - 1) ? C4[e] : B3[e];
                    Nx[e] = inner(c3 + 1) ? C2[e] : B3[e];
                }
                p8_phase<0>(F, Pv, Nx, K3, r, nrows, omega, om1);
                p8_phase<1>(F, Pv, Nx, K3, r, nrows, omega, om1);
            }
            p8_replicate_rows(F, r, nrows, VEC);
            if (store_lane) {
                rb_store4<VEC>(F, P.x_out, c3, r, nrows);
                if (c3 == 1) rb_store4<VEC>(F, P.x_out, 0, r, nrows);
                if (c3 == ncols - 2) rb_store4<VEC>(F, P.x_out, ncols - 1, r, nrows);
            }
        }
#pragma unroll
        for (int e = 0; e < 4; e++) {
            C4[e] = C3[e];
            C3[e] = C2[e];
            B3[e] = B2[e];
            B2[e] = B1[e];
            A2[e] = A1[e];
            A1[e] = A0[e];
            Om[e] = Oc[e];
            Oc[e] = Op[e];
            Op[e] = On[e];
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

struct Pde8Layout {
    static constexpr int NCF = ModelPde8::NCF;
    static constexpr int CST = NCF * 64 * EX_STR;  // coefficient chunk: [plane][column][20] (16 rows used)
    static constexpr int XST = 65 * EX_STR;        // X chunk: 64 own columns + the east edge column, 20 rows each
    static constexpr int WED = EX_STR;             // west edge column (new values), 20 rows
    static constexpr int BUF = CST + XST + WED;
    static constexpr int OUTB = 64 * EX_STR;
    static constexpr size_t LDS_BYTES = (size_t)(2 * BUF + 2 * OUTB) * sizeof(float);
};

// One workgroup = one tile (a,b,t) = two waves (mover + compute), four chunks of 16 steps; the same
// structure as k_sor_exact (pdeip_sor_exact.hpp) with a skew of two rows per lane.  The X chunk of a
// column holds 20 rows (its 16 centre rows + 4 of look-ahead), so the south tap and the three east
// taps (the neighbour lane's rows +1,+2,+3) are plain LDS reads of staged sweep t-1 values; the three
// west taps come from the neighbour lane's last three results through one DPP shift per step.
// Border cells hold the previous sweep's replicate: in sweeps t>0 the mover substitutes them from the
// ring side array while staging (tiles that touch the border take an element-wise path), and the
// compute wave records this sweep's ring.
__global__ void __launch_bounds__(128)
k_pde8_exact(Pde8Planes P, float *side, int nrows, int ncols, int A, int B, int T, int m, float omega,
             size_t frame_stride)
{
    using L = Pde8Layout;
    constexpr int NCF = ModelPde8::NCF;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *outb_base = smem + 2 * L::BUF;

    const int lane = threadIdx.x & 63;
    const bool mover = (threadIdx.x >> 6) == 1;
    const int b = blockIdx.x % B, t = blockIdx.x / B;
    const int a = m - P8_G * b - P8_H * t;
    if (a < 0 || a >= A) return;
    const size_t fo = (size_t)blockIdx.y * frame_stride;
    float *x = P.x + fo;
    const float *cfp[NCF];
#pragma unroll
    for (int f = 0; f < NCF; f++) cfp[f] = P.cf[f] + fo;
    const size_t sstride = pde8_side_stride(nrows, ncols);
    // border ring after sweep t-1 (read) and after sweep t (written): top[ncols] bot[ncols] left[nrows] right[nrows]
    float *ring_w = side + ((size_t)blockIdx.y * T + t) * sstride;
    const float *ring_r = (t > 0) ? ring_w - sstride : nullptr;

    const int jbase = 1 + 64 * b;
    const int i00 = 1 + a * P8_R;
    auto crow = [&](int i) { return i < 0 ? 0 : (i > nrows - 1 ? nrows - 1 : i); };
    auto ccol = [&](int jj) { return jj < 0 ? 0 : (jj > ncols - 1 ? ncols - 1 : jj); };
    // value of cell (ii,jj) as sweep t sees a cell it has not relaxed: border cells hold the replicate of
    // sweep t-1 (the caller's cell in sweep 0).  Out-of-image coordinates are clamped (value unused).
    auto cell = [&](int ii, int jj) __attribute__((always_inline)) -> float {
        ii = crow(ii);
        jj = ccol(jj);
        if (ring_r != nullptr) {
            if (ii == 0) return ring_r[jj];
            if (ii == nrows - 1) return ring_r[ncols + jj];
            if (jj == 0) return ring_r[2 * ncols + ii];
            if (jj == ncols - 1) return ring_r[2 * ncols + nrows + ii];
        }
        return x[(size_t)jj * nrows + ii];
    };
    const int lcol = lane >> 2, lrq = lane & 3;

    if (mover) {
        // ================================ mover wave ===========================================
        f4u cA[NCF][4], cB[NCF][4], xA[6], xB[6];
        // does the tile's X traffic touch border cells (ring substitution, t>0) or could a 16-byte access
        // leave the plane?  Decided once per tile (static load counts per path).
        const bool xborder = (i00 - 126 <= 0) || (i00 + P8_R + 4 >= nrows - 1) || (jbase - 1 == 0) || (jbase + 64 >= ncols - 1);
        const int jmax = jbase + 64 < ncols - 1 ? jbase + 64 : ncols - 1;
        const bool inside_plane = (long)jmax * nrows + (i00 + P8_R + 8) < (long)nrows * ncols && (jbase - 1) * (long)nrows + i00 - 1 >= 0;
        const bool xfast = inside_plane && !(t > 0 && xborder);
        // path 2: everything 16-byte; path 1: coefficients 16-byte, X element-wise with ring substitution
        // (border tiles of sweeps t>0); path 0: everything element-wise and row-clamped (end of the plane)
        // replace the border cells among rows row..row+3 of column jj by the previous sweep's ring values
        auto patch = [&](f4u &v, int row, int jj) __attribute__((always_inline)) {
            const bool colb = (jj == 0) || (jj == ncols - 1);
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const int ii = row + e;
                if (ii >= 0 && ii <= nrows - 1 && jj >= 0 && jj <= ncols - 1 && (colb || ii == 0 || ii == nrows - 1)) v.v[e] = cell(ii, jj);
            }
        };
        auto fetch = [&](int k, f4u (&cpre)[NCF][4], f4u (&xpre)[6], auto path_tag) __attribute__((always_inline)) {
            constexpr bool CFAST = decltype(path_tag)::value >= 1, FAST = decltype(path_tag)::value >= 2;
#pragma unroll
            for (int g = 0; g < 4; g++) {
                const int col = 16 * g + lcol;
                const int jj = ccol(jbase + col);
                const int row = i00 - P8_SKEW * col + EX_CH * k + 4 * lrq;
#pragma unroll
                for (int f = 0; f < NCF; f++) {
                    const float *src = cfp[f] + (size_t)jj * nrows;
                    if (CFAST) cpre[f][g].load(src + row);
                    else {
#pragma unroll
                        for (int e = 0; e < 4; e++) cpre[f][g].v[e] = src[crow(row + e)];
                    }
                }
                if (CFAST) {
                    xpre[g].load(x + (size_t)jj * nrows + row);
                    if (!FAST) patch(xpre[g], row, jbase + col);
                } else {
#pragma unroll
                    for (int e = 0; e < 4; e++) xpre[g].v[e] = cell(row + e, jbase + col);
                }
            }
            { // fifth quad (rows +16..+19) of every own column: lane -> column
                const int row = i00 - P8_SKEW * lane + EX_CH * k + 16;
                if (CFAST) {
                    xpre[4].load(x + (size_t)ccol(jbase + lane) * nrows + row);
                    if (!FAST) patch(xpre[4], row, jbase + lane);
                } else {
#pragma unroll
                    for (int e = 0; e < 4; e++) xpre[4].v[e] = cell(row + e, jbase + lane);
                }
            }
            { // lanes 0-4: east edge column (as lane 64); lanes 5-9: west edge column, rows i00-1 ...; others repeat
                const int w = lane % 10;
                const bool west = w >= 5;
                const int jj = west ? jbase - 1 : jbase + 64;
                const int row = (west ? i00 - 1 + 4 * (w - 5) : i00 - P8_SKEW * 64 + 4 * w) + EX_CH * k;
                if (CFAST) {
                    xpre[5].load(x + (size_t)ccol(jj) * nrows + row);
                    if (!FAST) patch(xpre[5], row, jj);
                } else {
#pragma unroll
                    for (int e = 0; e < 4; e++) xpre[5].v[e] = cell(row + e, jj);
                }
            }
        };
        auto stash = [&](const f4u (&cpre)[NCF][4], const f4u (&xpre)[6], int buf) __attribute__((always_inline)) {
            float *cst = smem + buf * L::BUF, *xst = cst + L::CST, *wed = xst + L::XST;
            auto put = [&](float *dst, const f4u &v) { *reinterpret_cast<float4 *>(dst) = make_float4(v.v[0], v.v[1], v.v[2], v.v[3]); };
#pragma unroll
            for (int g = 0; g < 4; g++) {
                const int col = 16 * g + lcol;
#pragma unroll
                for (int f = 0; f < NCF; f++) put(&cst[(f * 64 + col) * EX_STR + 4 * lrq], cpre[f][g]);
                put(&xst[col * EX_STR + 4 * lrq], xpre[g]);
            }
            put(&xst[lane * EX_STR + 16], xpre[4]);
            if (lane < 5) put(&xst[64 * EX_STR + 4 * lane], xpre[5]);
            else if (lane < 10) put(&wed[4 * (lane - 5)], xpre[5]);
        };
        auto store_out = [&](int k) __attribute__((always_inline)) {
            const float *outb = outb_base + (k & 1) * L::OUTB;
            const int lo_row = i00 - 126 + EX_CH * k, hi_row = i00 + EX_CH * k + EX_CH - 1;
            const bool all_valid = (lo_row >= 1) && (hi_row <= nrows - 2) && (jbase + 63 <= ncols - 2);
#pragma unroll
            for (int g = 0; g < 4; g++) {
                const int col = 16 * g + lcol;
                const int jj = jbase + col;
                const int row = i00 - P8_SKEW * col + EX_CH * k + 4 * lrq;
                const float4 v = *reinterpret_cast<const float4 *>(&outb[col * EX_STR + 4 * lrq]);
                float *dst = x + (size_t)ccol(jj) * nrows;
                if (all_valid) {
                    f4u o;
                    o.v[0] = v.x; o.v[1] = v.y; o.v[2] = v.z; o.v[3] = v.w;
                    o.store(dst + row);
                } else if (jj <= ncols - 2) {
                    if (row >= 1 && row <= nrows - 2) dst[row] = v.x;
                    if (row + 1 >= 1 && row + 1 <= nrows - 2) dst[row + 1] = v.y;
                    if (row + 2 >= 1 && row + 2 <= nrows - 2) dst[row + 2] = v.z;
                    if (row + 3 >= 1 && row + 3 <= nrows - 2) dst[row + 3] = v.w;
                }
            }
        };
        auto run = [&](auto fast_tag) __attribute__((always_inline)) {
            fetch(0, cA, xA, fast_tag);
            stash(cA, xA, 0);
            fetch(1, cB, xB, fast_tag);
            lds_barrier(); // #0
            fetch(2, cA, xA, fast_tag);
            stash(cB, xB, 1);
            lds_barrier(); // #1
            fetch(3, cB, xB, fast_tag);
            stash(cA, xA, 0);
            store_out(0);
            lds_barrier(); // #2
            stash(cB, xB, 1);
            store_out(1);
            lds_barrier(); // #3
            store_out(2);
            lds_barrier(); // #4
            store_out(3);
        };
        if (xfast) run(std::integral_constant<int, 2>{});
        else if (inside_plane) run(std::integral_constant<int, 1>{});
        else run(std::integral_constant<int, 0>{});
        return;
    }

    // ================================== compute wave ===========================================
    const int j = jbase + lane;
    const bool col_ok = j <= ncols - 2;
    const float om1 = 1.0f - omega;
    const int i0 = i00 - P8_SKEW * lane;
    // state at step 0: values relaxed by earlier launches (tile a-1 / strip b-1) or border cells
    float prev = 0.0f;
    float n0 = cell(i0 - 1, j), nw = cell(i0 - 1, j - 1), w = cell(i0, j - 1), sw0 = cell(i0 + 1, j - 1);
    const float topb = cell(0, j), topbW = cell(0, j - 1), botbW = cell(nrows - 1, j - 1);
    lds_barrier(); // #0

    for (int k = 0; k < P8_R / EX_CH; k++) {
        const float *cst = smem + (k & 1) * L::BUF, *xst = cst + L::CST, *wed = xst + L::XST;
        float *outb = outb_base + (k & 1) * L::OUTB;
        auto relax_chunk = [&](auto interior_tag) __attribute__((always_inline)) {
            constexpr bool INTERIOR = decltype(interior_tag)::value;
#pragma unroll
            for (int mq = 0; mq < EX_CH / 4; mq++) {
                float4 ck[NCF], res;
                float xo[8], xe[8], we[8];
#pragma unroll
                for (int f = 0; f < NCF; f++) ck[f] = *reinterpret_cast<const float4 *>(&cst[(f * 64 + lane) * EX_STR + 4 * mq]);
                {
                    const float4 a0 = *reinterpret_cast<const float4 *>(&xst[lane * EX_STR + 4 * mq]);
                    const float4 a1 = *reinterpret_cast<const float4 *>(&xst[lane * EX_STR + 4 * mq + 4]);
                    const float4 b0 = *reinterpret_cast<const float4 *>(&xst[(lane + 1) * EX_STR + 4 * mq]);
                    const float4 b1 = *reinterpret_cast<const float4 *>(&xst[(lane + 1) * EX_STR + 4 * mq + 4]);
                    const float4 c0 = *reinterpret_cast<const float4 *>(&wed[4 * mq]);
                    const float4 c1 = *reinterpret_cast<const float4 *>(&wed[4 * mq + 4]);
                    xo[0] = a0.x; xo[1] = a0.y; xo[2] = a0.z; xo[3] = a0.w; xo[4] = a1.x; xo[5] = a1.y; xo[6] = a1.z; xo[7] = a1.w;
                    xe[0] = b0.x; xe[1] = b0.y; xe[2] = b0.z; xe[3] = b0.w; xe[4] = b1.x; xe[5] = b1.y; xe[6] = b1.z; xe[7] = b1.w;
                    we[0] = c0.x; we[1] = c0.y; we[2] = c0.z; we[3] = c0.w; we[4] = c1.x; we[5] = c1.y; we[6] = c1.z; we[7] = c1.w;
                }
#pragma unroll
                for (int xq = 0; xq < 4; xq++) {
                    const int q = EX_CH * k + 4 * mq + xq;
                    const int i = i0 + q;
                    const bool active = INTERIOR || (col_ok && (i >= 1) && (i <= nrows - 2));
                    auto el = [&](const float4 &v) { return xq == 0 ? v.x : (xq == 1 ? v.y : (xq == 2 ? v.z : v.w)); };
                    // south-west (new): lane l-1 relaxed (i+1, j-1) one step ago; lane 0 reads the west edge column
                    float sw = dpp_from_lower_lane(prev, we[xq + 2]);
                    if (q == 0) {
                        if (lane != 0) sw = sw0;
                        else { nw = we[0]; w = we[1]; }
                    }
                    if (!INTERIOR && lane != 0) { // the tap is a border cell of the top/bottom ring
                        if (i + 1 == 0) sw = topbW;
                        if (i + 1 == nrows - 1) sw = botbW;
                    }
                    float nn = (q == 0) ? n0 : prev;
                    if (!INTERIOR && i == 1) nn = topb;
                    float kk[NCF];
#pragma unroll
                    for (int f = 0; f < NCF; f++) kk[f] = el(ck[f]);
                    // update(xc, xW, xE, xN, xS, xNW, xNE, xSW, xSE)
                    const float v = ModelPde8::update(xo[xq], w, xe[xq + 2], nn, xo[xq + 1], nw, xe[xq + 1], sw, xe[xq + 3], kk, omega, om1);
                    if (active) prev = v;
                    if (xq == 0) res.x = v; else if (xq == 1) res.y = v; else if (xq == 2) res.z = v; else res.w = v;
                    if (!INTERIOR && active) { // border ring after this sweep: nearest-interior replicate (pdeSolvers.c:249-262)
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
                    nw = w;
                    w = sw;
                }
                *reinterpret_cast<float4 *>(&outb[lane * EX_STR + 4 * mq]) = res;
            }
        };
        {
            const int lo_row = i00 - 126 + EX_CH * k, hi_row = i00 + EX_CH * k + EX_CH - 1;
            const bool interior = (lo_row >= 2) && (hi_row <= nrows - 3) && (jbase >= 2) && (jbase + 63 <= ncols - 3);
            if (interior) relax_chunk(std::true_type{});
            else relax_chunk(std::false_type{});
        }
        lds_barrier(); // #k+1
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
    // > 64 KiB of dynamic LDS needs an explicit opt-in, per device (pdeip_ctx: ensure_lds)
    if (ensure_lds(reinterpret_cast<const void *>(&k_pde8_exact), Pde8Layout::LDS_BYTES) != PDEIP_OK) return -1;
    int launches = 0;
    for (int m = 0; m <= last_m; m++) {
        hipLaunchKernelGGL(k_pde8_exact, grid, dim3(128), Pde8Layout::LDS_BYTES, s, P, side, nrows, ncols, A, B, iter, m, omega, n);
        launches++;
    }
    const int nb = 2 * ncols + 2 * (nrows - 2);
    hipLaunchKernelGGL(k_fill_borders, dim3((nb + 255) / 256, nframes, 1), dim3(256), 0, s, P.x, P.x, 1, nrows, ncols, n);
    return launches + 1;
}

} // namespace pdeip
