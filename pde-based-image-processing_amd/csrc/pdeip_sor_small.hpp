// pdeip_sor_small.hpp -- all `iter` red-black sweeps of a small frame in ONE workgroup (gfx950).
//
// The coarse scales of the drivers' pyramids (68x120 and below in the 4K multigrid run: 240 of its 324 solver calls) are
// bound by what a launch costs, not by what it computes: the marching kernels need 2-4 launches of 20-40 us per call there.
// A frame whose iterate fits LDS is relaxed by one 1024-thread workgroup instead:
//   * the iterate fields (and the read-only neighbour fields of the late-linearisation models) live in LDS, whole frame,
//     borders included, in the MATLAB layout;
//   * every thread owns up to Q pixels of each colour (slot s = thread + 1024 q of the colour's column-major enumeration)
//     and keeps their coefficients -- divisors derived once, as the reference's first sweep does -- in registers for the
//     whole call, so a half-sweep is: five LDS reads per field, Mdl::update(), one LDS write;
//   * one workgroup barrier per half-sweep, the border replicate (rows, then columns: opticalflowSolvers.c:161-179) in LDS
//     after every sweep, one coalesced write-back at the end.  In place or out of place, any nrows (no 16-byte alignment
//     needed), any `iter`, multi-frame planes (blockIdx.x = frame).
// Same per-pixel arithmetic as every other ordering (Mdl::update): bit-identical to the marching kernels.
#pragma once
#include "pdeip_models.hpp"

namespace pdeip {

constexpr int SMALL_THREADS = 1024;
constexpr int SMALL_Q = 4; // pixels of one colour per thread

template <class Mdl> struct SmallLayout {
    static constexpr int NF = Mdl::NIT + Mdl::NRO;
    // slots per column of one colour, frame fits?
    static int pr(int nrows) { return (nrows - 2 + 1) / 2; }
    static bool fits(int nrows, int ncols)
    {
        const long slots = (long)pr(nrows) * (ncols - 2);
        return slots <= (long)SMALL_THREADS * SMALL_Q && lds_bytes(nrows, ncols) <= (size_t)150 * 1024;
    }
    static size_t lds_bytes(int nrows, int ncols) { return (size_t)NF * nrows * ncols * sizeof(float); }
};

template <class Mdl>
__global__ void __launch_bounds__(SMALL_THREADS)
k_sor_small(SweepPlanes<Mdl> P, int nrows, int ncols, int iter, float omega, int col0, size_t frame_stride)
{
    constexpr int NIT = Mdl::NIT, NRO = Mdl::NRO, NRO1 = at_least_one<NRO>::value, NCF = Mdl::NCF, Q = SMALL_Q;
    extern __shared__ __attribute__((aligned(16))) float small_lds[]; // [NIT + NRO][ncols][nrows]
    const int tid = threadIdx.x;
    const int N = nrows * ncols;
    const size_t fo = (size_t)blockIdx.x * frame_stride;
    const float om1 = 1.0f - omega;

    for (int idx = tid; idx < N; idx += SMALL_THREADS) {
#pragma unroll
        for (int f = 0; f < NIT; f++) small_lds[f * N + idx] = P.it_in[f][fo + idx];
#pragma unroll
        for (int f = 0; f < NRO; f++) small_lds[(NIT + f) * N + idx] = P.ro[f][fo + idx];
    }

    // my pixels: colour c, slot q -> column j = 1 + s / PR, row i = first row of colour c in column j + 2 (s % PR)
    const int PR = (nrows - 2 + 1) / 2;
    int pos[2][Q];    // j * nrows + i, or -1
    float cf[2][Q][NCF];
#pragma unroll
    for (int c = 0; c < 2; c++)
#pragma unroll
        for (int q = 0; q < Q; q++) {
            const int s = tid + SMALL_THREADS * q;
            const int j = 1 + s / PR, k = s % PR;
            const int i = 1 + ((1 + j + col0 + c) & 1) + 2 * k;
            const bool ok = (j <= ncols - 2) && (i <= nrows - 2);
            pos[c][q] = ok ? j * nrows + i : -1;
            const int p = ok ? j * nrows + i : 0;
            float kk[NCF];
#pragma unroll
            for (int f = 0; f < NCF; f++) kk[f] = P.cf[f][fo + p];
            Mdl::derive(kk); // the divisor planes of the reference's first sweep (:111-127), once per call
#pragma unroll
            for (int f = 0; f < NCF; f++) cf[c][q][f] = kk[f];
        }
    __syncthreads();

    for (int sweep = 0; sweep < iter; sweep++) {
#pragma unroll
        for (int c = 0; c < 2; c++) {
#pragma unroll
            for (int q = 0; q < Q; q++) {
                const int p = pos[c][q];
                if (p >= 0) {
                    float cc[NIT], w[NIT], e[NIT], n[NIT], s[NIT];
                    float rc[NRO1], rw[NRO1], re[NRO1], rn[NRO1], rs[NRO1];
#pragma unroll
                    for (int f = 0; f < NIT; f++) {
                        const float *F = small_lds + f * N;
                        cc[f] = F[p];
                        w[f] = F[p - nrows];
                        e[f] = F[p + nrows];
                        n[f] = F[p - 1];
                        s[f] = F[p + 1];
                    }
#pragma unroll
                    for (int f = 0; f < NRO1; f++) {
                        if (NRO > 0) {
                            const float *F = small_lds + (NIT + (NRO > 0 ? f : 0)) * N;
                            rc[f] = F[p];
                            rw[f] = F[p - nrows];
                            re[f] = F[p + nrows];
                            rn[f] = F[p - 1];
                            rs[f] = F[p + 1];
                        } else {
                            rc[f] = rw[f] = re[f] = rn[f] = rs[f] = 0.0f;
                        }
                    }
                    Mdl::update(cc, w, e, n, s, rc, rw, re, rn, rs, cf[c][q], omega, om1);
#pragma unroll
                    for (int f = 0; f < NIT; f++) small_lds[f * N + p] = cc[f];
                }
            }
            __syncthreads();
        }
        // border replicate: rows first, then columns (:161-179)
        for (int j = tid; j < ncols; j += SMALL_THREADS)
#pragma unroll
            for (int f = 0; f < NIT; f++) {
                float *F = small_lds + f * N + j * nrows;
                F[0] = F[1];
                F[nrows - 1] = F[nrows - 2];
            }
        __syncthreads();
        for (int i = tid; i < nrows; i += SMALL_THREADS)
#pragma unroll
            for (int f = 0; f < NIT; f++) {
                float *F = small_lds + f * N;
                F[i] = F[nrows + i];
                F[(ncols - 1) * nrows + i] = F[(ncols - 2) * nrows + i];
            }
        __syncthreads();
    }
    for (int idx = tid; idx < N; idx += SMALL_THREADS)
#pragma unroll
        for (int f = 0; f < NIT; f++) P.it_out[f][fo + idx] = small_lds[f * N + idx];
}

} // namespace pdeip
