// pdeip_alr.hpp -- alternating line relaxation (solver = 2 of the gateways).
//
// Reference: GS_ALR_SOR_{elin4,llin4,llin8}_2d (opticalflowSolvers.c:196,690,1677) with their
// {west,middle,east}Column_* / {north,middle,south}Row_* line solvers (:1763-3914), the disparity twin
// (disparitySolvers.c:154, :1376-2029) and GS_ALR_SOR_{4,8}_2d (pdeSolvers.c:277,344, :409-1393).
//
// Every line solver of the reference is one Thomas (TDMA) solve along an image column or row in which
// all pixels of the line are unknowns and a neighbour outside the image drops out of the diagonal
// and the right-hand side; the variants differ only in which terms are present and in which order
// they are added.  `Model::coef()` rebuilds one tridiagonal row (a,b,c,d) with the reference's term
// order; two kernels consume it:
//
//   k_alr_lex   EXACT_ORDER: the reference's line order.  Line l needs the finished line l-1, and the
//               Thomas recurrences are serial along the line, so the dependency chain crosses the whole
//               frame: there is no parallel schedule that keeps the arithmetic.  One workgroup per frame
//               walks the lines (see below).  Bit-identical, CPU-class speed.
//   k_alr_zebra RED_BLACK: "zebra" order -- every even line, then every odd line.  Lines of one colour
//               only read the other colour, so they are solved concurrently, one lane per line, with the
//               same per-line arithmetic.
#pragma once
#include <hip/hip_runtime.h>

#include <type_traits>

#include "pdeip_models.hpp"

namespace pdeip {

struct Tri {
    float a, b, c, d;
};

// Plane addressing.  The column pass works on the planes as MATLAB hands them over (element (i,j) at
// j*nrows + i: a column is contiguous).  The row pass works on TRANSPOSED copies (element (i,j) at
// i*ncols + j: a row is contiguous) that the host keeps in step -- a line is then contiguous in both
// passes, and the strided walk along image rows (one cache line and one address translation per pixel
// and plane) never happens.  AlrAt (below) is the only place that knows where a pixel sits.

// out[f][a][b] = in[f][b][a]: `in` is [F][nb][na] with a fastest; 32x32 tiles through LDS, coalesced both ways.
__global__ void __launch_bounds__(256) k_alr_transpose(float *__restrict__ out, const float *__restrict__ in, int na, int nb)
{
    __shared__ float tile[32][33];
    const size_t fo = (size_t)blockIdx.z * na * nb;
    const int a0 = blockIdx.x * 32, b0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5; // 32 x 8
#pragma unroll
    for (int r = 0; r < 32; r += 8) {
        const int a = a0 + tx, b = b0 + ty + r;
        if (a < na && b < nb) tile[ty + r][tx] = in[fo + (size_t)b * na + a];
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 32; r += 8) {
        const int b = b0 + tx, a = a0 + ty + r;
        if (a < na && b < nb) out[fo + (size_t)a * nb + b] = tile[tx][ty + r];
    }
}

// Several planes in one launch (a coarse multigrid scale's call is bound by its number of launches): blockIdx.z = plane * F + frame.
constexpr int ALR_TB_MAX = 16;
struct AlrTransposeBatch {
    float *out[ALR_TB_MAX];
    const float *in[ALR_TB_MAX];
};

__global__ void __launch_bounds__(256) k_alr_transpose_batch(AlrTransposeBatch B, int na, int nb, int nframes)
{
    __shared__ float tile[32][33];
    const int plane = blockIdx.z / nframes, frame = blockIdx.z % nframes;
    const float *__restrict__ in = B.in[plane] + (size_t)frame * na * nb;
    float *__restrict__ out = B.out[plane] + (size_t)frame * na * nb;
    const int a0 = blockIdx.x * 32, b0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
#pragma unroll
    for (int r = 0; r < 32; r += 8) {
        const int a = a0 + tx, b = b0 + ty + r;
        if (a < na && b < nb) tile[ty + r][tx] = in[(size_t)b * na + a];
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 32; r += 8) {
        const int b = b0 + tx, a = a0 + ty + r;
        if (a < na && b < nb) out[(size_t)a * nb + b] = tile[tx][ty + r];
    }
}

// 16-byte load at 4-byte alignment (a vector type with reduced alignment keeps it one global_load_dwordx4)
typedef float alr_v4 __attribute__((ext_vector_type(4), aligned(4)));
__device__ __forceinline__ void alr_ld4(const float *p, float (&v)[4])
{
    const alr_v4 t = *reinterpret_cast<const alr_v4 *>(p);
    v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
}

// Where element k of line l sits and what surrounds it, in line terms: `before`/`after` are the
// neighbours on the line, `prev`/`next` the same element of the neighbouring lines.  On a column (MATLAB
// layout) before/after/prev/next = N/S/W/E, on a row (transposed layout) W/E/N/S; in both a line is
// contiguous and the neighbouring lines are n elements away.
struct AlrAt {
    size_t pos;
    long n; // line length = distance to the same element of the next line
    bool hasBefore, hasAfter, hasPrev, hasNext;
    __device__ __forceinline__ AlrAt(int l, int k, int n_, int nlines)
        : pos((size_t)l * n_ + k), n(n_), hasBefore(k > 0), hasAfter(k < n_ - 1), hasPrev(l > 0), hasNext(l < nlines - 1)
    {
    }
    __device__ __forceinline__ size_t prev() const { return hasPrev ? pos - n : pos; } // clamped: loads stay unconditional
    __device__ __forceinline__ size_t next() const { return hasNext ? pos + n : pos; }
    __device__ __forceinline__ size_t before() const { return hasBefore ? pos - 1 : pos; }
    __device__ __forceinline__ size_t after() const { return hasAfter ? pos + 1 : pos; }
};

// "the terms that are present, in this order", as a C expression t1 + t2 + ... evaluates them
__device__ __forceinline__ void acc_add(float &v, bool &have, float t)
{
    v = have ? v + t : t;
    have = true;
}

// (line_coef / line_coef4 below the models)

// ---- early linearisation, 4 neighbours (opticalflowSolvers.c:1763-2410) ------------------------------
struct AlrElin4 {
    struct Ctx {
        const float *X, *O, *M, *C, *D, *wW, *wN, *wE, *wS; // X: the field being solved, O: the other field
        __device__ void shift(size_t) {}
    };
    static constexpr bool INTERIOR_LINES = false;
    static constexpr bool SOUTH_TRUEDIV = false; // southRow_elin4 multiplies by a reciprocal like every other line (:2364-2366)
    struct In { // x1, x2: the two off-line neighbours of X in the order they enter d: W,E on a column; S,N on a row
        float wN, wS, wE, wW, x1, x2, C, D, M, O;
    };
    template <bool vertical>
    __device__ __forceinline__ static Tri math(const In &v, bool hasN, bool hasS, bool hasW, bool hasE)
    {
        float b = 0.0f, d = 0.0f;
        bool hb = false, hd = false;
        Tri t;
        if (hasN) acc_add(b, hb, v.wN); // b = wN + wS + wE + wW, missing ones skipped (:1917)
        if (hasS) acc_add(b, hb, v.wS);
        if (hasE) acc_add(b, hb, v.wE);
        if (hasW) acc_add(b, hb, v.wW);
        if (vertical) { // d = wW*U_w + wE*U_e (:1919)
            if (hasW) acc_add(d, hd, v.wW * v.x1);
            if (hasE) acc_add(d, hd, v.wE * v.x2);
            t.a = hasN ? -v.wN : 0.0f;
            t.c = hasS ? -v.wS : 0.0f;
        } else { // d = wS*U_s + wN*U_n (:2247)
            if (hasS) acc_add(d, hd, v.wS * v.x1);
            if (hasN) acc_add(d, hd, v.wN * v.x2);
            t.a = hasW ? -v.wW : 0.0f;
            t.c = hasE ? -v.wE : 0.0f;
        }
        const float MO = v.M * v.O;
        if (!is_nan(v.C)) { // :1921-1926
            b += v.D;
            d += v.C;
            d -= MO;
        }
        t.b = b;
        t.d = d;
        return t;
    }
    template <bool vertical> __device__ __forceinline__ static Tri coef(const Ctx &q, const AlrAt &at)
    {
        const size_t p = at.pos;
        const In v{q.wN[p], q.wS[p], q.wE[p], q.wW[p], q.X[vertical ? at.prev() : at.next()], q.X[vertical ? at.next() : at.prev()],
                   q.C[p], q.D[p], q.M[p], q.O[p]};
        return vertical ? math<true>(v, at.hasBefore, at.hasAfter, at.hasPrev, at.hasNext)
                        : math<false>(v, at.hasPrev, at.hasNext, at.hasBefore, at.hasAfter);
    }
    // elements k..k+3 of one line (all four inside it), operands fetched 16 bytes at a time
    template <bool vertical> __device__ __forceinline__ static void coef4(const Ctx &q, int l, int k, int n, int nlines, Tri (&out)[4])
    {
        const AlrAt at(l, k, n, nlines);
        const size_t p = at.pos;
        float wN[4], wS[4], wE[4], wW[4], x1[4], x2[4], C[4], D[4], M[4], O[4];
        alr_ld4(q.wN + p, wN); alr_ld4(q.wS + p, wS); alr_ld4(q.wE + p, wE); alr_ld4(q.wW + p, wW);
        alr_ld4(q.X + (vertical ? at.prev() : at.next()), x1); alr_ld4(q.X + (vertical ? at.next() : at.prev()), x2);
        alr_ld4(q.C + p, C); alr_ld4(q.D + p, D); alr_ld4(q.M + p, M); alr_ld4(q.O + p, O);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const In v{wN[e], wS[e], wE[e], wW[e], x1[e], x2[e], C[e], D[e], M[e], O[e]};
            const bool hasBefore = k + e > 0, hasAfter = k + e < n - 1;
            out[e] = vertical ? math<true>(v, hasBefore, hasAfter, at.hasPrev, at.hasNext)
                              : math<false>(v, at.hasPrev, at.hasNext, hasBefore, hasAfter);
        }
    }
};

// ---- late linearisation, 4 neighbours: flow (opticalflowSolvers.c:2415-3100); M == nullptr: disparity
//      (disparitySolvers.c:1376-2029, the same lines without the coupling term) ---------------------------
template <bool COUPLED> struct AlrLlin4T {
    struct Ctx {
        const float *U, *X, *O, *M, *C, *D, *wW, *wN, *wE, *wS; // U: base field, X: increment being solved
        __device__ void shift(size_t) {}
    };
    static constexpr bool INTERIOR_LINES = false;
    // southRow_llin4 (opticalflowSolvers.c:3059-3060) and southRow4 (disparitySolvers.c:1986-1987) DIVIDE in the forward
    // elimination of their middle elements: cp = c / den, dp = (d - dp' a) / den.  Every other line function multiplies by
    // div = 1 / den (audit: DESIGN.md section 5.5).
    static constexpr bool SOUTH_TRUEDIV = true;
    struct In { // d1, d2: the increments of the two off-line neighbours: W,E on a column; S,N on a row
        float wN, wS, wE, wW, Uc, Uw, Ue, Us, Un, d1, d2, C, D, M, O;
    };
    template <bool vertical>
    __device__ __forceinline__ static Tri math(const In &v, bool hasN, bool hasS, bool hasW, bool hasE)
    {
        float b = 0.0f, d = 0.0f;
        bool hb = false, hd = false;
        Tri t;
        if (hasN) acc_add(b, hb, v.wN);
        if (hasS) acc_add(b, hb, v.wS);
        if (hasE) acc_add(b, hb, v.wE);
        if (hasW) acc_add(b, hb, v.wW);
        // d: W, E, S, N; neighbours that are not on the line carry their increment (:2589-2592, :2933-2936)
        if (vertical) {
            if (hasW) acc_add(d, hd, v.wW * (v.Uw - v.Uc + v.d1));
            if (hasE) acc_add(d, hd, v.wE * (v.Ue - v.Uc + v.d2));
            if (hasS) acc_add(d, hd, v.wS * (v.Us - v.Uc));
            if (hasN) acc_add(d, hd, v.wN * (v.Un - v.Uc));
            t.a = hasN ? -v.wN : 0.0f;
            t.c = hasS ? -v.wS : 0.0f;
        } else {
            if (hasW) acc_add(d, hd, v.wW * (v.Uw - v.Uc));
            if (hasE) acc_add(d, hd, v.wE * (v.Ue - v.Uc));
            if (hasS) acc_add(d, hd, v.wS * (v.Us - v.Uc + v.d1));
            if (hasN) acc_add(d, hd, v.wN * (v.Un - v.Uc + v.d2));
            t.a = hasW ? -v.wW : 0.0f;
            t.c = hasE ? -v.wE : 0.0f;
        }
        float MO = 0.0f;
        if constexpr (COUPLED) MO = v.M * v.O;
        if (!is_nan(v.C)) {
            b += v.D;
            d += v.C;
            if constexpr (COUPLED) d -= MO;
        }
        t.b = b;
        t.d = d;
        return t;
    }
    template <bool vertical> __device__ __forceinline__ static Tri coef(const Ctx &q, const AlrAt &at)
    {
        const size_t p = at.pos;
        // geographic neighbours: on a column W/E = prev/next line and N/S = before/after; on a row N/S = prev/next, W/E = before/after
        const size_t pw = vertical ? at.prev() : at.before(), pe = vertical ? at.next() : at.after();
        const size_t pn = vertical ? at.before() : at.prev(), ps = vertical ? at.after() : at.next();
        In v{q.wN[p], q.wS[p], q.wE[p], q.wW[p], q.U[p], q.U[pw], q.U[pe], q.U[ps], q.U[pn],
             q.X[vertical ? pw : ps], q.X[vertical ? pe : pn], q.C[p], q.D[p], 0.0f, 0.0f};
        if constexpr (COUPLED) {
            v.M = q.M[p];
            v.O = q.O[p];
        }
        return vertical ? math<true>(v, at.hasBefore, at.hasAfter, at.hasPrev, at.hasNext)
                        : math<false>(v, at.hasPrev, at.hasNext, at.hasBefore, at.hasAfter);
    }
    template <bool vertical> __device__ __forceinline__ static void coef4(const Ctx &q, int l, int k, int n, int nlines, Tri (&out)[4])
    {
        const AlrAt at(l, k, n, nlines);
        const size_t p = at.pos;
        float wN[4], wS[4], wE[4], wW[4], Uc[4], Up[4], Un[4], dp[4], dn[4], C[4], D[4], M[4] = {0, 0, 0, 0}, O[4] = {0, 0, 0, 0};
        alr_ld4(q.wN + p, wN); alr_ld4(q.wS + p, wS); alr_ld4(q.wE + p, wE); alr_ld4(q.wW + p, wW);
        alr_ld4(q.U + p, Uc); alr_ld4(q.U + at.prev(), Up); alr_ld4(q.U + at.next(), Un);
        alr_ld4(q.X + at.prev(), dp); alr_ld4(q.X + at.next(), dn);
        alr_ld4(q.C + p, C); alr_ld4(q.D + p, D);
        if constexpr (COUPLED) {
            alr_ld4(q.M + p, M);
            alr_ld4(q.O + p, O);
        }
        const float Ubefore = q.U[at.before()];                 // element k-1 (or k itself on the first element: unused then)
        const float Uafter = q.U[k + 4 <= n - 1 ? p + 4 : p + 3]; // element k+4 (or k+3 on the last group: unused then)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float ub = e == 0 ? Ubefore : Uc[e - 1], ua = e == 3 ? Uafter : Uc[e + 1];
            const bool hasBefore = k + e > 0, hasAfter = k + e < n - 1;
            if (vertical) { // W/E = prev/next line, N/S = before/after
                const In v{wN[e], wS[e], wE[e], wW[e], Uc[e], Up[e], Un[e], ua, ub, dp[e], dn[e], C[e], D[e], M[e], O[e]};
                out[e] = math<true>(v, hasBefore, hasAfter, at.hasPrev, at.hasNext);
            } else { // N/S = prev/next line, W/E = before/after
                const In v{wN[e], wS[e], wE[e], wW[e], Uc[e], ub, ua, Un[e], Up[e], dn[e], dp[e], C[e], D[e], M[e], O[e]};
                out[e] = math<false>(v, at.hasPrev, at.hasNext, hasBefore, hasAfter);
            }
        }
    }
};

using AlrLlin4 = AlrLlin4T<true>;  // optical flow
using AlrDisp4 = AlrLlin4T<false>; // disparity: M, O unused

// ---- late linearisation, 8 neighbours (opticalflowSolvers.c:3104-3914) -------------------------------
enum { DN = 0, DS, DE, DW, DNW, DNE, DSW, DSE, DEND };
// term orders of the 18 cases: [pass: 0 column, 1 row][line: first/middle/last][element: first/middle/last][b, d]
#define PDEIP_L8(...) {__VA_ARGS__, DEND}
__device__ const signed char ALR_L8[2][3][3][2][9] = {
    { // column pass: west column (:3104), middle columns (:3237), east column (:3380)
        {{PDEIP_L8(DS, DE, DSE), PDEIP_L8(DS, DE, DSE)},
         {PDEIP_L8(DN, DS, DE, DNE, DSE), PDEIP_L8(DS, DN, DNE, DE, DSE)},
         {PDEIP_L8(DN, DE, DNE), PDEIP_L8(DN, DNE, DE)}},
        {{PDEIP_L8(DS, DE, DW, DSE, DSW), PDEIP_L8(DS, DW, DE, DSE, DSW)},
         {PDEIP_L8(DN, DS, DE, DW, DNW, DNE, DSW, DSE), PDEIP_L8(DN, DS, DW, DNW, DNE, DE, DSW, DSE)},
         {PDEIP_L8(DN, DE, DW, DNW, DNE), PDEIP_L8(DN, DW, DNW, DNE, DE)}},
        {{PDEIP_L8(DS, DW, DSW), PDEIP_L8(DS, DW, DSW)},
         {PDEIP_L8(DN, DS, DW, DNW, DSW), PDEIP_L8(DS, DN, DW, DNW, DSW)},
         {PDEIP_L8(DN, DW, DNW), PDEIP_L8(DN, DW, DNW)}},
    },
    { // row pass: north row (:3513), middle rows (:3646), south row (:3789)
        {{PDEIP_L8(DS, DE, DSE), PDEIP_L8(DE, DSE, DS)},
         {PDEIP_L8(DS, DE, DW, DSW, DSE), PDEIP_L8(DW, DE, DSW, DSE, DS)},
         {PDEIP_L8(DS, DW, DSW), PDEIP_L8(DW, DSW, DS)}},
        {{PDEIP_L8(DN, DS, DE, DNE, DSE), PDEIP_L8(DE, DNE, DSE, DS, DN)},
         {PDEIP_L8(DN, DS, DE, DW, DNW, DNE, DSW, DSE), PDEIP_L8(DW, DE, DNW, DNE, DSW, DSE, DS, DN)},
         {PDEIP_L8(DN, DS, DW, DNW, DSW), PDEIP_L8(DW, DNW, DSW, DS, DN)}},
        {{PDEIP_L8(DN, DE, DNE), PDEIP_L8(DE, DNE, DN)},
         {PDEIP_L8(DN, DE, DW, DNW, DNE), PDEIP_L8(DW, DE, DNW, DNE, DN)},
         {PDEIP_L8(DN, DW, DNW), PDEIP_L8(DW, DNW, DN)}},
    },
};
#undef PDEIP_L8

struct AlrLlin8 {
    struct Ctx {
        const float *U, *X, *O, *M, *C, *D;
        const float *w[8]; // indexed DN..DSE
        __device__ void shift(size_t) {}
    };
    static constexpr bool INTERIOR_LINES = false;
    static constexpr bool SOUTH_TRUEDIV = true; // southRow_llin8 (opticalflowSolvers.c:3871-3872) divides, like southRow_llin4
    __device__ __forceinline__ static int third(int k, int n) { return k == 0 ? 0 : (k == n - 1 ? 2 : 1); }
    // everything one pixel needs, by direction (DN..DSE): weight, base-flow difference U_nb - U_c, increment of the neighbour
    struct In {
        float w[8], g[8], dx[8], C, D, M, O;
    };
    __device__ __forceinline__ static float pick8(const float (&a)[8], int k) // a[k] without indexing registers dynamically
    {
        float r = a[0];
#pragma unroll
        for (int q = 1; q < 8; ++q) r = (k == q) ? a[q] : r;
        return r;
    }
    template <bool vertical>
    __device__ __forceinline__ static Tri math(const In &v, int line_third, int elem_third, bool hasBefore, bool hasAfter)
    {
        float b, d;
        if (line_third == 1 && elem_third == 1) { // the interior case, term orders of :3308-3318 (columns) / :3717-3727 (rows)
            b = v.w[DN] + v.w[DS];
            b = b + v.w[DE];
            b = b + v.w[DW];
            b = b + v.w[DNW];
            b = b + v.w[DNE];
            b = b + v.w[DSW];
            b = b + v.w[DSE];
            if (vertical) { // N, S, W, NW, NE, E, SW, SE; N and S are the unknowns' own line: no increment
                d = v.w[DN] * v.g[DN] + v.w[DS] * v.g[DS];
                d = d + v.w[DW] * (v.g[DW] + v.dx[DW]);
                d = d + v.w[DNW] * (v.g[DNW] + v.dx[DNW]);
                d = d + v.w[DNE] * (v.g[DNE] + v.dx[DNE]);
                d = d + v.w[DE] * (v.g[DE] + v.dx[DE]);
                d = d + v.w[DSW] * (v.g[DSW] + v.dx[DSW]);
                d = d + v.w[DSE] * (v.g[DSE] + v.dx[DSE]);
            } else { // W, E, NW, NE, SW, SE, S, N
                d = v.w[DW] * v.g[DW] + v.w[DE] * v.g[DE];
                d = d + v.w[DNW] * (v.g[DNW] + v.dx[DNW]);
                d = d + v.w[DNE] * (v.g[DNE] + v.dx[DNE]);
                d = d + v.w[DSW] * (v.g[DSW] + v.dx[DSW]);
                d = d + v.w[DSE] * (v.g[DSE] + v.dx[DSE]);
                d = d + v.w[DS] * (v.g[DS] + v.dx[DS]);
                d = d + v.w[DN] * (v.g[DN] + v.dx[DN]);
            }
        } else { // the 17 edge cases: table-driven
            const signed char(*cs)[9] = ALR_L8[vertical ? 0 : 1][line_third][elem_third];
            b = 0.0f;
            d = 0.0f;
            bool more_b = true, more_d = true;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int db = cs[0][k], dd = cs[1][k];
                more_b = more_b && db != DEND;
                more_d = more_d && dd != DEND;
                if (more_b) {
                    const float w = pick8(v.w, db);
                    b = k ? b + w : w;
                }
                if (more_d) {
                    const bool on_line = vertical ? (dd == DN || dd == DS) : (dd == DW || dd == DE);
                    const float w = pick8(v.w, dd), g = pick8(v.g, dd), x = pick8(v.dx, dd);
                    const float term = on_line ? w * g : w * (g + x);
                    d = k ? d + term : term;
                }
            }
        }
        Tri t;
        t.a = hasBefore ? -(vertical ? v.w[DN] : v.w[DW]) : 0.0f;
        t.c = hasAfter ? -(vertical ? v.w[DS] : v.w[DE]) : 0.0f;
        const float MO = v.M * v.O;
        if (!is_nan(v.C)) {
            b += v.D;
            d += v.C;
            d -= MO;
        }
        t.b = b;
        t.d = d;
        return t;
    }
    // direction -> (line offset, element offset) in line terms: on a column before/after = N/S and prev/next = W/E;
    // on a row before/after = W/E and prev/next = N/S
    template <bool vertical> __device__ __forceinline__ static void where(int dir, int &dl, int &dk)
    {
        const int dj = (dir == DE || dir == DNE || dir == DSE) ? 1 : ((dir == DW || dir == DNW || dir == DSW) ? -1 : 0);
        const int di = (dir == DS || dir == DSW || dir == DSE) ? 1 : ((dir == DN || dir == DNW || dir == DNE) ? -1 : 0);
        dl = vertical ? dj : di;
        dk = vertical ? di : dj;
    }
    template <bool vertical> __device__ __forceinline__ static Tri coef(const Ctx &q, const AlrAt &at)
    {
        const size_t p = at.pos;
        In v;
        const float Uc = q.U[p];
#pragma unroll
        for (int dir = 0; dir < 8; ++dir) {
            int dl, dk;
            where<vertical>(dir, dl, dk);
            // clamped neighbour: a direction that leaves the image is in no term list of its case
            const size_t line = dl < 0 ? at.prev() : (dl > 0 ? at.next() : p);
            const long ek = dk < 0 ? (at.hasBefore ? -1 : 0) : (dk > 0 ? (at.hasAfter ? 1 : 0) : 0);
            v.w[dir] = q.w[dir][p];
            v.g[dir] = q.U[line + ek] - Uc;
            v.dx[dir] = q.X[line + ek];
        }
        v.C = q.C[p]; v.D = q.D[p]; v.M = q.M[p]; v.O = q.O[p];
        const int lt = at.hasPrev ? (at.hasNext ? 1 : 2) : 0, et = at.hasBefore ? (at.hasAfter ? 1 : 2) : 0;
        return math<vertical>(v, lt, et, at.hasBefore, at.hasAfter);
    }
    template <bool vertical> __device__ __forceinline__ static void coef4(const Ctx &q, int l, int k, int n, int nlines, Tri (&out)[4])
    {
        const AlrAt at(l, k, n, nlines);
        const size_t p = at.pos;
        float w[8][4], Uc[4], Up[4], Un[4], Xp[4], Xn[4], C[4], D[4], M[4], O[4];
#pragma unroll
        for (int dir = 0; dir < 8; ++dir) alr_ld4(q.w[dir] + p, w[dir]);
        alr_ld4(q.U + p, Uc); alr_ld4(q.U + at.prev(), Up); alr_ld4(q.U + at.next(), Un);
        alr_ld4(q.X + at.prev(), Xp); alr_ld4(q.X + at.next(), Xn);
        alr_ld4(q.C + p, C); alr_ld4(q.D + p, D); alr_ld4(q.M + p, M); alr_ld4(q.O + p, O);
        const long ob = at.hasBefore ? -1 : 0, oa = k + 4 <= n - 1 ? 4 : 3; // element k-1 / k+4, clamped (unused when clamped)
        const float Uc_b = q.U[p + ob], Uc_a = q.U[p + oa];
        const float Up_b = q.U[at.prev() + ob], Up_a = q.U[at.prev() + oa], Un_b = q.U[at.next() + ob], Un_a = q.U[at.next() + oa];
        const float Xp_b = q.X[at.prev() + ob], Xp_a = q.X[at.prev() + oa], Xn_b = q.X[at.next() + ob], Xn_a = q.X[at.next() + oa];
        const int lt = at.hasPrev ? (at.hasNext ? 1 : 2) : 0;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            // values of the three lines at elements e-1, e, e+1
            const float uc[3] = {e == 0 ? Uc_b : Uc[e - 1], Uc[e], e == 3 ? Uc_a : Uc[e + 1]};
            const float up[3] = {e == 0 ? Up_b : Up[e - 1], Up[e], e == 3 ? Up_a : Up[e + 1]};
            const float un[3] = {e == 0 ? Un_b : Un[e - 1], Un[e], e == 3 ? Un_a : Un[e + 1]};
            const float xp[3] = {e == 0 ? Xp_b : Xp[e - 1], Xp[e], e == 3 ? Xp_a : Xp[e + 1]};
            const float xn[3] = {e == 0 ? Xn_b : Xn[e - 1], Xn[e], e == 3 ? Xn_a : Xn[e + 1]};
            const bool hasBefore = k + e > 0, hasAfter = k + e < n - 1;
            In v;
#pragma unroll
            for (int dir = 0; dir < 8; ++dir) {
                int dl, dk;
                where<vertical>(dir, dl, dk);
                // a direction that leaves the line at its ends is in no term list there: read the centre element instead
                const int kk = (dk < 0 && !hasBefore) || (dk > 0 && !hasAfter) ? 1 : 1 + dk;
                v.w[dir] = w[dir][e];
                v.g[dir] = (dl < 0 ? up[kk] : (dl > 0 ? un[kk] : uc[kk])) - uc[1];
                v.dx[dir] = dl < 0 ? xp[kk] : (dl > 0 ? xn[kk] : 0.0f);
            }
            v.C = C[e]; v.D = D[e]; v.M = M[e]; v.O = O[e];
            const int et = hasBefore ? (hasAfter ? 1 : 2) : 0;
            out[e] = math<vertical>(v, lt, et, hasBefore, hasAfter);
        }
    }
};

// ---- PDE solvers (pdeSolvers.c:409-1393); planes are [nrows x ncols x F], frames independent ----------
struct AlrPde4 {
    struct Ctx {
        const float *X, *T, *B, *wW, *wN, *wE, *wS;
        __device__ void shift(size_t o)
        {
            X += o; T += o; B += o; wW += o; wN += o; wE += o; wS += o;
        }
    };
    static constexpr bool INTERIOR_LINES = false;
    static constexpr bool SOUTH_TRUEDIV = false; // TDMA_srow_ALR_4 multiplies (pdeSolvers.c:1083-1085)
    struct In { // x1, x2: the two off-line neighbours in the order they enter d: W,E on a column; S,N on a row
        float wN, wS, wE, wW, x1, x2, T, B;
    };
    template <bool vertical>
    __device__ __forceinline__ static Tri math(const In &v, bool hasN, bool hasS, bool hasW, bool hasE)
    {
        float b = 0.0f, d = 0.0f;
        bool hb = false, hd = false;
        Tri t;
        if (vertical) { // pdeSolvers.c:593
            if (hasW) acc_add(d, hd, v.wW * v.x1);
            if (hasE) acc_add(d, hd, v.wE * v.x2);
            t.a = hasN ? -v.wN : 0.0f;
            t.c = hasS ? -v.wS : 0.0f;
        } else { // :956
            if (hasS) acc_add(d, hd, v.wS * v.x1);
            if (hasN) acc_add(d, hd, v.wN * v.x2);
            t.a = hasW ? -v.wW : 0.0f;
            t.c = hasE ? -v.wE : 0.0f;
        }
        if (!is_nan(v.T)) { // :595-599
            b = v.T;
            d += v.B;
        } else { // :601-603: wN + wS + wW + wE, missing ones skipped
            if (hasN) acc_add(b, hb, v.wN);
            if (hasS) acc_add(b, hb, v.wS);
            if (hasW) acc_add(b, hb, v.wW);
            if (hasE) acc_add(b, hb, v.wE);
        }
        t.b = b;
        t.d = d;
        return t;
    }
    template <bool vertical> __device__ __forceinline__ static Tri coef(const Ctx &q, const AlrAt &at)
    {
        const size_t p = at.pos;
        const In v{q.wN[p], q.wS[p], q.wE[p], q.wW[p], q.X[vertical ? at.prev() : at.next()], q.X[vertical ? at.next() : at.prev()],
                   q.T[p], q.B[p]};
        return vertical ? math<true>(v, at.hasBefore, at.hasAfter, at.hasPrev, at.hasNext)
                        : math<false>(v, at.hasPrev, at.hasNext, at.hasBefore, at.hasAfter);
    }
    template <bool vertical> __device__ __forceinline__ static void coef4(const Ctx &q, int l, int k, int n, int nlines, Tri (&out)[4])
    {
        const AlrAt at(l, k, n, nlines);
        const size_t p = at.pos;
        float wN[4], wS[4], wE[4], wW[4], x1[4], x2[4], T[4], B[4];
        alr_ld4(q.wN + p, wN); alr_ld4(q.wS + p, wS); alr_ld4(q.wE + p, wE); alr_ld4(q.wW + p, wW);
        alr_ld4(q.X + (vertical ? at.prev() : at.next()), x1); alr_ld4(q.X + (vertical ? at.next() : at.prev()), x2);
        alr_ld4(q.T + p, T); alr_ld4(q.B + p, B);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const In v{wN[e], wS[e], wE[e], wW[e], x1[e], x2[e], T[e], B[e]};
            const bool hasBefore = k + e > 0, hasAfter = k + e < n - 1;
            out[e] = vertical ? math<true>(v, hasBefore, hasAfter, at.hasPrev, at.hasNext)
                              : math<false>(v, at.hasPrev, at.hasNext, hasBefore, hasAfter);
        }
    }
};

struct AlrPde8 {
    struct Ctx {
        const float *X, *T, *B, *wW, *wNW, *wN, *wNE, *wE, *wSE, *wS, *wSW;
        __device__ void shift(size_t o)
        {
            X += o; T += o; B += o; wW += o; wNW += o; wN += o; wNE += o; wE += o; wSE += o; wS += o; wSW += o;
        }
    };
    static constexpr bool INTERIOR_LINES = true; // interior columns, then interior rows (pdeSolvers.c:1153, :1290)
    static constexpr bool SOUTH_TRUEDIV = false; // TDMArow_ALR_8 multiplies (pdeSolvers.c:1347-1349)
    // X on the two neighbouring lines: same element (P, N) and the elements before / after it (Pb, Pa, Nb, Na).
    // On a column the previous line is the west column (P = W, Pb = NW, Pa = SW, N = E, Nb = NE, Na = SE); on a row
    // it is the north row (P = N, Pb = NW, Pa = NE, N = S, Nb = SW, Na = SE).
    struct In {
        float wW, wNW, wN, wNE, wE, wSE, wS, wSW, xP, xPb, xPa, xN, xNb, xNa, T, B;
    };
    template <bool vertical> __device__ __forceinline__ static Tri math(const In &v, bool hasBefore, bool hasAfter)
    {
        Tri t;
        float d;
        if (vertical) { // interior columns: W and E exist (:1171-1173, :1195-1197, :1227-1228)
            d = v.wW * v.xP + v.wE * v.xN;
            if (hasAfter) d += v.wSW * v.xPa + v.wSE * v.xNa;
            if (hasBefore) d += v.wNW * v.xPb + v.wNE * v.xNb;
            t.a = hasBefore ? -v.wN : 0.0f;
            t.c = hasAfter ? -v.wS : 0.0f;
        } else { // interior rows: N and S exist (:1309-1310, :1335-1337, :1364-1365)
            d = v.wS * v.xN + v.wN * v.xP;
            if (hasBefore) d += v.wSW * v.xNb + v.wNW * v.xPb;
            if (hasAfter) d += v.wSE * v.xNa + v.wNE * v.xPa;
            t.a = hasBefore ? -v.wW : 0.0f;
            t.c = hasAfter ? -v.wE : 0.0f;
        }
        if (!is_nan(v.T)) {
            t.b = v.T;
            d += v.B;
        } else { // :1181-1182 as written: wNW twice, wNE never, all eight terms at every position
            float b = v.wN + v.wS + v.wW + v.wE;
            b += v.wNW + v.wNW + v.wSW + v.wSE;
            t.b = b;
        }
        t.d = d;
        return t;
    }
    template <bool vertical> __device__ __forceinline__ static Tri coef(const Ctx &q, const AlrAt &at)
    {
        const size_t p = at.pos;
        const long ob = at.hasBefore ? -1 : 0, oa = at.hasAfter ? 1 : 0; // clamped: every load is unconditional
        const float *XP = q.X + at.prev(), *XN = q.X + at.next();
        const In v{q.wW[p], q.wNW[p], q.wN[p], q.wNE[p], q.wE[p], q.wSE[p], q.wS[p], q.wSW[p],
                   XP[0], XP[ob], XP[oa], XN[0], XN[ob], XN[oa], q.T[p], q.B[p]};
        return math<vertical>(v, at.hasBefore, at.hasAfter);
    }
    template <bool vertical> __device__ __forceinline__ static void coef4(const Ctx &q, int l, int k, int n, int nlines, Tri (&out)[4])
    {
        const AlrAt at(l, k, n, nlines);
        const size_t p = at.pos;
        float wW[4], wNW[4], wN[4], wNE[4], wE[4], wSE[4], wS[4], wSW[4], xP[4], xN[4], T[4], B[4];
        alr_ld4(q.wW + p, wW); alr_ld4(q.wNW + p, wNW); alr_ld4(q.wN + p, wN); alr_ld4(q.wNE + p, wNE);
        alr_ld4(q.wE + p, wE); alr_ld4(q.wSE + p, wSE); alr_ld4(q.wS + p, wS); alr_ld4(q.wSW + p, wSW);
        alr_ld4(q.X + at.prev(), xP); alr_ld4(q.X + at.next(), xN);
        alr_ld4(q.T + p, T); alr_ld4(q.B + p, B);
        const long ob = at.hasBefore ? -1 : 0, oa = k + 4 <= n - 1 ? 4 : 3; // element k-1 / k+4 (clamped; unused when clamped)
        const float xPbefore = q.X[at.prev() + ob], xNbefore = q.X[at.next() + ob];
        const float xPafter = q.X[at.prev() + oa], xNafter = q.X[at.next() + oa];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const In v{wW[e], wNW[e], wN[e], wNE[e], wE[e], wSE[e], wS[e], wSW[e],
                       xP[e], e == 0 ? xPbefore : xP[e - 1], e == 3 ? xPafter : xP[e + 1],
                       xN[e], e == 0 ? xNbefore : xN[e - 1], e == 3 ? xNafter : xN[e + 1], T[e], B[e]};
            out[e] = math<vertical>(v, k + e > 0, k + e < n - 1);
        }
    }
};

// element k of line l: (i,j) = (k,l) on a column, (l,k) on a row
template <class Mdl, bool VERT>
__device__ __forceinline__ Tri line_coef(const typename Mdl::Ctx &q, int l, int k, int nrows, int ncols)
{
    return Mdl::template coef<VERT>(q, AlrAt(l, k, VERT ? nrows : ncols, VERT ? ncols : nrows));
}

// ------------------------------------------------------------------------------------------------
// Zebra order: one lane per line of the active colour.  cp/dp are scratch planes with x's layout.
// Thomas recurrences and the lagged SOR blend as opticalflowSolvers.c:1890-1958.
// ------------------------------------------------------------------------------------------------
template <class Mdl, bool VERT>
__global__ void __launch_bounds__(64) k_alr_zebra(typename Mdl::Ctx q, float *x, float *__restrict__ cp,
                                                  float *__restrict__ dp, int nrows, int ncols, size_t frame_stride,
                                                  int lo, int hi, int colour, float omega)
{
    constexpr bool vertical = VERT;
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int l = lo + (((lo & 1) != colour) ? 1 : 0) + 2 * t;
    if (l > hi) return;
    const size_t fo = (size_t)blockIdx.y * frame_stride;
    q.shift(fo);
    x += fo;
    cp += fo;
    dp += fo;
    const int n = vertical ? nrows : ncols; // line length; line l starts at l * n in either layout
    constexpr size_t stride = 1;
    const size_t base = (size_t)l * n;
    const float om1 = 1.0f - omega;
    const bool tdiv = Mdl::SOUTH_TRUEDIV && !VERT && l == nrows - 1; // the south row of these models divides (see the model)

    // The recurrences are serial, the loads are not: fetch the coefficients of ZCH steps at once so that
    // one memory latency is paid per chunk instead of per step.
    constexpr int ZCH = 8;
    Tri c0 = line_coef<Mdl, VERT>(q, l, 0, nrows, ncols);
    float cpv = c0.c / c0.b;
    float dpv = c0.d / c0.b;
    cp[base] = cpv;
    dp[base] = dpv;
    for (int k0 = 1; k0 <= n - 2; k0 += ZCH) {
        Tri c[ZCH];
#pragma unroll
        for (int u = 0; u < ZCH; ++u) {
            const int k = min(k0 + u, n - 2);
            c[u] = line_coef<Mdl, VERT>(q, l, k, nrows, ncols);
        }
#pragma unroll
        for (int u = 0; u < ZCH; ++u) {
            const int k = k0 + u;
            if (k <= n - 2) {
                const float den = c[u].b - cpv * c[u].a;
                if (tdiv) {
                    cpv = c[u].c / den;
                    dpv = (c[u].d - dpv * c[u].a) / den;
                } else {
                    const float div = 1.0f / den;
                    cpv = c[u].c * div;
                    dpv = (c[u].d - dpv * c[u].a) * div;
                }
                cp[base + k * stride] = cpv;
                dp[base + k * stride] = dpv;
            }
        }
    }
    {
        const int k = n - 1;
        const Tri c = line_coef<Mdl, VERT>(q, l, k, nrows, ncols);
        dpv = (c.d - dpv * c.a) / (c.b - cpv * c.a);
    }
    // back-substitution; element k+1 gets its blend once it has been used
    float xs = dpv;
    float old = x[base + (size_t)(n - 1) * stride];
    for (int k0 = n - 2; k0 >= 0; k0 -= ZCH) {
        float dpk[ZCH], cpk[ZCH], xo[ZCH];
#pragma unroll
        for (int u = 0; u < ZCH; ++u) {
            const size_t pos = base + (size_t)max(k0 - u, 0) * stride;
            dpk[u] = dp[pos];
            cpk[u] = cp[pos];
            xo[u] = x[pos];
        }
#pragma unroll
        for (int u = 0; u < ZCH; ++u) {
            const int k = k0 - u;
            if (k >= 0) {
                const size_t pos = base + (size_t)k * stride;
                const float xk = dpk[u] - cpk[u] * xs;
                x[pos + stride] = omega * xs + om1 * old;
                old = xo[u];
                xs = xk;
            }
        }
    }
    x[base] = omega * xs + om1 * old;
}

// ------------------------------------------------------------------------------------------------
// Zebra order on a small frame: the whole call in ONE launch, one workgroup per frame.
//
// On the coarse scales of the drivers' pyramids a zebra call is ~45 launches (coefficient transposes, factor passes, per
// iteration four colour passes per field and two iterate transposes) of ~10 us each for a few microseconds of work: 0.4-0.5 ms
// per call whatever the frame size below ~135 x 240, and these scales take most of the calls (240 of 324 in the 4K multigrid
// run).  Here one workgroup of 1024 threads runs the same sequence with a workgroup barrier where the launches were:
//   * the coefficient planes are transposed into their twins once, the iterate planes around every row pass (as run_alr does);
//   * the coefficient-only half of the Thomas recurrences (cp and the divisors) is run once per call into factor planes,
//     as k_alr_zebra3<ZB_FACTOR> does;
//   * a colour pass has three stages: every thread builds tridiagonal rows of the colour's lines -- the same Model::coef as
//     everywhere -- into LDS next to the line's factors and old x, in parallel over lines AND elements; one lane per line
//     runs the two short recurrences out of LDS (three dependent instructions per element going down, two coming back);
//     every thread copies the blended lines back to the plane.
// Bit-identical to k_alr_zebra3 / k_alr_zebra.  Frames whose largest colour pass fits 150 KB of LDS.
// ------------------------------------------------------------------------------------------------
constexpr int ALR_SMALL_THREADS = 1024;
constexpr int ALR_SMALL_MAXTR = 24;
template <class Mdl> struct AlrSmallArgs {
    typename Mdl::Ctx q[2], qt[2];
    float *x[2], *xt[2];
    float *cp[2][2], *dv[2][2]; // per-call factor planes [field][column pass 0 / row pass 1]: k_alr_zebra3<ZB_FACTOR>'s contents
    const float *tin[ALR_SMALL_MAXTR];
    float *tout[ALR_SMALL_MAXTR];
    int ntr, nch, nrows, ncols, iter;
    float omega;
    size_t fs;
};

// floats the row part of the LDS image takes (the old / new x part follows it)
__host__ __device__ inline size_t alr_small_lds_bytes_dev(int nrows, int ncols, bool interior_lines)
{
    const int lo = interior_lines ? 1 : 0;
    const size_t col_lines = (size_t)(ncols - 2 * lo + 1) / 2, row_lines = (size_t)(nrows - 2 * lo + 1) / 2;
    const size_t a = col_lines * (size_t)(nrows | 1), b = row_lines * (size_t)(ncols | 1);
    return (a > b ? a : b) * 4;
}
// LDS per colour pass: a row (a, divisor, cp, d) and the old / new x of every element of the colour's lines; lines padded to an
// odd number of rows so that the lanes of the recurrence stage (one per line) hit different banks
__host__ __device__ inline int alr_small_stride(int n) { return n | 1; }
inline size_t alr_small_lds_bytes(int nrows, int ncols, bool interior_lines)
{
    const int lo = interior_lines ? 1 : 0;
    const size_t col_lines = (size_t)(ncols - 2 * lo + 1) / 2, row_lines = (size_t)(nrows - 2 * lo + 1) / 2;
    const size_t a = col_lines * alr_small_stride(nrows), b = row_lines * alr_small_stride(ncols);
    return (a > b ? a : b) * (sizeof(float4) + sizeof(float));
}

// FACTOR = true: once per call and direction, the coefficient-only recurrence of every line of one colour:
//   cp[0] = c/b, divisor[0] = b;  middle: den = b - cp' a, cp = c * (1/den), divisor = 1/den  (a dividing south row: cp = c/den,
//   divisor = den);  last: divisor = b - cp' a.                                     -> the cp / dv planes
// FACTOR = false: one relaxation of the lines of one colour with those planes:
//   dp[0] = d / divisor[0];  middle: dp = (d - dp' a) * divisor (dividing south row: / divisor);  last: dp = (d - dp' a) / divisor;
//   back-substitution x[k] = dp[k] - cp[k] x[k+1] with the lagged SOR blend (opticalflowSolvers.c:1890-1958).
// The same statements as k_alr_zebra with 1/den hoisted out of the iteration, as k_alr_zebra3 does.
template <class Mdl, bool VERT, bool FACTOR>
__device__ __forceinline__ void alr_small_pass(const typename Mdl::Ctx &q, float *x, float *cpP, float *dvP, float4 *T, float *X, int nrows,
                                               int ncols, int colour, float omega)
{
    const int tid = threadIdx.x;
    const int lo = Mdl::INTERIOR_LINES ? 1 : 0, hi = (VERT ? ncols : nrows) - 1 - lo;
    const int first = lo + (((lo & 1) != colour) ? 1 : 0);
    if (first > hi) return; // uniform
    const int count = (hi - first) / 2 + 1, n = VERT ? nrows : ncols, S = alr_small_stride(n);
    for (int e = tid; e < count * n; e += ALR_SMALL_THREADS) { // stage 1, all threads: the rows of every line of this colour
        const int li = e / n, k = e - li * n;
        const int l = first + 2 * li;
        const Tri c = line_coef<Mdl, VERT>(q, l, k, nrows, ncols);
        const size_t pos = (size_t)l * n + k;
        if (FACTOR) {
            T[li * S + k] = make_float4(c.a, c.b, c.c, 0.0f);
        } else {
            T[li * S + k] = make_float4(c.a, dvP[pos], cpP[pos], c.d);
            X[li * S + k] = x[pos];
        }
    }
    __syncthreads();
    if (tid < count) { // stage 2, one lane per line: the recurrences, out of LDS
        const int l = first + 2 * tid;
        float4 *L = T + (size_t)tid * S;
        const bool tdiv = Mdl::SOUTH_TRUEDIV && !VERT && l == nrows - 1;
        if (FACTOR) {
            const size_t base = (size_t)l * n;
            float4 c = L[0];
            float cpv = c.z / c.y;
            cpP[base] = cpv;
            dvP[base] = c.y;
            for (int k = 1; k <= n - 2; k++) {
                c = L[k];
                const float den = c.y - cpv * c.x;
                float dvv;
                if (tdiv) {
                    cpv = c.z / den;
                    dvv = den;
                } else {
                    dvv = 1.0f / den;
                    cpv = c.z * dvv;
                }
                cpP[base + k] = cpv;
                dvP[base + k] = dvv;
            }
            c = L[n - 1];
            cpP[base + n - 1] = 0.0f;
            dvP[base + n - 1] = c.y - cpv * c.x;
        } else {
            float *Xl = X + (size_t)tid * S;
            const float om1 = 1.0f - omega;
            constexpr int CH = 8; // rows per trip: the LDS loads of a trip are issued together, ahead of the dependent chain
            float4 c = L[0];
            float dpv = c.w / c.y;
            L[0].w = dpv;
            int k = 1;
            for (; k + CH - 1 <= n - 2; k += CH) {
                float4 r[CH];
                float dpo[CH];
#pragma unroll
                for (int u = 0; u < CH; u++) r[u] = L[k + u];
#pragma unroll
                for (int u = 0; u < CH; u++) {
                    dpv = tdiv ? (r[u].w - dpv * r[u].x) / r[u].y : (r[u].w - dpv * r[u].x) * r[u].y;
                    dpo[u] = dpv;
                }
#pragma unroll
                for (int u = 0; u < CH; u++) L[k + u].w = dpo[u];
            }
            for (; k <= n - 2; k++) {
                c = L[k];
                dpv = tdiv ? (c.w - dpv * c.x) / c.y : (c.w - dpv * c.x) * c.y;
                L[k].w = dpv;
            }
            c = L[n - 1];
            dpv = (c.w - dpv * c.x) / c.y;
            // back-substitution; element k+1 gets its blend once it has been used
            float xs = dpv, old = Xl[n - 1];
            k = n - 2;
            for (; k - (CH - 1) >= 0; k -= CH) {
                float dpk[CH], cpk[CH], xo[CH], xn[CH];
#pragma unroll
                for (int u = 0; u < CH; u++) {
                    const float4 t = L[k - u];
                    dpk[u] = t.w;
                    cpk[u] = t.z;
                    xo[u] = Xl[k - u];
                }
#pragma unroll
                for (int u = 0; u < CH; u++) {
                    const float xk = dpk[u] - cpk[u] * xs;
                    xn[u] = omega * xs + om1 * old; // the blended x of element k - u + 1
                    old = xo[u];
                    xs = xk;
                }
#pragma unroll
                for (int u = 0; u < CH; u++) Xl[k - u + 1] = xn[u];
            }
            for (; k >= 0; k--) {
                const float xk = L[k].w - L[k].z * xs;
                const float xo = Xl[k];
                Xl[k + 1] = omega * xs + om1 * old;
                old = xo;
                xs = xk;
            }
            Xl[0] = omega * xs + om1 * old;
        }
    }
    __syncthreads();
    if (!FACTOR) {
        for (int e = tid; e < count * n; e += ALR_SMALL_THREADS) { // stage 3, all threads: the relaxed lines back to the plane
            const int li = e / n, k = e - li * n;
            x[(size_t)(first + 2 * li) * n + k] = X[li * S + k];
        }
        __syncthreads();
    }
}

template <class Mdl>
__global__ void __launch_bounds__(ALR_SMALL_THREADS) k_alr_small(AlrSmallArgs<Mdl> A)
{
    extern __shared__ __attribute__((aligned(16))) float4 alr_small_lds[];
    const int tid = threadIdx.x, nrows = A.nrows, ncols = A.ncols, n = nrows * ncols;
    const size_t fo = (size_t)blockIdx.x * A.fs;
    float4 *const T = alr_small_lds;
    float *const X = reinterpret_cast<float *>(alr_small_lds) + alr_small_lds_bytes_dev(nrows, ncols, Mdl::INTERIOR_LINES);
    // two fields at most; named copies rather than arrays indexed at run time (those would live in scratch memory)
    typename Mdl::Ctx q0 = A.q[0], q1 = A.q[1], qt0 = A.qt[0], qt1 = A.qt[1];
    q0.shift(fo);
    q1.shift(fo);
    qt0.shift(fo);
    qt1.shift(fo);
    const bool two = A.nch > 1;
    float *const x0 = A.x[0] + fo, *const xt0 = A.xt[0] + fo;
    float *const x1 = two ? A.x[1] + fo : nullptr, *const xt1 = two ? A.xt[1] + fo : nullptr;
    float *const cp00 = A.cp[0][0] + fo, *const dv00 = A.dv[0][0] + fo, *const cp01 = A.cp[0][1] + fo, *const dv01 = A.dv[0][1] + fo;
    float *const cp10 = two ? A.cp[1][0] + fo : nullptr, *const dv10 = two ? A.dv[1][0] + fo : nullptr;
    float *const cp11 = two ? A.cp[1][1] + fo : nullptr, *const dv11 = two ? A.dv[1][1] + fo : nullptr;
    // out(i, j) at i * ncols + j  <-  in(i, j) at j * nrows + i
    auto to_rows = [&](float *out, const float *in) {
        for (int idx = tid; idx < n; idx += ALR_SMALL_THREADS) {
            const int j = idx / nrows, i = idx - j * nrows;
            out[(size_t)i * ncols + j] = in[idx];
        }
    };
    auto to_cols = [&](float *out, const float *in) {
        for (int idx = tid; idx < n; idx += ALR_SMALL_THREADS) {
            const int j = idx / nrows, i = idx - j * nrows;
            out[idx] = in[(size_t)i * ncols + j];
        }
    };
    for (int k = 0; k < A.ntr; k++) to_rows(A.tout[k] + fo, A.tin[k] + fo);
    __syncthreads();
    for (int colour = 0; colour < 2; colour++) { // the factor planes: coefficients only, once per call
        alr_small_pass<Mdl, true, true>(q0, nullptr, cp00, dv00, T, X, nrows, ncols, colour, A.omega);
        alr_small_pass<Mdl, false, true>(qt0, nullptr, cp01, dv01, T, X, nrows, ncols, colour, A.omega);
        if (two) {
            alr_small_pass<Mdl, true, true>(q1, nullptr, cp10, dv10, T, X, nrows, ncols, colour, A.omega);
            alr_small_pass<Mdl, false, true>(qt1, nullptr, cp11, dv11, T, X, nrows, ncols, colour, A.omega);
        }
    }
    for (int it = 0; it < A.iter; it++) {
        // columns of field 0 then field 1, rows of field 1 then field 0 (opticalflowSolvers.c:231-258)
        alr_small_pass<Mdl, true, false>(q0, x0, cp00, dv00, T, X, nrows, ncols, 0, A.omega);
        alr_small_pass<Mdl, true, false>(q0, x0, cp00, dv00, T, X, nrows, ncols, 1, A.omega);
        if (two) {
            alr_small_pass<Mdl, true, false>(q1, x1, cp10, dv10, T, X, nrows, ncols, 0, A.omega);
            alr_small_pass<Mdl, true, false>(q1, x1, cp10, dv10, T, X, nrows, ncols, 1, A.omega);
        }
        to_rows(xt0, x0);
        if (two) to_rows(xt1, x1);
        __syncthreads();
        if (two) {
            alr_small_pass<Mdl, false, false>(qt1, xt1, cp11, dv11, T, X, nrows, ncols, 0, A.omega);
            alr_small_pass<Mdl, false, false>(qt1, xt1, cp11, dv11, T, X, nrows, ncols, 1, A.omega);
        }
        alr_small_pass<Mdl, false, false>(qt0, xt0, cp01, dv01, T, X, nrows, ncols, 0, A.omega);
        alr_small_pass<Mdl, false, false>(qt0, xt0, cp01, dv01, T, X, nrows, ncols, 1, A.omega);
        to_cols(x0, xt0);
        if (two) to_cols(x1, xt1);
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------
// Zebra order, 4-neighbour models: one workgroup = 16 lines of the active colour, one SOLVER wave
// (lane = line, runs the recurrences) fed by seven MOVER waves.
//
// A lane that walks its own line makes every load instruction touch 64 cache lines for 256 useful
// bytes, and one wave can keep only ~63 such loads in flight: k_alr_zebra is bound by that, not by the
// recurrences.  Here the movers fetch the operands of [16 lines x 32 elements] tiles with 16-byte loads
// (eight lanes cover 128 contiguous bytes of a line), turn them into (a,b,c,d) rows with the same
// Model::math as everywhere else and park them in LDS; the solver reads its line's rows back and leaves
// cp,dp in their place, which the movers write out with 16-byte stores during the next round.
// Back-substitution runs the same pipeline downwards on (cp, dp, old x) and writes the blended x.
// Rounds of 7 tiles are double-buffered; one barrier per round is the only synchronisation.  Only 16
// lines per workgroup because the frame has only ~1000-2000 lines per colour: the recurrences cost the
// same however many lanes run them, but 120 workgroups pull operands through 120 CUs' memory paths
// instead of 30.  Arithmetic and operand order are those of k_alr_zebra (bit-identical).
// ------------------------------------------------------------------------------------------------
constexpr int ZB_NM = 7;                        // mover waves = tiles per round
constexpr int ZB_LW = 8;                        // lines per workgroup
constexpr int ZB_TE = 32;                       // elements per tile
constexpr int ZB_GP = ZB_TE / 4;                // 4-element groups per line of a tile
constexpr int ZB_LP = 64 / ZB_GP;               // lines one mover pass covers
constexpr int ZB_NP = ZB_LW / ZB_LP;            // mover passes per tile
constexpr int ZB_THREADS = 64 * (1 + ZB_NM);
static_assert(ZB_NP * ZB_LP == ZB_LW, "zebra tile geometry");

__device__ __forceinline__ void alr_st4(float *p, float a, float b, float c, float d)
{
    alr_v4 t;
    t.x = a; t.y = b; t.z = c; t.w = d;
    *reinterpret_cast<alr_v4 *>(p) = t;
}

// FACTOR: forward recurrence of the coefficient-only part, once per call: cp -> cp plane; the divisor
//         1/(b - cp' a) -> dv plane (b itself for the first element, the bare denominator for the last:
//         those two are divided by, as in the reference).  No back-substitution.
// APPLY:  one relaxation of the lines first, first+lstep, ... <= lastc with those planes: the forward
//         recurrence is down to dp = (d - dp' a) * dv -- three dependent instructions instead of a division.
enum { ZB_FACTOR = 1, ZB_APPLY = 2 };

// ------------------------------------------------------------------------------------------------
// k_alr_zebra3: k_alr_zebra2 (both modes) with the solver wave's instruction count cut to the bone.
//
// With the movers disabled-in-reverse (solver switched off) a 4K colour pass takes 70-100 us; with the
// solver on, 150-270 us: the one wave that runs the recurrences is bound by instruction ISSUE (a wave gets
// one issue slot every ~4 cycles), and k_alr_zebra2 spends ~9 instructions per element there: element-wise
// (a,div,cp,d) rows read one dword pair at a time, per-sub-block scalar bookkeeping, the SOR blend.  Here
//   * a tile keeps each line as three rows of 32 floats (a | divisor | d->dp going down, old x | cp | dp->x
//     coming back), so the solver moves four elements per ds_read_b128 / ds_write_b128;
//   * the SOR blend moves to the movers (they hold old x and get the raw x from LDS);
//   * rounds that contain neither end of the line run a branch-free body, one tile (32 elements) per
//     loop trip;
// which leaves 4 instructions per element going down (3 of them the dependent mul-sub-mul) and 2.75 coming
// back.  Arithmetic, operand order and results are those of k_alr_zebra2 (bit-identical).
// ------------------------------------------------------------------------------------------------
constexpr int Z3_LSF = 3 * ZB_TE + 4;              // floats per line of a tile; +4 keeps the 16 lines' b128 reads on distinct banks
constexpr int Z3_TILE = ZB_LW * Z3_LSF;            // floats per tile
constexpr size_t Z3_LDS_BYTES = (size_t)2 * ZB_NM * Z3_TILE * sizeof(float);

template <class Mdl, bool VERT, int MODE>
__device__ __forceinline__ void alr_zebra3_body(typename Mdl::Ctx q, float *x, float *__restrict__ cp, float *__restrict__ dv,
                                                float *__restrict__ dp, int nrows, int ncols, size_t frame_stride, int first, int lastc,
                                                int lstep, float omega)
{
    extern __shared__ float z3_lds[]; // [2][ZB_NM][Z3_TILE]
    const size_t fo = (size_t)blockIdx.y * frame_stride;
    q.shift(fo);
    x += fo;
    cp += fo;
    dv += fo;
    dp += fo;
    const int n = VERT ? nrows : ncols, nlines = VERT ? ncols : nrows;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int line0 = first + lstep * ZB_LW * (int)blockIdx.x;
    const float om1 = 1.0f - omega;
    constexpr int RE = ZB_NM * ZB_TE, TE = ZB_TE;
    const int nrounds = (n + RE - 1) / RE;
    const int mslot = wave - 1;
    const int mg = lane % ZB_GP, ml = lane / ZB_GP;
    auto tile_of = [&](int buf, int slot) __attribute__((always_inline)) { return z3_lds + ((size_t)buf * ZB_NM + slot) * Z3_TILE; };
    auto lds_st4 = [](float *p, float a, float b, float c, float d) __attribute__((always_inline)) { *reinterpret_cast<float4 *>(p) = make_float4(a, b, c, d); };
    auto lds_ld4 = [](const float *p, float (&v)[4]) __attribute__((always_inline)) {
        const float4 t = *reinterpret_cast<const float4 *>(p);
        v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
    };

    // ---- movers ---------------------------------------------------------------------------------
    auto produce = [&](int r, int buf) __attribute__((always_inline)) { // operands of round r, slot mslot -> rows a | divisor | d  (FACTOR: a | b | c)
        const int k0 = (r * ZB_NM + mslot) * TE;
        if (k0 >= n) return;
        float *T = tile_of(buf, mslot);
        if (k0 + TE - 1 <= n - 1) {
            Tri t[ZB_NP][4];
            float fd[ZB_NP][4];
#pragma unroll
            for (int rr = 0; rr < ZB_NP; ++rr) {
                const int l = min(line0 + lstep * (ml + ZB_LP * rr), lastc);
                Mdl::template coef4<VERT>(q, l, k0 + 4 * mg, n, nlines, t[rr]);
                if (MODE == ZB_APPLY) alr_ld4(dv + (size_t)l * n + k0 + 4 * mg, fd[rr]);
            }
#pragma unroll
            for (int rr = 0; rr < ZB_NP; ++rr) {
                float *row = T + (ml + ZB_LP * rr) * Z3_LSF + 4 * mg;
                lds_st4(row, t[rr][0].a, t[rr][1].a, t[rr][2].a, t[rr][3].a);
                if (MODE == ZB_APPLY) {
                    lds_st4(row + TE, fd[rr][0], fd[rr][1], fd[rr][2], fd[rr][3]);
                    lds_st4(row + 2 * TE, t[rr][0].d, t[rr][1].d, t[rr][2].d, t[rr][3].d);
                } else {
                    lds_st4(row + TE, t[rr][0].b, t[rr][1].b, t[rr][2].b, t[rr][3].b);
                    lds_st4(row + 2 * TE, t[rr][0].c, t[rr][1].c, t[rr][2].c, t[rr][3].c);
                }
            }
            return;
        }
#pragma unroll
        for (int rr = 0; rr < ZB_NP; ++rr) {
            const int L = ml + ZB_LP * rr, k = k0 + 4 * mg;
            const int l = min(line0 + lstep * L, lastc);
            float *row = T + L * Z3_LSF + 4 * mg;
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (k + e <= n - 1) {
                    const Tri t = Mdl::template coef<VERT>(q, AlrAt(l, k + e, n, nlines));
                    row[e] = t.a;
                    row[TE + e] = MODE == ZB_APPLY ? dv[(size_t)l * n + k + e] : t.b;
                    row[2 * TE + e] = MODE == ZB_APPLY ? t.d : t.c;
                }
        }
    };
    auto load_bwd = [&](int r, int buf) __attribute__((always_inline)) { // rows old x | cp | dp of round r
        const int k0 = (r * ZB_NM + mslot) * TE;
        if (k0 >= n) return;
        float *T = tile_of(buf, mslot);
        if (k0 + TE - 1 <= n - 1) {
            float c[ZB_NP][4], d[ZB_NP][4], o[ZB_NP][4];
#pragma unroll
            for (int rr = 0; rr < ZB_NP; ++rr) {
                const size_t pos = (size_t)min(line0 + lstep * (ml + ZB_LP * rr), lastc) * n + k0 + 4 * mg;
                alr_ld4(cp + pos, c[rr]); alr_ld4(dp + pos, d[rr]); alr_ld4(x + pos, o[rr]);
            }
#pragma unroll
            for (int rr = 0; rr < ZB_NP; ++rr) {
                float *row = T + (ml + ZB_LP * rr) * Z3_LSF + 4 * mg;
                lds_st4(row, o[rr][0], o[rr][1], o[rr][2], o[rr][3]);
                lds_st4(row + TE, c[rr][0], c[rr][1], c[rr][2], c[rr][3]);
                lds_st4(row + 2 * TE, d[rr][0], d[rr][1], d[rr][2], d[rr][3]);
            }
            return;
        }
#pragma unroll
        for (int rr = 0; rr < ZB_NP; ++rr) {
            const int L = ml + ZB_LP * rr, k = k0 + 4 * mg;
            const size_t pos = (size_t)min(line0 + lstep * L, lastc) * n + min(k, n - 1);
            float *row = T + L * Z3_LSF + 4 * mg;
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (k + e <= n - 1) {
                    row[e] = x[pos + e];
                    row[TE + e] = cp[pos + e];
                    row[2 * TE + e] = dp[pos + e];
                }
        }
    };
    // results of a finished tile: into registers first (the tile is about to be refilled), to global after the
    // next tile's loads have been consumed.  Going down: dp.  Coming back: the blended x (:1951-1958).
    float res[ZB_NP][4], res2[ZB_NP][4]; // FACTOR: cp and the divisor
    auto grab = [&](int buf, bool fwd) __attribute__((always_inline)) {
        const float *T = tile_of(buf, mslot);
#pragma unroll
        for (int rr = 0; rr < ZB_NP; ++rr) {
            const float *row = T + (ml + ZB_LP * rr) * Z3_LSF + 4 * mg;
            lds_ld4(row + 2 * TE, res[rr]);
            if (MODE == ZB_FACTOR) lds_ld4(row + TE, res2[rr]);
            if (!fwd) {
                float old[4];
                lds_ld4(row, old);
#pragma unroll
                for (int e = 0; e < 4; ++e) res[rr][e] = omega * res[rr][e] + om1 * old[e];
            }
        }
    };
    auto put = [&](int r, bool fwd) __attribute__((always_inline)) {
        const int k0 = (r * ZB_NM + mslot) * TE;
        float *out = MODE == ZB_FACTOR ? cp : (fwd ? dp : x);
#pragma unroll
        for (int rr = 0; rr < ZB_NP; ++rr) {
            const int k = k0 + 4 * mg;
            const int l = line0 + lstep * (ml + ZB_LP * rr);
            const size_t pos = (size_t)min(l, lastc) * n + min(k, n - 1);
            if (l <= lastc && k + 3 <= n - 1) {
                alr_st4(out + pos, res[rr][0], res[rr][1], res[rr][2], res[rr][3]);
                if (MODE == ZB_FACTOR) alr_st4(dv + pos, res2[rr][0], res2[rr][1], res2[rr][2], res2[rr][3]);
            } else if (l <= lastc) {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (k + e <= n - 1) {
                        out[pos + e] = res[rr][e];
                        if (MODE == ZB_FACTOR) dv[pos + e] = res2[rr][e];
                    }
            }
        }
    };
    auto lds_barrier = [&]() __attribute__((always_inline)) { __asm__ volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };

    // ---- solver ---------------------------------------------------------------------------------
    const bool solver = wave == 0 && lane < ZB_LW;
    // The south row of the late-linearisation models divides where every other line multiplies by a reciprocal
    // (Mdl::SOUTH_TRUEDIV).  It is one lane of one workgroup of the row pass: that workgroup (wave-uniform `wg_south`) runs
    // the SOUTH instantiation of the forward bodies, whose middle elements select per lane; its divisor plane holds the bare
    // denominator for that line.  Everybody else runs the unchanged fast bodies.
    constexpr bool CAN_SOUTH = Mdl::SOUTH_TRUEDIV && !VERT;
    bool tdiv = false, wg_south = false;
    if constexpr (CAN_SOUTH) {
        const int lsolve = line0 + lstep * lane;
        tdiv = lane < ZB_LW && lsolve <= lastc && lsolve == nlines - 1;
        wg_south = __builtin_amdgcn_ballot_w64(tdiv) != 0;
    }
    auto ld8 = [&](const float *p, float (&v)[8]) __attribute__((always_inline)) {
        const float4 lo = *reinterpret_cast<const float4 *>(p), hi = *reinterpret_cast<const float4 *>(p + 4);
        v[0] = lo.x; v[1] = lo.y; v[2] = lo.z; v[3] = lo.w; v[4] = hi.x; v[5] = hi.y; v[6] = hi.z; v[7] = hi.w;
    };
    auto st8 = [&](float *p, const float (&v)[8]) __attribute__((always_inline)) {
        *reinterpret_cast<float4 *>(p) = make_float4(v[0], v[1], v[2], v[3]);
        *reinterpret_cast<float4 *>(p + 4) = make_float4(v[4], v[5], v[6], v[7]);
    };

    // going down (opticalflowSolvers.c:1890-1950): dp_k = (d_k - dp_{k-1} a_k) * divisor_k; the first element is divided
    // by b, the last by its bare denominator.  Tiles that hold neither end of the line take the branch-free body.
    float dpv = 0.0f;
    auto fwd_edge = [&](auto south, float *P, int kr, int s0, int s1) __attribute__((always_inline)) {
        constexpr bool S = decltype(south)::value;
#pragma unroll 1
        for (int j = s0 * (TE / 8); j < s1 * (TE / 8); ++j) {
            const int k0 = kr + 8 * j;
            if (k0 > n - 1) break;
            float *Q = P + (j / (TE / 8)) * Z3_TILE + (j % (TE / 8)) * 8;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int k = k0 + e; // wave-uniform
                if (k <= n - 1) {
                    const float a = Q[e], v = Q[TE + e], d = Q[2 * TE + e];
                    if (MODE == ZB_APPLY) {
                        if (k == 0) dpv = d / v;
                        else if (k == n - 1) dpv = (d - dpv * a) / v;
                        else if (S && tdiv) dpv = (d - dpv * a) / v; // v = the bare denominator on this line
                        else dpv = (d - dpv * a) * v;
                    } else { // rows a | b | c; dpv carries cp.  The first element is divided by b, the last by its bare denominator
                        float dvv;
                        if (k == 0) {
                            dvv = v;
                            dpv = d / v;
                        } else if (k == n - 1) {
                            dvv = v - dpv * a;
                            dpv = 0.0f; // cp = 0 closes the back-substitution
                        } else if (S && tdiv) {
                            dvv = v - dpv * a;
                            dpv = d / dvv;
                        } else {
                            dvv = 1.0f / (v - dpv * a);
                            dpv = d * dvv;
                        }
                        Q[TE + e] = dvv;
                    }
                    Q[2 * TE + e] = dpv;
                }
            }
        }
    };
    // one element of the branch-free body: (a, v, d) -> v (FACTOR: the divisor to store), d (dp / cp), dpv
    auto fwd_step = [&](auto south, float a, float &v, float &d) __attribute__((always_inline)) {
        constexpr bool S = decltype(south)::value;
        if (MODE == ZB_APPLY) {
            const float t = d - dpv * a;
            if constexpr (S) dpv = tdiv ? t / v : t * v;
            else dpv = t * v;
        } else {
            const float den = v - dpv * a;
            const float r = 1.0f / den;
            if constexpr (S) {
                const float cq = d / den;
                v = tdiv ? den : r;
                dpv = tdiv ? cq : d * r;
            } else {
                v = r;
                dpv = d * r;
            }
        }
        d = dpv;
    };
    auto fwd_fast = [&](auto south, float *P, int s0, int s1) __attribute__((always_inline)) {
        if (s0 >= s1) return;
        float a0[8], v0[8], d0[8], a1[8], v1[8], d1[8];
        ld8(P + s0 * Z3_TILE, a0); ld8(P + s0 * Z3_TILE + TE, v0); ld8(P + s0 * Z3_TILE + 2 * TE, d0);
#pragma unroll 1
        for (int s = s0; s < s1; ++s) {
            float *Q = P + s * Z3_TILE;
            const float *N = P + min(s + 1, s1 - 1) * Z3_TILE;
#pragma unroll
            for (int h = 0; h < TE / 16; ++h) {
                ld8(Q + 16 * h + 8, a1); ld8(Q + 16 * h + 8 + TE, v1); ld8(Q + 16 * h + 8 + 2 * TE, d1);
#pragma unroll
                for (int e = 0; e < 8; ++e) fwd_step(south, a0[e], v0[e], d0[e]);
                st8(Q + 16 * h + 2 * TE, d0);
                if (MODE == ZB_FACTOR) st8(Q + 16 * h + TE, v0);
                const float *F = h == TE / 16 - 1 ? N : Q + 16 * (h + 1);
                ld8(F, a0); ld8(F + TE, v0); ld8(F + 2 * TE, d0);
#pragma unroll
                for (int e = 0; e < 8; ++e) fwd_step(south, a1[e], v1[e], d1[e]);
                st8(Q + 16 * h + 8 + 2 * TE, d1);
                if (MODE == ZB_FACTOR) st8(Q + 16 * h + 8 + TE, v1);
            }
        }
    };
    if (wave > 0) produce(0, 0);
    lds_barrier();
    for (int r = 0; r < nrounds; ++r) {
        if (solver) {
            float *P = tile_of(r & 1, 0) + lane * Z3_LSF;
            const int kr = __builtin_amdgcn_readfirstlane(r * RE);
            const int s0 = kr == 0 ? 1 : 0;                                   // tile 0 of round 0 holds the first element
            const int s1 = max(s0, min(ZB_NM, (n - 1 - kr) / TE));            // tiles [s0, s1) end at or before element n-2
            if (CAN_SOUTH && wg_south) {
                fwd_edge(std::true_type{}, P, kr, 0, s0);
                fwd_fast(std::true_type{}, P, s0, s1);
                fwd_edge(std::true_type{}, P, kr, s1, ZB_NM);
            } else {
                fwd_edge(std::false_type{}, P, kr, 0, s0);
                fwd_fast(std::false_type{}, P, s0, s1);
                fwd_edge(std::false_type{}, P, kr, s1, ZB_NM);
            }
        } else if (wave > 0) {
            if (r >= 1) grab((r - 1) & 1, true);
            if (r + 1 < nrounds) produce(r + 1, (r + 1) & 1);
            if (r >= 1) put(r - 1, true);
        }
        lds_barrier();
    }
    if (wave > 0) {
        grab((nrounds - 1) & 1, true);
        put(nrounds - 1, true);
    }
    if (MODE == ZB_FACTOR) return;
    if (wave > 0) {
        __threadfence_block(); // every dp this thread reloads below was stored by this thread: drain them once
        load_bwd(nrounds - 1, (nrounds - 1) & 1);
    }
    lds_barrier();

    // coming back: x_k = dp_k - cp_k x_{k+1} (cp of the last element is 0); the blend is the movers' (grab)
    float xs = 0.0f;
    auto bwd_edge = [&](float *P, int kr, int s0, int s1) __attribute__((always_inline)) { // tiles s1-1 down to s0
#pragma unroll 1
        for (int j = s1 * (TE / 8) - 1; j >= s0 * (TE / 8); --j) {
            const int k0 = kr + 8 * j;
            if (k0 > n - 1) continue;
            float *Q = P + (j / (TE / 8)) * Z3_TILE + (j % (TE / 8)) * 8;
#pragma unroll
            for (int e = 7; e >= 0; --e)
                if (k0 + e <= n - 1) { // wave-uniform
                    xs = Q[2 * TE + e] - Q[TE + e] * xs;
                    Q[2 * TE + e] = xs;
                }
        }
    };
    auto bwd_fast = [&](float *P, int s1) __attribute__((always_inline)) { // tiles s1-1 down to 0, all inside the line
        if (s1 <= 0) return;
        float c0[8], d0[8], c1[8], d1[8];
        ld8(P + (s1 - 1) * Z3_TILE + TE - 8 + TE, c0); ld8(P + (s1 - 1) * Z3_TILE + TE - 8 + 2 * TE, d0);
#pragma unroll 1
        for (int s = s1 - 1; s >= 0; --s) {
            float *Q = P + s * Z3_TILE;
            const float *N = P + max(s - 1, 0) * Z3_TILE + TE - 8;
#pragma unroll
            for (int h = TE / 16 - 1; h >= 0; --h) {
                ld8(Q + 16 * h + TE, c1); ld8(Q + 16 * h + 2 * TE, d1);
#pragma unroll
                for (int e = 7; e >= 0; --e) {
                    xs = d0[e] - c0[e] * xs;
                    d0[e] = xs;
                }
                st8(Q + 16 * h + 8 + 2 * TE, d0);
                const float *F = h == 0 ? N : Q + 16 * (h - 1) + 8;
                ld8(F + TE, c0); ld8(F + 2 * TE, d0);
#pragma unroll
                for (int e = 7; e >= 0; --e) {
                    xs = d1[e] - c1[e] * xs;
                    d1[e] = xs;
                }
                st8(Q + 16 * h + 2 * TE, d1);
            }
        }
    };
    for (int r = nrounds - 1; r >= 0; --r) {
        if (solver) {
            float *P = tile_of(r & 1, 0) + lane * Z3_LSF;
            const int kr = __builtin_amdgcn_readfirstlane(r * RE);
            const int s1 = min(ZB_NM, (n - kr) / TE);                         // tiles [0, s1) lie wholly inside the line
            bwd_edge(P, kr, s1, ZB_NM);
            bwd_fast(P, s1);
        } else if (wave > 0) {
            if (r + 1 <= nrounds - 1) grab((r + 1) & 1, false);
            if (r - 1 >= 0) load_bwd(r - 1, (r - 1) & 1);
            if (r + 1 <= nrounds - 1) put(r + 1, false);
        }
        lds_barrier();
    }
    if (wave > 0) {
        grab(0, false);
        put(0, false);
    }
}

template <class Mdl, bool VERT, int MODE>
__global__ void __launch_bounds__(ZB_THREADS) k_alr_zebra3(typename Mdl::Ctx q, float *x, float *__restrict__ cp,
                                                           float *__restrict__ dv, float *__restrict__ dp, int nrows, int ncols,
                                                           size_t frame_stride, int first, int lastc, int lstep, float omega)
{
    alr_zebra3_body<Mdl, VERT, MODE>(q, x, cp, dv, dp, nrows, ncols, frame_stride, first, lastc, lstep, omega);
}

// One colour of BOTH fields of a coupled solver in one launch: field A's lines, then field B's same lines, by the same workgroup.
// The passes run [A even, A odd, B even, B odd]; [A odd] and [B even] touch disjoint data (A odd writes A on odd lines and reads A on
// even lines + B on odd lines; B even writes B on even lines and reads B on odd lines + A on even lines), so [A even, B even,
// A odd, B odd] gives the same bits -- and B's lines need A only at their own pixels (the coupling term), which this workgroup has
// just written.  Half the launches of a zebra iteration: at the drivers' scales a pass is 20-40 us of dependent steps and a launch
// boundary ~5 us.
template <class Mdl, bool VERT, int MODE>
__global__ void __launch_bounds__(ZB_THREADS) k_alr_zebra3_pair(typename Mdl::Ctx qa, float *xa, float *__restrict__ cpa, float *__restrict__ dva,
                                                                typename Mdl::Ctx qb, float *xb, float *__restrict__ cpb, float *__restrict__ dvb,
                                                                float *__restrict__ dp, int nrows, int ncols, size_t frame_stride, int first,
                                                                int lastc, int lstep, float omega)
{
    alr_zebra3_body<Mdl, VERT, MODE>(qa, xa, cpa, dva, dp, nrows, ncols, frame_stride, first, lastc, lstep, omega);
    __syncthreads(); // field A's stores have left this workgroup's waves; field B's coefficients read them back
    alr_zebra3_body<Mdl, VERT, MODE>(qb, xb, cpb, dvb, dp, nrows, ncols, frame_stride, first, lastc, lstep, omega);
}

// The factor passes of the two fields of a coupled solver in one launch (blockIdx.z = field): they are independent and each
// is bound by its recurrence, so side by side they take the time of one.
template <class Mdl, bool VERT>
__global__ void __launch_bounds__(ZB_THREADS) k_alr_factor_pair(typename Mdl::Ctx q0, typename Mdl::Ctx q1, float *__restrict__ cp0,
                                                                float *__restrict__ dv0, float *__restrict__ cp1, float *__restrict__ dv1,
                                                                int nrows, int ncols, size_t frame_stride, int first, int lastc)
{
    if (blockIdx.z == 0) alr_zebra3_body<Mdl, VERT, ZB_FACTOR>(q0, nullptr, cp0, dv0, nullptr, nrows, ncols, frame_stride, first, lastc, 1, 0.0f);
    else alr_zebra3_body<Mdl, VERT, ZB_FACTOR>(q1, nullptr, cp1, dv1, nullptr, nrows, ncols, frame_stride, first, lastc, 1, 0.0f);
}

// ------------------------------------------------------------------------------------------------
// Reference line order.
//
// cp[k] = c/(b - cp[k-1] a) depends on the coefficient planes only, not on the iterate, so it is the same
// in every iteration of a call: k_alr_zebra3<ZB_FACTOR> runs that recurrence once per call for every line
// and stores cp and the per-element divisor (1/(b - cp a); b itself for
// the first element, the plain denominator for the last -- those two are divided by, as in the
// reference).  What is left per line and iteration is the right-hand side (parallel along the line)
// and two short recurrences, dp = (d - dp' a) div and x = dp - cp x', which are serial along the
// line AND from line to line (line l needs the finished line l-1): one workgroup per frame walks the
// lines, all threads build the line into LDS and apply the SOR blend, one lane per chain runs the
// recurrences.  For the two-field solvers the second field's pass runs one line behind the first
// field's in the same workgroup (it only needs the first field's finished line l and its own l-1).
// ------------------------------------------------------------------------------------------------
constexpr int ALR_LEX_THREADS = 1024;

template <class Mdl> struct AlrChain {
    typename Mdl::Ctx q;
    float *x;            // the plane this chain solves (q reads it too)
    const float *cp, *dv; // the factor planes (k_alr_zebra3<ZB_FACTOR>) of this field and direction
};
template <class Mdl, int NCH> struct AlrChains {
    AlrChain<Mdl> c[NCH];
};

__device__ __forceinline__ float alr_from_lower_lane(float v)
{ // lane l <- lane l-1 (DPP wave_shr:1, folds into the consuming multiply); lane 0 reads 0
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x138, 0xf, 0xf, true));
}
__device__ __forceinline__ float alr_from_upper_lane(float v)
{ // lane l <- lane l+1 (wave_shl:1); lane 63 reads 0
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x130, 0xf, 0xf, true));
}

// Both recurrences of one line out of LDS, run by ONE WAVE; element = (a, div, cp, d) on entry, .w = the
// (unblended) solution on exit.
//
// The recurrences are serial, so only one value is "live" at a time -- but fetching each element's
// operands into the lane that holds that value costs more than the arithmetic.  Instead lane u keeps the
// operands of one element of a 63-element block, and the live value travels: 63 times over, every lane
// evaluates  dp = (d - dp[lane-1] * a) * div  with its lower neighbour's current value (DPP wave_shr:1).
// Lane 0 carries the value entering the block (a = 0, div = 1, d = carry reproduce it exactly), lane 1 is
// right after the first pass, lane 2 after the second, ... and a lane that is already right recomputes
// the same number, so after 63 passes all are final: three dependent VALU instructions per element,
// each with exactly the operands and the rounding of the serial loop.
constexpr int ALR_BLK = 63;

// `tdiv` (wave-uniform): the line is the south row of a model whose south-row function divides (Mdl::SOUTH_TRUEDIV); its
// middle elements then carry the bare denominator in .y and dp = (d - dp' a) / den.
__device__ __forceinline__ void alr_serial_wave(float4 *L, int n, int lane, bool tdiv)
{
    float carry; // wave-uniform: dp of the element before the current block
    {
        const float4 e = L[0];
        carry = e.w / e.y; // first element: d / b
        if (lane == 0) L[0].w = carry;
    }
    const int nmid = n - 2; // elements 1 .. n-2
    const int nblk = (nmid + ALR_BLK - 1) / ALR_BLK;
    // lane u >= 1 of block m holds element 1 + 63 m + (u - 1)
    float4 nxt = L[min(lane, n - 2)]; // block 0 (lane 0's slot is overwritten below)
    for (int m = 0; m < nblk; ++m) {
        const int k0 = 1 + ALR_BLK * m, k = k0 + lane - 1;
        const int cnt = min(ALR_BLK, n - 1 - k0);
        const bool valid = lane >= 1 && lane <= cnt;
        float4 e = nxt;
        nxt = L[min(k + ALR_BLK, n - 2)]; // next block's operands, in flight during this block's passes
        if (lane == 0) e = make_float4(0.0f, 1.0f, 0.0f, carry);
        float dp = 0.0f;
        if (tdiv) { // lane 0: (carry - 0 * 0) / 1 = carry exactly
            for (int u = 0; u <= cnt; ++u) dp = (e.w - alr_from_lower_lane(dp) * e.x) / e.y;
        } else if (cnt == ALR_BLK) {
#pragma unroll
            for (int u = 0; u <= ALR_BLK; ++u) dp = (e.w - alr_from_lower_lane(dp) * e.x) * e.y;
        } else {
            for (int u = 0; u <= cnt; ++u) dp = (e.w - alr_from_lower_lane(dp) * e.x) * e.y;
        }
        if (valid) L[k].w = dp;
        carry = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(dp), cnt));
    }
    float xs; // wave-uniform: solution of the element after the current block
    {
        const float4 e = L[n - 1];
        xs = (e.w - carry * e.x) / e.y; // last element: divided, not multiplied by a reciprocal
        if (lane == 0) L[n - 1].w = xs;
    }
    // back-substitution x = dp - cp * x[lane+1]: lane 63 carries the value entering the block from above
    // (dp = xs, cp = 0); lane u <= 62 of block m holds element 1 + 63 m + u
    nxt = L[min(max(1 + ALR_BLK * (nblk - 1) + lane, 1), n - 2)];
    for (int m = nblk - 1; m >= 0; --m) {
        const int k0 = 1 + ALR_BLK * m, k = k0 + lane;
        const int cnt = min(ALR_BLK, n - 1 - k0);
        const bool valid = lane < cnt;
        float4 e = nxt; // .w = dp, .z = cp
        nxt = L[min(max(k - ALR_BLK, 1), n - 2)];
        if (!valid) e = make_float4(0.0f, 0.0f, 0.0f, xs); // lane 63 (and lanes past a short block) hold the entering value
        float x = valid ? 0.0f : xs;
        if (cnt == ALR_BLK) { // lane 63: xs - 0 * 0 = xs, exactly, whatever xs is
#pragma unroll
            for (int u = 0; u <= ALR_BLK; ++u) x = e.w - e.z * alr_from_upper_lane(x);
        } else {
            for (int u = 0; u < cnt; ++u) {
                const float t = e.w - e.z * alr_from_upper_lane(x);
                x = valid ? t : xs;
            }
        }
        if (valid) L[k].w = x;
        xs = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(x)));
    }
    {
        const float4 e = L[0];
        const float x0 = e.w - e.z * xs;
        if (lane == 0) L[0].w = x0;
    }
}

// GL: the line buffers live in global memory (`gline`: [frames][NCH][n] float4) instead of LDS -- lines of more than 10 240 pixels,
// which the reference does not forbid; one workgroup works on a frame, so its barriers order those accesses as they order LDS.
// Slow (every step of the serial recurrence is a global round trip) and only taken then.
template <class Mdl, int NCH, bool VERT, bool GL = false>
__global__ void __launch_bounds__(ALR_LEX_THREADS) k_alr_lex(AlrChains<Mdl, NCH> ch, int nrows, int ncols, size_t frame_stride,
                                                             int lo, int hi, float omega, float4 *gline = nullptr)
{
    constexpr bool vertical = VERT;
    extern __shared__ float4 alr_lds_raw[]; // NCH lines of n elements
    float4 *const alr_lds = GL ? gline + (size_t)blockIdx.x * NCH * (vertical ? nrows : ncols) : alr_lds_raw;
    const size_t fo = (size_t)blockIdx.x * frame_stride;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        ch.c[c].q.shift(fo);
        ch.c[c].x += fo;
        ch.c[c].cp += fo;
        ch.c[c].dv += fo;
    }
    const int n = vertical ? nrows : ncols; // line length; line l starts at l * n in either layout
    constexpr size_t stride = 1;
    const float om1 = 1.0f - omega;
    const int tid = threadIdx.x;

    float sink = 0.0f;
    for (int s = lo; s <= hi + NCH - 1; ++s) {
#pragma unroll
        for (int c = 0; c < NCH; ++c) { // chain c works on line s - c
            const int l = s - c;
            if (l < lo || l > hi) continue;
            float4 *L = alr_lds + (size_t)c * n;
            const size_t base = (size_t)l * n;
            for (int k = tid; k < n; k += ALR_LEX_THREADS) {
                const Tri t = line_coef<Mdl, VERT>(ch.c[c].q, l, k, nrows, ncols);
                const size_t pos = base + (size_t)k * stride;
                L[k] = make_float4(t.a, ch.c[c].dv[pos], ch.c[c].cp[pos], t.d);
            }
        }
        __syncthreads();
        if ((tid >> 6) < NCH) {
            const int c = tid >> 6, l = s - c;
            if (l >= lo && l <= hi) alr_serial_wave(alr_lds + (size_t)c * n, n, tid & 63, Mdl::SOUTH_TRUEDIV && !VERT && l == nrows - 1);
        } else {
            // The other waves have nothing to do while the recurrences run: they touch everything the next
            // step's build will read (with values that are still stale, hence discarded), so that build
            // finds its operands in L2 with the address translations cached instead of paying HBM latency
            // on the critical path.
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
                const int l = s + 1 - c;
                if (l < lo || l > hi) continue;
                const size_t base = (size_t)l * n;
                for (int k = tid - 64 * NCH; k < n; k += ALR_LEX_THREADS - 64 * NCH) {
                    const Tri t = line_coef<Mdl, VERT>(ch.c[c].q, l, k, nrows, ncols);
                    const size_t pos = base + (size_t)k * stride;
                    sink += t.a + t.b + t.c + t.d + ch.c[c].dv[pos] + ch.c[c].cp[pos];
                }
            }
        }
        __syncthreads();
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            const int l = s - c;
            if (l < lo || l > hi) continue;
            const float4 *L = alr_lds + (size_t)c * n;
            float *x = ch.c[c].x;
            const size_t base = (size_t)l * n;
            for (int k = tid; k < n; k += ALR_LEX_THREADS) {
                const size_t pos = base + (size_t)k * stride;
                x[pos] = omega * L[k].w + om1 * x[pos];
            }
        }
        __syncthreads(); // the next lines' right-hand sides read these results
    }
    if (sink == 1.2345e-30f) ch.c[0].x[0] = sink; // keeps the warm-up loads alive; never true in practice
}

} // namespace pdeip
