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
//               walks the lines; all its threads build the line's coefficients into LDS and apply the
//               SOR blend, one thread runs the two recurrences out of LDS.  Bit-identical, CPU-class speed.
//   k_alr_zebra RED_BLACK: "zebra" order -- every even line, then every odd line.  Lines of one colour
//               only read the other colour, so they are solved concurrently, one lane per line, with the
//               same per-line arithmetic.
#pragma once
#include <hip/hip_runtime.h>

#include "pdeip_models.hpp"

namespace pdeip {

struct Tri {
    float a, b, c, d;
};

// "the terms that are present, in this order", as a C expression t1 + t2 + ... evaluates them
__device__ __forceinline__ void acc_add(float &v, bool &have, float t)
{
    v = have ? v + t : t;
    have = true;
}

// ---- early linearisation, 4 neighbours (opticalflowSolvers.c:1763-2410) ------------------------------
struct AlrElin4 {
    struct Ctx {
        const float *X, *O, *M, *C, *D, *wW, *wN, *wE, *wS; // X: the field being solved, O: the other field
        __device__ void shift(size_t) {}
    };
    static constexpr bool INTERIOR_LINES = false;
    __device__ __forceinline__ static Tri coef(const Ctx &q, int i, int j, int nrows, int ncols, bool vertical)
    {
        const size_t pos = (size_t)j * nrows + i;
        const bool hasN = i > 0, hasS = i < nrows - 1, hasW = j > 0, hasE = j < ncols - 1;
        const float wN = q.wN[pos], wS = q.wS[pos], wE = q.wE[pos], wW = q.wW[pos];
        float b = 0.0f, d = 0.0f;
        bool hb = false, hd = false;
        Tri t;
        if (hasN) acc_add(b, hb, wN); // b = wN + wS + wE + wW, missing ones skipped (:1917)
        if (hasS) acc_add(b, hb, wS);
        if (hasE) acc_add(b, hb, wE);
        if (hasW) acc_add(b, hb, wW);
        if (vertical) { // d = wW*U_w + wE*U_e (:1919)
            if (hasW) acc_add(d, hd, wW * q.X[pos - nrows]);
            if (hasE) acc_add(d, hd, wE * q.X[pos + nrows]);
            t.a = hasN ? -wN : 0.0f;
            t.c = hasS ? -wS : 0.0f;
        } else { // d = wS*U_s + wN*U_n (:2247)
            if (hasS) acc_add(d, hd, wS * q.X[pos + 1]);
            if (hasN) acc_add(d, hd, wN * q.X[pos - 1]);
            t.a = hasW ? -wW : 0.0f;
            t.c = hasE ? -wE : 0.0f;
        }
        const float C = q.C[pos];
        if (!is_nan(C)) { // :1921-1926
            b += q.D[pos];
            d += C;
            d -= q.M[pos] * q.O[pos];
        }
        t.b = b;
        t.d = d;
        return t;
    }
};

// ---- late linearisation, 4 neighbours: flow (opticalflowSolvers.c:2415-3100); M == nullptr: disparity
//      (disparitySolvers.c:1376-2029, the same lines without the coupling term) ---------------------------
struct AlrLlin4 {
    struct Ctx {
        const float *U, *X, *O, *M, *C, *D, *wW, *wN, *wE, *wS; // U: base field, X: increment being solved
        __device__ void shift(size_t) {}
    };
    static constexpr bool INTERIOR_LINES = false;
    __device__ __forceinline__ static Tri coef(const Ctx &q, int i, int j, int nrows, int ncols, bool vertical)
    {
        const size_t pos = (size_t)j * nrows + i;
        const size_t wpos = pos - nrows, epos = pos + nrows, npos = pos - 1, spos = pos + 1;
        const bool hasN = i > 0, hasS = i < nrows - 1, hasW = j > 0, hasE = j < ncols - 1;
        const float wN = q.wN[pos], wS = q.wS[pos], wE = q.wE[pos], wW = q.wW[pos];
        const float *U = q.U, *dU = q.X;
        const float Uc = U[pos];
        float b = 0.0f, d = 0.0f;
        bool hb = false, hd = false;
        Tri t;
        if (hasN) acc_add(b, hb, wN);
        if (hasS) acc_add(b, hb, wS);
        if (hasE) acc_add(b, hb, wE);
        if (hasW) acc_add(b, hb, wW);
        // d: W, E, S, N; neighbours that are not on the line carry their increment (:2589-2592, :2933-2936)
        if (vertical) {
            if (hasW) acc_add(d, hd, wW * (U[wpos] - Uc + dU[wpos]));
            if (hasE) acc_add(d, hd, wE * (U[epos] - Uc + dU[epos]));
            if (hasS) acc_add(d, hd, wS * (U[spos] - Uc));
            if (hasN) acc_add(d, hd, wN * (U[npos] - Uc));
            t.a = hasN ? -wN : 0.0f;
            t.c = hasS ? -wS : 0.0f;
        } else {
            if (hasW) acc_add(d, hd, wW * (U[wpos] - Uc));
            if (hasE) acc_add(d, hd, wE * (U[epos] - Uc));
            if (hasS) acc_add(d, hd, wS * (U[spos] - Uc + dU[spos]));
            if (hasN) acc_add(d, hd, wN * (U[npos] - Uc + dU[npos]));
            t.a = hasW ? -wW : 0.0f;
            t.c = hasE ? -wE : 0.0f;
        }
        const float C = q.C[pos];
        if (!is_nan(C)) {
            b += q.D[pos];
            d += C;
            if (q.M) d -= q.M[pos] * q.O[pos];
        }
        t.b = b;
        t.d = d;
        return t;
    }
};

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
    __device__ __forceinline__ static int third(int k, int n) { return k == 0 ? 0 : (k == n - 1 ? 2 : 1); }
    __device__ __forceinline__ static Tri coef(const Ctx &q, int i, int j, int nrows, int ncols, bool vertical)
    {
        const size_t pos = (size_t)j * nrows + i;
        const long off[8] = {-1, 1, nrows, -(long)nrows, -(long)nrows - 1, (long)nrows - 1, -(long)nrows + 1, (long)nrows + 1};
        const signed char(*cs)[9] = vertical ? ALR_L8[0][third(j, ncols)][third(i, nrows)] : ALR_L8[1][third(i, nrows)][third(j, ncols)];
        const float Uc = q.U[pos];
        float b = 0.0f, d = 0.0f;
        for (int k = 0; cs[0][k] != DEND; ++k) {
            const float w = q.w[cs[0][k]][pos];
            b = k ? b + w : w;
        }
        for (int k = 0; cs[1][k] != DEND; ++k) {
            const int dir = cs[1][k];
            const size_t nb = pos + off[dir];
            const bool on_line = vertical ? (dir == DN || dir == DS) : (dir == DW || dir == DE);
            const float w = q.w[dir][pos];
            const float term = on_line ? w * (q.U[nb] - Uc) : w * (q.U[nb] - Uc + q.X[nb]);
            d = k ? d + term : term;
        }
        Tri t;
        if (vertical) {
            t.a = i > 0 ? -q.w[DN][pos] : 0.0f;
            t.c = i < nrows - 1 ? -q.w[DS][pos] : 0.0f;
        } else {
            t.a = j > 0 ? -q.w[DW][pos] : 0.0f;
            t.c = j < ncols - 1 ? -q.w[DE][pos] : 0.0f;
        }
        const float C = q.C[pos];
        if (!is_nan(C)) {
            b += q.D[pos];
            d += C;
            d -= q.M[pos] * q.O[pos];
        }
        t.b = b;
        t.d = d;
        return t;
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
    __device__ __forceinline__ static Tri coef(const Ctx &q, int i, int j, int nrows, int ncols, bool vertical)
    {
        const size_t pos = (size_t)j * nrows + i;
        const bool hasN = i > 0, hasS = i < nrows - 1, hasW = j > 0, hasE = j < ncols - 1;
        const float wN = q.wN[pos], wS = q.wS[pos], wE = q.wE[pos], wW = q.wW[pos];
        float b = 0.0f, d = 0.0f;
        bool hb = false, hd = false;
        Tri t;
        if (vertical) { // pdeSolvers.c:593
            if (hasW) acc_add(d, hd, wW * q.X[pos - nrows]);
            if (hasE) acc_add(d, hd, wE * q.X[pos + nrows]);
            t.a = hasN ? -wN : 0.0f;
            t.c = hasS ? -wS : 0.0f;
        } else { // :956
            if (hasS) acc_add(d, hd, wS * q.X[pos + 1]);
            if (hasN) acc_add(d, hd, wN * q.X[pos - 1]);
            t.a = hasW ? -wW : 0.0f;
            t.c = hasE ? -wE : 0.0f;
        }
        const float T = q.T[pos];
        if (!is_nan(T)) { // :595-599
            b = T;
            d += q.B[pos];
        } else { // :601-603: wN + wS + wW + wE, missing ones skipped
            if (hasN) acc_add(b, hb, wN);
            if (hasS) acc_add(b, hb, wS);
            if (hasW) acc_add(b, hb, wW);
            if (hasE) acc_add(b, hb, wE);
        }
        t.b = b;
        t.d = d;
        return t;
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
    __device__ __forceinline__ static Tri coef(const Ctx &q, int i, int j, int nrows, int ncols, bool vertical)
    {
        const size_t pos = (size_t)j * nrows + i;
        const size_t wpos = pos - nrows, epos = pos + nrows, npos = pos - 1, spos = pos + 1;
        const bool hasN = i > 0, hasS = i < nrows - 1, hasW = j > 0, hasE = j < ncols - 1;
        const float *X = q.X;
        Tri t;
        float d;
        if (vertical) { // :1171-1173, :1195-1197, :1227-1228
            d = q.wW[pos] * X[wpos] + q.wE[pos] * X[epos];
            if (hasS) d += q.wSW[pos] * X[wpos + 1] + q.wSE[pos] * X[epos + 1];
            if (hasN) d += q.wNW[pos] * X[wpos - 1] + q.wNE[pos] * X[epos - 1];
            t.a = hasN ? -q.wN[pos] : 0.0f;
            t.c = hasS ? -q.wS[pos] : 0.0f;
        } else { // :1309-1310, :1335-1337, :1364-1365
            d = q.wS[pos] * X[spos] + q.wN[pos] * X[npos];
            if (hasW) d += q.wSW[pos] * X[spos - nrows] + q.wNW[pos] * X[npos - nrows];
            if (hasE) d += q.wSE[pos] * X[spos + nrows] + q.wNE[pos] * X[npos + nrows];
            t.a = hasW ? -q.wW[pos] : 0.0f;
            t.c = hasE ? -q.wE[pos] : 0.0f;
        }
        const float T = q.T[pos];
        if (!is_nan(T)) {
            t.b = T;
            d += q.B[pos];
        } else { // :1181-1182 as written: wNW twice, wNE never, all eight terms at every position
            float b = q.wN[pos] + q.wS[pos] + q.wW[pos] + q.wE[pos];
            b += q.wNW[pos] + q.wNW[pos] + q.wSW[pos] + q.wSE[pos];
            t.b = b;
        }
        t.d = d;
        return t;
    }
};

// ------------------------------------------------------------------------------------------------
// Zebra order: one lane per line of the active colour.  cp/dp are scratch planes with x's layout.
// Thomas recurrences and the lagged SOR blend as opticalflowSolvers.c:1890-1958.
// ------------------------------------------------------------------------------------------------
template <class Mdl>
__global__ void __launch_bounds__(64) k_alr_zebra(typename Mdl::Ctx q, float *x, float *__restrict__ cp,
                                                  float *__restrict__ dp, int nrows, int ncols, size_t frame_stride,
                                                  int vertical, int lo, int hi, int colour, float omega)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int l = lo + (((lo & 1) != colour) ? 1 : 0) + 2 * t;
    if (l > hi) return;
    const size_t fo = (size_t)blockIdx.y * frame_stride;
    q.shift(fo);
    x += fo;
    cp += fo;
    dp += fo;
    const int n = vertical ? nrows : ncols;
    const size_t stride = vertical ? 1 : (size_t)nrows;
    const size_t base = vertical ? (size_t)l * nrows : (size_t)l;
    const float om1 = 1.0f - omega;

    Tri c0 = vertical ? Mdl::coef(q, 0, l, nrows, ncols, true) : Mdl::coef(q, l, 0, nrows, ncols, false);
    float cpv = c0.c / c0.b;
    float dpv = c0.d / c0.b;
    cp[base] = cpv;
    dp[base] = dpv;
    int k;
    for (k = 1; k <= n - 2; ++k) {
        const Tri c = vertical ? Mdl::coef(q, k, l, nrows, ncols, true) : Mdl::coef(q, l, k, nrows, ncols, false);
        const float div = 1.0f / (c.b - cpv * c.a);
        cpv = c.c * div;
        dpv = (c.d - dpv * c.a) * div;
        cp[base + k * stride] = cpv;
        dp[base + k * stride] = dpv;
    }
    {
        const Tri c = vertical ? Mdl::coef(q, k, l, nrows, ncols, true) : Mdl::coef(q, l, k, nrows, ncols, false);
        dpv = (c.d - dpv * c.a) / (c.b - cpv * c.a);
    }
    // back-substitution; element k+1 gets its blend once it has been used
    float xs = dpv;
    float old = x[base + (size_t)k * stride];
    for (k = n - 2; k >= 0; --k) {
        const size_t pos = base + (size_t)k * stride;
        const float xk = dp[pos] - cp[pos] * xs;
        x[pos + stride] = omega * xs + om1 * old;
        old = x[pos];
        xs = xk;
    }
    x[base] = omega * xs + om1 * old;
}

// ------------------------------------------------------------------------------------------------
// Reference line order: one workgroup per frame walks the lines lo..hi.  LDS holds one float4 per
// line element: (a,b,c,d) after the parallel build, (xs,.,cp,dp) after the serial recurrences.
// ------------------------------------------------------------------------------------------------
constexpr int ALR_LEX_THREADS = 1024;

template <class Mdl>
__global__ void __launch_bounds__(ALR_LEX_THREADS) k_alr_lex(typename Mdl::Ctx q, float *x, int nrows, int ncols,
                                                             size_t frame_stride, int vertical, int lo, int hi, float omega)
{
    extern __shared__ float4 alr_line[];
    const size_t fo = (size_t)blockIdx.x * frame_stride;
    q.shift(fo);
    x += fo;
    const int n = vertical ? nrows : ncols;
    const size_t stride = vertical ? 1 : (size_t)nrows;
    const float om1 = 1.0f - omega;
    const int tid = threadIdx.x;

    for (int l = lo; l <= hi; ++l) {
        const size_t base = vertical ? (size_t)l * nrows : (size_t)l;
        for (int k = tid; k < n; k += ALR_LEX_THREADS) {
            const Tri c = vertical ? Mdl::coef(q, k, l, nrows, ncols, true) : Mdl::coef(q, l, k, nrows, ncols, false);
            alr_line[k] = make_float4(c.a, c.b, c.c, c.d);
        }
        __syncthreads();
        if (tid == 0) {
            float4 c = alr_line[0];
            float cpv = c.z / c.y;
            float dpv = c.w / c.y;
            alr_line[0].z = cpv;
            alr_line[0].w = dpv;
            float4 nx = alr_line[1];
            int k;
            for (k = 1; k <= n - 2; ++k) {
                c = nx;
                nx = alr_line[k + 1]; // in flight while the recurrence step runs
                const float div = 1.0f / (c.y - cpv * c.x);
                cpv = c.z * div;
                dpv = (c.w - dpv * c.x) * div;
                alr_line[k].z = cpv;
                alr_line[k].w = dpv;
            }
            c = nx;
            dpv = (c.w - dpv * c.x) / (c.y - cpv * c.x);
            float xs = dpv;
            alr_line[k].x = xs;
            for (k = n - 2; k >= 0; --k) {
                const float4 e = alr_line[k];
                xs = e.w - e.z * xs;
                alr_line[k].x = xs;
            }
        }
        __syncthreads();
        for (int k = tid; k < n; k += ALR_LEX_THREADS) {
            const size_t pos = base + (size_t)k * stride;
            x[pos] = omega * alr_line[k].x + om1 * x[pos];
        }
        __syncthreads(); // the next line's build reads this line's result
    }
}

} // namespace pdeip
