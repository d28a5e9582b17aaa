/*
 * pdeip_oracle_alr.c -- CPU oracle, alternating line relaxation (solver = 2 of the gateways).
 *
 * TEST INFRASTRUCTURE ONLY, PARITY UNPINNED: see pdeip_oracle.h.
 *
 * Restates, in plain C99, the line solvers of the reference library:
 *   GS_ALR_SOR_elin4_2d   opticalflowSolvers.c:196-262   + {west,middle,east}Column_elin4, {north,middle,south}Row_elin4 :1763-2410
 *   GS_ALR_SOR_llin4_2d   opticalflowSolvers.c:690-759   + *_llin4 :2415-3100
 *   GS_ALR_SOR_llin8_2d   opticalflowSolvers.c:1677-1750 + *_llin8 :3104-3914
 *   GS_ALR_SOR_llin4_2d   disparitySolvers.c:154-211     + *Column4 / *Row4 :1376-2029
 *   GS_ALR_SOR_4_2d       pdeSolvers.c:277-335           + TDMA_{w,m,e}column_ALR_4, TDMA_{n,m,s}row_ALR_4 :409-1131
 *   GS_ALR_SOR_8_2d       pdeSolvers.c:344-402           + TDMAcolumn_ALR_8, TDMArow_ALR_8 :1132-1393
 *
 * Every line function of the reference has the same shape: a Thomas (TDMA) solve along one image
 * column or row in which ALL pixels of the line, border pixels included, are unknowns; a
 * neighbour that lies outside the image simply drops out of the diagonal `b` and of the
 * right-hand side `d` (Neumann).  What differs between the "west / middle / east" (or "north /
 * middle / south") variants and between the first / middle / last element of a line is only WHICH
 * terms are present and in WHICH ORDER they are added.  The restatement therefore has one Thomas
 * routine (`thomas_line`, arithmetic of e.g. opticalflowSolvers.c:1905-1958) and one coefficient
 * function per model that adds the present terms in the reference's order.  For the 4-neighbour
 * models the order is a fixed sequence with the missing terms skipped; the 8-neighbour late-
 * linearisation lines use a different order in each of their 18 cases, transcribed as tables.
 *
 * Line order:
 *   ORC_ORDER_LEX    : the reference's: lines one after the other (west->east, north->south).
 *   ORC_ORDER_COLOUR : "zebra": all even-indexed lines, then all odd-indexed lines, same per-line
 *                      arithmetic.  Not in the reference; it defines the product's RED_BLACK mode.
 */
#include "pdeip_oracle.h"

#include <stddef.h>
#include <stdlib.h>

#define ORC_ISNAN(x) ((x) != (x))

/* One tridiagonal row: a*x[k-1] + b*x[k] + c*x[k+1] = d  (a of the first and c of the last are unused). */
typedef struct {
    float a, b, c, d;
} tri_t;

/* accumulate "the terms that are present, in this order" the way a C expression t1 + t2 + ... does */
typedef struct {
    float v;
    int have;
} acc_t;
static void acc_add(acc_t *s, float t)
{
    if (!s->have) {
        s->v = t;
        s->have = 1;
    } else {
        s->v = s->v + t;
    }
}

typedef tri_t (*coef_fn)(const void *ctx, int i, int j, int vertical);

/* Thomas solve of one line + lagged SOR blend during back-substitution
 * (opticalflowSolvers.c:1890-1958 for a column, :2219-2296 for a row).  x points at element 0 of the
 * line, consecutive elements are `stride` floats apart. */
static void thomas_line(float *x, ptrdiff_t stride, int n, coef_fn coef, const void *ctx, int fixed, int vertical,
                        float omega, float *cp, float *dp, int truediv)
{
    int k;
    tri_t t;
    float div, temp1, temp2;
    ptrdiff_t pos;

    t = vertical ? coef(ctx, 0, fixed, 1) : coef(ctx, fixed, 0, 0);
    cp[0] = t.c / t.b;
    dp[0] = t.d / t.b;
    for (k = 1; k <= n - 2; k++) {
        t = vertical ? coef(ctx, k, fixed, 1) : coef(ctx, fixed, k, 0);
        if (truediv) {
            /* southRow_llin4 (opticalflowSolvers.c:3059-3060), southRow_llin8 (:3871-3872), southRow4
             * (disparitySolvers.c:1986-1987): the only line functions whose middle elements divide
             * instead of multiplying by a reciprocal (audit table: DESIGN.md section 5.5). */
            cp[k] = t.c / (t.b - cp[k - 1] * t.a);
            dp[k] = (t.d - dp[k - 1] * t.a) / (t.b - cp[k - 1] * t.a);
        } else {
            div = 1 / (t.b - cp[k - 1] * t.a);
            cp[k] = t.c * div;
            dp[k] = (t.d - dp[k - 1] * t.a) * div;
        }
    }
    t = vertical ? coef(ctx, k, fixed, 1) : coef(ctx, fixed, k, 0);
    dp[k] = (t.d - dp[k - 1] * t.a) / (t.b - cp[k - 1] * t.a);
    pos = (ptrdiff_t)k * stride;
    temp1 = x[pos];
    x[pos] = dp[k];
    for (k = n - 2; k >= 0; k--) {
        pos = (ptrdiff_t)k * stride;
        temp2 = x[pos];
        x[pos] = dp[k] - cp[k] * x[pos + stride];
        x[pos + stride] = omega * x[pos + stride] + (1.0f - omega) * temp1;
        temp1 = temp2;
    }
    x[pos] = omega * x[pos] + (1.0f - omega) * temp1;
}

/* All lines [lo, hi] of one direction over the plane x (nrows x ncols), in the given order. */
/* south_div: the model's south-row function (the row pass on line nrows-1) uses true division. */
static void line_pass(float *x, int nrows, int ncols, int vertical, int lo, int hi, int order, coef_fn coef,
                      const void *ctx, float omega, float *cp, float *dp, int south_div)
{
    int pass, l;
    int colour = order & 1;
    for (pass = 0; pass < (colour ? 2 : 1); pass++) {
        for (l = lo; l <= hi; l++) {
            if (colour && (l & 1) != pass)
                continue;
            if (vertical)
                thomas_line(x + (size_t)l * nrows, 1, nrows, coef, ctx, l, 1, omega, cp, dp, 0);
            else
                thomas_line(x + l, nrows, ncols, coef, ctx, l, 0, omega, cp, dp, south_div && l == nrows - 1);
        }
    }
}

static int alloc_scratch(int nrows, int ncols, float **cp, float **dp)
{
    int n = nrows > ncols ? nrows : ncols; /* opticalflowSolvers.c:219 */
    *cp = (float *)malloc((size_t)n * sizeof(float));
    *dp = (float *)malloc((size_t)n * sizeof(float));
    if (!*cp || !*dp) {
        free(*cp);
        free(*dp);
        return 0;
    }
    return 1;
}

/* ------------------------------------------------------------------------------------------------
 * Early linearisation, 4 neighbours (opticalflowSolvers.c:1763-2410)
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
    const float *U, *V, *M, *C, *D, *wW, *wN, *wE, *wS; /* U = the field being solved, V = the other one */
    int nrows, ncols;
} elin4_ctx;

static tri_t elin4_coef(const void *vctx, int i, int j, int vertical)
{
    const elin4_ctx *q = (const elin4_ctx *)vctx;
    int nrows = q->nrows;
    size_t pos = (size_t)j * nrows + i;
    int hasN = i > 0, hasS = i < nrows - 1, hasW = j > 0, hasE = j < q->ncols - 1;
    acc_t b = {0.0f, 0}, d = {0.0f, 0};
    tri_t t;
    /* b = wN + wS + wE + wW, missing ones skipped (:1917, :1897, :1942 ...) */
    if (hasN) acc_add(&b, q->wN[pos]);
    if (hasS) acc_add(&b, q->wS[pos]);
    if (hasE) acc_add(&b, q->wE[pos]);
    if (hasW) acc_add(&b, q->wW[pos]);
    if (vertical) { /* d = wW*U_w + wE*U_e (:1919) */
        if (hasW) acc_add(&d, q->wW[pos] * q->U[pos - nrows]);
        if (hasE) acc_add(&d, q->wE[pos] * q->U[pos + nrows]);
        t.a = hasN ? -q->wN[pos] : 0.0f;
        t.c = hasS ? -q->wS[pos] : 0.0f;
    } else { /* d = wS*U_s + wN*U_n (:2247) */
        if (hasS) acc_add(&d, q->wS[pos] * q->U[pos + 1]);
        if (hasN) acc_add(&d, q->wN[pos] * q->U[pos - 1]);
        t.a = hasW ? -q->wW[pos] : 0.0f;
        t.c = hasE ? -q->wE[pos] : 0.0f;
    }
    t.b = b.v;
    t.d = d.v;
    if (!ORC_ISNAN(q->C[pos])) { /* :1921-1926 -- note: tests Cu only, unlike the point solver */
        t.b += q->D[pos];
        t.d += q->C[pos];
        t.d -= q->M[pos] * q->V[pos];
    }
    return t;
}

void orc_oflow_alr_elin4(float *U, float *V, const float *M, const float *Cu, const float *Cv, const float *Du,
                         const float *Dv, const float *wW, const float *wN, const float *wE, const float *wS,
                         int nrows, int ncols, int iter, float omega, int order)
{
    float *cp, *dp;
    int it;
    elin4_ctx qu = {U, V, M, Cu, Du, wW, wN, wE, wS, nrows, ncols};
    elin4_ctx qv = {V, U, M, Cv, Dv, wW, wN, wE, wS, nrows, ncols};
    if (nrows < 2 || ncols < 2 || !alloc_scratch(nrows, ncols, &cp, &dp))
        return;
    for (it = 0; it < iter; it++) { /* :231-258: columns U then V, rows V then U */
        line_pass(U, nrows, ncols, 1, 0, ncols - 1, order, elin4_coef, &qu, omega, cp, dp, 0);
        line_pass(V, nrows, ncols, 1, 0, ncols - 1, order, elin4_coef, &qv, omega, cp, dp, 0);
        line_pass(V, nrows, ncols, 0, 0, nrows - 1, order, elin4_coef, &qv, omega, cp, dp, 0);
        line_pass(U, nrows, ncols, 0, 0, nrows - 1, order, elin4_coef, &qu, omega, cp, dp, 0);
    }
    free(cp);
    free(dp);
}

/* ------------------------------------------------------------------------------------------------
 * Late linearisation, 4 neighbours: optical flow (opticalflowSolvers.c:2415-3100) and disparity
 * (disparitySolvers.c:1376-2029; the same lines without the M*dV coupling)
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
    const float *U, *dU, *dV, *M, *C, *D, *wW, *wN, *wE, *wS; /* M == NULL: disparity */
    int nrows, ncols;
} llin4_ctx;

static tri_t llin4_coef(const void *vctx, int i, int j, int vertical)
{
    const llin4_ctx *q = (const llin4_ctx *)vctx;
    int nrows = q->nrows;
    size_t pos = (size_t)j * nrows + i;
    size_t wpos = pos - nrows, epos = pos + nrows, npos = pos - 1, spos = pos + 1;
    int hasN = i > 0, hasS = i < nrows - 1, hasW = j > 0, hasE = j < q->ncols - 1;
    const float *U = q->U, *dU = q->dU;
    acc_t b = {0.0f, 0}, d = {0.0f, 0};
    tri_t t;
    if (hasN) acc_add(&b, q->wN[pos]);
    if (hasS) acc_add(&b, q->wS[pos]);
    if (hasE) acc_add(&b, q->wE[pos]);
    if (hasW) acc_add(&b, q->wW[pos]);
    /* d: W, E, S, N; the neighbours that are NOT on the line carry their increment (:2589-2592 columns,
     * :2933-2936 rows) */
    if (vertical) {
        if (hasW) acc_add(&d, q->wW[pos] * (U[wpos] - U[pos] + dU[wpos]));
        if (hasE) acc_add(&d, q->wE[pos] * (U[epos] - U[pos] + dU[epos]));
        if (hasS) acc_add(&d, q->wS[pos] * (U[spos] - U[pos]));
        if (hasN) acc_add(&d, q->wN[pos] * (U[npos] - U[pos]));
        t.a = hasN ? -q->wN[pos] : 0.0f;
        t.c = hasS ? -q->wS[pos] : 0.0f;
    } else {
        if (hasW) acc_add(&d, q->wW[pos] * (U[wpos] - U[pos]));
        if (hasE) acc_add(&d, q->wE[pos] * (U[epos] - U[pos]));
        if (hasS) acc_add(&d, q->wS[pos] * (U[spos] - U[pos] + dU[spos]));
        if (hasN) acc_add(&d, q->wN[pos] * (U[npos] - U[pos] + dU[npos]));
        t.a = hasW ? -q->wW[pos] : 0.0f;
        t.c = hasE ? -q->wE[pos] : 0.0f;
    }
    t.b = b.v;
    t.d = d.v;
    if (!ORC_ISNAN(q->C[pos])) {
        t.b += q->D[pos];
        t.d += q->C[pos];
        if (q->M)
            t.d -= q->M[pos] * q->dV[pos];
    }
    return t;
}

void orc_oflow_alr_llin4(const float *U, const float *V, float *dU, float *dV, const float *M, const float *Cu,
                         const float *Cv, const float *Du, const float *Dv, const float *wW, const float *wN,
                         const float *wE, const float *wS, int nrows, int ncols, int iter, float omega, int order)
{
    float *cp, *dp;
    int it;
    llin4_ctx qu = {U, dU, dV, M, Cu, Du, wW, wN, wE, wS, nrows, ncols};
    llin4_ctx qv = {V, dV, dU, M, Cv, Dv, wW, wN, wE, wS, nrows, ncols};
    if (nrows < 2 || ncols < 2 || !alloc_scratch(nrows, ncols, &cp, &dp))
        return;
    for (it = 0; it < iter; it++) { /* :728-755 */
        line_pass(dU, nrows, ncols, 1, 0, ncols - 1, order, llin4_coef, &qu, omega, cp, dp, 1);
        line_pass(dV, nrows, ncols, 1, 0, ncols - 1, order, llin4_coef, &qv, omega, cp, dp, 1);
        line_pass(dV, nrows, ncols, 0, 0, nrows - 1, order, llin4_coef, &qv, omega, cp, dp, 1);
        line_pass(dU, nrows, ncols, 0, 0, nrows - 1, order, llin4_coef, &qu, omega, cp, dp, 1);
    }
    free(cp);
    free(dp);
}

void orc_disp_alr_llin4(const float *U, float *dU, const float *Cu, const float *Du, const float *wW,
                        const float *wN, const float *wE, const float *wS, int nrows, int ncols, int iter,
                        float omega, int order)
{
    float *cp, *dp;
    int it;
    llin4_ctx q = {U, dU, NULL, NULL, Cu, Du, wW, wN, wE, wS, nrows, ncols};
    if (nrows < 2 || ncols < 2 || !alloc_scratch(nrows, ncols, &cp, &dp))
        return;
    for (it = 0; it < iter; it++) { /* disparitySolvers.c:186-204: columns, then rows */
        line_pass(dU, nrows, ncols, 1, 0, ncols - 1, order, llin4_coef, &q, omega, cp, dp, 1);
        line_pass(dU, nrows, ncols, 0, 0, nrows - 1, order, llin4_coef, &q, omega, cp, dp, 1);
    }
    free(cp);
    free(dp);
}

/* ------------------------------------------------------------------------------------------------
 * Late linearisation, 8 neighbours (opticalflowSolvers.c:3104-3914).  Term orders per case.
 * ---------------------------------------------------------------------------------------------- */
enum { DN, DS, DE, DW, DNW, DNE, DSW, DSE, DEND };

/* [vertical? 0 column pass : 1 row pass][line: first/middle/last][element: first/middle/last] */
typedef struct {
    signed char b[9], d[9];
} l8_case;
#define L8(...) {__VA_ARGS__, DEND}
static const l8_case L8_TABLE[2][3][3] = {
    { /* column pass: west column (:3104), middle columns (:3237), east column (:3380) */
        {{L8(DS, DE, DSE), L8(DS, DE, DSE)},
         {L8(DN, DS, DE, DNE, DSE), L8(DS, DN, DNE, DE, DSE)},
         {L8(DN, DE, DNE), L8(DN, DNE, DE)}},
        {{L8(DS, DE, DW, DSE, DSW), L8(DS, DW, DE, DSE, DSW)},
         {L8(DN, DS, DE, DW, DNW, DNE, DSW, DSE), L8(DN, DS, DW, DNW, DNE, DE, DSW, DSE)},
         {L8(DN, DE, DW, DNW, DNE), L8(DN, DW, DNW, DNE, DE)}},
        {{L8(DS, DW, DSW), L8(DS, DW, DSW)},
         {L8(DN, DS, DW, DNW, DSW), L8(DS, DN, DW, DNW, DSW)},
         {L8(DN, DW, DNW), L8(DN, DW, DNW)}},
    },
    { /* row pass: north row (:3513), middle rows (:3646), south row (:3789) */
        {{L8(DS, DE, DSE), L8(DE, DSE, DS)},
         {L8(DS, DE, DW, DSW, DSE), L8(DW, DE, DSW, DSE, DS)},
         {L8(DS, DW, DSW), L8(DW, DSW, DS)}},
        {{L8(DN, DS, DE, DNE, DSE), L8(DE, DNE, DSE, DS, DN)},
         {L8(DN, DS, DE, DW, DNW, DNE, DSW, DSE), L8(DW, DE, DNW, DNE, DSW, DSE, DS, DN)},
         {L8(DN, DS, DW, DNW, DSW), L8(DW, DNW, DSW, DS, DN)}},
        {{L8(DN, DE, DNE), L8(DE, DNE, DN)},
         {L8(DN, DE, DW, DNW, DNE), L8(DW, DE, DNW, DNE, DN)},
         {L8(DN, DW, DNW), L8(DW, DNW, DN)}},
    },
};

typedef struct {
    const float *U, *dU, *dV, *M, *C, *D;
    const float *w[8]; /* indexed by DN..DSE */
    int nrows, ncols;
} llin8_ctx;

static int third(int k, int n) { return k == 0 ? 0 : (k == n - 1 ? 2 : 1); }

static tri_t llin8_coef(const void *vctx, int i, int j, int vertical)
{
    const llin8_ctx *q = (const llin8_ctx *)vctx;
    int nrows = q->nrows;
    size_t pos = (size_t)j * nrows + i;
    ptrdiff_t off[8];
    const l8_case *cs;
    const float *U = q->U, *dU = q->dU;
    float b = 0.0f, d = 0.0f;
    int k;
    tri_t t;
    off[DN] = -1;
    off[DS] = 1;
    off[DE] = nrows;
    off[DW] = -nrows;
    off[DNW] = -nrows - 1;
    off[DNE] = nrows - 1;
    off[DSW] = -nrows + 1;
    off[DSE] = nrows + 1;
    cs = vertical ? &L8_TABLE[0][third(j, q->ncols)][third(i, nrows)] : &L8_TABLE[1][third(i, nrows)][third(j, q->ncols)];
    for (k = 0; cs->b[k] != DEND; k++) {
        float w = q->w[cs->b[k]][pos];
        b = k ? b + w : w;
    }
    for (k = 0; cs->d[k] != DEND; k++) {
        int dir = cs->d[k];
        size_t nb = pos + off[dir];
        /* the two neighbours on the line itself enter without their increment (they are the unknowns) */
        int on_line = vertical ? (dir == DN || dir == DS) : (dir == DW || dir == DE);
        float term = on_line ? q->w[dir][pos] * (U[nb] - U[pos]) : q->w[dir][pos] * (U[nb] - U[pos] + dU[nb]);
        d = k ? d + term : term;
    }
    if (vertical) {
        t.a = i > 0 ? -q->w[DN][pos] : 0.0f;
        t.c = i < nrows - 1 ? -q->w[DS][pos] : 0.0f;
    } else {
        t.a = j > 0 ? -q->w[DW][pos] : 0.0f;
        t.c = j < q->ncols - 1 ? -q->w[DE][pos] : 0.0f;
    }
    t.b = b;
    t.d = d;
    if (!ORC_ISNAN(q->C[pos])) {
        t.b += q->D[pos];
        t.d += q->C[pos];
        t.d -= q->M[pos] * q->dV[pos];
    }
    return t;
}

void orc_oflow_alr_llin8(const float *U, const float *V, float *dU, float *dV, const float *M, const float *Cu,
                         const float *Cv, const float *Du, const float *Dv, const float *wW, const float *wNW,
                         const float *wN, const float *wNE, const float *wE, const float *wSE, const float *wS,
                         const float *wSW, int nrows, int ncols, int iter, float omega, int order)
{
    float *cp, *dp;
    int it;
    llin8_ctx qu = {U, dU, dV, M, Cu, Du, {wN, wS, wE, wW, wNW, wNE, wSW, wSE}, nrows, ncols};
    llin8_ctx qv = {V, dV, dU, M, Cv, Dv, {wN, wS, wE, wW, wNW, wNE, wSW, wSE}, nrows, ncols};
    if (nrows < 2 || ncols < 2 || !alloc_scratch(nrows, ncols, &cp, &dp))
        return;
    for (it = 0; it < iter; it++) { /* :1718-1746 */
        line_pass(dU, nrows, ncols, 1, 0, ncols - 1, order, llin8_coef, &qu, omega, cp, dp, 1);
        line_pass(dV, nrows, ncols, 1, 0, ncols - 1, order, llin8_coef, &qv, omega, cp, dp, 1);
        line_pass(dV, nrows, ncols, 0, 0, nrows - 1, order, llin8_coef, &qv, omega, cp, dp, 1);
        line_pass(dU, nrows, ncols, 0, 0, nrows - 1, order, llin8_coef, &qu, omega, cp, dp, 1);
    }
    free(cp);
    free(dp);
}

/* ------------------------------------------------------------------------------------------------
 * PDE (denoising) solvers (pdeSolvers.c:409-1393)
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
    const float *X, *T, *B, *wW, *wNW, *wN, *wNE, *wE, *wSE, *wS, *wSW; /* frame-offset pointers */
    int nrows, ncols;
} pde_ctx;

static tri_t pde4_coef(const void *vctx, int i, int j, int vertical)
{
    const pde_ctx *q = (const pde_ctx *)vctx;
    int nrows = q->nrows;
    size_t pos = (size_t)j * nrows + i;
    int hasN = i > 0, hasS = i < nrows - 1, hasW = j > 0, hasE = j < q->ncols - 1;
    acc_t b = {0.0f, 0}, d = {0.0f, 0};
    tri_t t;
    if (vertical) { /* pdeSolvers.c:593 */
        if (hasW) acc_add(&d, q->wW[pos] * q->X[pos - nrows]);
        if (hasE) acc_add(&d, q->wE[pos] * q->X[pos + nrows]);
        t.a = hasN ? -q->wN[pos] : 0.0f;
        t.c = hasS ? -q->wS[pos] : 0.0f;
    } else { /* :956 */
        if (hasS) acc_add(&d, q->wS[pos] * q->X[pos + 1]);
        if (hasN) acc_add(&d, q->wN[pos] * q->X[pos - 1]);
        t.a = hasW ? -q->wW[pos] : 0.0f;
        t.c = hasE ? -q->wE[pos] : 0.0f;
    }
    t.d = d.v;
    if (!ORC_ISNAN(q->T[pos])) { /* :595-599 */
        t.b = q->T[pos];
        t.d += q->B[pos];
    } else { /* :601-603: wN + wS + wW + wE, missing ones skipped */
        if (hasN) acc_add(&b, q->wN[pos]);
        if (hasS) acc_add(&b, q->wS[pos]);
        if (hasW) acc_add(&b, q->wW[pos]);
        if (hasE) acc_add(&b, q->wE[pos]);
        t.b = b.v;
    }
    return t;
}

void orc_pde_alr4(float *X, const float *TRACE, const float *B, const float *wW, const float *wN, const float *wE,
                  const float *wS, int nrows, int ncols, int nframes, int iter, float omega, int order)
{
    float *cp, *dp;
    int it, k;
    if (nrows < 2 || ncols < 2 || !alloc_scratch(nrows, ncols, &cp, &dp))
        return;
    /* :308-327.  The reference loops the frames inside each of its six line functions; frames do
     * not interact, so looping them outside the two passes visits every frame in the same order. */
    for (it = 0; it < iter; it++) {
        for (k = 0; k < nframes; k++) {
            size_t o = (size_t)k * nrows * ncols;
            pde_ctx q = {X + o, TRACE + o, B + o, wW + o, NULL, wN + o, NULL, wE + o, NULL, wS + o, NULL, nrows, ncols};
            line_pass(X + o, nrows, ncols, 1, 0, ncols - 1, order, pde4_coef, &q, omega, cp, dp, 0);
        }
        for (k = 0; k < nframes; k++) {
            size_t o = (size_t)k * nrows * ncols;
            pde_ctx q = {X + o, TRACE + o, B + o, wW + o, NULL, wN + o, NULL, wE + o, NULL, wS + o, NULL, nrows, ncols};
            line_pass(X + o, nrows, ncols, 0, 0, nrows - 1, order, pde4_coef, &q, omega, cp, dp, 0);
        }
    }
    free(cp);
    free(dp);
}

static tri_t pde8_coef(const void *vctx, int i, int j, int vertical)
{
    const pde_ctx *q = (const pde_ctx *)vctx;
    int nrows = q->nrows;
    size_t pos = (size_t)j * nrows + i;
    size_t wpos = pos - nrows, epos = pos + nrows, npos = pos - 1, spos = pos + 1;
    int hasN = i > 0, hasS = i < nrows - 1, hasW = j > 0, hasE = j < q->ncols - 1;
    const float *X = q->X;
    tri_t t;
    float d;
    if (vertical) { /* interior columns only: W and E exist (pdeSolvers.c:1171-1173, :1195-1197, :1227-1228) */
        d = q->wW[pos] * X[wpos] + q->wE[pos] * X[epos];
        if (hasS) d += q->wSW[pos] * X[wpos + 1] + q->wSE[pos] * X[epos + 1];
        if (hasN) d += q->wNW[pos] * X[wpos - 1] + q->wNE[pos] * X[epos - 1];
        t.a = hasN ? -q->wN[pos] : 0.0f;
        t.c = hasS ? -q->wS[pos] : 0.0f;
    } else { /* interior rows only: N and S exist (:1309-1310, :1335-1337, :1364-1365) */
        d = q->wS[pos] * X[spos] + q->wN[pos] * X[npos];
        if (hasW) d += q->wSW[pos] * X[spos - nrows] + q->wNW[pos] * X[npos - nrows];
        if (hasE) d += q->wSE[pos] * X[spos + nrows] + q->wNE[pos] * X[npos + nrows];
        t.a = hasW ? -q->wW[pos] : 0.0f;
        t.c = hasE ? -q->wE[pos] : 0.0f;
    }
    if (!ORC_ISNAN(q->T[pos])) {
        t.b = q->T[pos];
        d += q->B[pos];
    } else { /* :1181-1182, as written: wNW twice, wNE never, all eight terms at every position */
        t.b = q->wN[pos] + q->wS[pos] + q->wW[pos] + q->wE[pos];
        t.b += q->wNW[pos] + q->wNW[pos] + q->wSW[pos] + q->wSE[pos];
    }
    t.d = d;
    return t;
}

void orc_pde_alr8(float *X, const float *TRACE, const float *B, const float *wW, const float *wNW, const float *wN,
                  const float *wNE, const float *wE, const float *wSE, const float *wS, const float *wSW, int nrows,
                  int ncols, int nframes, int iter, float omega, int order)
{
    float *cp, *dp;
    int k;
    (void)iter; /* pdeSolvers.c:362: `iterations = 1`, whatever the caller asked for */
    if (nrows < 3 || ncols < 3 || !alloc_scratch(nrows, ncols, &cp, &dp))
        return;
    for (k = 0; k < nframes; k++) { /* interior columns :1153, all their rows */
        size_t o = (size_t)k * nrows * ncols;
        pde_ctx q = {X + o, TRACE + o, B + o, wW + o, wNW + o, wN + o, wNE + o, wE + o, wSE + o, wS + o, wSW + o, nrows, ncols};
        line_pass(X + o, nrows, ncols, 1, 1, ncols - 2, order, pde8_coef, &q, omega, cp, dp, 0);
    }
    for (k = 0; k < nframes; k++) { /* interior rows :1290, all their columns */
        size_t o = (size_t)k * nrows * ncols;
        pde_ctx q = {X + o, TRACE + o, B + o, wW + o, wNW + o, wN + o, wNE + o, wE + o, wSE + o, wS + o, wSW + o, nrows, ncols};
        line_pass(X + o, nrows, ncols, 0, 1, nrows - 2, order, pde8_coef, &q, omega, cp, dp, 0);
    }
    free(cp);
    free(dp);
}
