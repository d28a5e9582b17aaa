/*
 * pdeip_oracle.c -- CPU oracle (plain C restatement) for the MEX-side stencil hot path.
 *
 * TEST INFRASTRUCTURE ONLY -- see pdeip_oracle.h.  PARITY UNPINNED: restated from the
 * reference's C sources read as text (file:line cited per function); the reference
 * itself cannot be built in this image (it needs MATLAB's mex.h / matrix.h).
 *
 * Build: gcc -O2 -ffp-contract=off -fno-fast-math -shared -fPIC (see Makefile).  Every
 * float expression below is written one operation per statement where the reference
 * fixes an association, so the compiler has nothing to reassociate or contract.
 */
#include "pdeip_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define ORC_ISNAN(x) ((x) != (x)) /* opticalflowSolvers.c:31-33 */

/* Border replicate: rows first, then columns (opticalflowSolvers.c:161-179). */
static void fill_borders(float *P, int nrows, int ncols)
{
    int i, j;
    for (j = 0; j < ncols; j++) {
        float *c = P + (size_t)j * nrows;
        c[0] = c[1];
        c[nrows - 1] = c[nrows - 2];
    }
    for (i = 0; i < nrows; i++) {
        P[i] = P[i + nrows];
        P[i + (size_t)(ncols - 1) * nrows] = P[i + (size_t)(ncols - 2) * nrows];
    }
}

/* bit 1 of `order`: parity of the global index of local column 0 (slab decomposition) */
#define ORC_CPAR(order) (((order) >> 1) & 1)

/* Visit the interior in the requested order: lexicographic (j outer, i inner) in one
 * pass, or ncolours passes where pass c touches the pixels whose colour is c. */
#define FOR_INTERIOR(order, ncolours, COLOUR_EXPR, ...)                               \
    do {                                                                              \
        int pass_, npass_ = (((order) & 1) == ORC_ORDER_LEX) ? 1 : (ncolours);        \
        for (pass_ = 0; pass_ < npass_; pass_++) {                                    \
            int i, j;                                                                 \
            for (j = 1; j < ncols - 1; j++)                                           \
                for (i = 1; i < nrows - 1; i++) {                                     \
                    if (((order) & 1) != ORC_ORDER_LEX && (COLOUR_EXPR) != pass_) continue; \
                    __VA_ARGS__                                                       \
                }                                                                     \
        }                                                                             \
    } while (0)

/* Divisor planes of the coupled (u,v) solvers, built by the reference during sweep 0
 * (opticalflowSolvers.c:111-127 and :606-622): 1/((wW+wE)+(wN+wS) [+ D unless NaN]). */
static void oflow_divisors(float *divU, float *divV, const float *Du, const float *Dv,
                           const float *wW, const float *wN, const float *wE, const float *wS,
                           int nrows, int ncols)
{
    int i, j;
    for (j = 1; j < ncols - 1; j++)
        for (i = 1; i < nrows - 1; i++) {
            size_t pos = (size_t)j * nrows + i;
            float t1 = wW[pos] + wE[pos];
            float t2 = wN[pos] + wS[pos];
            t1 += t2;
            divU[pos] = ORC_ISNAN(Du[pos]) ? 1.0f / t1 : 1.0f / (t1 + Du[pos]);
            divV[pos] = ORC_ISNAN(Dv[pos]) ? 1.0f / t1 : 1.0f / (t1 + Dv[pos]);
        }
}

/* One pixel of GS_SOR_elin4_2d (opticalflowSolvers.c:89-152), in place. */
static inline void elin4_pixel(float *U, float *V, const float *M, const float *Cu, const float *Cv, const float *divU,
                               const float *divV, const float *wW, const float *wN, const float *wE, const float *wS,
                               size_t pos, int nrows, float omega)
{
    float nbU, nbV, t1, t2, t3, Unew, Vnew;
    /* opticalflowSolvers.c:89-97 */
    nbU = U[pos - nrows] * wW[pos];
    t1 = U[pos + nrows] * wE[pos];
    nbU += t1;
    t2 = U[pos - 1] * wN[pos];
    t3 = U[pos + 1] * wS[pos];
    t2 += t3;
    nbU += t2;
    /* :100-108 */
    nbV = V[pos - nrows] * wW[pos];
    t1 = V[pos + nrows] * wE[pos];
    nbV += t1;
    t2 = V[pos - 1] * wN[pos];
    t3 = V[pos + 1] * wS[pos];
    t2 += t3;
    nbV += t2;
    /* :129-149 -- both new values use the pre-update U[pos], V[pos] */
    if (ORC_ISNAN(Cu[pos])) {
        Unew = nbU * divU[pos];
    } else {
        t1 = nbU + Cu[pos];
        t2 = M[pos] * V[pos];
        t1 = t1 - t2;
        Unew = t1 * divU[pos];
    }
    if (ORC_ISNAN(Cv[pos])) {
        Vnew = nbV * divV[pos];
    } else {
        t1 = nbV + Cv[pos];
        t2 = M[pos] * U[pos];
        t1 = t1 - t2;
        Vnew = t1 * divV[pos];
    }
    /* :151-152 */
    t1 = (1.0f - omega) * U[pos];
    t2 = omega * Unew;
    U[pos] = t1 + t2;
    t1 = (1.0f - omega) * V[pos];
    t2 = omega * Vnew;
    V[pos] = t1 + t2;
}

void orc_oflow_sor_elin4(float *U, float *V, const float *M, const float *Cu, const float *Cv,
                         const float *Du, const float *Dv, const float *wW, const float *wN,
                         const float *wE, const float *wS, int nrows, int ncols, int iter,
                         float omega, int order)
{
    size_t n = (size_t)nrows * ncols;
    float *divU, *divV;
    int it;
    if (iter <= 0) return;
    divU = (float *)malloc(n * sizeof(float));
    divV = (float *)malloc(n * sizeof(float));
    if (!divU || !divV) { free(divU); free(divV); return; }
    oflow_divisors(divU, divV, Du, Dv, wW, wN, wE, wS, nrows, ncols);

    for (it = 0; it < iter; it++) {
        FOR_INTERIOR(order, 2, ((i + j + ORC_CPAR(order)) & 1), {
            elin4_pixel(U, V, M, Cu, Cv, divU, divV, wW, wN, wE, wS, (size_t)j * nrows + i, nrows, omega);
        });
        fill_borders(U, nrows, ncols);
        fill_borders(V, nrows, ncols);
    }
    free(divU);
    free(divV);
}

/* The red-black ordering of orc_oflow_sor_elin4 (order = ORC_ORDER_COLOUR) on `nthreads` host threads: pixels of one
 * colour do not read each other, so the columns of a colour pass are relaxed concurrently -- same arithmetic, same
 * results bit for bit as the serial colour-ordered loop.  NOT in the reference (its flow solvers are single-threaded
 * lexicographic loops; its only OpenMP code is the level-set library): it exists as the reported all-core CPU
 * comparator of the GPU's red-black number (bench.py cpu_baseline.red_black_all_cores).
 * Returns the number of threads actually used (0: built without OpenMP, ran serially). */
int orc_oflow_sor_elin4_rb_omp(float *U, float *V, const float *M, const float *Cu, const float *Cv, const float *Du,
                               const float *Dv, const float *wW, const float *wN, const float *wE, const float *wS,
                               int nrows, int ncols, int iter, float omega, int nthreads)
{
    size_t n = (size_t)nrows * ncols;
    float *divU, *divV;
    int it, colour, used = 0;
    if (iter <= 0) return 0;
    divU = (float *)malloc(n * sizeof(float));
    divV = (float *)malloc(n * sizeof(float));
    if (!divU || !divV) { free(divU); free(divV); return -1; }
    oflow_divisors(divU, divV, Du, Dv, wW, wN, wE, wS, nrows, ncols);
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#else
    (void)nthreads;
#endif
    for (it = 0; it < iter; it++) {
        for (colour = 0; colour < 2; colour++) {
            int j;
#ifdef _OPENMP
#pragma omp parallel for schedule(static)
#endif
            for (j = 1; j < ncols - 1; j++) {
                int i;
#ifdef _OPENMP
                if (j == 1) used = omp_get_num_threads();
#endif
                for (i = 1 + ((1 + j + colour) & 1); i < nrows - 1; i += 2)
                    elin4_pixel(U, V, M, Cu, Cv, divU, divV, wW, wN, wE, wS, (size_t)j * nrows + i, nrows, omega);
            }
        }
        fill_borders(U, nrows, ncols);
        fill_borders(V, nrows, ncols);
    }
    free(divU);
    free(divV);
    return used;
}

/* Copies a [ncols][nrows] plane with the thread-to-column assignment of orc_oflow_sor_elin4_rb_omp, so that a destination
 * whose pages have not been touched yet ends up spread over the NUMA nodes the way the sweep will read it (first touch).
 * Only used to prepare the buffers of the all-core CPU comparator. */
void orc_plane_copy_omp(float *dst, const float *src, int nrows, int ncols, int nthreads)
{
    int j;
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#pragma omp parallel for schedule(static)
#else
    (void)nthreads;
#endif
    for (j = 0; j < ncols; j++) memcpy(dst + (size_t)j * nrows, src + (size_t)j * nrows, (size_t)nrows * sizeof(float));
}

/* Neighbourhood term of the late-linearization solvers (opticalflowSolvers.c:563-580):
 * ((dW+W-c)*wW + (dE+E-c)*wE) + ((dN+N-c)*wN + (dS+S-c)*wS). */
static float llin_neigh(const float *U, const float *dU, size_t pos, int nrows, const float *wW,
                        const float *wN, const float *wE, const float *wS)
{
    float a = dU[pos - nrows] + U[pos - nrows];
    float b = dU[pos + nrows] + U[pos + nrows];
    float c = dU[pos - 1] + U[pos - 1];
    float d = dU[pos + 1] + U[pos + 1];
    a -= U[pos];
    b -= U[pos];
    c -= U[pos];
    d -= U[pos];
    a *= wW[pos];
    b *= wE[pos];
    c *= wN[pos];
    d *= wS[pos];
    a += b;
    c += d;
    a += c;
    return a;
}

void orc_oflow_sor_llin4(const float *U, const float *V, float *dU, float *dV, const float *M,
                         const float *Cu, const float *Cv, const float *Du, const float *Dv,
                         const float *wW, const float *wN, const float *wE, const float *wS,
                         int nrows, int ncols, int iter, float omega, int order)
{
    size_t n = (size_t)nrows * ncols;
    float *divU, *divV;
    int it;
    if (iter <= 0) return;
    divU = (float *)malloc(n * sizeof(float));
    divV = (float *)malloc(n * sizeof(float));
    if (!divU || !divV) { free(divU); free(divV); return; }
    oflow_divisors(divU, divV, Du, Dv, wW, wN, wE, wS, nrows, ncols);

    for (it = 0; it < iter; it++) {
        FOR_INTERIOR(order, 2, ((i + j + ORC_CPAR(order)) & 1), {
            size_t pos = (size_t)j * nrows + i;
            float nbU = llin_neigh(U, dU, pos, nrows, wW, wN, wE, wS);
            float nbV = llin_neigh(V, dV, pos, nrows, wW, wN, wE, wS);
            float t1, t2, dUnew, dVnew;
            /* opticalflowSolvers.c:624-644 */
            if (ORC_ISNAN(Cu[pos])) {
                dUnew = nbU * divU[pos];
            } else {
                t1 = nbU + Cu[pos];
                t2 = M[pos] * dV[pos];
                t1 -= t2;
                dUnew = t1 * divU[pos];
            }
            if (ORC_ISNAN(Cv[pos])) {
                dVnew = nbV * divV[pos];
            } else {
                t1 = nbV + Cv[pos];
                t2 = M[pos] * dU[pos];
                t1 -= t2;
                dVnew = t1 * divV[pos];
            }
            /* :646-647 */
            t1 = (1.0f - omega) * dU[pos];
            t2 = omega * dUnew;
            dU[pos] = t1 + t2;
            t1 = (1.0f - omega) * dV[pos];
            t2 = omega * dVnew;
            dV[pos] = t1 + t2;
        });
        fill_borders(dU, nrows, ncols);
        fill_borders(dV, nrows, ncols);
    }
    free(divU);
    free(divV);
}

void orc_oflow_res_elin4(float *RU, float *RV, const float *U, const float *V, const float *M,
                         const float *Cu, const float *Cv, const float *Du, const float *Dv,
                         const float *wW, const float *wN, const float *wE, const float *wS,
                         int nrows, int ncols, int nframes)
{
    size_t n = (size_t)nrows * ncols;
    int i, j, k;
    for (k = 0; k < nframes; k++) {
        size_t fo = (size_t)k * n;
        for (j = 1; j < ncols - 1; j++)
            for (i = 1; i < nrows - 1; i++) {
                size_t pos = (size_t)j * nrows + i, os = pos + fo;
                float nbU, nbV, s, s2, t;
                /* opticalflowSolvers.c:320-333, strictly left to right */
                nbU = U[pos - nrows] * wW[pos];
                nbU = nbU + U[pos + nrows] * wE[pos];
                nbU = nbU + U[pos - 1] * wN[pos];
                nbU = nbU + U[pos + 1] * wS[pos];
                nbV = V[pos - nrows] * wW[pos];
                nbV = nbV + V[pos + nrows] * wE[pos];
                nbV = nbV + V[pos - 1] * wN[pos];
                nbV = nbV + V[pos + 1] * wS[pos];
                s = wW[pos] + wE[pos];
                s2 = wN[pos] + wS[pos];
                s += s2;
                /* :336-345 */
                if (!ORC_ISNAN(Cu[os])) {
                    t = Cu[os] - M[os] * V[pos];
                    t = t + nbU;
                    RU[os] = t - (Du[os] + s) * U[pos];
                } else {
                    RU[os] = nbU - s * U[pos];
                }
                if (!ORC_ISNAN(Cv[os])) {
                    t = Cv[os] - M[os] * U[pos];
                    t = t + nbV;
                    RV[os] = t - (Dv[os] + s) * V[pos];
                } else {
                    RV[os] = nbV - s * V[pos];
                }
            }
    }
    for (k = 0; k < nframes; k++) { /* :355-379 */
        fill_borders(RU + (size_t)k * n, nrows, ncols);
        fill_borders(RV + (size_t)k * n, nrows, ncols);
    }
}

void orc_oflow_lhs_elin4(float *AU, float *AV, const float *U, const float *V, const float *M,
                         const float *Du, const float *Dv, const float *wW, const float *wN,
                         const float *wE, const float *wS, int nrows, int ncols, int nframes)
{
    size_t n = (size_t)nrows * ncols;
    int i, j, k;
    for (k = 0; k < nframes; k++) {
        size_t fo = (size_t)k * n;
        for (j = 1; j < ncols - 1; j++)
            for (i = 1; i < nrows - 1; i++) {
                size_t pos = (size_t)j * nrows + i, os = pos + fo;
                float nbU, nbV, s, s2, t;
                /* opticalflowSolvers.c:436-449 */
                nbU = U[pos - nrows] * wW[pos];
                nbU = nbU + U[pos + nrows] * wE[pos];
                nbU = nbU + U[pos - 1] * wN[pos];
                nbU = nbU + U[pos + 1] * wS[pos];
                nbV = V[pos - nrows] * wW[pos];
                nbV = nbV + V[pos + nrows] * wE[pos];
                nbV = nbV + V[pos - 1] * wN[pos];
                nbV = nbV + V[pos + 1] * wS[pos];
                s = wW[pos] + wE[pos];
                s2 = wN[pos] + wS[pos];
                s += s2;
                /* :452-461 */
                if (!ORC_ISNAN(Du[os])) {
                    t = M[os] * V[pos] - nbU;
                    AU[os] = t + (Du[os] + s) * U[pos];
                } else {
                    AU[os] = -nbU + s * U[pos];
                }
                if (!ORC_ISNAN(Dv[os])) {
                    t = M[os] * U[pos] - nbV;
                    AV[os] = t + (Dv[os] + s) * V[pos];
                } else {
                    AV[os] = -nbV + s * V[pos];
                }
            }
    }
    for (k = 0; k < nframes; k++) { /* :471-494 */
        fill_borders(AU + (size_t)k * n, nrows, ncols);
        fill_borders(AV + (size_t)k * n, nrows, ncols);
    }
}

void orc_oflow_res_llin4(float *RU, float *RV, const float *U, const float *V, const float *dU,
                         const float *dV, const float *M, const float *Cu, const float *Cv,
                         const float *Du, const float *Dv, const float *wW, const float *wN,
                         const float *wE, const float *wS, int nrows, int ncols, int nframes)
{
    size_t n = (size_t)nrows * ncols;
    int i, j, k;
    for (k = 0; k < nframes; k++) {
        size_t fo = (size_t)k * n;
        for (j = 1; j < ncols - 1; j++)
            for (i = 1; i < nrows - 1; i++) {
                size_t pos = (size_t)j * nrows + i, os = pos + fo;
                float nbU = llin_neigh(U, dU, pos, nrows, wW, wN, wE, wS); /* :824-841 */
                float nbV = llin_neigh(V, dV, pos, nrows, wW, wN, wE, wS); /* :848-865 */
                float s = wW[pos] + wE[pos], s2 = wN[pos] + wS[pos], t;
                s += s2;
                /* :873-882 */
                if (!ORC_ISNAN(Cu[os])) {
                    t = Cu[os] - M[os] * dV[pos];
                    t = t + nbU;
                    RU[os] = t - (Du[os] + s) * dU[pos];
                } else {
                    RU[os] = nbU - s * dU[pos];
                }
                if (!ORC_ISNAN(Cv[os])) {
                    t = Cv[os] - M[os] * dU[pos];
                    t = t + nbV;
                    RV[os] = t - (Dv[os] + s) * dV[pos];
                } else {
                    RV[os] = nbV - s * dV[pos];
                }
            }
    }
    for (k = 0; k < nframes; k++) { /* :891-915 */
        size_t fo = (size_t)k * n;
        float *ru = RU + fo, *rv = RV + fo;
        for (j = 0; j < ncols; j++) {
            size_t p = (size_t)j * nrows;
            ru[p] = ru[p + 1];
            ru[p + nrows - 1] = ru[p + nrows - 2];
            rv[p] = rv[p + 1];
            rv[p + nrows - 1] = rv[p + nrows - 2];
        }
        for (i = 0; i < nrows; i++) {
            ru[i] = ru[i + nrows];
            ru[i + (size_t)(ncols - 1) * nrows] = ru[i + (size_t)(ncols - 2) * nrows];
            rv[i] = RV[i + nrows]; /* :912 reads frame 0 whatever k is (reference quirk) */
            rv[i + (size_t)(ncols - 1) * nrows] = rv[i + (size_t)(ncols - 2) * nrows];
        }
    }
}

void orc_oflow_lhs_llin4(float *AU, float *AV, const float *U, const float *V, const float *dU,
                         const float *dV, const float *M, const float *Du, const float *Dv,
                         const float *wW, const float *wN, const float *wE, const float *wS,
                         int nrows, int ncols, int nframes)
{
    size_t n = (size_t)nrows * ncols;
    /* `pos` as the triple loop leaves it: the last interior pixel of frame 0 (:961). */
    size_t stale = (size_t)(ncols - 2) * nrows + (size_t)(nrows - 2);
    int i, j, k;
    for (k = 0; k < nframes; k++) {
        size_t fo = (size_t)k * n;
        for (j = 1; j < ncols - 1; j++)
            for (i = 1; i < nrows - 1; i++) {
                size_t pos = (size_t)j * nrows + i, os = pos + fo;
                float nbU = llin_neigh(U, dU, pos, nrows, wW, wN, wE, wS); /* :979-996 */
                float nbV = llin_neigh(V, dV, pos, nrows, wW, wN, wE, wS); /* :1003-1020 */
                float s = wW[pos] + wE[pos], s2 = wN[pos] + wS[pos], t;
                s += s2;
                /* :1028-1037 */
                if (!ORC_ISNAN(Du[os])) {
                    t = M[os] * dV[pos] - nbU;
                    AU[os] = t + (Du[os] + s) * dU[pos];
                } else {
                    AU[os] = -nbU + s * dU[pos];
                }
                if (!ORC_ISNAN(Dv[os])) {
                    t = M[os] * dU[pos] - nbV;
                    AV[os] = t + (Dv[os] + s) * dV[pos];
                } else {
                    AV[os] = -nbV + s * dV[pos];
                }
            }
    }
    for (k = 0; k < nframes; k++) { /* :1046-1069 */
        size_t fo = (size_t)k * n;
        float *au = AU + fo, *av = AV + fo;
        for (j = 0; j < ncols; j++) {
            size_t p = (size_t)j * nrows;
            au[p] = au[p + 1];
            au[p + nrows - 1] = au[p + nrows - 2];
            av[p] = AV[stale + 1]; /* :1056 uses the stale, frame-less `pos` (reference quirk) */
            av[p + nrows - 1] = av[p + nrows - 2];
        }
        for (i = 0; i < nrows; i++) {
            au[i] = au[i + nrows];
            au[i + (size_t)(ncols - 1) * nrows] = au[i + (size_t)(ncols - 2) * nrows];
            av[i] = av[i + nrows];
            av[i + (size_t)(ncols - 1) * nrows] = av[i + (size_t)(ncols - 2) * nrows];
        }
    }
}

/* disparitySolvers.c:89-92: left-to-right sum in the order E, W, S, N. */
static float disp_neigh(const float *U, const float *dU, size_t pos, int nrows, const float *wW,
                        const float *wN, const float *wE, const float *wS)
{
    float e = ((U[pos + nrows] + dU[pos + nrows]) - U[pos]) * wE[pos];
    float w = ((U[pos - nrows] + dU[pos - nrows]) - U[pos]) * wW[pos];
    float s = ((U[pos + 1] + dU[pos + 1]) - U[pos]) * wS[pos];
    float nn = ((U[pos - 1] + dU[pos - 1]) - U[pos]) * wN[pos];
    float r = e + w;
    r = r + s;
    r = r + nn;
    return r;
}

void orc_disp_sor_llin4(const float *U, float *dU, const float *Cu, const float *Du,
                        const float *wW, const float *wN, const float *wE, const float *wS,
                        int nrows, int ncols, int iter, float omega, int order)
{
    size_t n = (size_t)nrows * ncols;
    float *div, *dividend;
    int it;
    if (iter <= 0) return;
    div = (float *)malloc(n * sizeof(float));
    dividend = (float *)malloc(n * sizeof(float));
    if (!div || !dividend) { free(div); free(dividend); return; }
    { /* disparitySolvers.c:94-113, built by the reference during sweep 0 */
        int i, j;
        for (j = 1; j < ncols - 1; j++)
            for (i = 1; i < nrows - 1; i++) {
                size_t pos = (size_t)j * nrows + i;
                float t;
                if (!ORC_ISNAN(Cu[pos])) {
                    dividend[pos] = Cu[pos];
                    t = Du[pos] + wE[pos];
                } else {
                    dividend[pos] = 0.0f;
                    t = wE[pos];
                }
                t = t + wW[pos];
                t = t + wS[pos];
                t = t + wN[pos];
                div[pos] = 1.0f / t;
            }
    }
    for (it = 0; it < iter; it++) {
        FOR_INTERIOR(order, 2, ((i + j + ORC_CPAR(order)) & 1), {
            size_t pos = (size_t)j * nrows + i;
            float nb = disp_neigh(U, dU, pos, nrows, wW, wN, wE, wS);
            /* :116-118 */
            float A = (1.0f - omega) * dU[pos];
            float B = omega * (nb + dividend[pos]);
            B = B * div[pos];
            dU[pos] = A + B;
        });
        fill_borders(dU, nrows, ncols);
    }
    free(div);
    free(dividend);
}

/* disparitySolvers.c:301-460 (GS_SOR_llinsym4_2d): the two fields never read each other; per pixel field 0 then
 * field 1.  Same neighbour sum and divisors as above, but omega multiplies the finished quotient (:425-429). */
void orc_disp_sor_llinsym4(const float *U0, float *dU0, const float *Cu0, const float *Du0, const float *wW0,
                           const float *wN0, const float *wE0, const float *wS0, const float *U1, float *dU1,
                           const float *Cu1, const float *Du1, const float *wW1, const float *wN1, const float *wE1,
                           const float *wS1, int nrows, int ncols, int iter, float omega, int order)
{
    size_t n = (size_t)nrows * ncols;
    const float *U[2], *Cu[2], *Du[2], *wW[2], *wN[2], *wE[2], *wS[2];
    float *dU[2], *div[2], *dividend[2];
    int it, k;
    U[0] = U0; U[1] = U1; dU[0] = dU0; dU[1] = dU1; Cu[0] = Cu0; Cu[1] = Cu1; Du[0] = Du0; Du[1] = Du1;
    wW[0] = wW0; wW[1] = wW1; wN[0] = wN0; wN[1] = wN1; wE[0] = wE0; wE[1] = wE1; wS[0] = wS0; wS[1] = wS1;
    if (iter <= 0) return;
    for (k = 0; k < 2; k++) {
        div[k] = (float *)calloc(n, sizeof(float));
        dividend[k] = (float *)calloc(n, sizeof(float)); /* stays 0 where Cu is NaN (:357, :386) */
    }
    if (div[0] && div[1] && dividend[0] && dividend[1]) {
        int i, j;
        for (k = 0; k < 2; k++) /* :381-412, built by the reference during sweep 0 */
            for (j = 1; j < ncols - 1; j++)
                for (i = 1; i < nrows - 1; i++) {
                    size_t pos = (size_t)j * nrows + i;
                    float t;
                    if (!ORC_ISNAN(Cu[k][pos])) {
                        dividend[k][pos] = Cu[k][pos];
                        t = Du[k][pos] + wE[k][pos];
                    } else {
                        t = wE[k][pos];
                    }
                    t = t + wW[k][pos];
                    t = t + wS[k][pos];
                    t = t + wN[k][pos];
                    div[k][pos] = 1.0f / t;
                }
        for (it = 0; it < iter; it++) {
            FOR_INTERIOR(order, 2, ((i + j + ORC_CPAR(order)) & 1), {
                size_t pos = (size_t)j * nrows + i;
                for (k = 0; k < 2; k++) {
                    float nb = disp_neigh(U[k], dU[k], pos, nrows, wW[k], wN[k], wE[k], wS[k]);
                    float approx = nb + dividend[k][pos];
                    float A, B;
                    approx = approx * div[k][pos];
                    A = (1.0f - omega) * dU[k][pos];
                    B = omega * approx;
                    dU[k][pos] = A + B;
                }
            });
            fill_borders(dU[0], nrows, ncols);
            fill_borders(dU[1], nrows, ncols);
        }
    }
    for (k = 0; k < 2; k++) {
        free(div[k]);
        free(dividend[k]);
    }
}

void orc_disp_res_llin4(float *RU, const float *U, const float *dU, const float *Cu,
                        const float *Du, const float *wW, const float *wN, const float *wE,
                        const float *wS, int nrows, int ncols)
{
    int i, j;
    for (j = 1; j < ncols - 1; j++)
        for (i = 1; i < nrows - 1; i++) {
            size_t pos = (size_t)j * nrows + i;
            float nb = disp_neigh(U, dU, pos, nrows, wW, wN, wE, wS); /* :250-253 */
            float t;
            if (!ORC_ISNAN(Cu[pos])) { /* :256-262 */
                t = Du[pos] + wW[pos];
                t = t + wN[pos];
                t = t + wE[pos];
                t = t + wS[pos];
                RU[pos] = (Cu[pos] + nb) - dU[pos] * t;
            } else { /* :265-268 */
                t = wW[pos] + wN[pos];
                t = t + wE[pos];
                t = t + wS[pos];
                RU[pos] = nb - dU[pos] * t;
            }
        }
    fill_borders(RU, nrows, ncols); /* :279-291 */
}

void orc_pde_sor4(float *X, const float *TRACE, const float *B, const float *wW, const float *wN,
                  const float *wE, const float *wS, int nrows, int ncols, int nframes, int iter,
                  float omega, int order)
{
    size_t n = (size_t)nrows * ncols, total = n * (size_t)nframes;
    float *inv, *bt;
    int it, k;
    if (iter <= 0) return;
    inv = (float *)malloc(total * sizeof(float));
    bt = (float *)malloc(total * sizeof(float));
    if (!inv || !bt) { free(inv); free(bt); return; }
    for (k = 0; k < nframes; k++) { /* pdeSolvers.c:99-115, built during sweep 0 */
        int i, j;
        for (j = 1; j < ncols - 1; j++)
            for (i = 1; i < nrows - 1; i++) {
                size_t pos = (size_t)j * nrows + i + (size_t)k * n;
                if (!ORC_ISNAN(TRACE[pos])) {
                    inv[pos] = 1.0f / TRACE[pos];
                    bt[pos] = B[pos];
                } else {
                    float t = wE[pos] + wW[pos];
                    t += wS[pos] + wN[pos];
                    inv[pos] = 1.0f / t;
                    bt[pos] = 0.0f;
                }
            }
    }
    for (it = 0; it < iter; it++)
        for (k = 0; k < nframes; k++) {
            size_t fo = (size_t)k * n;
            FOR_INTERIOR(order, 2, ((i + j + ORC_CPAR(order)) & 1), {
                size_t pos = (size_t)j * nrows + i + fo;
                float nb, t;
                /* :94-97 */
                nb = X[pos + nrows] * wE[pos] + X[pos - nrows] * wW[pos];
                nb += X[pos + 1] * wS[pos] + X[pos - 1] * wN[pos];
                /* :117-118 */
                X[pos] = (1.0f - omega) * X[pos];
                t = omega * (bt[pos] + nb);
                t = t * inv[pos];
                X[pos] += t;
            });
            fill_borders(X + fo, nrows, ncols); /* :127-140 */
        }
    free(inv);
    free(bt);
}

void orc_pde_sor8(float *X, const float *TRACE, const float *B, const float *wW, const float *wNW,
                  const float *wN, const float *wNE, const float *wE, const float *wSE,
                  const float *wS, const float *wSW, int nrows, int ncols, int nframes, int iter,
                  float omega, int order)
{
    size_t n = (size_t)nrows * ncols, total = n * (size_t)nframes;
    float *inv, *bt;
    int it, k;
    if (iter <= 0) return;
    inv = (float *)malloc(total * sizeof(float));
    bt = (float *)malloc(total * sizeof(float));
    if (!inv || !bt) { free(inv); free(bt); return; }
    for (k = 0; k < nframes; k++) { /* pdeSolvers.c:217-237 */
        int i, j;
        for (j = 1; j < ncols - 1; j++)
            for (i = 1; i < nrows - 1; i++) {
                size_t pos = (size_t)j * nrows + i + (size_t)k * n;
                if (!ORC_ISNAN(TRACE[pos])) {
                    inv[pos] = 1.0f / TRACE[pos];
                    bt[pos] = B[pos];
                } else {
                    float t = wE[pos] + wW[pos];
                    t += wS[pos] + wN[pos];
                    t += wSW[pos] + wNW[pos];
                    t += wSE[pos] + wNE[pos];
                    inv[pos] = 1.0f / t;
                    bt[pos] = 0.0f;
                }
            }
    }
    for (it = 0; it < iter; it++)
        for (k = 0; k < nframes; k++) {
            size_t fo = (size_t)k * n;
            FOR_INTERIOR(order, 4, ((i & 1) | (((j + ORC_CPAR(order)) & 1) << 1)), {
                size_t pos = (size_t)j * nrows + i + fo;
                size_t wpos = pos - nrows, epos = pos + nrows;
                float nb, t;
                /* :208-215 */
                nb = X[epos] * wE[pos] + X[wpos] * wW[pos];
                nb += X[pos + 1] * wS[pos] + X[pos - 1] * wN[pos];
                nb += X[wpos + 1] * wSW[pos] + X[wpos - 1] * wNW[pos];
                nb += X[epos + 1] * wSE[pos] + X[epos - 1] * wNE[pos];
                /* :239-240 */
                X[pos] = (1.0f - omega) * X[pos];
                t = omega * (bt[pos] + nb);
                t = t * inv[pos];
                X[pos] += t;
            });
            fill_borders(X + fo, nrows, ncols); /* :249-262 */
        }
    free(inv);
    free(bt);
}

/* 1/sqrt(t+eps) as the reference writes it: float add, double sqrt, cast, float divide
 * (imageDiffusionWeights.c:156). */
static float inv_sqrt_eps(float t, float eps)
{
    float s = t + eps;
    float r = (float)sqrt((double)s);
    return 1.0f / r;
}

void orc_diffweights6(float *wW, float *wN, float *wE, float *wS, const float *D, int nrows,
                      int ncols, int nframes, float eps)
{
    size_t n = (size_t)nrows * ncols, total = n * (size_t)nframes;
    float *ver = (float *)malloc(total * sizeof(float));
    float *hor = (float *)malloc(total * sizeof(float));
    float *temp = (float *)malloc(total * sizeof(float));
    int i, j, k;
    if (!ver || !hor || !temp) { free(ver); free(hor); free(temp); return; }

    for (k = 0; k < nframes; k++) { /* Dver :44-69, Dhor :85-108 (replicate ends) */
        const float *d = D + (size_t)k * n;
        float *v = ver + (size_t)k * n, *h = hor + (size_t)k * n;
        for (j = 0; j < ncols; j++)
            for (i = 0; i < nrows; i++) {
                size_t pos = (size_t)j * nrows + i;
                float up = d[i > 0 ? pos - 1 : pos];
                float dn = d[i < nrows - 1 ? pos + 1 : pos];
                float lf = d[j > 0 ? pos - nrows : pos];
                float rt = d[j < ncols - 1 ? pos + nrows : pos];
                float A = 0.25f * up, Bv = -0.25f * dn;
                v[pos] = A + Bv;
                A = 0.25f * lf;
                Bv = -0.25f * rt;
                h[pos] = A + Bv;
            }
    }
    memset(wW, 0, n * sizeof(float));
    memset(wN, 0, n * sizeof(float));
    memset(wE, 0, n * sizeof(float));
    memset(wS, 0, n * sizeof(float));

    /* Calc_wW :123-161: columns 1..ncols-1, max over frames folded into frame 0 */
    for (k = 0; k < nframes; k++) {
        const float *d = D + (size_t)k * n;
        const float *v = ver + (size_t)k * n;
        float *t = temp + (size_t)k * n;
        for (i = 0; i < nrows; i++)
            for (j = 1; j < ncols; j++) {
                size_t pos = (size_t)j * nrows + i;
                float A = d[pos] - d[pos - nrows];
                float Bv = v[pos] + v[pos - nrows];
                A = A * A;
                Bv = Bv * Bv;
                t[pos] = A + Bv;
                if (k >= 1 && t[pos] > temp[pos]) temp[pos] = t[pos];
            }
    }
    for (i = 0; i < nrows; i++)
        for (j = 1; j < ncols; j++) {
            size_t pos = (size_t)j * nrows + i;
            wW[pos] = inv_sqrt_eps(temp[pos], eps);
        }
    /* Calc_wN :177-213: rows 1..nrows-1 */
    for (k = 0; k < nframes; k++) {
        const float *d = D + (size_t)k * n;
        const float *h = hor + (size_t)k * n;
        float *t = temp + (size_t)k * n;
        for (j = 0; j < ncols; j++)
            for (i = 1; i < nrows; i++) {
                size_t pos = (size_t)j * nrows + i;
                float A = d[pos] - d[pos - 1];
                float Bv = h[pos] + h[pos - 1];
                A = A * A;
                Bv = Bv * Bv;
                t[pos] = A + Bv;
                if (k >= 1 && t[pos] > temp[pos]) temp[pos] = t[pos];
            }
    }
    for (j = 0; j < ncols; j++)
        for (i = 1; i < nrows; i++) {
            size_t pos = (size_t)j * nrows + i;
            wN[pos] = inv_sqrt_eps(temp[pos], eps);
        }
    /* Calc_wE :236-273: columns 0..ncols-2 */
    for (k = 0; k < nframes; k++) {
        const float *d = D + (size_t)k * n;
        const float *v = ver + (size_t)k * n;
        float *t = temp + (size_t)k * n;
        for (i = 0; i < nrows; i++)
            for (j = 0; j < ncols - 1; j++) {
                size_t pos = (size_t)j * nrows + i;
                float A = d[pos] - d[pos + nrows];
                float Bv = v[pos] + v[pos + nrows];
                A = A * A;
                Bv = Bv * Bv;
                t[pos] = A + Bv;
                if (k >= 1 && t[pos] > temp[pos]) temp[pos] = t[pos];
            }
    }
    for (i = 0; i < nrows; i++)
        for (j = 0; j < ncols - 1; j++) {
            size_t pos = (size_t)j * nrows + i;
            wE[pos] = inv_sqrt_eps(temp[pos], eps);
        }
    /* Calc_wS :289-326: rows 0..nrows-2 */
    for (k = 0; k < nframes; k++) {
        const float *d = D + (size_t)k * n;
        const float *h = hor + (size_t)k * n;
        float *t = temp + (size_t)k * n;
        for (j = 0; j < ncols; j++)
            for (i = 0; i < nrows - 1; i++) {
                size_t pos = (size_t)j * nrows + i;
                float A = d[pos] - d[pos + 1];
                float Bv = h[pos] + h[pos + 1];
                A = A * A;
                Bv = Bv * Bv;
                t[pos] = A + Bv;
                if (k >= 1 && t[pos] > temp[pos]) temp[pos] = t[pos];
            }
    }
    for (j = 0; j < ncols; j++)
        for (i = 0; i < nrows - 1; i++) {
            size_t pos = (size_t)j * nrows + i;
            wS[pos] = inv_sqrt_eps(temp[pos], eps);
        }
    free(ver);
    free(hor);
    free(temp);
}

void orc_warp_bilinear(float *Iout, const float *Iin, const float *X, const float *Y, int nrows,
                       int ncols, int nframes)
{
    size_t n = (size_t)nrows * ncols;
    int i, j, k;
    for (j = 0; j < ncols; j++)
        for (i = 0; i < nrows; i++) {
            size_t pos = (size_t)j * nrows + i;
            /* imageInterpolation.c:82-86.  The reference casts floor() to unsigned and lets
             * negative values wrap out of range; that is stated here as an explicit range
             * test.  Non-finite coordinates are out of range (the reference's cast is
             * undefined for them). */
            float xm = X[pos] - 1.0f, ym = Y[pos] - 1.0f;
            double fx = floor((double)xm), fy = floor((double)ym);
            if (fx >= 0.0 && fx < (double)ncols && fy >= 0.0 && fy < (double)nrows) {
                unsigned x = (unsigned)fx, y = (unsigned)fy;
                /* :89-96 */
                float xf = xm - (float)x, yf = ym - (float)y;
                float w00 = (1.0f - xf) * (1.0f - yf);
                float w10 = xf * (1.0f - yf);
                float w01 = (1.0f - xf) * yf;
                float w11 = xf * yf;
                for (k = 0; k < nframes; k++) { /* :99-124 */
                    size_t fo = (size_t)k * n;
                    size_t p00 = fo + (size_t)nrows * x + y;
                    size_t p10 = p00, p01 = p00, p11 = p00;
                    float r;
                    if (x < (unsigned)ncols - 1) p10 += nrows;
                    if (y < (unsigned)nrows - 1) p01 += 1;
                    if (x < (unsigned)ncols - 1 && y < (unsigned)nrows - 1) p11 = p11 + nrows + 1;
                    r = w00 * Iin[p00];
                    r = r + w10 * Iin[p10];
                    r = r + w01 * Iin[p01];
                    r = r + w11 * Iin[p11];
                    Iout[fo + pos] = r;
                }
            } else {
                for (k = 0; k < nframes; k++) Iout[(size_t)k * n + pos] = NAN; /* :129-135 */
            }
        }
}

/* ---- Simoncelli derivatives (next row f1) -------------------------------------------------------- */

static const float SIM_SMOOTH[5] = {0.037659f, 0.249724f, 0.439911f, 0.249724f, 0.037659f};  /* FstDerivatives5.c:59 */
static const float SIM_D1[5] = {-0.104550f, -0.292315f, 0.0f, 0.292315f, 0.104550f};          /* :60 */
static const float SIM_D2[5] = {0.232905f, 0.002668f, -0.471147f, 0.002668f, 0.232905f};      /* SndDerivatives5.c:67 */
static const float SIM_DT[2] = {0.50f, -0.50f};                                               /* :61 / :68 */

static int clampi(int x, int hi) { return x < 0 ? 0 : (x > hi ? hi : x); }

/* VerticalConvWO5 (imageDerivatives.c:66-119): 5-tap correlation along the rows of each column, the two
 * rows past either end replicate the end row; five products summed left to right. */
static void conv_vertical5(float *out, const float *in, const float *op, int nrows, int ncols)
{
    int i, j, k;
    for (j = 0; j < ncols; j++)
        for (i = 0; i < nrows; i++) {
            const float *c = in + (size_t)j * nrows;
            float t[5], r;
            for (k = 0; k < 5; k++) t[k] = c[clampi(i - 2 + k, nrows - 1)] * op[k];
            r = t[0] + t[1];
            r = r + t[2];
            r = r + t[3];
            r = r + t[4];
            out[(size_t)j * nrows + i] = r;
        }
}

/* HorizontalConvWO5 (imageDerivatives.c:125-211): the same along the columns of each row. */
static void conv_horizontal5(float *out, const float *in, const float *op, int nrows, int ncols)
{
    int i, j, k;
    for (i = 0; i < nrows; i++)
        for (j = 0; j < ncols; j++) {
            float t[5], r;
            for (k = 0; k < 5; k++) t[k] = in[(size_t)clampi(j - 2 + k, ncols - 1) * nrows + i] * op[k];
            r = t[0] + t[1];
            r = r + t[2];
            r = r + t[3];
            r = r + t[4];
            out[(size_t)j * nrows + i] = r;
        }
}

/* TemporalConvWO2 (imageDerivatives.c:44-60) */
static void conv_temporal2(float *out, const float *a, const float *b, const float *op, size_t n)
{
    size_t p;
    for (p = 0; p < n; p++) out[p] = a[p] * op[0] + b[p] * op[1];
}

void orc_fst_derivatives5(float *Idt, float *Idx, float *Idy, const float *It0, const float *It1,
                          int nrows, int ncols, int nframes)
{
    size_t n = (size_t)nrows * ncols;
    float *temp = (float *)malloc(n * sizeof(float));
    int k;
    if (!temp) return;
    for (k = 0; k < nframes; k++) { /* imageDerivatives.c:369-385 */
        size_t o = (size_t)k * n;
        conv_temporal2(Idt + o, It0 + o, It1 + o, SIM_DT, n);
        conv_vertical5(temp, It1 + o, SIM_SMOOTH, nrows, ncols);
        conv_horizontal5(Idx + o, temp, SIM_D1, nrows, ncols);
        conv_horizontal5(temp, It1 + o, SIM_SMOOTH, nrows, ncols);
        conv_vertical5(Idy + o, temp, SIM_D1, nrows, ncols);
    }
    free(temp);
}

void orc_snd_derivatives5(float *Idxt, float *Idyt, float *Idxx, float *Idyy, float *Idxy,
                          const float *It0, const float *It1, int nrows, int ncols, int nframes)
{
    size_t n = (size_t)nrows * ncols;
    float *t1 = (float *)malloc(n * sizeof(float)), *t2 = (float *)malloc(n * sizeof(float));
    int k;
    if (!t1 || !t2) { free(t1); free(t2); return; }
    for (k = 0; k < nframes; k++) { /* imageDerivatives.c:454-480 */
        size_t o = (size_t)k * n;
        conv_vertical5(t2, It0 + o, SIM_SMOOTH, nrows, ncols);
        conv_horizontal5(t1, t2, SIM_D1, nrows, ncols);
        conv_vertical5(Idxt + o, It1 + o, SIM_SMOOTH, nrows, ncols);
        conv_horizontal5(t2, Idxt + o, SIM_D1, nrows, ncols);
        conv_temporal2(Idxt + o, t1, t2, SIM_DT, n);

        conv_horizontal5(t2, It0 + o, SIM_SMOOTH, nrows, ncols);
        conv_vertical5(t1, t2, SIM_D1, nrows, ncols);
        conv_horizontal5(Idyt + o, It1 + o, SIM_SMOOTH, nrows, ncols);
        conv_vertical5(t2, Idyt + o, SIM_D1, nrows, ncols);
        conv_temporal2(Idyt + o, t1, t2, SIM_DT, n);

        conv_vertical5(t1, It1 + o, SIM_SMOOTH, nrows, ncols);
        conv_horizontal5(Idxx + o, t1, SIM_D2, nrows, ncols);
        conv_horizontal5(t1, It1 + o, SIM_SMOOTH, nrows, ncols);
        conv_vertical5(Idyy + o, t1, SIM_D2, nrows, ncols);
        conv_horizontal5(t1, It1 + o, SIM_D1, nrows, ncols);
        conv_vertical5(Idxy + o, t1, SIM_D1, nrows, ncols);
    }
    free(t1);
    free(t2);
}
