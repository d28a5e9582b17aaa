// pdeip_line.hip -- libpdeip.so: alternating line relaxation (solver = 2): launch logic and the *_dev entry points.
//
// Build (build.py): hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -c, one object per translation unit.
// -ffp-contract=off is part of the parity contract: the reference is plain C built without FMA.
#include "pdeip_ctx.hpp"

#include "pdeip_alr.hpp"

using namespace pdeip;

// ------------------------------------------------------------------------------------------------
// alternating line relaxation (solver 2): pdeip_alr.hpp
// ------------------------------------------------------------------------------------------------
// exact-order line relaxation holds a line as one float4 per element: in LDS up to 10 240 elements (160 KB), in a global scratch
// buffer beyond (k_alr_lex<.., GL = true>: correct, slow)
static int check_alr_line(const char *, int, int, int) { return PDEIP_OK; }

// Workspace of one call: per (chain, direction) the cp and divisor planes (k_alr_zebra3<ZB_FACTOR>).
struct AlrFactors {
    float *cp[2][2], *dv[2][2]; // [chain][vertical ? 0 : 1]
};

template <class Mdl, bool VERT, int MODE>
static int zebra3_launch(hipStream_t s, const typename Mdl::Ctx &q, float *x, float *cp, float *dv, float *dp, int nrows, int ncols, int nframes,
                         int first, int lastc, int lstep, float omega)
{
    RC(ensure_lds(reinterpret_cast<const void *>(&k_alr_zebra3<Mdl, VERT, MODE>), Z3_LDS_BYTES));
    const int count = (lastc - first) / lstep + 1;
    hipLaunchKernelGGL((k_alr_zebra3<Mdl, VERT, MODE>), dim3((unsigned)((count + ZB_LW - 1) / ZB_LW), (unsigned)nframes), dim3(ZB_THREADS),
                       Z3_LDS_BYTES, s, q, x, cp, dv, dp, nrows, ncols, (size_t)nrows * ncols, first, lastc, lstep, omega);
    tls.last_launches++;
    HIPCHK(hipGetLastError());
    return PDEIP_OK;
}

// cp and divisor planes of every (field, direction): the part of the Thomas recurrence that depends on the
// coefficient planes only, once per call (pdeip_alr.hpp).  Column planes from q, row planes from the transposed qt.
template <class Mdl>
static int alr_factor(hipStream_t s, const typename Mdl::Ctx *q, const typename Mdl::Ctx *qt, int nch, int nrows, int ncols, int nframes,
                      AlrFactors *f)
{
    const size_t fs = (size_t)nrows * ncols, plane = fs * nframes;
    float *base;
    RC(ws_get(WS_ALR, plane * 8 * sizeof(float), &base));
    const int lo = Mdl::INTERIOR_LINES ? 1 : 0;
    for (int c = 0; c < nch; c++)
        for (int d = 0; d < 2; d++) {
            f->cp[c][d] = base + plane * (size_t)((c * 2 + d) * 2);
            f->dv[c][d] = f->cp[c][d] + plane;
        }
    if (nch == 2) { // both fields of a coupled solver in one launch per direction
        for (int d = 0; d < 2; d++) {
            const int hi = (d == 0 ? ncols : nrows) - 1 - lo, count = hi - lo + 1;
            const dim3 grid((unsigned)((count + ZB_LW - 1) / ZB_LW), (unsigned)nframes, 2);
            if (d == 0)
                hipLaunchKernelGGL((k_alr_factor_pair<Mdl, true>), grid, dim3(ZB_THREADS), Z3_LDS_BYTES, s, q[0], q[1], f->cp[0][0], f->dv[0][0],
                                   f->cp[1][0], f->dv[1][0], nrows, ncols, fs, lo, hi);
            else
                hipLaunchKernelGGL((k_alr_factor_pair<Mdl, false>), grid, dim3(ZB_THREADS), Z3_LDS_BYTES, s, qt[0], qt[1], f->cp[0][1], f->dv[0][1],
                                   f->cp[1][1], f->dv[1][1], nrows, ncols, fs, lo, hi);
            tls.last_launches++;
        }
        HIPCHK(hipGetLastError());
        return PDEIP_OK;
    }
    for (int c = 0; c < nch; c++)
        for (int d = 0; d < 2; d++) {
            const int hi = (d == 0 ? ncols : nrows) - 1 - lo;
            if (d == 0) RC((zebra3_launch<Mdl, true, ZB_FACTOR>(s, q[c], nullptr, f->cp[c][d], f->dv[c][d], nullptr, nrows, ncols, nframes, lo, hi, 1, 0.0f)));
            else RC((zebra3_launch<Mdl, false, ZB_FACTOR>(s, qt[c], nullptr, f->cp[c][d], f->dv[c][d], nullptr, nrows, ncols, nframes, lo, hi, 1, 0.0f)));
        }
    HIPCHK(hipGetLastError());
    return PDEIP_OK;
}

// One direction, reference line order, for the `nch` fields in `order` (chain 1 trails chain 0 by a line).
template <class Mdl>
static int alr_lex_pass(hipStream_t s, const typename Mdl::Ctx *q, float *const *x, const AlrFactors &f, const int *order,
                        int nch, int nrows, int ncols, int nframes, bool vertical, float omega)
{
    const int lo = Mdl::INTERIOR_LINES ? 1 : 0;
    const int hi = (vertical ? ncols : nrows) - 1 - lo;
    const int n = vertical ? nrows : ncols;
    const size_t fs = (size_t)nrows * ncols;
    const int d = vertical ? 0 : 1;
    const size_t line_bytes = (size_t)n * sizeof(float4);
    if (line_bytes > 160 * 1024) { // a line longer than LDS holds: the line buffer in global memory, one chain per launch
        float *g = nullptr;
        RC(ws_get(WS_LEX, line_bytes * nframes, &g));
        for (int c = 0; c < nch; c++) {
            AlrChains<Mdl, 1> ch;
            ch.c[0] = AlrChain<Mdl>{q[order[c]], x[order[c]], f.cp[order[c]][d], f.dv[order[c]][d]};
            if (vertical) hipLaunchKernelGGL((k_alr_lex<Mdl, 1, true, true>), dim3((unsigned)nframes), dim3(ALR_LEX_THREADS), 0, s, ch, nrows, ncols, fs, lo, hi, omega, reinterpret_cast<float4 *>(g));
            else hipLaunchKernelGGL((k_alr_lex<Mdl, 1, false, true>), dim3((unsigned)nframes), dim3(ALR_LEX_THREADS), 0, s, ch, nrows, ncols, fs, lo, hi, omega, reinterpret_cast<float4 *>(g));
            tls.last_launches++;
        }
        HIPCHK(hipGetLastError());
        return PDEIP_OK;
    }
    if (nch == 2 && 2 * line_bytes <= 160 * 1024) {
        AlrChains<Mdl, 2> ch;
        for (int c = 0; c < 2; c++) ch.c[c] = AlrChain<Mdl>{q[order[c]], x[order[c]], f.cp[order[c]][d], f.dv[order[c]][d]};
        if (2 * line_bytes > 64 * 1024)
            RC(ensure_lds(vertical ? reinterpret_cast<const void *>(&k_alr_lex<Mdl, 2, true>) : reinterpret_cast<const void *>(&k_alr_lex<Mdl, 2, false>), 2 * line_bytes));
        if (vertical) hipLaunchKernelGGL((k_alr_lex<Mdl, 2, true>), dim3((unsigned)nframes), dim3(ALR_LEX_THREADS), 2 * line_bytes, s, ch, nrows, ncols, fs, lo, hi, omega);
        else hipLaunchKernelGGL((k_alr_lex<Mdl, 2, false>), dim3((unsigned)nframes), dim3(ALR_LEX_THREADS), 2 * line_bytes, s, ch, nrows, ncols, fs, lo, hi, omega);
        tls.last_launches++;
    } else {
        if (line_bytes > 64 * 1024)
            RC(ensure_lds(vertical ? reinterpret_cast<const void *>(&k_alr_lex<Mdl, 1, true>) : reinterpret_cast<const void *>(&k_alr_lex<Mdl, 1, false>), line_bytes));
        for (int c = 0; c < nch; c++) {
            AlrChains<Mdl, 1> ch;
            ch.c[0] = AlrChain<Mdl>{q[order[c]], x[order[c]], f.cp[order[c]][d], f.dv[order[c]][d]};
            if (vertical) hipLaunchKernelGGL((k_alr_lex<Mdl, 1, true>), dim3((unsigned)nframes), dim3(ALR_LEX_THREADS), line_bytes, s, ch, nrows, ncols, fs, lo, hi, omega);
            else hipLaunchKernelGGL((k_alr_lex<Mdl, 1, false>), dim3((unsigned)nframes), dim3(ALR_LEX_THREADS), line_bytes, s, ch, nrows, ncols, fs, lo, hi, omega);
            tls.last_launches++;
        }
    }
    HIPCHK(hipGetLastError());
    return PDEIP_OK;
}

// One direction of one field in zebra order: even lines, then odd lines: k_alr_zebra3 with the per-call factor
// planes (cpf, dvf); the others: one lane per line (k_alr_zebra).
template <class Mdl>
static int alr_zebra_pass(hipStream_t s, const typename Mdl::Ctx &q, float *x, const float *cpf, const float *dvf, int nrows, int ncols,
                          int nframes, bool vertical, float omega)
{
    const int lo = Mdl::INTERIOR_LINES ? 1 : 0;
    const int hi = (vertical ? ncols : nrows) - 1 - lo;
    const size_t fs = (size_t)nrows * ncols;
    float *cp, *dp;
    RC(ws_get(WS_AUX0, fs * nframes * sizeof(float), &cp));
    RC(ws_get(WS_AUX1, fs * nframes * sizeof(float), &dp));
    for (int colour = 0; colour < 2; colour++) {
        const int first = lo + (((lo & 1) != colour) ? 1 : 0);
        if (first > hi) continue;
        const int lastc = hi - (((hi & 1) != colour) ? 1 : 0);
        {
            if (cpf) {
                if (vertical) RC((zebra3_launch<Mdl, true, ZB_APPLY>(s, q, x, const_cast<float *>(cpf), const_cast<float *>(dvf), dp, nrows, ncols, nframes, first, lastc, 2, omega)));
                else RC((zebra3_launch<Mdl, false, ZB_APPLY>(s, q, x, const_cast<float *>(cpf), const_cast<float *>(dvf), dp, nrows, ncols, nframes, first, lastc, 2, omega)));
                continue;
            }
        }
        const int count = (hi - first) / 2 + 1;
        const dim3 grid((unsigned)((count + 63) / 64), (unsigned)nframes);
        if (vertical) hipLaunchKernelGGL((k_alr_zebra<Mdl, true>), grid, dim3(64), 0, s, q, x, cp, dp, nrows, ncols, fs, lo, hi, colour, omega);
        else hipLaunchKernelGGL((k_alr_zebra<Mdl, false>), grid, dim3(64), 0, s, q, x, cp, dp, nrows, ncols, fs, lo, hi, colour, omega);
        tls.last_launches++;
    }
    HIPCHK(hipGetLastError());
    return PDEIP_OK;
}

// Both fields of a coupled solver, one colour per launch (k_alr_zebra3_pair): field a first, then field b, as the per-field passes
// would run them.  PDEIP_ALR_PAIR=0: one launch per field and colour.
template <class Mdl>
static int alr_zebra_pass_pair(hipStream_t s, const typename Mdl::Ctx &qa, float *xa, const float *cpa, const float *dva, const typename Mdl::Ctx &qb,
                               float *xb, const float *cpb, const float *dvb, int nrows, int ncols, int nframes, bool vertical, float omega)
{
    const int lo = Mdl::INTERIOR_LINES ? 1 : 0;
    const int hi = (vertical ? ncols : nrows) - 1 - lo;
    const size_t fs = (size_t)nrows * ncols;
    float *dp;
    RC(ws_get(WS_AUX1, fs * nframes * sizeof(float), &dp));
    if (vertical) RC(ensure_lds(reinterpret_cast<const void *>(&k_alr_zebra3_pair<Mdl, true, ZB_APPLY>), Z3_LDS_BYTES));
    else RC(ensure_lds(reinterpret_cast<const void *>(&k_alr_zebra3_pair<Mdl, false, ZB_APPLY>), Z3_LDS_BYTES));
    for (int colour = 0; colour < 2; colour++) {
        const int first = lo + (((lo & 1) != colour) ? 1 : 0);
        if (first > hi) continue;
        const int lastc = hi - (((hi & 1) != colour) ? 1 : 0);
        const int count = (lastc - first) / 2 + 1;
        const dim3 grid((unsigned)((count + ZB_LW - 1) / ZB_LW), (unsigned)nframes);
        if (vertical)
            hipLaunchKernelGGL((k_alr_zebra3_pair<Mdl, true, ZB_APPLY>), grid, dim3(ZB_THREADS), Z3_LDS_BYTES, s, qa, xa, const_cast<float *>(cpa), const_cast<float *>(dva), qb, xb,
                               const_cast<float *>(cpb), const_cast<float *>(dvb), dp, nrows, ncols, fs, first, lastc, 2, omega);
        else
            hipLaunchKernelGGL((k_alr_zebra3_pair<Mdl, false, ZB_APPLY>), grid, dim3(ZB_THREADS), Z3_LDS_BYTES, s, qa, xa, const_cast<float *>(cpa), const_cast<float *>(dva), qb, xb,
                               const_cast<float *>(cpb), const_cast<float *>(dvb), dp, nrows, ncols, fs, first, lastc, 2, omega);
        tls.last_launches++;
    }
    HIPCHK(hipGetLastError());
    return PDEIP_OK;
}

// The row passes run on transposed copies of every plane (pdeip_alr.hpp).  A model's Ctx is a plain
// struct of plane pointers; each distinct plane gets one transposed twin in the WS_ALR_T workspace.  The
// coefficient planes are transposed once per call, the iterate planes around every row pass.
struct AlrTwin {
    static constexpr int MAXP = 40;
    const float *orig[MAXP];
    float *twin[MAXP];
    int count = 0;
    float *find(const float *p) const
    {
        for (int k = 0; k < count; k++)
            if (orig[k] == p) return twin[k];
        return nullptr;
    }
};

static int alr_transpose(hipStream_t s, float *out, const float *in, int na, int nb, int nframes)
{
    hipLaunchKernelGGL(k_alr_transpose, dim3((unsigned)((na + 31) / 32), (unsigned)((nb + 31) / 32), (unsigned)nframes), dim3(256), 0, s, out, in, na, nb);
    tls.last_launches++;
    HIPCHK(hipGetLastError());
    return PDEIP_OK;
}

// up to ALR_TB_MAX planes per launch
static int alr_transpose_many(hipStream_t s, float *const *out, const float *const *in, int count, int na, int nb, int nframes)
{
    for (int k0 = 0; k0 < count; k0 += ALR_TB_MAX) {
        AlrTransposeBatch B{};
        const int m = count - k0 < ALR_TB_MAX ? count - k0 : ALR_TB_MAX;
        for (int k = 0; k < m; k++) {
            B.out[k] = out[k0 + k];
            B.in[k] = in[k0 + k];
        }
        hipLaunchKernelGGL(k_alr_transpose_batch, dim3((unsigned)((na + 31) / 32), (unsigned)((nb + 31) / 32), (unsigned)(m * nframes)), dim3(256), 0,
                           s, B, na, nb, nframes);
        tls.last_launches++;
    }
    HIPCHK(hipGetLastError());
    return PDEIP_OK;
}

template <class Ctx>
static int alr_make_twins(hipStream_t s, const Ctx *q, Ctx *qt, int nch, float *const *x, float **xt, int nrows, int ncols, int nframes,
                          AlrTwin *tw, const float **defer_in = nullptr, float **defer_out = nullptr, int *defer_n = nullptr)
{
    static_assert(sizeof(Ctx) % sizeof(float *) == 0, "a line-relaxation context is a struct of plane pointers");
    constexpr int NP = (int)(sizeof(Ctx) / sizeof(float *));
    const size_t plane = (size_t)nrows * ncols * nframes;
    const float *ptrs[2][NP];
    for (int c = 0; c < nch; c++) {
        memcpy(ptrs[c], &q[c], sizeof(Ctx));
        for (int k = 0; k < NP; k++)
            if (ptrs[c][k] && !tw->find(ptrs[c][k])) {
                if (tw->count == AlrTwin::MAXP) return set_err(PDEIP_ERR_ARG, "line relaxation: too many planes");
                tw->orig[tw->count++] = ptrs[c][k];
            }
    }
    float *base;
    RC(ws_get(WS_ALR_T, plane * tw->count * sizeof(float), &base));
    float *outs[AlrTwin::MAXP];
    const float *ins[AlrTwin::MAXP];
    int nco = 0;
    for (int k = 0; k < tw->count; k++) {
        tw->twin[k] = base + plane * k;
        bool iterate = false;
        for (int c = 0; c < nch; c++) iterate = iterate || tw->orig[k] == x[c];
        if (!iterate) { // coefficient plane: once per call
            outs[nco] = tw->twin[k];
            ins[nco++] = tw->orig[k];
        }
    }
    if (defer_n) { // the caller's kernel transposes the coefficient planes itself (k_alr_small)
        for (int k = 0; k < nco; k++) {
            defer_in[k] = ins[k];
            defer_out[k] = outs[k];
        }
        *defer_n = nco;
    } else {
        RC(alr_transpose_many(s, outs, ins, nco, nrows, ncols, nframes));
    }
    for (int c = 0; c < nch; c++) {
        const float *tp[NP];
        for (int k = 0; k < NP; k++) tp[k] = ptrs[c][k] ? tw->find(ptrs[c][k]) : nullptr;
        memcpy(&qt[c], tp, sizeof(Ctx));
        xt[c] = tw->find(x[c]);
    }
    return PDEIP_OK;
}

// The iteration loop shared by every line-relaxation entry point.  q[c] / x[c]: context and iterate
// plane of field c; the reference relaxes columns of field 0 then field 1, rows of field 1 then
// field 0 (opticalflowSolvers.c:231-258); single-field solvers: columns, then rows.
template <class Mdl>
static int run_alr(const char *who, hipStream_t s, const typename Mdl::Ctx *q, float *const *x, int nch, int nrows, int ncols,
                   int nframes, int iter, float omega, int mode)
{
    RC(check_dims(who, nrows, ncols, nframes));
    RC(check_mode(who, mode));
    RC(check_alr_line(who, mode, nrows, ncols));
    tls.last_launches = 0;
    if (iter <= 0) return PDEIP_OK;
    const int fwd[2] = {0, 1}, rev[2] = {1, 0};
    typename Mdl::Ctx qt[2];
    float *xt[2] = {nullptr, nullptr};
    AlrTwin tw;
    // zebra order on a small frame: the whole call in one launch (k_alr_small); PDEIP_ALR_SMALL=0 disables
    if (mode == PDEIP_MODE_RED_BLACK && env_int("PDEIP_ALR_SMALL", 1) != 0) {
        const size_t lds = alr_small_lds_bytes(nrows, ncols, Mdl::INTERIOR_LINES);
        // one CU evaluates every row of the call: worth it where a pass is a few microseconds of work, i.e. up to ~60 x 100
        // (tools/time_alr_small.py: 34x60 348 -> 161 us, 17x30 294 -> 94 us, 61x108 407 -> 367 us, 68x120 392 -> 434 us)
        if (lds <= (size_t)150 * 1024 && nrows >= 3 && ncols >= 3 && (long)nrows * ncols <= 6144) {
            AlrSmallArgs<Mdl> A{};
            const float *tin[AlrTwin::MAXP];
            float *tout[AlrTwin::MAXP];
            int ntr = 0;
            RC(alr_make_twins(s, q, qt, nch, x, xt, nrows, ncols, nframes, &tw, tin, tout, &ntr));
            if (ntr <= ALR_SMALL_MAXTR) {
                for (int c = 0; c < nch; c++) {
                    A.q[c] = q[c];
                    A.qt[c] = qt[c];
                    A.x[c] = x[c];
                    A.xt[c] = xt[c];
                }
                for (int k = 0; k < ntr; k++) {
                    A.tin[k] = tin[k];
                    A.tout[k] = tout[k];
                }
                float *fbase;
                const size_t fplane = (size_t)nrows * ncols * nframes;
                RC(ws_get(WS_ALR, fplane * 8 * sizeof(float), &fbase));
                for (int c = 0; c < nch; c++)
                    for (int d = 0; d < 2; d++) {
                        A.cp[c][d] = fbase + fplane * (size_t)((c * 2 + d) * 2);
                        A.dv[c][d] = A.cp[c][d] + fplane;
                    }
                A.ntr = ntr; A.nch = nch; A.nrows = nrows; A.ncols = ncols; A.iter = iter; A.omega = omega;
                A.fs = (size_t)nrows * ncols;
                RC(ensure_lds(reinterpret_cast<const void *>(&k_alr_small<Mdl>), lds));
                SweepTimer timer(s);
                hipLaunchKernelGGL(k_alr_small<Mdl>, dim3((unsigned)nframes), dim3(ALR_SMALL_THREADS), lds, s, A);
                timer.stop(iter);
                tls.last_launches++;
                HIPCHK(hipGetLastError());
                return PDEIP_OK;
            }
            tw = AlrTwin{}; // too many planes for the argument block: the launch-per-pass path
        }
    }
    RC(alr_make_twins(s, q, qt, nch, x, xt, nrows, ncols, nframes, &tw));
    AlrFactors f{};
    static const bool zebra1 = env_int("PDEIP_ALR_ZEBRA1", 0) != 0; // the one-lane-per-line kernel for every model (A/B timing)
    if (mode == PDEIP_MODE_EXACT_ORDER || !zebra1) RC(alr_factor<Mdl>(s, q, qt, nch, nrows, ncols, nframes, &f));
    // zebra order, two coupled fields, factor planes present: one launch per colour for both fields (k_alr_zebra3_pair)
    const bool pair = mode != PDEIP_MODE_EXACT_ORDER && nch == 2 && f.cp[0][0] != nullptr && f.cp[1][0] != nullptr && env_int("PDEIP_ALR_PAIR", 1) != 0;
    SweepTimer timer(s);
    for (int it = 0; it < iter; it++) {
        if (mode == PDEIP_MODE_EXACT_ORDER)
            RC(alr_lex_pass<Mdl>(s, q, x, f, fwd, nch, nrows, ncols, nframes, true, omega));
        else if (pair)
            RC(alr_zebra_pass_pair<Mdl>(s, q[0], x[0], f.cp[0][0], f.dv[0][0], q[1], x[1], f.cp[1][0], f.dv[1][0], nrows, ncols, nframes, true, omega));
        else
            for (int c = 0; c < nch; c++) RC(alr_zebra_pass<Mdl>(s, q[c], x[c], f.cp[c][0], f.dv[c][0], nrows, ncols, nframes, true, omega));
        RC(alr_transpose_many(s, xt, x, nch, nrows, ncols, nframes));
        if (mode == PDEIP_MODE_EXACT_ORDER)
            RC(alr_lex_pass<Mdl>(s, qt, xt, f, nch == 2 ? rev : fwd, nch, nrows, ncols, nframes, false, omega));
        else if (pair)
            RC(alr_zebra_pass_pair<Mdl>(s, qt[1], xt[1], f.cp[1][1], f.dv[1][1], qt[0], xt[0], f.cp[0][1], f.dv[0][1], nrows, ncols, nframes, false, omega));
        else
            for (int c = nch - 1; c >= 0; c--) RC(alr_zebra_pass<Mdl>(s, qt[c], xt[c], f.cp[c][1], f.dv[c][1], nrows, ncols, nframes, false, omega));
        RC(alr_transpose_many(s, x, xt, nch, ncols, nrows, nframes));
    }
    timer.stop(iter);
    return PDEIP_OK;
}

extern "C" int pdeip_oflow_alr_elin4_dev(void *stream, float *U, float *V, const float *M, const float *Cu,
                                         const float *Cv, const float *Du, const float *Dv, const float *wW,
                                         const float *wN, const float *wE, const float *wS, int nrows, int ncols,
                                         int iter, float omega, int mode)
{
    const AlrElin4::Ctx q[2] = {{U, V, M, Cu, Du, wW, wN, wE, wS}, {V, U, M, Cv, Dv, wW, wN, wE, wS}};
    float *const x[2] = {U, V};
    return run_alr<AlrElin4>("pdeip_oflow_alr_elin4_dev", static_cast<hipStream_t>(stream), q, x, 2, nrows, ncols, 1, iter, omega, mode);
}

extern "C" int pdeip_oflow_alr_llin4_dev(void *stream, const float *U, const float *V, float *dU, float *dV,
                                         const float *M, const float *Cu, const float *Cv, const float *Du,
                                         const float *Dv, const float *wW, const float *wN, const float *wE,
                                         const float *wS, int nrows, int ncols, int iter, float omega, int mode)
{
    const AlrLlin4::Ctx q[2] = {{U, dU, dV, M, Cu, Du, wW, wN, wE, wS}, {V, dV, dU, M, Cv, Dv, wW, wN, wE, wS}};
    float *const x[2] = {dU, dV};
    return run_alr<AlrLlin4>("pdeip_oflow_alr_llin4_dev", static_cast<hipStream_t>(stream), q, x, 2, nrows, ncols, 1, iter, omega, mode);
}

extern "C" int pdeip_oflow_alr_llin8_dev(void *stream, const float *U, const float *V, float *dU, float *dV,
                                         const float *M, const float *Cu, const float *Cv, const float *Du,
                                         const float *Dv, const float *wW, const float *wNW, const float *wN,
                                         const float *wNE, const float *wE, const float *wSE, const float *wS,
                                         const float *wSW, int nrows, int ncols, int iter, float omega, int mode)
{
    const AlrLlin8::Ctx q[2] = {{U, dU, dV, M, Cu, Du, {wN, wS, wE, wW, wNW, wNE, wSW, wSE}},
                                {V, dV, dU, M, Cv, Dv, {wN, wS, wE, wW, wNW, wNE, wSW, wSE}}};
    float *const x[2] = {dU, dV};
    return run_alr<AlrLlin8>("pdeip_oflow_alr_llin8_dev", static_cast<hipStream_t>(stream), q, x, 2, nrows, ncols, 1, iter, omega, mode);
}

extern "C" int pdeip_disp_alr_llin4_dev(void *stream, const float *U, float *dU, const float *Cu, const float *Du,
                                        const float *wW, const float *wN, const float *wE, const float *wS,
                                        int nrows, int ncols, int iter, float omega, int mode)
{
    const AlrDisp4::Ctx q[1] = {{U, dU, nullptr, nullptr, Cu, Du, wW, wN, wE, wS}};
    float *const x[1] = {dU};
    return run_alr<AlrDisp4>("pdeip_disp_alr_llin4_dev", static_cast<hipStream_t>(stream), q, x, 1, nrows, ncols, 1, iter, omega, mode);
}

extern "C" int pdeip_pde_alr4_dev(void *stream, float *X, const float *TRACE, const float *B, const float *wW,
                                  const float *wN, const float *wE, const float *wS, int nrows, int ncols,
                                  int nframes, int iter, float omega, int mode)
{
    const AlrPde4::Ctx q[1] = {{X, TRACE, B, wW, wN, wE, wS}};
    float *const x[1] = {X};
    return run_alr<AlrPde4>("pdeip_pde_alr4_dev", static_cast<hipStream_t>(stream), q, x, 1, nrows, ncols, nframes, iter, omega, mode);
}

// One iteration whatever `iter` says (pdeSolvers.c:362), interior columns then interior rows.
extern "C" int pdeip_pde_alr8_dev(void *stream, float *X, const float *TRACE, const float *B, const float *wW,
                                  const float *wNW, const float *wN, const float *wNE, const float *wE,
                                  const float *wSE, const float *wS, const float *wSW, int nrows, int ncols,
                                  int nframes, int iter, float omega, int mode)
{
    (void)iter;
    const AlrPde8::Ctx q[1] = {{X, TRACE, B, wW, wNW, wN, wNE, wE, wSE, wS, wSW}};
    float *const x[1] = {X};
    return run_alr<AlrPde8>("pdeip_pde_alr8_dev", static_cast<hipStream_t>(stream), q, x, 1, nrows, ncols, nframes, 1, omega, mode);
}

