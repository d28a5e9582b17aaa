// pdeip_capi.hip -- libpdeip.so: context, launch logic and the extern "C" boundary (include/pdeip.h).
//
// Build (see build.py): hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -shared.
// -ffp-contract=off is part of the parity contract: the reference is plain C built without FMA.
#include "../../include/pdeip.h"

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include <rocprim/rocprim.hpp>

#include "pdeip_alr.hpp"
#include "pdeip_flow.hpp"
#include "pdeip_fas.hpp"
#include "pdeip_sym.hpp"
#include "pdeip_pyr.hpp"
#include "pdeip_tv.hpp"
#include "pdeip_models.hpp"
#include "pdeip_pointwise.hpp"
#include "pdeip_sor_exact.hpp"
#include "pdeip_sor_pde8.hpp"
#include "pdeip_sor_rb.hpp"

using namespace pdeip;

// ------------------------------------------------------------------------------------------------
// context
// ------------------------------------------------------------------------------------------------
namespace {

enum { WS_AUX0 = 0, WS_AUX1, WS_PING, WS_ARENA, WS_CTL, WS_ORDER, WS_ALR, WS_ALR_T, WS_TV, WS_NSLOT };

struct Context {
    int device = 0;
    int mode = PDEIP_MODE_EXACT_ORDER;
    int last_launches = 0;
    char err[512] = "";
    void *ws[WS_NSLOT] = {};
    size_t ws_bytes[WS_NSLOT] = {};
    int order_B = 0, order_T = 0; // shape of the cached persistent-kernel schedule table
    bool persist_used = false;    // a persistent launch happened since the last pdeip_persist_error()
    int rb_tj = 0; // columns per red-black unit (0 = default)
    // sweep-kernel timing (pdeip_profile_*)
    bool profile = false;
    static constexpr int MAX_EV = 4096;
    hipEvent_t ev[MAX_EV][2];
    int ev_launches[MAX_EV];
    int n_ev = 0, n_ev_created = 0;
};
Context g;

// Records an event pair around a run of sweep launches when profiling is on.
struct SweepTimer {
    hipStream_t s;
    int slot = -1;
    explicit SweepTimer(hipStream_t stream) : s(stream)
    {
        if (!g.profile || g.n_ev >= Context::MAX_EV) return;
        if (g.n_ev == g.n_ev_created) {
            if (hipEventCreate(&g.ev[g.n_ev][0]) != hipSuccess || hipEventCreate(&g.ev[g.n_ev][1]) != hipSuccess) return;
            g.n_ev_created++;
        }
        slot = g.n_ev++;
        (void)hipEventRecord(g.ev[slot][0], s);
    }
    void stop(int launches)
    {
        if (slot < 0) return;
        (void)hipEventRecord(g.ev[slot][1], s);
        g.ev_launches[slot] = launches;
    }
};

int set_err(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g.err, sizeof g.err, fmt, ap);
    va_end(ap);
    return code;
}

#define HIPCHK(expr)                                                                             \
    do {                                                                                         \
        hipError_t e_ = (expr);                                                                  \
        if (e_ != hipSuccess)                                                                    \
            return set_err(PDEIP_ERR_DEVICE, "%s failed: %s", #expr, hipGetErrorString(e_));     \
    } while (0)

#define RC(expr)                  \
    do {                          \
        int rc_ = (expr);         \
        if (rc_) return rc_;      \
    } while (0)

// Grow-only device workspace.  Growing synchronises the device (old buffer may be in use).
int ws_get(int slot, size_t bytes, float **out)
{
    if (g.ws_bytes[slot] < bytes) {
        HIPCHK(hipDeviceSynchronize());
        if (g.ws[slot]) HIPCHK(hipFree(g.ws[slot]));
        g.ws[slot] = nullptr;
        g.ws_bytes[slot] = 0;
        hipError_t e = hipMalloc(&g.ws[slot], bytes);
        if (e != hipSuccess) return set_err(PDEIP_ERR_NOMEM, "hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
        g.ws_bytes[slot] = bytes;
    }
    *out = static_cast<float *>(g.ws[slot]);
    return PDEIP_OK;
}

int check_dims(const char *who, int nrows, int ncols, int nframes)
{
    if (nrows < 3 || ncols < 3)
        return set_err(PDEIP_ERR_ARG, "%s: image must be at least 3x3 (got %dx%d)", who, nrows, ncols);
    if (nframes < 1) return set_err(PDEIP_ERR_ARG, "%s: number of frames must be >= 1 (got %d)", who, nframes);
    if ((long long)nrows * ncols * nframes > 0x7fffffffLL)
        return set_err(PDEIP_ERR_ARG, "%s: more than 2^31-1 elements", who);
    return PDEIP_OK;
}

int check_mode(const char *who, int mode)
{
    if (mode != PDEIP_MODE_EXACT_ORDER && mode != PDEIP_MODE_RED_BLACK)
        return set_err(PDEIP_ERR_ARG, "%s: unknown sweep ordering %d", who, mode);
    return PDEIP_OK;
}

// The gateways' solver switch (e.g. Oflow_sor_elin4_2d.c:328-338).
int check_solver(const char *who, int solver)
{
    if (solver == PDEIP_SOLVER_SOR || solver == PDEIP_SOLVER_ALR) return PDEIP_OK;
    return set_err(PDEIP_ERR_SOLVER, "%s: no such solver", who);
}

bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

int env_int(const char *name, int dflt)
{
    const char *s = getenv(name);
    return (s && *s) ? atoi(s) : dflt;
}

// Columns per red-black unit.  Narrow strips mean more waves in flight but more halo re-reads
// ((TJ+2)/TJ coefficient, (TJ+4)/TJ iterate columns).  12 is the measured optimum at 4K (2880 units)
// and at 1080p (10-12 equal, 6-8 slower); a strip stride that is a multiple of a large power of two
// aliases on HBM channels (TJ=16 at nrows=2160 is 15 % slower than 12).  PDEIP_RB_TJ overrides.
int pick_rb_tj(int nrows, int ncols)
{
    (void)nrows;
    (void)ncols;
    const int forced = g.rb_tj > 0 ? g.rb_tj : env_int("PDEIP_RB_TJ", 0);
    return forced > 0 ? (forced < 2 ? 2 : forced) : 12;
}

// Strip width of the two-sweeps-per-launch kernel.  That kernel holds four column stages in registers
// (one wave per SIMD) and is bound by its instruction stream, not by HBM: a launch takes
// ceil(units / resident waves) rounds of (TJ + 6) steps, so the best TJ is the one that fills the last
// round (4K: 12 -> 3 rounds of 18 steps, 133 us; 34 -> 1 round of 40 steps, 113 us; 33 -> 2 rounds, 182 us).
template <class Mdl>
int pick_rb2_tj(int nrows, int ncols, int nframes, int ntiles_r)
{
    (void)nrows;
    const int forced = g.rb_tj > 0 ? g.rb_tj : env_int("PDEIP_RB_TJ", 0);
    if (forced > 0) return forced < 2 ? 2 : forced;
    static int slots = 0; // waves of this kernel the device holds at once
    if (slots == 0) {
        int blocks = 0, dev = 0;
        hipDeviceProp_t prop;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks, k_sor_rb<Mdl, true, false, true>, 64 * RB_WAVES_PER_BLOCK, 0) != hipSuccess) blocks = 1;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) prop.multiProcessorCount = 256;
        slots = (blocks > 0 ? blocks : 1) * RB_WAVES_PER_BLOCK * prop.multiProcessorCount;
    }
    int best = 12;
    long best_cost = -1;
    for (int tj = 2; tj <= 64; tj++) {
        const long units = (long)ntiles_r * ((ncols + tj - 1) / tj) * nframes;
        const long cost = ((units + slots - 1) / slots) * (tj + 6);
        if (best_cost < 0 || cost <= best_cost) { // ties: the wider strip re-reads fewer halo columns
            best_cost = cost;
            best = tj;
        }
    }
    return best;
}

// ------------------------------------------------------------------------------------------------
// sweep drivers (5-point models)
// ------------------------------------------------------------------------------------------------

dim3 pixel_grid(int nrows, int ncols, int nz) { return dim3((unsigned)((nrows + 255) / 256), (unsigned)ncols, (unsigned)nz); }

// Runs `iter` sweeps of model Mdl on the iterate buffers P.it_out (in place from the caller's
// point of view).  P.ro and P.cf must be set, with the RAW planes in the two derived slots
// (Mdl::D0, Mdl::D1); the derived planes (divisors) are built into workspace here.
template <class Mdl>
int run_sweeps(hipStream_t s, SweepPlanes<Mdl> P, int nrows, int ncols, int nframes, int iter,
               float omega, int mode, int col0)
{
    constexpr int NIT = Mdl::NIT;
    const size_t n = (size_t)nrows * ncols;
    g.last_launches = 0;
    if (iter <= 0) return PDEIP_OK;
    float *aux0 = nullptr, *aux1 = nullptr;
    RC(ws_get(WS_AUX0, n * nframes * sizeof(float), &aux0));
    RC(ws_get(WS_AUX1, n * nframes * sizeof(float), &aux1));

    if (mode == PDEIP_MODE_EXACT_ORDER) {
        hipLaunchKernelGGL(k_derive<Mdl>, pixel_grid(nrows, ncols, nframes), dim3(256), 0, s, P, aux0, aux1, nrows, ncols, n);
        g.last_launches++;
        P.cf[Mdl::D0] = aux0;
        P.cf[Mdl::D1] = aux1;
        const int A = (nrows - 2 + 63 + EX_R - 1) / EX_R;
        const int B = (ncols - 2 + 63) / 64;
        const int last_m = (A - 1) + 2 * (B - 1) + 3 * (iter - 1);
        for (int f = 0; f < NIT; f++) P.it_in[f] = P.it_out[f];
        // Launch-per-front or persistent?  The persistent form wins at every iter and frame size (tools/time_exact_persist.py:
        // 4K 2.70 vs 2.87 ms at iter=4 -- each strip has to trail its west neighbour by 64 rows plus the hand-off latency either
        // way --, 1.4x at iter=20, 1.3x at 1080p, 2.4x at 34x60: no per-front launch, sweeps overlap more tightly).
        // PDEIP_EXACT_PERSIST = 0 falls back to one launch per front.
        const bool persist = env_int("PDEIP_EXACT_PERSIST", 1) != 0;
        if (persist && B <= 0xffff && iter <= 0x7fff && n * sizeof(float) < 0xffffffffull) {
            // ---- persistent form: one launch, progress counters instead of one launch per front ----
            const int NC = (nrows - 2 + 63 + EX_CH - 1) / EX_CH;
            float *ctl_f = nullptr, *order_f = nullptr;
            const size_t nprog = (size_t)nframes * iter * B;
            RC(ws_get(WS_CTL, (4 + nprog) * sizeof(unsigned), &ctl_f));
            RC(ws_get(WS_ORDER, (size_t)B * iter * sizeof(int), &order_f));
            if (g.order_B != B || g.order_T != iter) { // (b,t) in an order where every dependency comes earlier
                std::vector<int> ord;
                ord.reserve((size_t)B * iter);
                for (int key = 0; key <= (B - 1) + 2 * (iter - 1); key++)
                    for (int t = 0; t < iter; t++) {
                        const int b = key - 2 * t;
                        if (b >= 0 && b < B) ord.push_back(b | (t << 16));
                    }
                HIPCHK(hipMemcpyAsync(order_f, ord.data(), ord.size() * sizeof(int), hipMemcpyHostToDevice, s));
                HIPCHK(hipStreamSynchronize(s)); // `ord` is about to go out of scope
                g.order_B = B;
                g.order_T = iter;
            }
            HIPCHK(hipMemsetAsync(ctl_f, 0, (4 + nprog) * sizeof(unsigned), s));
            PersistCtl ctl;
            ctl.ticket = reinterpret_cast<unsigned *>(ctl_f);
            ctl.abort_flag = ctl.ticket + 1;
            ctl.progress = ctl.ticket + 4;
            ctl.order = reinterpret_cast<const int *>(order_f);
            constexpr size_t plds = ExactLayout<Mdl>::LDS_BYTES + 16;
            static bool plds_opt_in = false;
            if (!plds_opt_in) {
                HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_sor_exact_persist<Mdl>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)plds));
                plds_opt_in = true;
            }
            g.persist_used = true;
            SweepTimer timer(s);
            hipLaunchKernelGGL(k_sor_exact_persist<Mdl>, dim3((unsigned)(B * iter * nframes)), dim3(128), plds, s, P, ctl, nrows, ncols, B, iter, NC, nframes, omega, n);
            timer.stop(1);
            g.last_launches++;
            const int nb = 2 * ncols + 2 * (nrows - 2);
            hipLaunchKernelGGL(k_fill_borders, dim3((nb + 255) / 256, nframes, NIT), dim3(256), 0, s,
                               P.it_out[0], P.it_out[NIT - 1], NIT, nrows, ncols, n);
            g.last_launches++;
            HIPCHK(hipGetLastError());
            return PDEIP_OK;
        }
        const dim3 grid((unsigned)(B * iter), (unsigned)nframes);
        constexpr size_t lds = ExactLayout<Mdl>::LDS_BYTES;
        static bool lds_opt_in = false; // > 64 KiB of dynamic LDS needs an explicit opt-in, once per kernel
        if (!lds_opt_in) {
            HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_sor_exact<Mdl>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            lds_opt_in = true;
        }
        SweepTimer timer(s);
        for (int m = 0; m <= last_m; m++) {
            hipLaunchKernelGGL(k_sor_exact<Mdl>, grid, dim3(128), lds, s, P, nrows, ncols, A, B, iter, m, omega, n);
            g.last_launches++;
        }
        timer.stop(last_m + 1);
        const int nb = 2 * ncols + 2 * (nrows - 2);
        hipLaunchKernelGGL(k_fill_borders, dim3((nb + 255) / 256, nframes, NIT), dim3(256), 0, s,
                           P.it_out[0], P.it_out[NIT - 1], NIT, nrows, ncols, n);
        g.last_launches++;
        HIPCHK(hipGetLastError());
        return PDEIP_OK;
    }

    // red-black: ping-pong between the caller's buffers and a scratch copy
    float *scratch = nullptr;
    int rc = ws_get(WS_PING, (size_t)NIT * n * nframes * sizeof(float), &scratch);
    if (rc) return rc;
    float *bufA[NIT], *bufB[NIT];
    bool vec = (nrows % 4 == 0);
    for (int f = 0; f < NIT; f++) {
        bufA[f] = P.it_out[f];
        bufB[f] = scratch + (size_t)f * n * nframes;
        vec = vec && aligned16(bufA[f]) && aligned16(bufB[f]);
    }
    for (int f = 0; f < Mdl::NCF; f++) vec = vec && aligned16(P.cf[f]);
    vec = vec && aligned16(aux0) && aligned16(aux1);
    for (int f = 0; f < Mdl::NRO; f++) vec = vec && aligned16(P.ro[f]);

    const int ntiles_r = (nrows + RB_OWN_ROWS - 1) / RB_OWN_ROWS;
    const dim3 block(64 * RB_WAVES_PER_BLOCK);
    // Two sweeps per launch where the model allows it (pdeip_sor_rb.hpp, rb_march2): same results, about
    // two thirds of the traffic per sweep.  PDEIP_RB_FUSE=0 keeps one sweep per launch.
    static const bool fuse_enabled = env_int("PDEIP_RB_FUSE", 1) != 0;
    const bool fuse = fuse_enabled;
    const int TJ1 = pick_rb_tj(nrows, ncols), TJ2 = fuse ? pick_rb2_tj<Mdl>(nrows, ncols, nframes, ntiles_r) : TJ1;
    SweepTimer timer(s);
    int nlaunch = 0, flips = 0; // flips: how many times the iterate changed buffers
    for (int it = 0; it < iter;) {
        const bool two = fuse && it + 2 <= iter;
        const int TJ = two ? TJ2 : TJ1;
        const int nunits = ntiles_r * ((ncols + TJ - 1) / TJ);
        const dim3 grid((unsigned)((nunits + RB_WAVES_PER_BLOCK - 1) / RB_WAVES_PER_BLOCK), (unsigned)nframes);
        for (int f = 0; f < NIT; f++) {
            P.it_in[f] = (flips & 1) ? bufB[f] : bufA[f];
            P.it_out[f] = (flips & 1) ? bufA[f] : bufB[f];
        }
        const bool first = it == 0; // sweep 0 also builds the divisor planes
        float *d0 = first ? aux0 : nullptr, *d1 = first ? aux1 : nullptr;
#define PDEIP_RB_LAUNCH(V, F, T) hipLaunchKernelGGL((k_sor_rb<Mdl, V, F, T>), grid, block, 0, s, P, d0, d1, nrows, ncols, TJ, ntiles_r, nunits, omega, col0, n)
        if (two) {
            if (vec) { if (first) PDEIP_RB_LAUNCH(true, true, true); else PDEIP_RB_LAUNCH(true, false, true); }
            else     { if (first) PDEIP_RB_LAUNCH(false, true, true); else PDEIP_RB_LAUNCH(false, false, true); }
        } else {
            if (vec) { if (first) PDEIP_RB_LAUNCH(true, true, false); else PDEIP_RB_LAUNCH(true, false, false); }
            else     { if (first) PDEIP_RB_LAUNCH(false, true, false); else PDEIP_RB_LAUNCH(false, false, false); }
        }
#undef PDEIP_RB_LAUNCH
        if (first) {
            P.cf[Mdl::D0] = aux0;
            P.cf[Mdl::D1] = aux1;
        }
        it += two ? 2 : 1;
        flips++;
        nlaunch++;
        g.last_launches++;
    }
    timer.stop(nlaunch);
    if (flips & 1) // the last launch wrote the scratch copy
        for (int f = 0; f < NIT; f++)
            HIPCHK(hipMemcpyAsync(bufA[f], bufB[f], n * nframes * sizeof(float), hipMemcpyDeviceToDevice, s));
    HIPCHK(hipGetLastError());
    return PDEIP_OK;
}

// ------------------------------------------------------------------------------------------------
// host staging: one arena per call, carved sequentially
// ------------------------------------------------------------------------------------------------
struct Arena {
    float *base = nullptr;
    size_t cap = 0, used = 0;
    int init(size_t nfloats)
    {
        // every plane starts 16-byte aligned so the vector path stays available
        cap = nfloats;
        used = 0;
        return ws_get(WS_ARENA, nfloats * sizeof(float), &base);
    }
    float *take(size_t nfloats)
    {
        float *p = base + used;
        used += (nfloats + 3) & ~(size_t)3;
        return p;
    }
};
size_t pad4(size_t n) { return (n + 3) & ~(size_t)3; }

int upload(float *dst, const float *src, size_t nfloats)
{
    HIPCHK(hipMemcpy(dst, src, nfloats * sizeof(float), hipMemcpyHostToDevice));
    return PDEIP_OK;
}
int download(float *dst, const float *src, size_t nfloats)
{
    HIPCHK(hipMemcpy(dst, src, nfloats * sizeof(float), hipMemcpyDeviceToHost));
    return PDEIP_OK;
}

#define NONNULL(who, p)                                                                    \
    do {                                                                                   \
        if ((p) == nullptr) return set_err(PDEIP_ERR_ARG, "%s: argument '%s' is NULL", who, #p); \
    } while (0)

int use_device()
{
    HIPCHK(hipSetDevice(g.device));
    return PDEIP_OK;
}

} // namespace

// ------------------------------------------------------------------------------------------------
// library state
// ------------------------------------------------------------------------------------------------
extern "C" const char *pdeip_version(void) { return "pdeip-mi355x 0.1 (gfx950)"; }
extern "C" const char *pdeip_last_error(void) { return g.err; }
extern "C" int pdeip_set_mode(int mode)
{
    RC(check_mode("pdeip_set_mode", mode));
    g.mode = mode;
    return PDEIP_OK;
}
extern "C" int pdeip_get_mode(void) { return g.mode; }
extern "C" int pdeip_set_device(int device_id)
{
    int n = 0;
    HIPCHK(hipGetDeviceCount(&n));
    if (device_id < 0 || device_id >= n) return set_err(PDEIP_ERR_ARG, "pdeip_set_device: no device %d (have %d)", device_id, n);
    if (device_id != g.device) pdeip_release();
    g.device = device_id;
    return PDEIP_OK;
}
extern "C" int pdeip_release(void)
{
    for (int s = 0; s < WS_NSLOT; s++) {
        if (g.ws[s]) (void)hipFree(g.ws[s]);
        g.ws[s] = nullptr;
        g.ws_bytes[s] = 0;
    }
    return PDEIP_OK;
}
extern "C" int pdeip_last_launch_count(void) { return g.last_launches; }
extern "C" int pdeip_persist_error(void)
{ // waits for the device, then reports whether a bounded spin of the persistent kernel timed out
    if (!g.ws[WS_CTL] || !g.persist_used) return PDEIP_OK;
    g.persist_used = false;
    unsigned words[2] = {0, 0};
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpy(words, g.ws[WS_CTL], sizeof words, hipMemcpyDeviceToHost));
    if (words[1] != 0) return set_err(PDEIP_ERR_DEVICE, "persistent exact-order kernel: a dependency wait timed out (results are invalid)");
    return PDEIP_OK;
}
#ifdef PDEIP_EXACT_STAMPS
extern "C" int pdeip_debug_read_stamps(unsigned long long *out)
{
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_exact_stamps), 64 * sizeof(unsigned long long)));
    return PDEIP_OK;
}
#endif
extern "C" int pdeip_profile_enable(int on)
{
    g.profile = (on != 0);
    g.n_ev = 0;
    return PDEIP_OK;
}
extern "C" int pdeip_profile_read(double *elapsed_ms, int *sweep_launches)
{
    double ms = 0.0;
    int launches = 0;
    for (int k = 0; k < g.n_ev; k++) {
        HIPCHK(hipEventSynchronize(g.ev[k][1]));
        float t = 0.0f;
        HIPCHK(hipEventElapsedTime(&t, g.ev[k][0], g.ev[k][1]));
        ms += t;
        launches += g.ev_launches[k];
    }
    g.n_ev = 0;
    if (elapsed_ms) *elapsed_ms = ms;
    if (sweep_launches) *sweep_launches = launches;
    return PDEIP_OK;
}

// ------------------------------------------------------------------------------------------------
// device-pointer entry points
// ------------------------------------------------------------------------------------------------
extern "C" int pdeip_oflow_sor_elin4_dev(void *stream, float *U, float *V, const float *M, const float *Cu,
                                         const float *Cv, const float *Du, const float *Dv, const float *wW,
                                         const float *wN, const float *wE, const float *wS, int nrows,
                                         int ncols, int iter, float omega, int mode, int col0)
{
    const char *who = "pdeip_oflow_sor_elin4_dev";
    RC(check_dims(who, nrows, ncols, 1));
    RC(check_mode(who, mode));
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (iter <= 0) return PDEIP_OK;
    SweepPlanes<ModelElin4> P{};
    P.it_out[0] = U;
    P.it_out[1] = V;
    const float *cf[9] = {M, Cu, Cv, Du, Dv, wW, wN, wE, wS}; // Du,Dv: raw planes in the divisor slots
    for (int f = 0; f < 9; f++) P.cf[f] = cf[f];
    RC(run_sweeps<ModelElin4>(s, P, nrows, ncols, 1, iter, omega, mode, col0));
    return PDEIP_OK;
}

extern "C" int pdeip_oflow_sor_llin4_dev(void *stream, const float *U, const float *V, float *dU, float *dV,
                                         const float *M, const float *Cu, const float *Cv, const float *Du,
                                         const float *Dv, const float *wW, const float *wN, const float *wE,
                                         const float *wS, int nrows, int ncols, int iter, float omega,
                                         int mode, int col0)
{
    const char *who = "pdeip_oflow_sor_llin4_dev";
    RC(check_dims(who, nrows, ncols, 1));
    RC(check_mode(who, mode));
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (iter <= 0) return PDEIP_OK;
    SweepPlanes<ModelLlin4> P{};
    P.it_out[0] = dU;
    P.it_out[1] = dV;
    P.ro[0] = U;
    P.ro[1] = V;
    const float *cf[9] = {M, Cu, Cv, Du, Dv, wW, wN, wE, wS}; // Du,Dv: raw planes in the divisor slots
    for (int f = 0; f < 9; f++) P.cf[f] = cf[f];
    RC(run_sweeps<ModelLlin4>(s, P, nrows, ncols, 1, iter, omega, mode, col0));
    return PDEIP_OK;
}

extern "C" int pdeip_disp_sor_llin4_dev(void *stream, const float *U, float *dU, const float *Cu,
                                        const float *Du, const float *wW, const float *wN, const float *wE,
                                        const float *wS, int nrows, int ncols, int iter, float omega,
                                        int mode, int col0)
{
    const char *who = "pdeip_disp_sor_llin4_dev";
    RC(check_dims(who, nrows, ncols, 1));
    RC(check_mode(who, mode));
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (iter <= 0) return PDEIP_OK;
    SweepPlanes<ModelDisp4> P{};
    P.it_out[0] = dU;
    P.ro[0] = U;
    const float *cf[6] = {Cu, Du, wW, wN, wE, wS}; // Cu,Du: raw planes in the dividend/divisor slots
    for (int f = 0; f < 6; f++) P.cf[f] = cf[f];
    RC(run_sweeps<ModelDisp4>(s, P, nrows, ncols, 1, iter, omega, mode, col0));
    return PDEIP_OK;
}

// Disp_sor_llin_sym4_2d: two disparity fields that do not read each other (disparitySolvers.c:301-548).
// solver 1: ModelDispSym4 on each; solver 2: the line solvers are the plain disparity ones (:503-540).
extern "C" int pdeip_disp_sor_llin_sym4_dev(void *stream, const float *U0, float *dU0, const float *Cu0, const float *Du0,
                                            const float *wW0, const float *wN0, const float *wE0, const float *wS0,
                                            const float *U1, float *dU1, const float *Cu1, const float *Du1,
                                            const float *wW1, const float *wN1, const float *wE1, const float *wS1,
                                            int nrows, int ncols, int iter, float omega, int solver, int mode, int col0)
{
    const char *who = "pdeip_disp_sor_llin_sym4_dev";
    RC(check_dims(who, nrows, ncols, 1));
    RC(check_mode(who, mode));
    RC(check_solver(who, solver));
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (iter <= 0) return PDEIP_OK;
    const float *U[2] = {U0, U1}, *cf[2][6] = {{Cu0, Du0, wW0, wN0, wE0, wS0}, {Cu1, Du1, wW1, wN1, wE1, wS1}};
    float *dU[2] = {dU0, dU1};
    int launches = 0;
    for (int k = 0; k < 2; k++) {
        if (solver == PDEIP_SOLVER_ALR) {
            RC(pdeip_disp_alr_llin4_dev(stream, U[k], dU[k], cf[k][0], cf[k][1], cf[k][2], cf[k][3], cf[k][4], cf[k][5], nrows, ncols, iter, omega, mode));
        } else {
            SweepPlanes<ModelDispSym4> P{};
            P.it_out[0] = dU[k];
            P.ro[0] = U[k];
            for (int f = 0; f < 6; f++) P.cf[f] = cf[k][f];
            RC(run_sweeps<ModelDispSym4>(s, P, nrows, ncols, 1, iter, omega, mode, col0));
        }
        launches += g.last_launches;
    }
    g.last_launches = launches;
    return PDEIP_OK;
}

extern "C" int pdeip_pde_sor4_dev(void *stream, float *X, const float *TRACE, const float *B, const float *wW,
                                  const float *wN, const float *wE, const float *wS, int nrows, int ncols,
                                  int nframes, int iter, float omega, int mode, int col0)
{
    const char *who = "pdeip_pde_sor4_dev";
    RC(check_dims(who, nrows, ncols, nframes));
    RC(check_mode(who, mode));
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (iter <= 0) return PDEIP_OK;
    SweepPlanes<ModelPde4> P{};
    P.it_out[0] = X;
    const float *cf[6] = {B, TRACE, wW, wN, wE, wS}; // B,TRACE: raw planes in the derived slots
    for (int f = 0; f < 6; f++) P.cf[f] = cf[f];
    RC(run_sweeps<ModelPde4>(s, P, nrows, ncols, nframes, iter, omega, mode, col0));
    return PDEIP_OK;
}

extern "C" int pdeip_pde_sor8_dev(void *stream, float *X, const float *TRACE, const float *B, const float *wW,
                                  const float *wNW, const float *wN, const float *wNE, const float *wE,
                                  const float *wSE, const float *wS, const float *wSW, int nrows, int ncols,
                                  int nframes, int iter, float omega, int mode, int col0)
{
    const char *who = "pdeip_pde_sor8_dev";
    RC(check_dims(who, nrows, ncols, nframes));
    RC(check_mode(who, mode));
    hipStream_t s = static_cast<hipStream_t>(stream);
    g.last_launches = 0;
    if (iter <= 0) return PDEIP_OK;
    const size_t n = (size_t)nrows * ncols, nf = n * (size_t)nframes;
    float *bt, *inv, *scratch = nullptr;
    RC(ws_get(WS_AUX0, nf * sizeof(float), &bt));
    RC(ws_get(WS_AUX1, nf * sizeof(float), &inv));
    if (mode == PDEIP_MODE_EXACT_ORDER) {
        hipLaunchKernelGGL(k_pde8_divisors, pixel_grid(nrows, ncols, nframes), dim3(256), 0, s, bt, inv, TRACE, B, wW, wNW, wN, wNE, wE, wSE, wS, wSW, nrows, ncols, n);
        g.last_launches++;
        Pde8Planes P{};
        P.x = X;
        const float *cf[ModelPde8::NCF] = {bt, inv, wW, wNW, wN, wNE, wE, wSE, wS, wSW};
        for (int f = 0; f < ModelPde8::NCF; f++) P.cf[f] = cf[f];
        RC(ws_get(WS_PING, pde8_exact_scratch_floats(nrows, ncols, nframes, iter) * sizeof(float), &scratch));
        SweepTimer timer(s);
        const int nl = pde8_run_exact(s, P, scratch, nrows, ncols, nframes, iter, omega);
        timer.stop(nl);
        g.last_launches += nl;
        HIPCHK(hipGetLastError());
        return PDEIP_OK;
    }
    // four-colour: one fused launch per sweep, ping-pong with a scratch copy; sweep 0 builds B_temp/INV_TRACE
    RC(ws_get(WS_PING, nf * sizeof(float), &scratch));
    Pde8SweepPlanes P{};
    const float *cf[ModelPde8::NCF] = {B, TRACE, wW, wNW, wN, wNE, wE, wSE, wS, wSW}; // raw planes in the derived slots
    bool vec = (nrows % 4 == 0) && aligned16(X) && aligned16(scratch) && aligned16(bt) && aligned16(inv);
    for (int f = 0; f < ModelPde8::NCF; f++) {
        P.cf[f] = cf[f];
        vec = vec && aligned16(cf[f]);
    }
    const dim3 block(64 * RB_WAVES_PER_BLOCK);
    static const bool fuse = env_int("PDEIP_RB_FUSE", 1) != 0; // two sweeps per launch (k_pde8_colour2), same results
    const int TJ1 = pick_rb_tj(nrows, ncols);
    int TJ2 = TJ1;
    const int ntiles1 = (nrows + RB_OWN_ROWS - 1) / RB_OWN_ROWS, ntiles2 = (nrows + P8_OWN_ROWS2 - 1) / P8_OWN_ROWS2;
    if (fuse && iter >= 2) {
        static int slots = 0; // resident waves of the fused kernel (see pick_rb2_tj)
        if (slots == 0) {
            int blocks = 0, dev = 0;
            hipDeviceProp_t prop;
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks, k_pde8_colour2<true, false>, 64 * RB_WAVES_PER_BLOCK, 0) != hipSuccess) blocks = 1;
            if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) prop.multiProcessorCount = 256;
            slots = (blocks > 0 ? blocks : 1) * RB_WAVES_PER_BLOCK * prop.multiProcessorCount;
        }
        const int forced = g.rb_tj > 0 ? g.rb_tj : env_int("PDEIP_RB_TJ", 0);
        if (forced > 0) TJ2 = forced < 2 ? 2 : forced;
        else {
            long best_cost = -1;
            for (int tj = 2; tj <= 64; tj++) {
                const long units = (long)ntiles2 * ((ncols + tj - 1) / tj) * nframes;
                const long cost = ((units + slots - 1) / slots) * (tj + 6);
                if (best_cost < 0 || cost <= best_cost) {
                    best_cost = cost;
                    TJ2 = tj;
                }
            }
        }
    }
    SweepTimer timer(s);
    int nlaunch = 0, flips = 0;
    for (int it = 0; it < iter;) {
        const bool two = fuse && it + 2 <= iter, first = it == 0;
        const int TJ = two ? TJ2 : TJ1, ntiles_r = two ? ntiles2 : ntiles1;
        const int nunits = ntiles_r * ((ncols + TJ - 1) / TJ);
        const dim3 grid((unsigned)((nunits + RB_WAVES_PER_BLOCK - 1) / RB_WAVES_PER_BLOCK), (unsigned)nframes);
        P.x_in = (flips & 1) ? scratch : X;
        P.x_out = (flips & 1) ? X : scratch;
        float *d0 = first ? bt : nullptr, *d1 = first ? inv : nullptr;
#define PDEIP_P8_LAUNCH(KERNEL, V, F) hipLaunchKernelGGL((KERNEL<V, F>), grid, block, 0, s, P, d0, d1, nrows, ncols, TJ, ntiles_r, nunits, omega, col0, n)
        if (two) {
            if (vec) { if (first) PDEIP_P8_LAUNCH(k_pde8_colour2, true, true); else PDEIP_P8_LAUNCH(k_pde8_colour2, true, false); }
            else     { if (first) PDEIP_P8_LAUNCH(k_pde8_colour2, false, true); else PDEIP_P8_LAUNCH(k_pde8_colour2, false, false); }
        } else {
            if (vec) { if (first) PDEIP_P8_LAUNCH(k_pde8_colour, true, true); else PDEIP_P8_LAUNCH(k_pde8_colour, true, false); }
            else     { if (first) PDEIP_P8_LAUNCH(k_pde8_colour, false, true); else PDEIP_P8_LAUNCH(k_pde8_colour, false, false); }
        }
#undef PDEIP_P8_LAUNCH
        if (first) { // sweep 0 built B_temp / INV_TRACE
            P.cf[ModelPde8::cB] = bt;
            P.cf[ModelPde8::cInv] = inv;
        }
        it += two ? 2 : 1;
        flips++;
        nlaunch++;
        g.last_launches++;
    }
    timer.stop(nlaunch);
    if (flips & 1) HIPCHK(hipMemcpyAsync(X, scratch, nf * sizeof(float), hipMemcpyDeviceToDevice, s));
    HIPCHK(hipGetLastError());
    return PDEIP_OK;
}

// ------------------------------------------------------------------------------------------------
// alternating line relaxation (solver 2): pdeip_alr.hpp
// ------------------------------------------------------------------------------------------------
constexpr int ALR_LEX_MAX_LINE = 10000; // one float4 per line element in LDS (160 KB per workgroup)

static int check_alr_line(const char *who, int mode, int nrows, int ncols)
{
    const int n = nrows > ncols ? nrows : ncols;
    if (mode == PDEIP_MODE_EXACT_ORDER && n > ALR_LEX_MAX_LINE)
        return set_err(PDEIP_ERR_UNSUPPORTED, "%s: exact-order line relaxation holds one line in LDS: at most %d pixels per line (got %d)",
                       who, ALR_LEX_MAX_LINE, n);
    return PDEIP_OK;
}

// Workspace of one call: per (chain, direction) the cp and divisor planes (k_alr_zebra3<ZB_FACTOR>).
struct AlrFactors {
    float *cp[2][2], *dv[2][2]; // [chain][vertical ? 0 : 1]
};

template <class Mdl, bool VERT, int MODE>
static int zebra3_launch(hipStream_t s, const typename Mdl::Ctx &q, float *x, float *cp, float *dv, float *dp, int nrows, int ncols, int nframes,
                         int first, int lastc, int lstep, float omega)
{
    static bool attr_set = false;
    if (!attr_set) {
        HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_alr_zebra3<Mdl, VERT, MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)Z3_LDS_BYTES));
        attr_set = true;
    }
    const int count = (lastc - first) / lstep + 1;
    hipLaunchKernelGGL((k_alr_zebra3<Mdl, VERT, MODE>), dim3((unsigned)((count + ZB_LW - 1) / ZB_LW), (unsigned)nframes), dim3(ZB_THREADS),
                       Z3_LDS_BYTES, s, q, x, cp, dv, dp, nrows, ncols, (size_t)nrows * ncols, first, lastc, lstep, omega);
    g.last_launches++;
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
            g.last_launches++;
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
    if (nch == 2 && 2 * line_bytes <= 160 * 1024) {
        AlrChains<Mdl, 2> ch;
        for (int c = 0; c < 2; c++) ch.c[c] = AlrChain<Mdl>{q[order[c]], x[order[c]], f.cp[order[c]][d], f.dv[order[c]][d]};
        static size_t lds_set[2] = {0, 0};
        if (2 * line_bytes > 64 * 1024 && 2 * line_bytes > lds_set[d]) {
            const void *fn = vertical ? reinterpret_cast<const void *>(&k_alr_lex<Mdl, 2, true>) : reinterpret_cast<const void *>(&k_alr_lex<Mdl, 2, false>);
            HIPCHK(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(2 * line_bytes)));
            lds_set[d] = 2 * line_bytes;
        }
        if (vertical) hipLaunchKernelGGL((k_alr_lex<Mdl, 2, true>), dim3((unsigned)nframes), dim3(ALR_LEX_THREADS), 2 * line_bytes, s, ch, nrows, ncols, fs, lo, hi, omega);
        else hipLaunchKernelGGL((k_alr_lex<Mdl, 2, false>), dim3((unsigned)nframes), dim3(ALR_LEX_THREADS), 2 * line_bytes, s, ch, nrows, ncols, fs, lo, hi, omega);
        g.last_launches++;
    } else {
        static size_t lds_set[2] = {0, 0};
        if (line_bytes > 64 * 1024 && line_bytes > lds_set[d]) {
            const void *fn = vertical ? reinterpret_cast<const void *>(&k_alr_lex<Mdl, 1, true>) : reinterpret_cast<const void *>(&k_alr_lex<Mdl, 1, false>);
            HIPCHK(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)line_bytes));
            lds_set[d] = line_bytes;
        }
        for (int c = 0; c < nch; c++) {
            AlrChains<Mdl, 1> ch;
            ch.c[0] = AlrChain<Mdl>{q[order[c]], x[order[c]], f.cp[order[c]][d], f.dv[order[c]][d]};
            if (vertical) hipLaunchKernelGGL((k_alr_lex<Mdl, 1, true>), dim3((unsigned)nframes), dim3(ALR_LEX_THREADS), line_bytes, s, ch, nrows, ncols, fs, lo, hi, omega);
            else hipLaunchKernelGGL((k_alr_lex<Mdl, 1, false>), dim3((unsigned)nframes), dim3(ALR_LEX_THREADS), line_bytes, s, ch, nrows, ncols, fs, lo, hi, omega);
            g.last_launches++;
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
        g.last_launches++;
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
    g.last_launches++;
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
        g.last_launches++;
    }
    HIPCHK(hipGetLastError());
    return PDEIP_OK;
}

template <class Ctx>
static int alr_make_twins(hipStream_t s, const Ctx *q, Ctx *qt, int nch, float *const *x, float **xt, int nrows, int ncols, int nframes,
                          AlrTwin *tw)
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
    RC(alr_transpose_many(s, outs, ins, nco, nrows, ncols, nframes));
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
    g.last_launches = 0;
    if (iter <= 0) return PDEIP_OK;
    const int fwd[2] = {0, 1}, rev[2] = {1, 0};
    typename Mdl::Ctx qt[2];
    float *xt[2] = {nullptr, nullptr};
    AlrTwin tw;
    RC(alr_make_twins(s, q, qt, nch, x, xt, nrows, ncols, nframes, &tw));
    AlrFactors f{};
    static const bool zebra1 = env_int("PDEIP_ALR_ZEBRA1", 0) != 0; // the one-lane-per-line kernel for every model (A/B timing)
    if (mode == PDEIP_MODE_EXACT_ORDER || !zebra1) RC(alr_factor<Mdl>(s, q, qt, nch, nrows, ncols, nframes, &f));
    SweepTimer timer(s);
    for (int it = 0; it < iter; it++) {
        if (mode == PDEIP_MODE_EXACT_ORDER)
            RC(alr_lex_pass<Mdl>(s, q, x, f, fwd, nch, nrows, ncols, nframes, true, omega));
        else
            for (int c = 0; c < nch; c++) RC(alr_zebra_pass<Mdl>(s, q[c], x[c], f.cp[c][0], f.dv[c][0], nrows, ncols, nframes, true, omega));
        RC(alr_transpose_many(s, xt, x, nch, nrows, ncols, nframes));
        if (mode == PDEIP_MODE_EXACT_ORDER)
            RC(alr_lex_pass<Mdl>(s, qt, xt, f, nch == 2 ? rev : fwd, nch, nrows, ncols, nframes, false, omega));
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

extern "C" int pdeip_oflow_res_elin4_dev(void *stream, float *RU, float *RV, const float *U, const float *V,
                                         const float *M, const float *Cu, const float *Cv, const float *Du,
                                         const float *Dv, const float *wW, const float *wN, const float *wE,
                                         const float *wS, int nrows, int ncols, int nframes_coef)
{
    RC(check_dims("pdeip_oflow_res_elin4_dev", nrows, ncols, nframes_coef));
    hipLaunchKernelGGL((k_oflow_operator<false, false>), pixel_grid(nrows, ncols, nframes_coef), dim3(256), 0,
                       static_cast<hipStream_t>(stream), RU, RV, U, V, nullptr, nullptr, M, Cu, Cv, Du, Dv, wW,
                       wN, wE, wS, nrows, ncols, (size_t)nrows * ncols);
    HIPCHK(hipGetLastError());
    return PDEIP_OK;
}

extern "C" int pdeip_oflow_lhs_elin4_dev(void *stream, float *AU, float *AV, const float *U, const float *V,
                                         const float *M, const float *Du, const float *Dv, const float *wW,
                                         const float *wN, const float *wE, const float *wS, int nrows,
                                         int ncols, int nframes_coef)
{
    RC(check_dims("pdeip_oflow_lhs_elin4_dev", nrows, ncols, nframes_coef));
    hipLaunchKernelGGL((k_oflow_operator<false, true>), pixel_grid(nrows, ncols, nframes_coef), dim3(256), 0,
                       static_cast<hipStream_t>(stream), AU, AV, U, V, nullptr, nullptr, M, nullptr, nullptr, Du, Dv,
                       wW, wN, wE, wS, nrows, ncols, (size_t)nrows * ncols);
    HIPCHK(hipGetLastError());
    return PDEIP_OK;
}

extern "C" int pdeip_oflow_res_llin4_dev(void *stream, float *RU, float *RV, const float *U, const float *V,
                                         const float *dU, const float *dV, const float *M, const float *Cu,
                                         const float *Cv, const float *Du, const float *Dv, const float *wW,
                                         const float *wN, const float *wE, const float *wS, int nrows,
                                         int ncols, int nframes_coef)
{
    RC(check_dims("pdeip_oflow_res_llin4_dev", nrows, ncols, nframes_coef));
    hipLaunchKernelGGL((k_oflow_operator<true, false>), pixel_grid(nrows, ncols, nframes_coef), dim3(256), 0,
                       static_cast<hipStream_t>(stream), RU, RV, U, V, dU, dV, M, Cu, Cv, Du, Dv, wW, wN, wE, wS,
                       nrows, ncols, (size_t)nrows * ncols);
    HIPCHK(hipGetLastError());
    return PDEIP_OK;
}

extern "C" int pdeip_oflow_lhs_llin4_dev(void *stream, float *AU, float *AV, const float *U, const float *V,
                                         const float *dU, const float *dV, const float *M, const float *Du,
                                         const float *Dv, const float *wW, const float *wN, const float *wE,
                                         const float *wS, int nrows, int ncols, int nframes_coef)
{
    RC(check_dims("pdeip_oflow_lhs_llin4_dev", nrows, ncols, nframes_coef));
    hipLaunchKernelGGL((k_oflow_operator<true, true>), pixel_grid(nrows, ncols, nframes_coef), dim3(256), 0,
                       static_cast<hipStream_t>(stream), AU, AV, U, V, dU, dV, M, nullptr, nullptr, Du, Dv, wW, wN,
                       wE, wS, nrows, ncols, (size_t)nrows * ncols);
    HIPCHK(hipGetLastError());
    return PDEIP_OK;
}

extern "C" int pdeip_diffweights6_dev(void *stream, const float *D, int nrows, int ncols, int nframes,
                                      float eps, float *wW, float *wN, float *wE, float *wS)
{
    RC(check_dims("pdeip_diffweights6_dev", nrows, ncols, nframes));
    hipLaunchKernelGGL(k_diffweights6, pixel_grid(nrows, ncols, 1), dim3(256), 0, static_cast<hipStream_t>(stream),
                       wW, wN, wE, wS, D, nrows, ncols, nframes, eps);
    HIPCHK(hipGetLastError());
    return PDEIP_OK;
}

extern "C" int pdeip_warp_bilinear_dev(void *stream, const float *Iin, const float *X, const float *Y,
                                       int nrows, int ncols, int nframes, float *Iout)
{
    RC(check_dims("pdeip_warp_bilinear_dev", nrows, ncols, nframes));
    hipLaunchKernelGGL(k_warp_bilinear, pixel_grid(nrows, ncols, 1), dim3(256), 0, static_cast<hipStream_t>(stream),
                       Iout, Iin, X, Y, nrows, ncols, nframes);
    HIPCHK(hipGetLastError());
    return PDEIP_OK;
}

static int check_deriv_dims(const char *who, int nrows, int ncols, int nframes)
{
    RC(check_dims(who, nrows, ncols, nframes));
    if (nrows < 4 || ncols < 4) return set_err(PDEIP_ERR_ARG, "%s: the 5-tap filters need at least 4x4 pixels (got %dx%d)", who, nrows, ncols);
    return PDEIP_OK;
}

extern "C" int pdeip_fst_derivatives5_dev(void *stream, const float *It0, const float *It1, int nrows, int ncols,
                                          int nframes, float *Idt, float *Idx, float *Idy)
{
    RC(check_deriv_dims("pdeip_fst_derivatives5_dev", nrows, ncols, nframes));
    hipLaunchKernelGGL(k_fst_derivatives5, pixel_grid(nrows, ncols, nframes), dim3(256), 0, static_cast<hipStream_t>(stream),
                       Idt, Idx, Idy, It0, It1, nrows, ncols, (size_t)nrows * ncols);
    HIPCHK(hipGetLastError());
    return PDEIP_OK;
}

extern "C" int pdeip_snd_derivatives5_dev(void *stream, const float *It0, const float *It1, int nrows, int ncols,
                                          int nframes, float *Idxt, float *Idyt, float *Idxx, float *Idyy, float *Idxy)
{
    RC(check_deriv_dims("pdeip_snd_derivatives5_dev", nrows, ncols, nframes));
    hipLaunchKernelGGL(k_snd_derivatives5, pixel_grid(nrows, ncols, nframes), dim3(256), 0, static_cast<hipStream_t>(stream),
                       Idxt, Idyt, Idxx, Idyy, Idxy, It0, It1, nrows, ncols, (size_t)nrows * ncols);
    HIPCHK(hipGetLastError());
    return PDEIP_OK;
}

// ------------------------------------------------------------------------------------------------
// MATLAB-side stages of one late-linearisation pyramid level (pdeip_flow.hpp), device-resident only
// ------------------------------------------------------------------------------------------------
extern "C" int pdeip_flow_coords_dev(void *stream, const float *U, const float *V, int nrows, int ncols, float *X, float *Y)
{
    RC(check_dims("pdeip_flow_coords_dev", nrows, ncols, 1));
    hipLaunchKernelGGL(k_flow_coords, pixel_grid(nrows, ncols, 1), dim3(256), 0, static_cast<hipStream_t>(stream), X, Y, U, V, nrows, ncols);
    HIPCHK(hipGetLastError());
    return PDEIP_OK;
}

extern "C" int pdeip_flow_assemble_dev(void *stream, const float *It1, const float *Ix1, const float *Iy1, int C1, float b1,
                                       const float *It2, const float *Ix2, const float *Iy2, int C2, float b2, const float *dU,
                                       const float *dV, float alpha, int nrows, int ncols, float *MGd, float *CuGd, float *CvGd,
                                       float *DuGd, float *DvGd)
{
    const char *who = "pdeip_flow_assemble_dev";
    RC(check_dims(who, nrows, ncols, C1));
    if (C2 < 0 || (C2 > 0 && (!It2 || !Ix2 || !Iy2))) return set_err(PDEIP_ERR_ARG, "%s: second data term needs its three derivative arrays", who);
    const FlowTerm t1{It1, Ix1, Iy1, C1, b1}, t2{It2, Ix2, Iy2, C2, b2};
    hipLaunchKernelGGL(k_flow_assemble, pixel_grid(nrows, ncols, 1), dim3(256), 0, static_cast<hipStream_t>(stream), MGd, CuGd, CvGd, DuGd,
                       DvGd, t1, t2, dU, dV, alpha, nrows, ncols);
    HIPCHK(hipGetLastError());
    return PDEIP_OK;
}

extern "C" int pdeip_disp_assemble_dev(void *stream, const float *It1, const float *Ix1, int C1, float b1, const float *It2,
                                       const float *Ix2, int C2, float b2, const float *dU, float alpha, int nrows, int ncols,
                                       float *CuGd, float *DuGd)
{
    const char *who = "pdeip_disp_assemble_dev";
    RC(check_dims(who, nrows, ncols, C1));
    if (C2 < 0 || (C2 > 0 && (!It2 || !Ix2))) return set_err(PDEIP_ERR_ARG, "%s: second data term needs its derivative arrays", who);
    const FlowTerm t1{It1, Ix1, nullptr, C1, b1}, t2{It2, Ix2, nullptr, C2, b2};
    hipLaunchKernelGGL(k_disp_assemble, pixel_grid(nrows, ncols, 1), dim3(256), 0, static_cast<hipStream_t>(stream), CuGd, DuGd, t1, t2, dU,
                       alpha, nrows, ncols);
    HIPCHK(hipGetLastError());
    return PDEIP_OK;
}

extern "C" int pdeip_flow_assemble_gradmag_dev(void *stream, const float *It1, const float *Ix1, const float *Iy1, int C1, float b1,
                                               const float *Ixt, const float *Iyt, const float *Ixx, const float *Iyy, const float *Ixy, int C2,
                                               float b2, const float *dU, const float *dV, float alpha, int nrows, int ncols, float *MGd,
                                               float *CuGd, float *CvGd, float *DuGd, float *DvGd)
{
    const char *who = "pdeip_flow_assemble_gradmag_dev";
    RC(check_dims(who, nrows, ncols, C1));
    if (C2 < 1 || !Ixt || !Iyt || !Ixx || !Iyy || !Ixy) return set_err(PDEIP_ERR_ARG, "%s: the gradient-magnitude term needs its five derivative arrays", who);
    const FlowTerm t1{It1, Ix1, Iy1, C1, b1}, t2{Ixt, Iyt, Ixx, C2, b2, Iyy, Ixy};
    hipLaunchKernelGGL(k_flow_assemble, pixel_grid(nrows, ncols, 1), dim3(256), 0, static_cast<hipStream_t>(stream), MGd, CuGd, CvGd, DuGd,
                       DvGd, t1, t2, dU, dV, alpha, nrows, ncols);
    HIPCHK(hipGetLastError());
    return PDEIP_OK;
}

extern "C" int pdeip_disp_assemble_gradmag_dev(void *stream, const float *It1, const float *Ix1, int C1, float b1, const float *Ixt,
                                               const float *Iyt, const float *Ixx, const float *Ixy, int C2, float b2, const float *dU, float alpha,
                                               int nrows, int ncols, float *CuGd, float *DuGd)
{
    const char *who = "pdeip_disp_assemble_gradmag_dev";
    RC(check_dims(who, nrows, ncols, C1));
    if (C2 < 1 || !Ixt || !Iyt || !Ixx || !Ixy) return set_err(PDEIP_ERR_ARG, "%s: the gradient-magnitude term needs its four derivative arrays", who);
    const FlowTerm t1{It1, Ix1, nullptr, C1, b1}, t2{Ixt, Iyt, Ixx, C2, b2, nullptr, Ixy};
    hipLaunchKernelGGL(k_disp_assemble, pixel_grid(nrows, ncols, 1), dim3(256), 0, static_cast<hipStream_t>(stream), CuGd, DuGd, t1, t2, dU,
                       alpha, nrows, ncols);
    HIPCHK(hipGetLastError());
    return PDEIP_OK;
}

extern "C" int pdeip_flow_apriori_dev(void *stream, const double *Us, const float *U, const float *dU, double gammaS, double alpha,
                                      double as_diff, int u_double, int du_double, int nrows, int ncols, float *CGd, float *DGd)
{
    RC(check_dims("pdeip_flow_apriori_dev", nrows, ncols, 1));
    if (!Us || !U || !dU || !CGd || !DGd) return set_err(PDEIP_ERR_ARG, "pdeip_flow_apriori_dev: null plane");
    hipLaunchKernelGGL(k_flow_apriori, pixel_grid(nrows, ncols, 1), dim3(256), 0, static_cast<hipStream_t>(stream), CGd, DGd, Us, U, dU, gammaS,
                       alpha, as_diff * as_diff, u_double, du_double, nrows, ncols);
    HIPCHK(hipGetLastError());
    return PDEIP_OK;
}

extern "C" int pdeip_rgb2grad_dev(void *stream, const float *in, int nrows, int ncols, int nframes, float *out)
{
    RC(check_dims("pdeip_rgb2grad_dev", nrows, ncols, nframes));
    if (in == out) return set_err(PDEIP_ERR_ARG, "pdeip_rgb2grad_dev: output must not alias the input");
    hipLaunchKernelGGL(k_rgb2grad, pixel_grid(nrows, ncols, nframes), dim3(256), 0, static_cast<hipStream_t>(stream), out, in, nrows, ncols);
    HIPCHK(hipGetLastError());
    return PDEIP_OK;
}

extern "C" int pdeip_add_dev(void *stream, const float *A, const float *B, int nrows, int ncols, float *out)
{
    RC(check_dims("pdeip_add_dev", nrows, ncols, 1));
    hipLaunchKernelGGL(k_add2, pixel_grid(nrows, ncols, 1), dim3(256), 0, static_cast<hipStream_t>(stream), out, A, B, nrows, ncols);
    HIPCHK(hipGetLastError());
    return PDEIP_OK;
}

extern "C" int pdeip_hs_assemble_dev(void *stream, const float *It0, const float *It1, int C, float b1, float b2, int nrows, int ncols,
                                     float *MGd, float *CuGd, float *CvGd, float *DuGd, float *DvGd)
{
    RC(check_dims("pdeip_hs_assemble_dev", nrows, ncols, C));
    hipLaunchKernelGGL(k_hs_assemble, pixel_grid(nrows, ncols, 1), dim3(256), 0, static_cast<hipStream_t>(stream), MGd, CuGd, CvGd, DuGd, DvGd,
                       It0, It1, C, b1, b2, nrows, ncols);
    HIPCHK(hipGetLastError());
    return PDEIP_OK;
}

// ---- FAS full-multigrid driver stages (pdeip_fas.hpp) ----------------------------------------------
extern "C" int pdeip_fas_gauss5_dev(void *stream, const float *in, int nrows, int ncols, int frames, const float *g25, float *out)
{
    RC(check_dims("pdeip_fas_gauss5_dev", nrows, ncols, frames));
    if (!g25 || in == out) return set_err(PDEIP_ERR_ARG, "pdeip_fas_gauss5_dev: kernel missing or output aliases the input");
    FasTaps25 T;
    for (int b = 0; b < 5; ++b)
        for (int a = 0; a < 5; ++a) T.g[b * 5 + a] = g25[(4 - b) * 5 + (4 - a)]; // 'conv' flips the kernel
    hipLaunchKernelGGL(k_fas_gauss5, pixel_grid(nrows, ncols, frames), dim3(256), 0, static_cast<hipStream_t>(stream), out, in, T, nrows, ncols);
    HIPCHK(hipGetLastError());
    return PDEIP_OK;
}

extern "C" int pdeip_fas_down_dev(void *stream, const float *in, int nrows, int ncols, int frames, float *out)
{
    RC(check_dims("pdeip_fas_down_dev", nrows, ncols, frames));
    const int nr = (nrows + 1) / 2, nc = (ncols + 1) / 2;
    hipLaunchKernelGGL(k_fas_down, pixel_grid(nr, nc, frames), dim3(256), 0, static_cast<hipStream_t>(stream), out, in, nrows, ncols, nr, nc);
    HIPCHK(hipGetLastError());
    return PDEIP_OK;
}

extern "C" int pdeip_fas_prepare_dev(void *stream, const float *It0, const float *It1, int nrows, int ncols, int frames, float b1, float b2,
                                     float *planes)
{
    RC(check_dims("pdeip_fas_prepare_dev", nrows, ncols, frames));
    hipLaunchKernelGGL(k_fas_prepare, pixel_grid(nrows, ncols, frames), dim3(256), 0, static_cast<hipStream_t>(stream), planes, It0, It1, frames,
                       b1, b2, nrows, ncols);
    HIPCHK(hipGetLastError());
    return PDEIP_OK;
}

extern "C" int pdeip_fas_assemble_dev(void *stream, const float *planes, const float *Cu, const float *Cv, const float *U, const float *V,
                                      int nrows, int ncols, int frames, float b1, float b2, float k, int per_frame, float *MGd, float *CuGd,
                                      float *CvGd, float *DuGd, float *DvGd, float *gd)
{
    RC(check_dims("pdeip_fas_assemble_dev", nrows, ncols, frames));
    if (!MGd || !DuGd || !DvGd || (Cu && !CuGd) || (Cv && !CvGd))
        return set_err(PDEIP_ERR_ARG, "pdeip_fas_assemble_dev: missing output plane");
    const auto s = static_cast<hipStream_t>(stream);
    if (per_frame)
        hipLaunchKernelGGL(k_fas_assemble<true>, pixel_grid(nrows, ncols, 1), dim3(256), 0, s, MGd, CuGd, CvGd, DuGd, DvGd, gd, planes, Cu, Cv, U,
                           V, frames, b1, b2, k, nrows, ncols);
    else
        hipLaunchKernelGGL(k_fas_assemble<false>, pixel_grid(nrows, ncols, 1), dim3(256), 0, s, MGd, CuGd, CvGd, DuGd, DvGd, gd, planes, Cu, Cv, U,
                           V, frames, b1, b2, k, nrows, ncols);
    HIPCHK(hipGetLastError());
    return PDEIP_OK;
}

extern "C" int pdeip_fas_restrict_dev(void *stream, const float *in, int nrows, int ncols, int frames, float scale, float *out)
{
    RC(check_dims("pdeip_fas_restrict_dev", nrows, ncols, frames));
    const int nr = (nrows + 1) / 2, nc = (ncols + 1) / 2;
    hipLaunchKernelGGL(k_fas_restrict, pixel_grid(nr, nc, frames), dim3(256), 0, static_cast<hipStream_t>(stream), out, in, scale, nrows, ncols,
                       nr, nc);
    HIPCHK(hipGetLastError());
    return PDEIP_OK;
}

extern "C" int pdeip_fas_rhs_dev(void *stream, const float *R, const float *A, const float *gd, int nrows, int ncols, int frames, float *out)
{
    RC(check_dims("pdeip_fas_rhs_dev", nrows, ncols, frames));
    hipLaunchKernelGGL(k_fas_rhs, pixel_grid(nrows, ncols, frames), dim3(256), 0, static_cast<hipStream_t>(stream), out, R, A, gd, nrows, ncols);
    HIPCHK(hipGetLastError());
    return PDEIP_OK;
}

extern "C" int pdeip_fas_prolong_add_dev(void *stream, float *U, int nrows, int ncols, const float *Uc, const float *Ures, int nrows_c,
                                         int ncols_c, float inv_scale)
{
    RC(check_dims("pdeip_fas_prolong_add_dev", nrows, ncols, 1));
    RC(check_dims("pdeip_fas_prolong_add_dev", nrows_c, ncols_c, 1));
    hipLaunchKernelGGL(k_fas_prolong_add, pixel_grid(nrows, ncols, 1), dim3(256), 0, static_cast<hipStream_t>(stream), U, Uc, Ures, inv_scale,
                       nrows_c, ncols_c, nrows, ncols);
    HIPCHK(hipGetLastError());
    return PDEIP_OK;
}

extern "C" int pdeip_fas_upscale_dev(void *stream, const float *in, int nrows, int ncols, float mul, int nrows_out, int ncols_out, float *out)
{
    RC(check_dims("pdeip_fas_upscale_dev", nrows, ncols, 1));
    RC(check_dims("pdeip_fas_upscale_dev", nrows_out, ncols_out, 1));
    if (nrows_out < nrows || ncols_out < ncols) return set_err(PDEIP_ERR_ARG, "pdeip_fas_upscale_dev: enlarging only");
    hipLaunchKernelGGL(k_fas_upscale, pixel_grid(nrows_out, ncols_out, 1), dim3(256), 0, static_cast<hipStream_t>(stream), out, in, mul, nrows,
                       ncols, nrows_out, ncols_out);
    HIPCHK(hipGetLastError());
    return PDEIP_OK;
}

extern "C" int pdeip_flow_opdiffweights_dev(void *stream, const float *U, const float *V, const float *dU, const float *dV, int nrows,
                                            int ncols, float *wW, float *wN, float *wS, float *wE)
{
    RC(check_dims("pdeip_flow_opdiffweights_dev", nrows, ncols, 1));
    hipLaunchKernelGGL(k_flow_opdiffweights, pixel_grid(nrows, ncols, 1), dim3(256), 0, static_cast<hipStream_t>(stream), wW, wN, wS, wE, U,
                       V, dU, dV, nrows, ncols);
    HIPCHK(hipGetLastError());
    return PDEIP_OK;
}

extern "C" int pdeip_median3_dev(void *stream, const float *A, const float *B, int nrows, int ncols, float *out)
{
    RC(check_dims("pdeip_median3_dev", nrows, ncols, 1));
    if (out == A || out == B) return set_err(PDEIP_ERR_ARG, "pdeip_median3_dev: output must not alias an input");
    hipLaunchKernelGGL(k_median3_sum, pixel_grid(nrows, ncols, 1), dim3(256), 0, static_cast<hipStream_t>(stream), out, A, B, nrows, ncols);
    HIPCHK(hipGetLastError());
    return PDEIP_OK;
}

// One lagged-diffusivity iteration's MATLAB-side work of TVdenoise8 (pdeip_tv.hpp): ADdiffWeights(Iout) incl. the
// quantile lambda, PsiData, TRACE, B and the alpha-scaled weights, ready for pdeip_pde_sor8_dev / pdeip_pde_alr8_dev.
extern "C" int pdeip_tv_assemble_dev(void *stream, const float *Iout, const float *Iin, int nrows, int ncols, int nframes,
                                     float alpha, float *TRACE, float *B, float *aW, float *aNW, float *aN, float *aNE,
                                     float *aE, float *aSE, float *aS, float *aSW)
{
    const char *who = "pdeip_tv_assemble_dev";
    RC(check_dims(who, nrows, ncols, nframes));
    hipStream_t s = static_cast<hipStream_t>(stream);
    const size_t n = (size_t)nrows * ncols;
    size_t temp_bytes = 0;
    double *nul = nullptr;
    HIPCHK(rocprim::radix_sort_keys(nullptr, temp_bytes, nul, nul, n, 0, 64, s));
    const size_t doubles = 4 * n + 2 + (temp_bytes + 7) / 8; // gx, gy, norm, sorted, lambda, sort workspace
    float *basef;
    RC(ws_get(WS_TV, doubles * sizeof(double), &basef));
    double *gx = reinterpret_cast<double *>(basef), *gy = gx + n, *nrm = gy + n, *sorted = nrm + n, *lambda = sorted + n;
    void *temp = lambda + 2;
    hipLaunchKernelGGL(k_tv_gradient, pixel_grid(nrows, ncols, 1), dim3(256), 0, s, gx, gy, nrm, Iout, nrows, ncols, nframes);
    HIPCHK(rocprim::radix_sort_keys(temp, temp_bytes, nrm, sorted, n, 0, 64, s));
    hipLaunchKernelGGL(k_tv_lambda, dim3(1), dim3(64), 0, s, lambda, sorted, n, -1.0);
    hipLaunchKernelGGL(k_tv_assemble, pixel_grid(nrows, ncols, 1), dim3(256), 0, s, TRACE, B, aW, aNW, aN, aNE, aE, aSE, aS, aSW, gx, gy,
                       nrm, lambda, Iout, Iin, alpha, nrows, ncols, nframes);
    HIPCHK(hipGetLastError());
    return PDEIP_OK;
}

// ---- image pyramid (pdeip_pyr.hpp) ---------------------------------------------------------------------
static int pyr_axis(const char *who, int n_in, int n_out, int cubic, PyrAxis *A)
{
    A->scale = (double)n_out / (double)n_in;
    A->stretch = A->scale >= 1.0 ? 1.0 : 1.0 / A->scale;
    A->width = (cubic ? 2.0 : 1.0) * A->stretch;
    A->T = (int)ceil(2.0 * A->width) + 2;
    A->cubic = cubic;
    if (A->T > PYR_TMAX) return set_err(PDEIP_ERR_UNSUPPORTED, "%s: shrinking %d -> %d needs %d taps (at most %d)", who, n_in, n_out, A->T, PYR_TMAX);
    return PDEIP_OK;
}

extern "C" int pdeip_pyr_resize_dev(void *stream, const float *in, int nrows, int ncols, int nframes, int nrows_out, int ncols_out, int cubic,
                                    float *out)
{
    const char *who = "pdeip_pyr_resize_dev";
    RC(check_dims(who, nrows, ncols, nframes));
    RC(check_dims(who, nrows_out, ncols_out, nframes));
    if (in == out) return set_err(PDEIP_ERR_ARG, "%s: output must not alias the input", who);
    PyrAxis R, C;
    RC(pyr_axis(who, nrows, nrows_out, cubic != 0, &R));
    RC(pyr_axis(who, ncols, ncols_out, cubic != 0, &C));
    hipLaunchKernelGGL(k_pyr_resize, pixel_grid(nrows_out, ncols_out, nframes), dim3(256), 0, static_cast<hipStream_t>(stream), out, in, R, C, nrows,
                       ncols, nrows_out, ncols_out);
    HIPCHK(hipGetLastError());
    return PDEIP_OK;
}

extern "C" int pdeip_pyr_smooth_dev(void *stream, const float *in, int nrows, int ncols, int nframes, const double *G, int size, float *out)
{
    const char *who = "pdeip_pyr_smooth_dev";
    RC(check_dims(who, nrows, ncols, nframes));
    if (!G || size < 1 || size > 7 || size % 2 == 0 || in == out) return set_err(PDEIP_ERR_ARG, "%s: odd mask of at most 7x7, output distinct from input", who);
    PyrMask M;
    for (int k = 0; k < size * size; k++) M.g[k] = G[k];
    M.size = size;
    hipLaunchKernelGGL(k_pyr_smooth, pixel_grid(nrows, ncols, nframes), dim3(256), 0, static_cast<hipStream_t>(stream), out, in, M, nrows, ncols);
    HIPCHK(hipGetLastError());
    return PDEIP_OK;
}

// ---- symmetric stereo driver stages (pdeip_sym.hpp) ----------------------------------------------------
extern "C" int pdeip_sym_warp_flow_dev(void *stream, const float *U, const float *Uq, int nrows, int ncols, double *out)
{
    RC(check_dims("pdeip_sym_warp_flow_dev", nrows, ncols, 1));
    if (ncols < 2) return set_err(PDEIP_ERR_ARG, "pdeip_sym_warp_flow_dev: needs at least two columns");
    hipLaunchKernelGGL(k_sym_warp_flow, pixel_grid(nrows, ncols, 1), dim3(256), 0, static_cast<hipStream_t>(stream), out, U, Uq, nrows, ncols);
    HIPCHK(hipGetLastError());
    return PDEIP_OK;
}

extern "C" int pdeip_sym_flow_terms_dev(void *stream, const float *U, const double *Uw, int nrows, int ncols, double *Udt, double *Udx,
                                        double *CuS, double *DuS)
{
    RC(check_dims("pdeip_sym_flow_terms_dev", nrows, ncols, 1));
    hipLaunchKernelGGL(k_sym_flow_terms, pixel_grid(nrows, ncols, 1), dim3(256), 0, static_cast<hipStream_t>(stream), Udt, Udx, CuS, DuS, U, Uw,
                       nrows, ncols);
    HIPCHK(hipGetLastError());
    return PDEIP_OK;
}

extern "C" int pdeip_sym_assemble_dev(void *stream, const float *Idt, const float *Idx, const float *Idxt, const float *Idyt, const float *Idxx,
                                      const float *Idxy, int C, const double *Udt, const double *Udx, const double *CuS, const double *DuS,
                                      const float *dU, float b1, float b2, float alpha, double kS, double sr2, int first, int nrows, int ncols,
                                      float *CuG, float *DuG)
{
    RC(check_dims("pdeip_sym_assemble_dev", nrows, ncols, C));
    const SymData d{Idt, Idx, Idxt, Idyt, Idxx, Idxy, C};
    hipLaunchKernelGGL(k_sym_assemble, pixel_grid(nrows, ncols, 1), dim3(256), 0, static_cast<hipStream_t>(stream), CuG, DuG, d, Udt, Udx, CuS, DuS,
                       dU, b1, b2, alpha, kS, sr2, first, nrows, ncols);
    HIPCHK(hipGetLastError());
    return PDEIP_OK;
}

extern "C" int pdeip_tv4_assemble_dev(void *stream, const float *Iout, const float *Iin, int nrows, int ncols, int nframes, float alpha,
                                      float *TRACE, float *B, float *aW, float *aN, float *aE, float *aS)
{
    RC(check_dims("pdeip_tv4_assemble_dev", nrows, ncols, nframes));
    hipLaunchKernelGGL(k_tv4_assemble, pixel_grid(nrows, ncols, 1), dim3(256), 0, static_cast<hipStream_t>(stream), TRACE, B, aW, aN, aE, aS, Iout,
                       Iin, alpha, nrows, ncols, nframes);
    HIPCHK(hipGetLastError());
    return PDEIP_OK;
}

extern "C" int pdeip_ad_weights_dev(void *stream, const float *D, int nrows, int ncols, int nframes, double quantile, float *wW, float *wNW,
                                    float *wN, float *wNE, float *wE, float *wSE, float *wS, float *wSW)
{
    const char *who = "pdeip_ad_weights_dev";
    RC(check_dims(who, nrows, ncols, nframes));
    if (!(quantile > 0.0 && quantile <= 1.0)) return set_err(PDEIP_ERR_ARG, "%s: quantile must be in (0, 1]", who);
    hipStream_t s = static_cast<hipStream_t>(stream);
    const size_t n = (size_t)nrows * ncols;
    size_t temp_bytes = 0;
    double *nul = nullptr;
    HIPCHK(rocprim::radix_sort_keys(nullptr, temp_bytes, nul, nul, n, 0, 64, s));
    const size_t doubles = 4 * n + 2 + (temp_bytes + 7) / 8;
    float *basef;
    RC(ws_get(WS_TV, doubles * sizeof(double), &basef));
    double *gx = reinterpret_cast<double *>(basef), *gy = gx + n, *nrm = gy + n, *sorted = nrm + n, *lambda = sorted + n;
    void *temp = lambda + 2;
    hipLaunchKernelGGL(k_tv_gradient, pixel_grid(nrows, ncols, 1), dim3(256), 0, s, gx, gy, nrm, D, nrows, ncols, nframes);
    HIPCHK(rocprim::radix_sort_keys(temp, temp_bytes, nrm, sorted, n, 0, 64, s));
    hipLaunchKernelGGL(k_tv_lambda, dim3(1), dim3(64), 0, s, lambda, sorted, n, quantile);
    hipLaunchKernelGGL(k_ad_weights, pixel_grid(nrows, ncols, 1), dim3(256), 0, s, wW, wNW, wN, wNE, wE, wSE, wS, wSW, gx, gy, nrm, lambda, nrows,
                       ncols);
    HIPCHK(hipGetLastError());
    return PDEIP_OK;
}

// ------------------------------------------------------------------------------------------------
// host-pointer drop-in entry points (gateway semantics)
// ------------------------------------------------------------------------------------------------

// Shared body of Oflow_sor_elin4_2d / Oflow_sor_llin4_2d / Oflow_sor_llin8_2d.
static int oflow_sor_host(const char *who, bool llin, bool fill_residuals, const float *U, const float *V,
                          const float *dU, const float *dV, const float *M, const float *Cu, const float *Cv,
                          const float *Du, const float *Dv, const float *wW, const float *wN, const float *wE,
                          const float *wS, int nrows, int ncols, int F, int iter, float omega, int solver,
                          float *o0, float *o1, float *RU, float *RV, const float *const *diag = nullptr)
{
    NONNULL(who, U); NONNULL(who, V); NONNULL(who, M); NONNULL(who, Cu); NONNULL(who, Cv); NONNULL(who, Du);
    NONNULL(who, Dv); NONNULL(who, wW); NONNULL(who, wN); NONNULL(who, wE); NONNULL(who, wS);
    NONNULL(who, o0); NONNULL(who, o1);
    if (llin) { NONNULL(who, dU); NONNULL(who, dV); }
    if ((RU == nullptr) != (RV == nullptr))
        return set_err(PDEIP_ERR_ARG, "%s: residual outputs RU and RV must be requested together", who);
    RC(check_dims(who, nrows, ncols, F));
    RC(check_solver(who, solver));
    RC(use_device());
    const size_t n = (size_t)nrows * ncols, nf = n * (size_t)F;

    Arena ar;
    RC(ar.init(pad4(n) * 16 + pad4(nf) * 7));
    float *dUin = ar.take(n), *dVin = ar.take(n), *ddU = ar.take(n), *ddV = ar.take(n);
    float *ddiag[4] = {nullptr, nullptr, nullptr, nullptr}; // wNW, wNE, wSE, wSW: only the line solvers read them
    if (diag && solver == PDEIP_SOLVER_ALR)
        for (int k = 0; k < 4; k++) {
            ddiag[k] = ar.take(n);
            RC(upload(ddiag[k], diag[k], n));
        }
    else
        diag = nullptr;
    float *dM = ar.take(nf), *dCu = ar.take(nf), *dCv = ar.take(nf), *dDu = ar.take(nf), *dDv = ar.take(nf);
    float *dwW = ar.take(n), *dwN = ar.take(n), *dwE = ar.take(n), *dwS = ar.take(n);
    float *do0 = ar.take(n), *do1 = ar.take(n), *dRU = ar.take(nf), *dRV = ar.take(nf);
    (void)ar.take(n); (void)ar.take(n);
    RC(upload(dUin, U, n)); RC(upload(dVin, V, n));
    if (llin) { RC(upload(ddU, dU, n)); RC(upload(ddV, dV, n)); }
    RC(upload(dM, M, nf)); RC(upload(dCu, Cu, nf)); RC(upload(dCv, Cv, nf)); RC(upload(dDu, Du, nf)); RC(upload(dDv, Dv, nf));
    RC(upload(dwW, wW, n)); RC(upload(dwN, wN, n)); RC(upload(dwE, wE, n)); RC(upload(dwS, wS, n));

    if (iter > 0) { // copy the iterate in, relax it in place (Oflow_sor_elin4_2d.c:341-346)
        HIPCHK(hipMemcpyAsync(do0, llin ? ddU : dUin, n * sizeof(float), hipMemcpyDeviceToDevice, 0));
        HIPCHK(hipMemcpyAsync(do1, llin ? ddV : dVin, n * sizeof(float), hipMemcpyDeviceToDevice, 0));
        if (solver == PDEIP_SOLVER_ALR && diag)
            RC(pdeip_oflow_alr_llin8_dev(nullptr, dUin, dVin, do0, do1, dM, dCu, dCv, dDu, dDv, dwW, ddiag[0], dwN, ddiag[1], dwE, ddiag[2], dwS, ddiag[3], nrows, ncols, iter, omega, g.mode));
        else if (solver == PDEIP_SOLVER_ALR && llin)
            RC(pdeip_oflow_alr_llin4_dev(nullptr, dUin, dVin, do0, do1, dM, dCu, dCv, dDu, dDv, dwW, dwN, dwE, dwS, nrows, ncols, iter, omega, g.mode));
        else if (solver == PDEIP_SOLVER_ALR)
            RC(pdeip_oflow_alr_elin4_dev(nullptr, do0, do1, dM, dCu, dCv, dDu, dDv, dwW, dwN, dwE, dwS, nrows, ncols, iter, omega, g.mode));
        else if (llin)
            RC(pdeip_oflow_sor_llin4_dev(nullptr, dUin, dVin, do0, do1, dM, dCu, dCv, dDu, dDv, dwW, dwN, dwE, dwS, nrows, ncols, iter, omega, g.mode, 0));
        else
            RC(pdeip_oflow_sor_elin4_dev(nullptr, do0, do1, dM, dCu, dCv, dDu, dDv, dwW, dwN, dwE, dwS, nrows, ncols, iter, omega, g.mode, 0));
        RC(download(o0, do0, n));
        RC(download(o1, do1, n));
        RC(pdeip_persist_error());
    } else { // outputs stay as mxCreateNumericArray made them: zero
        memset(o0, 0, n * sizeof(float));
        memset(o1, 0, n * sizeof(float));
    }
    if (RU) { // residuals of the INPUT iterate (:349-350)
        if (!fill_residuals) {
            memset(RU, 0, nf * sizeof(float));
            memset(RV, 0, nf * sizeof(float));
        } else {
            if (llin)
                RC(pdeip_oflow_res_llin4_dev(nullptr, dRU, dRV, dUin, dVin, ddU, ddV, dM, dCu, dCv, dDu, dDv, dwW, dwN, dwE, dwS, nrows, ncols, F));
            else
                RC(pdeip_oflow_res_elin4_dev(nullptr, dRU, dRV, dUin, dVin, dM, dCu, dCv, dDu, dDv, dwW, dwN, dwE, dwS, nrows, ncols, F));
            RC(download(RU, dRU, nf));
            RC(download(RV, dRV, nf));
        }
    }
    HIPCHK(hipDeviceSynchronize());
    return PDEIP_OK;
}

extern "C" int pdeip_oflow_sor_elin4(const float *U, const float *V, const float *M, const float *Cu,
                                     const float *Cv, const float *Du, const float *Dv, const float *wW,
                                     const float *wN, const float *wE, const float *wS, int nrows, int ncols,
                                     int nframes_coef, int iter, float omega, int solver, float *U_out,
                                     float *V_out, float *RU, float *RV)
{
    return oflow_sor_host("Oflow_sor_elin4_2d", false, true, U, V, nullptr, nullptr, M, Cu, Cv, Du, Dv, wW, wN, wE,
                          wS, nrows, ncols, nframes_coef, iter, omega, solver, U_out, V_out, RU, RV);
}

extern "C" int pdeip_oflow_sor_llin4(const float *U, const float *V, const float *dU, const float *dV,
                                     const float *M, const float *Cu, const float *Cv, const float *Du,
                                     const float *Dv, const float *wW, const float *wN, const float *wE,
                                     const float *wS, int nrows, int ncols, int nframes_coef, int iter,
                                     float omega, int solver, float *dU_out, float *dV_out, float *RU, float *RV)
{
    return oflow_sor_host("Oflow_sor_llin4_2d", true, true, U, V, dU, dV, M, Cu, Cv, Du, Dv, wW, wN, wE, wS, nrows,
                          ncols, nframes_coef, iter, omega, solver, dU_out, dV_out, RU, RV);
}

extern "C" int pdeip_oflow_sor_llin8(const float *U, const float *V, const float *dU, const float *dV,
                                     const float *M, const float *Cu, const float *Cv, const float *Du,
                                     const float *Dv, const float *wW, const float *wNW, const float *wN,
                                     const float *wNE, const float *wE, const float *wSE, const float *wS,
                                     const float *wSW, int nrows, int ncols, int nframes_coef, int iter,
                                     float omega, int solver, float *dU_out, float *dV_out, float *RU, float *RV)
{
    const char *who = "Oflow_sor_llin8_2d";
    NONNULL(who, wNW); NONNULL(who, wNE); NONNULL(who, wSE); NONNULL(who, wSW);
    // GS_SOR_llin8_2d never reads the diagonal weights (opticalflowSolvers.c:1550-1591) -- only the line
    // solvers do -- and the gateway leaves RU,RV unfilled (Oflow_sor_llin8_2d.c:466-488).
    const float *diag[4] = {wNW, wNE, wSE, wSW};
    return oflow_sor_host(who, true, false, U, V, dU, dV, M, Cu, Cv, Du, Dv, wW, wN, wE, wS, nrows, ncols,
                          nframes_coef, iter, omega, solver, dU_out, dV_out, RU, RV, diag);
}

static int oflow_lhs_host(const char *who, bool llin, const float *U, const float *V, const float *dU,
                          const float *dV, const float *M, const float *Du, const float *Dv, const float *wW,
                          const float *wN, const float *wE, const float *wS, int nrows, int ncols, int F,
                          float *AU, float *AV)
{
    NONNULL(who, U); NONNULL(who, V); NONNULL(who, M); NONNULL(who, Du); NONNULL(who, Dv); NONNULL(who, wW);
    NONNULL(who, wN); NONNULL(who, wE); NONNULL(who, wS); NONNULL(who, AU); NONNULL(who, AV);
    if (llin) { NONNULL(who, dU); NONNULL(who, dV); }
    RC(check_dims(who, nrows, ncols, F));
    RC(use_device());
    const size_t n = (size_t)nrows * ncols, nf = n * (size_t)F;
    Arena ar;
    RC(ar.init(pad4(n) * 8 + pad4(nf) * 5));
    float *dUin = ar.take(n), *dVin = ar.take(n), *ddU = ar.take(n), *ddV = ar.take(n);
    float *dM = ar.take(nf), *dDu = ar.take(nf), *dDv = ar.take(nf);
    float *dwW = ar.take(n), *dwN = ar.take(n), *dwE = ar.take(n), *dwS = ar.take(n);
    float *dAU = ar.take(nf), *dAV = ar.take(nf);
    RC(upload(dUin, U, n)); RC(upload(dVin, V, n));
    if (llin) { RC(upload(ddU, dU, n)); RC(upload(ddV, dV, n)); }
    RC(upload(dM, M, nf)); RC(upload(dDu, Du, nf)); RC(upload(dDv, Dv, nf));
    RC(upload(dwW, wW, n)); RC(upload(dwN, wN, n)); RC(upload(dwE, wE, n)); RC(upload(dwS, wS, n));
    if (llin) RC(pdeip_oflow_lhs_llin4_dev(nullptr, dAU, dAV, dUin, dVin, ddU, ddV, dM, dDu, dDv, dwW, dwN, dwE, dwS, nrows, ncols, F));
    else RC(pdeip_oflow_lhs_elin4_dev(nullptr, dAU, dAV, dUin, dVin, dM, dDu, dDv, dwW, dwN, dwE, dwS, nrows, ncols, F));
    RC(download(AU, dAU, nf));
    RC(download(AV, dAV, nf));
    return PDEIP_OK;
}

extern "C" int pdeip_oflow_lhs_elin4(const float *U, const float *V, const float *M, const float *Du,
                                     const float *Dv, const float *wW, const float *wN, const float *wE,
                                     const float *wS, int nrows, int ncols, int nframes_coef, float *AU, float *AV)
{
    return oflow_lhs_host("Oflow_lhs_elin4_2d", false, U, V, nullptr, nullptr, M, Du, Dv, wW, wN, wE, wS, nrows,
                          ncols, nframes_coef, AU, AV);
}

extern "C" int pdeip_oflow_lhs_llin4(const float *U, const float *V, const float *dU, const float *dV,
                                     const float *M, const float *Du, const float *Dv, const float *wW,
                                     const float *wN, const float *wE, const float *wS, int nrows, int ncols,
                                     int nframes_coef, float *AU, float *AV)
{
    return oflow_lhs_host("Oflow_lhs_llin4_2d", true, U, V, dU, dV, M, Du, Dv, wW, wN, wE, wS, nrows, ncols,
                          nframes_coef, AU, AV);
}

extern "C" int pdeip_disp_sor_llin4(const float *U, const float *dU, const float *Cu, const float *Du,
                                    const float *wW, const float *wN, const float *wE, const float *wS,
                                    int nrows, int ncols, int iter, float omega, int solver, float *dU_out, float *RU)
{
    const char *who = "Disp_sor_llin4_2d";
    NONNULL(who, U); NONNULL(who, dU); NONNULL(who, Cu); NONNULL(who, Du); NONNULL(who, wW); NONNULL(who, wN);
    NONNULL(who, wE); NONNULL(who, wS); NONNULL(who, dU_out);
    RC(check_dims(who, nrows, ncols, 1));
    RC(check_solver(who, solver));
    RC(use_device());
    const size_t n = (size_t)nrows * ncols;
    if (RU) memset(RU, 0, n * sizeof(float)); // allocated, never computed (Disp_sor_llin4_2d.c:251-281)
    if (iter <= 0) { // output stays zero (:276-280)
        memset(dU_out, 0, n * sizeof(float));
        return PDEIP_OK;
    }
    Arena ar;
    RC(ar.init(pad4(n) * 8));
    float *dUin = ar.take(n), *ddU = ar.take(n), *dCu = ar.take(n), *dDu = ar.take(n);
    float *dwW = ar.take(n), *dwN = ar.take(n), *dwE = ar.take(n), *dwS = ar.take(n);
    RC(upload(dUin, U, n)); RC(upload(ddU, dU, n)); RC(upload(dCu, Cu, n)); RC(upload(dDu, Du, n));
    RC(upload(dwW, wW, n)); RC(upload(dwN, wN, n)); RC(upload(dwE, wE, n)); RC(upload(dwS, wS, n));
    if (solver == PDEIP_SOLVER_ALR)
        RC(pdeip_disp_alr_llin4_dev(nullptr, dUin, ddU, dCu, dDu, dwW, dwN, dwE, dwS, nrows, ncols, iter, omega, g.mode));
    else
        RC(pdeip_disp_sor_llin4_dev(nullptr, dUin, ddU, dCu, dDu, dwW, dwN, dwE, dwS, nrows, ncols, iter, omega, g.mode, 0));
    RC(download(dU_out, ddU, n));
    RC(pdeip_persist_error());
    return PDEIP_OK;
}

// [dU0 dU1] = Disp_sor_llin_sym4_2d(U0,dU0,Cu0,Du0,wW0,wN0,wE0,wS0, U1,dU1,Cu1,Du1,wW1,wN1,wE1,wS1, iter,omega,solver)
// The gateway copies both increments in and solves unconditionally (Disp_sor_llin_sym4_2d.c:418-440): iter <= 0 returns copies.
extern "C" int pdeip_disp_sor_llin_sym4(const float *U0, const float *dU0, const float *Cu0, const float *Du0, const float *wW0,
                                        const float *wN0, const float *wE0, const float *wS0, const float *U1, const float *dU1,
                                        const float *Cu1, const float *Du1, const float *wW1, const float *wN1, const float *wE1,
                                        const float *wS1, int nrows, int ncols, int iter, float omega, int solver,
                                        float *dU_out0, float *dU_out1)
{
    const char *who = "Disp_sor_llin_sym4_2d";
    const float *in[16] = {U0, dU0, Cu0, Du0, wW0, wN0, wE0, wS0, U1, dU1, Cu1, Du1, wW1, wN1, wE1, wS1};
    for (int k = 0; k < 16; k++)
        if (!in[k]) return set_err(PDEIP_ERR_ARG, "%s: input %d is NULL", who, k + 1);
    NONNULL(who, dU_out0); NONNULL(who, dU_out1);
    RC(check_dims(who, nrows, ncols, 1));
    RC(check_solver(who, solver));
    RC(use_device());
    const size_t n = (size_t)nrows * ncols;
    if (iter <= 0) {
        memcpy(dU_out0, dU0, n * sizeof(float));
        memcpy(dU_out1, dU1, n * sizeof(float));
        return PDEIP_OK;
    }
    Arena ar;
    RC(ar.init(pad4(n) * 16));
    float *d[16];
    for (int k = 0; k < 16; k++) {
        d[k] = ar.take(n);
        RC(upload(d[k], in[k], n));
    }
    RC(pdeip_disp_sor_llin_sym4_dev(nullptr, d[0], d[1], d[2], d[3], d[4], d[5], d[6], d[7], d[8], d[9], d[10], d[11], d[12], d[13],
                                    d[14], d[15], nrows, ncols, iter, omega, solver, g.mode, 0));
    RC(download(dU_out0, d[1], n));
    RC(download(dU_out1, d[9], n));
    RC(pdeip_persist_error());
    return PDEIP_OK;
}

// The PDE gateways accept solver 3 and then call an unbound function pointer (PDEsolver4.c:228);
// that is rejected here like any other unknown solver.
extern "C" int pdeip_pde_sor4(const float *X, const float *TRACE, const float *B, const float *wW, const float *wN,
                              const float *wE, const float *wS, int nrows, int ncols, int nframes, int iter,
                              float omega, int solver, float *X_out)
{
    const char *who = "PDEsolver4";
    NONNULL(who, X); NONNULL(who, TRACE); NONNULL(who, B); NONNULL(who, wW); NONNULL(who, wN); NONNULL(who, wE);
    NONNULL(who, wS); NONNULL(who, X_out);
    RC(check_dims(who, nrows, ncols, nframes));
    RC(check_solver(who, solver));
    RC(use_device());
    const size_t nf = (size_t)nrows * ncols * nframes;
    if (iter <= 0) { // copy-in, zero sweeps (PDEsolver4.c:239-240)
        memcpy(X_out, X, nf * sizeof(float));
        return PDEIP_OK;
    }
    Arena ar;
    RC(ar.init(pad4(nf) * 7));
    float *dX = ar.take(nf), *dT = ar.take(nf), *dB = ar.take(nf);
    float *dwW = ar.take(nf), *dwN = ar.take(nf), *dwE = ar.take(nf), *dwS = ar.take(nf);
    RC(upload(dX, X, nf)); RC(upload(dT, TRACE, nf)); RC(upload(dB, B, nf));
    RC(upload(dwW, wW, nf)); RC(upload(dwN, wN, nf)); RC(upload(dwE, wE, nf)); RC(upload(dwS, wS, nf));
    if (solver == PDEIP_SOLVER_ALR)
        RC(pdeip_pde_alr4_dev(nullptr, dX, dT, dB, dwW, dwN, dwE, dwS, nrows, ncols, nframes, iter, omega, g.mode));
    else
        RC(pdeip_pde_sor4_dev(nullptr, dX, dT, dB, dwW, dwN, dwE, dwS, nrows, ncols, nframes, iter, omega, g.mode, 0));
    RC(download(X_out, dX, nf));
    RC(pdeip_persist_error());
    return PDEIP_OK;
}

extern "C" int pdeip_pde_sor8(const float *X, const float *TRACE, const float *B, const float *wW, const float *wNW,
                              const float *wN, const float *wNE, const float *wE, const float *wSE, const float *wS,
                              const float *wSW, int nrows, int ncols, int nframes, int iter, float omega,
                              int solver, float *X_out)
{
    const char *who = "PDEsolver8";
    NONNULL(who, X); NONNULL(who, TRACE); NONNULL(who, B); NONNULL(who, wW); NONNULL(who, wNW); NONNULL(who, wN);
    NONNULL(who, wNE); NONNULL(who, wE); NONNULL(who, wSE); NONNULL(who, wS); NONNULL(who, wSW); NONNULL(who, X_out);
    RC(check_dims(who, nrows, ncols, nframes));
    RC(check_solver(who, solver));
    RC(use_device());
    const size_t nf = (size_t)nrows * ncols * nframes;
    if (iter <= 0 && solver != PDEIP_SOLVER_ALR) { // GS_ALR_SOR_8_2d runs its one iteration regardless (pdeSolvers.c:362)
        memcpy(X_out, X, nf * sizeof(float));
        return PDEIP_OK;
    }
    Arena ar;
    RC(ar.init(pad4(nf) * 11));
    float *dX = ar.take(nf), *dT = ar.take(nf), *dB = ar.take(nf);
    const float *hw[8] = {wW, wNW, wN, wNE, wE, wSE, wS, wSW};
    float *dw[8];
    RC(upload(dX, X, nf)); RC(upload(dT, TRACE, nf)); RC(upload(dB, B, nf));
    for (int k = 0; k < 8; k++) {
        dw[k] = ar.take(nf);
        RC(upload(dw[k], hw[k], nf));
    }
    if (solver == PDEIP_SOLVER_ALR)
        RC(pdeip_pde_alr8_dev(nullptr, dX, dT, dB, dw[0], dw[1], dw[2], dw[3], dw[4], dw[5], dw[6], dw[7], nrows, ncols,
                              nframes, iter, omega, g.mode));
    else
        RC(pdeip_pde_sor8_dev(nullptr, dX, dT, dB, dw[0], dw[1], dw[2], dw[3], dw[4], dw[5], dw[6], dw[7], nrows, ncols,
                              nframes, iter, omega, g.mode, 0));
    RC(download(X_out, dX, nf));
    return PDEIP_OK;
}

extern "C" int pdeip_diffweights6(const float *D, int nrows, int ncols, int nframes, float eps, float *wW,
                                  float *wN, float *wE, float *wS)
{
    const char *who = "DdiffWeights";
    NONNULL(who, D); NONNULL(who, wW); NONNULL(who, wN); NONNULL(who, wE); NONNULL(who, wS);
    RC(check_dims(who, nrows, ncols, nframes));
    RC(use_device());
    const size_t n = (size_t)nrows * ncols, nf = n * (size_t)nframes;
    Arena ar;
    RC(ar.init(pad4(nf) + pad4(n) * 4));
    float *dD = ar.take(nf), *d0 = ar.take(n), *d1 = ar.take(n), *d2 = ar.take(n), *d3 = ar.take(n);
    RC(upload(dD, D, nf));
    RC(pdeip_diffweights6_dev(nullptr, dD, nrows, ncols, nframes, eps, d0, d1, d2, d3));
    float *outs[4] = {wW, wN, wE, wS};
    float *dev[4] = {d0, d1, d2, d3};
    for (int k = 0; k < 4; k++) {
        RC(download(outs[k], dev[k], n));
        // outputs carry D's dimensions; only frame 0 is written (DdiffWeights.c:97-138)
        if (nframes > 1) memset(outs[k] + n, 0, (nf - n) * sizeof(float));
    }
    return PDEIP_OK;
}

extern "C" int pdeip_warp_bilinear(const float *Iin, const float *X, const float *Y, int nrows, int ncols,
                                   int nframes, float *Iout)
{
    const char *who = "BilinInterp_2d";
    NONNULL(who, Iin); NONNULL(who, X); NONNULL(who, Y); NONNULL(who, Iout);
    RC(check_dims(who, nrows, ncols, nframes));
    RC(use_device());
    const size_t n = (size_t)nrows * ncols, nf = n * (size_t)nframes;
    Arena ar;
    RC(ar.init(pad4(nf) * 2 + pad4(n) * 2));
    float *dI = ar.take(nf), *dX = ar.take(n), *dY = ar.take(n), *dO = ar.take(nf);
    RC(upload(dI, Iin, nf)); RC(upload(dX, X, n)); RC(upload(dY, Y, n));
    RC(pdeip_warp_bilinear_dev(nullptr, dI, dX, dY, nrows, ncols, nframes, dO));
    RC(download(Iout, dO, nf));
    return PDEIP_OK;
}

extern "C" int pdeip_fst_derivatives5(const float *It0, const float *It1, int nrows, int ncols, int nframes,
                                      float *Idt, float *Idx, float *Idy)
{
    const char *who = "FstDerivatives5";
    NONNULL(who, It0); NONNULL(who, It1); NONNULL(who, Idt); NONNULL(who, Idx); NONNULL(who, Idy);
    RC(check_deriv_dims(who, nrows, ncols, nframes));
    RC(use_device());
    const size_t nf = (size_t)nrows * ncols * nframes;
    Arena ar;
    RC(ar.init(pad4(nf) * 5));
    float *d0 = ar.take(nf), *d1 = ar.take(nf), *o0 = ar.take(nf), *o1 = ar.take(nf), *o2 = ar.take(nf);
    RC(upload(d0, It0, nf)); RC(upload(d1, It1, nf));
    RC(pdeip_fst_derivatives5_dev(nullptr, d0, d1, nrows, ncols, nframes, o0, o1, o2));
    RC(download(Idt, o0, nf)); RC(download(Idx, o1, nf)); RC(download(Idy, o2, nf));
    return PDEIP_OK;
}

extern "C" int pdeip_snd_derivatives5(const float *It0, const float *It1, int nrows, int ncols, int nframes,
                                      float *Idxt, float *Idyt, float *Idxx, float *Idyy, float *Idxy)
{
    const char *who = "SndDerivatives5";
    NONNULL(who, It0); NONNULL(who, It1); NONNULL(who, Idxt); NONNULL(who, Idyt); NONNULL(who, Idxx); NONNULL(who, Idyy);
    NONNULL(who, Idxy);
    RC(check_deriv_dims(who, nrows, ncols, nframes));
    RC(use_device());
    const size_t nf = (size_t)nrows * ncols * nframes;
    Arena ar;
    RC(ar.init(pad4(nf) * 7));
    float *d0 = ar.take(nf), *d1 = ar.take(nf);
    float *o[5];
    for (int k = 0; k < 5; k++) o[k] = ar.take(nf);
    RC(upload(d0, It0, nf)); RC(upload(d1, It1, nf));
    RC(pdeip_snd_derivatives5_dev(nullptr, d0, d1, nrows, ncols, nframes, o[0], o[1], o[2], o[3], o[4]));
    float *h[5] = {Idxt, Idyt, Idxx, Idyy, Idxy};
    for (int k = 0; k < 5; k++) RC(download(h[k], o[k], nf));
    return PDEIP_OK;
}
