// pdeip_sor5.hip -- libpdeip.so: point SOR, 5-point models: launch logic (red-black / exact order) and the *_dev entry points.
//
// Build (build.py): hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -c, one object per translation unit.
// -ffp-contract=off is part of the parity contract: the reference is plain C built without FMA.
#include "pdeip_ctx.hpp"

#include "pdeip_models.hpp"
#include "pdeip_pointwise.hpp"
#include "pdeip_sor_exact.hpp"
#include "pdeip_walk_host.hpp"
#include "pdeip_persist_host.hpp"
#include "pdeip_sor_rb.hpp"
#include "pdeip_sor_rbp.hpp"
#include "pdeip_sor_small.hpp"

using namespace pdeip;

namespace {

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
    // waves of this kernel the device holds at once
    const int slots = resident_waves(reinterpret_cast<const void *>(&k_sor_rb<Mdl, true, false, true>), 64 * RB_WAVES_PER_BLOCK, RB_WAVES_PER_BLOCK);
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

// Strip width of the pipelined kernel (pdeip_sor_rbp.hpp): one workgroup per CU, a launch takes
// ceil(units / resident workgroups) rounds of nsteps(TJ) = TJ + 5S - 1 steps.
template <class Mdl, int S>
int pick_rbp_tj(int ncols, int nframes, int ntiles_r, const void *kernel)
{
    using L = RbpLayout<Mdl, S>;
    const int forced = env_int("PDEIP_RBP_TJ", 0);
    if (forced > 0) return forced < 2 ? 2 : forced;
    DeviceState *d = cur_dev();
    int slots;
    auto it = d->resident_waves.find(kernel);
    if (it != d->resident_waves.end()) slots = it->second;
    else {
        int blocks = 0, dev = 0;
        hipDeviceProp_t prop;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks, kernel, L::THREADS, L::LDS_BYTES) != hipSuccess) blocks = 1;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) prop.multiProcessorCount = 256;
        slots = (blocks > 0 ? blocks : 1) * prop.multiProcessorCount;
        d->resident_waves[kernel] = slots;
    }
    int best = 64;
    long best_cost = -1;
    for (int tj = 8; tj <= 1024; tj++) {
        const long units = (long)ntiles_r * ((ncols + tj - 1) / tj) * nframes;
        const long cost = ((units + slots - 1) / slots) * L::nsteps(tj);
        if (best_cost < 0 || cost < best_cost) {
            best_cost = cost;
            best = tj;
        }
    }
    return best;
}

// ------------------------------------------------------------------------------------------------
// sweep drivers (5-point models)
// ------------------------------------------------------------------------------------------------

// Runs `iter` sweeps of model Mdl on the iterate buffers P.it_out (in place from the caller's
// point of view).  P.ro and P.cf must be set, with the RAW planes in the two derived slots
// (Mdl::D0, Mdl::D1); the derived planes (divisors) are built into workspace here.
//
// `dst` (optional): NIT buffers that receive the result while the caller's iterate in P.it_out is only READ -- what a
// gateway does anyway (copy in, solve on the output: Oflow_sor_elin4_2d.c:341-346).  The red-black launches ping-pong
// between buffers, so with a separate destination the chain input -> (scratch | dst) ... -> dst needs no copy at all; in
// place, an odd number of launches ends in the scratch copy and costs one device-to-device copy of the iterate.
template <class Mdl>
int run_sweeps(hipStream_t s, SweepPlanes<Mdl> P, int nrows, int ncols, int nframes, int iter,
               float omega, int mode, int col0, float *const *dst = nullptr)
{
    constexpr int NIT = Mdl::NIT;
    const size_t n = (size_t)nrows * ncols;
    tls.last_launches = 0;
    if (dst != nullptr) {
        bool same = true;
        for (int f = 0; f < NIT; f++) same = same && dst[f] == P.it_out[f];
        if (same) dst = nullptr;
    }
    if (iter <= 0) {
        if (dst != nullptr)
            for (int f = 0; f < NIT; f++) RC(copy_d2d(s, dst[f], P.it_out[f], n * nframes));
        return PDEIP_OK;
    }
    if (dst != nullptr && mode == PDEIP_MODE_EXACT_ORDER) { // the wavefront kernels relax in place: on the destination
        for (int f = 0; f < NIT; f++) {
            RC(copy_d2d(s, dst[f], P.it_out[f], n * nframes));
            P.it_out[f] = dst[f];
        }
        dst = nullptr;
    }
    float *aux0 = nullptr, *aux1 = nullptr;
    RC(ws_get(WS_AUX0, n * nframes * sizeof(float), &aux0));
    RC(ws_get(WS_AUX1, n * nframes * sizeof(float), &aux1));

    if (mode == PDEIP_MODE_EXACT_ORDER) {
        const float *raw_cf[Mdl::NCF];
        for (int f = 0; f < Mdl::NCF; f++) raw_cf[f] = P.cf[f];
        const int A = (nrows - 2 + 63 + EX_R - 1) / EX_R;
        const int B = (ncols - 2 + 63) / 64;
        const int last_m = (A - 1) + 2 * (B - 1) + 3 * (iter - 1);
        for (int f = 0; f < NIT; f++) P.it_in[f] = P.it_out[f];
        // Launch-per-front or persistent?  The persistent form wins at every iter and frame size (tools/time_exact_persist.py:
        // 4K 2.70 vs 2.87 ms at iter=4 -- each strip has to trail its west neighbour by 64 rows plus the hand-off latency either
        // way --, 1.4x at iter=20, 1.3x at 1080p, 2.4x at 34x60: no per-front launch, sweeps overlap more tightly).
        // PDEIP_EXACT_PERSIST = 0 falls back to one launch per front.
        const bool persist = env_int("PDEIP_EXACT_PERSIST", 1) != 0;
        if (persist && B <= 0xffff && iter <= 0x7fff && n * Mdl::NCF * sizeof(float) < 0xffff0000ull && ncols <= 65535) {
            // pre-pass of the persistent form: the derived planes AND the raw ones, packed per pixel (k_pack_coefficients)
            float *pack = nullptr;
            for (int f = 0; f < Mdl::NCF; f++) P.cf[f] = raw_cf[f];
            RC(ws_get(WS_PACK, n * nframes * Mdl::NCF * sizeof(float), &pack));
            hipLaunchKernelGGL(k_pack_coefficients<Mdl>, pixel_grid(nrows, ncols, nframes), dim3(256), 0, s, P, pack, nrows, ncols, n);
            tls.last_launches++;
            // ---- persistent form: one launch, progress counters instead of one launch per front ----
            const int NC = (nrows - 2 + 63 + EX_CH - 1) / EX_CH;
            // Round 3's walker (pdeip_sor_walk.hpp, launched from pdeip_walk5.hip: LDS-DMA loader, three chunk buffers, strips of
            // W columns) is an opt-in, PDEIP_EXACT_WALK=1: measured against round 2's k_sor_exact_persist in the same runs it is
            // 15 % faster at 4K with one sweep per call, 2-5 % at iter = 4, and 5-25 % SLOWER on frames below 1080p and for the
            // single-field models (five waves and a longer prologue per walker) -- the walk is paced by what one compute unit's
            // memory pipeline takes per chunk and by the strips' start-up chain, not by the loader's instruction count
            // (exact order; profiles/NOTES.md).
            const bool walk = env_int("PDEIP_EXACT_WALK", 0) != 0;
            const int W = walk ? walk_width<Mdl>(nrows, ncols, nframes, iter) : 64;
            const int BW = (ncols - 2 + W - 1) / W;
            // schedule table, control block, mailbox: one 8-byte {value, tag} word per (frame, sweep, strip, field, row)
            PersistCtl ctl{};
            RC(persist_prepare(s, BW, iter, nframes, (size_t)nframes * iter * BW * NIT * (size_t)NC * EX_CH * sizeof(unsigned long long), &ctl));
            SweepTimer timer(s);
            if (walk) {
                RC(walk_launch<Mdl>(s, P, pack, ctl, nrows, ncols, BW, iter, NC, nframes, omega, n, W));
            } else {
                constexpr size_t plds = ExactLayout<Mdl>::LDS_BYTES + 16;
                RC(ensure_lds(reinterpret_cast<const void *>(&k_sor_exact_persist<Mdl>), plds));
                hipLaunchKernelGGL(k_sor_exact_persist<Mdl>, dim3((unsigned)(BW * iter * nframes)), dim3(exp_threads<Mdl>()), plds, s, P, pack, ctl, nrows, ncols, BW, iter, NC, nframes, omega, n);
            }
            timer.stop(1);
            tls.last_launches++;
            const int nb = 2 * ncols + 2 * (nrows - 2);
            hipLaunchKernelGGL(k_fill_borders, dim3((nb + 255) / 256, nframes, NIT), dim3(256), 0, s,
                               P.it_out[0], P.it_out[NIT - 1], NIT, nrows, ncols, n);
            tls.last_launches++;
            HIPCHK(hipGetLastError());
            return PDEIP_OK;
        }
        hipLaunchKernelGGL(k_derive<Mdl>, pixel_grid(nrows, ncols, nframes), dim3(256), 0, s, P, aux0, aux1, nrows, ncols, n);
        tls.last_launches++;
        P.cf[Mdl::D0] = aux0;
        P.cf[Mdl::D1] = aux1;
        const dim3 grid((unsigned)(B * iter), (unsigned)nframes);
        constexpr size_t lds = ExactLayout<Mdl>::LDS_BYTES;
        RC(ensure_lds(reinterpret_cast<const void *>(&k_sor_exact<Mdl>), lds)); // > 64 KiB of dynamic LDS needs an explicit opt-in
        SweepTimer timer(s);
        for (int m = 0; m <= last_m; m++) {
            hipLaunchKernelGGL(k_sor_exact<Mdl>, grid, dim3(128), lds, s, P, nrows, ncols, A, B, iter, m, omega, n);
            tls.last_launches++;
        }
        timer.stop(last_m + 1);
        const int nb = 2 * ncols + 2 * (nrows - 2);
        hipLaunchKernelGGL(k_fill_borders, dim3((nb + 255) / 256, nframes, NIT), dim3(256), 0, s,
                           P.it_out[0], P.it_out[NIT - 1], NIT, nrows, ncols, n);
        tls.last_launches++;
        HIPCHK(hipGetLastError());
        return PDEIP_OK;
    }

    // small frames: the iterate resident in LDS, one launch per (up to) four sweeps (pdeip_sor_small.hpp); PDEIP_RB_SMALL=0 disables
    const bool small_enabled = env_int("PDEIP_RB_SMALL", 1) != 0; // read per call: the tests switch it
    if (small_enabled) {
        using SL = SmallLayout<Mdl>;
        const int qpref = env_int("PDEIP_SMALL_Q", 1); // 1 measured fastest at every scale (tools/time_small.py): short workgroups beat little redundancy
        SmallPlan sp = SL::plan(nrows, ncols, iter, qpref < 1 ? 1 : qpref);
        int per_launch = iter;
        if (!sp.ok && iter > SMALL_MAX_SWEEPS) {
            sp = SL::plan(nrows, ncols, SMALL_MAX_SWEEPS, qpref < 1 ? 1 : qpref);
            per_launch = SMALL_MAX_SWEEPS;
        }
        // a cut frame relaxed in place needs every workgroup resident at once (the load counter below): at most one workgroup
        // per compute unit, with an eighth of the device left for whatever else is running
        if (sp.ok && sp.nslabs > 1) {
            DeviceState *d = cur_dev();
            if (d->num_cus == 0) {
                hipDeviceProp_t prop;
                d->num_cus = (hipGetDeviceProperties(&prop, d->device) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 1;
            }
            if ((long)sp.nslabs * nframes > (long)d->num_cus - d->num_cus / 8) sp.ok = false;
        }
        if (sp.ok) {
            RC(ensure_lds(reinterpret_cast<const void *>(&k_sor_small<Mdl>), sp.lds));
            unsigned *counter = nullptr, *abort_word = nullptr;
            if (sp.nslabs > 1) {
                float *p = nullptr;
                RC(ws_get(WS_SMALL, 64, &p));
                counter = reinterpret_cast<unsigned *>(p);
                RC(ws_get(WS_CTL, 16, &p));
                abort_word = reinterpret_cast<unsigned *>(p);
            }
            DeviceState *d = cur_dev();
            for (int f = 0; f < NIT; f++) {
                P.it_in[f] = P.it_out[f];
                if (dst) P.it_out[f] = dst[f];
            }
            SweepTimer timer(s);
            int nl = 0;
            for (int it = 0; it < iter; it += per_launch, nl++) {
                const int k = iter - it < per_launch ? iter - it : per_launch;
                const bool gated = sp.nslabs > 1 && P.it_in[0] == P.it_out[0];
                if (gated) d->persist_used = true; // a timed-out wait raises the sticky abort word
                hipLaunchKernelGGL(k_sor_small<Mdl>, dim3((unsigned)sp.nslabs, (unsigned)nframes), dim3(SL::THREADS), sp.lds, s, P, nrows, ncols, k,
                                   omega, col0, n, sp.W, gated ? counter : nullptr, abort_word);
                for (int f = 0; f < NIT; f++) P.it_in[f] = P.it_out[f]; // later launches of the call: in place on the result
            }
            timer.stop(nl);
            tls.last_launches += nl;
            HIPCHK(hipGetLastError());
            return PDEIP_OK;
        }
    }

    // red-black: ping-pong between the caller's buffers and a scratch copy
    float *scratch = nullptr;
    int rc = ws_get(WS_PING, (size_t)NIT * n * nframes * sizeof(float), &scratch);
    if (rc) return rc;
    float *bufA[NIT], *bufB[NIT], *bufD[NIT];
    bool vec = (nrows % 4 == 0);
    for (int f = 0; f < NIT; f++) {
        bufA[f] = P.it_out[f];
        bufB[f] = scratch + (size_t)f * n * nframes;
        bufD[f] = dst ? dst[f] : nullptr;
        vec = vec && aligned16(bufA[f]) && aligned16(bufB[f]) && (!dst || aligned16(bufD[f]));
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
    // Four sweeps per launch where the rings fit in LDS (pdeip_sor_rbp.hpp): the wave pipeline.  PDEIP_RB_PIPE=0 disables it.
    constexpr int PS = 4;
    const bool pipe_enabled = env_int("PDEIP_RB_PIPE", 1) != 0; // read per call: the tests switch it
    // single-field models run one wave per sweep (one wave per SIMD): the pipeline only pays on large frames there
    const bool pipe = pipe_enabled && fuse && vec && RbpLayout<Mdl, PS>::FITS && (RbpLayout<Mdl, PS>::NW == 2 || n >= (size_t)1 << 21);
    // launches of this call (the buffer chain below needs the count up front)
    int total_launches = 0;
    for (int it = 0; it < iter;) {
        const int k = (pipe && it + PS <= iter) ? PS : ((fuse && it + 2 <= iter) ? 2 : 1);
        it += k;
        total_launches++;
    }
    // launch number `flips` (0-based) reads src(flips) and writes out(flips).  In place: caller <-> scratch.  With a
    // destination: the caller's buffers are only read by launch 0, and the outputs alternate so that the last one is dst.
    auto buf_out = [&](int launch, int f) -> float * {
        if (dst) return ((total_launches - 1 - launch) & 1) ? bufB[f] : bufD[f];
        return (launch & 1) ? bufA[f] : bufB[f];
    };
    auto buf_in = [&](int launch, int f) -> const float * { return launch == 0 ? bufA[f] : buf_out(launch - 1, f); };
    SweepTimer timer(s);
    int nlaunch = 0, flips = 0; // flips: how many times the iterate changed buffers
    for (int it = 0; it < iter;) {
        if (pipe && it + PS <= iter) {
            using PL = RbpLayout<Mdl, PS>;
            const bool first = it == 0;
            const void *kfn = first ? reinterpret_cast<const void *>(&k_sor_rbp<Mdl, PS, true>) : reinterpret_cast<const void *>(&k_sor_rbp<Mdl, PS, false>);
            RC(ensure_lds(kfn, PL::LDS_BYTES));
            const int ntiles_p = (nrows + RBP_OWN_ROWS - 1) / RBP_OWN_ROWS;
            const int TJP = pick_rbp_tj<Mdl, PS>(ncols, nframes, ntiles_p, kfn);
            const int nunits = ntiles_p * ((ncols + TJP - 1) / TJP);
            for (int f = 0; f < NIT; f++) {
                P.it_in[f] = buf_in(flips, f);
                P.it_out[f] = buf_out(flips, f);
            }
            // the derived planes leave the kernel only if a later launch of this call reads them
            const bool keep = first && it + PS < iter;
            const dim3 pgrid((unsigned)nunits, (unsigned)nframes), pblock(PL::THREADS);
            // PDEIP_RBP_SERPENTINE = 1: alternate strips march backwards (k_sor_rbp, `mirror_mode`), 2: every strip (tests).  Same bits;
            // neighbouring strips then meet at the halo columns they share (-6 % bytes fetched), but at 4K the launch is not
            // faster for it (97.7 vs 94.6 us, same run) -- the wave pipeline's step, not the memory system, sets its time.  Off.
            const int serp = env_int("PDEIP_RBP_SERPENTINE", 0);
            const int mirror_mode = serp < 0 || serp > 2 ? 0 : serp;
            if (first) hipLaunchKernelGGL((k_sor_rbp<Mdl, PS, true>), pgrid, pblock, PL::LDS_BYTES, s, P, keep ? aux0 : nullptr, keep ? aux1 : nullptr, nrows, ncols, TJP, ntiles_p, nunits, omega, col0, n, mirror_mode);
            else hipLaunchKernelGGL((k_sor_rbp<Mdl, PS, false>), pgrid, pblock, PL::LDS_BYTES, s, P, nullptr, nullptr, nrows, ncols, TJP, ntiles_p, nunits, omega, col0, n, mirror_mode);
            if (first) {
                P.cf[Mdl::D0] = aux0;
                P.cf[Mdl::D1] = aux1;
            }
            it += PS;
            flips++;
            nlaunch++;
            tls.last_launches++;
            continue;
        }
        const bool two = fuse && it + 2 <= iter;
        const int TJ = two ? TJ2 : TJ1;
        const int nunits = ntiles_r * ((ncols + TJ - 1) / TJ);
        const dim3 grid((unsigned)((nunits + RB_WAVES_PER_BLOCK - 1) / RB_WAVES_PER_BLOCK), (unsigned)nframes);
        for (int f = 0; f < NIT; f++) {
            P.it_in[f] = buf_in(flips, f);
            P.it_out[f] = buf_out(flips, f);
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
        tls.last_launches++;
    }
    timer.stop(nlaunch);
    if (!dst && (flips & 1)) // in place and the last launch wrote the scratch copy
        for (int f = 0; f < NIT; f++)
            RC(copy_d2d(s, bufA[f], bufB[f], n * nframes));
    HIPCHK(hipGetLastError());
    return PDEIP_OK;
}

} // namespace

extern "C" int pdeip_debug_persist_order(int B, int T, int affine, int *table)
{
    if (B < 1 || T < 1 || table == nullptr) return set_err(PDEIP_ERR_ARG, "pdeip_debug_persist_order: bad arguments");
    RC(use_device());
    int *dev = nullptr;
    const size_t n = PERSIST_TABLE_HDR + (size_t)B * T;
    HIPCHK(hipMalloc(&dev, n * sizeof(int)));
    hipLaunchKernelGGL(k_persist_order, dim3((unsigned)((B * T + 255) / 256)), dim3(256), 0, nullptr, dev, B, T, affine ? 1 : 0);
    const hipError_t e = hipMemcpy(table, dev, n * sizeof(int), hipMemcpyDeviceToHost);
    (void)hipFree(dev);
    if (e != hipSuccess) return set_err(PDEIP_ERR_DEVICE, "pdeip_debug_persist_order: %s", hipGetErrorString(e));
    return PDEIP_OK;
}

namespace {
__global__ void k_rcp_check(unsigned long long *counts)
{
    const unsigned long long stride = (unsigned long long)gridDim.x * blockDim.x;
    unsigned long long n = 0, bad = 0;
    for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < (1ull << 32); i += stride) {
        const unsigned bits = (unsigned)i;
        const float d = __uint_as_float(bits);
        bool in_range = true;
        RcpRange{in_range}(d);
        if (in_range != (((bits >> 23) & 0xff) >= 1 && ((bits >> 23) & 0xff) <= 252)) bad++; // the range test is the exponent test
        if (!in_range) continue;
        n++;
        if (__float_as_uint(RcpFast()(d)) != __float_as_uint(RcpIeee()(d))) bad++;
    }
    atomicAdd(counts + 0, n);
    atomicAdd(counts + 1, bad);
}
} // namespace

extern "C" int pdeip_debug_rcp_check(unsigned long long *counts)
{
    if (counts == nullptr) return set_err(PDEIP_ERR_ARG, "pdeip_debug_rcp_check: null pointer");
    RC(use_device());
    unsigned long long *dev = nullptr;
    HIPCHK(hipMalloc(&dev, 2 * sizeof(unsigned long long)));
    hipError_t e = hipMemset(dev, 0, 2 * sizeof(unsigned long long));
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_rcp_check, dim3(4096), dim3(256), 0, nullptr, dev);
        e = hipMemcpy(counts, dev, 2 * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    }
    (void)hipFree(dev);
    if (e != hipSuccess) return set_err(PDEIP_ERR_DEVICE, "pdeip_debug_rcp_check: %s", hipGetErrorString(e));
    return PDEIP_OK;
}

#ifdef PDEIP_RBP_STAMPS
extern "C" int pdeip_debug_read_rbp_stamps(unsigned long long *out)
{
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_rbp_stamps), 256 * sizeof(unsigned long long)));
    return PDEIP_OK;
}
#endif
#ifdef PDEIP_P8_STAMPS
extern "C" int pdeip_debug_read_walk_stamps(unsigned long long *out)
{
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_p8_stamps), 4096 * sizeof(unsigned long long)));
    return PDEIP_OK;
}
#endif
#ifdef PDEIP_EXACT_STAMPS
extern "C" int pdeip_debug_read_stamps(unsigned long long *out)
{
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_exact_stamps), 64 * sizeof(unsigned long long)));
    return PDEIP_OK;
}
#endif

// ------------------------------------------------------------------------------------------------
// device-pointer entry points
// ------------------------------------------------------------------------------------------------
// Every 5-point solver has two device entry points: `_dev` relaxes the iterate in place, `_dev_to` reads the iterate and writes
// the relaxed one to separate planes (iter <= 0: a copy) -- the gateway's own shape (copy in, solve on the output) and, for the
// red-black launches, the one that needs no device-to-device copy of the iterate (run_sweeps).
extern "C" int pdeip_oflow_sor_elin4_dev_to(void *stream, const float *U, const float *V, float *U_out, float *V_out, const float *M,
                                            const float *Cu, const float *Cv, const float *Du, const float *Dv, const float *wW,
                                            const float *wN, const float *wE, const float *wS, int nrows, int ncols, int iter,
                                            float omega, int mode, int col0)
{
    const char *who = "pdeip_oflow_sor_elin4_dev";
    RC(check_dims(who, nrows, ncols, 1));
    RC(check_mode(who, mode));
    hipStream_t s = static_cast<hipStream_t>(stream);
    SweepPlanes<ModelElin4> P{};
    P.it_out[0] = const_cast<float *>(U);
    P.it_out[1] = const_cast<float *>(V);
    const float *cf[9] = {M, Cu, Cv, Du, Dv, wW, wN, wE, wS}; // Du,Dv: raw planes in the divisor slots
    for (int f = 0; f < 9; f++) P.cf[f] = cf[f];
    float *const dst[2] = {U_out, V_out};
    RC(run_sweeps<ModelElin4>(s, P, nrows, ncols, 1, iter, omega, mode, col0, dst));
    return PDEIP_OK;
}
extern "C" int pdeip_oflow_sor_elin4_dev(void *stream, float *U, float *V, const float *M, const float *Cu,
                                         const float *Cv, const float *Du, const float *Dv, const float *wW,
                                         const float *wN, const float *wE, const float *wS, int nrows,
                                         int ncols, int iter, float omega, int mode, int col0)
{
    return pdeip_oflow_sor_elin4_dev_to(stream, U, V, U, V, M, Cu, Cv, Du, Dv, wW, wN, wE, wS, nrows, ncols, iter, omega, mode, col0);
}

extern "C" int pdeip_oflow_sor_llin4_dev_to(void *stream, const float *U, const float *V, const float *dU, const float *dV,
                                            float *dU_out, float *dV_out, const float *M, const float *Cu, const float *Cv,
                                            const float *Du, const float *Dv, const float *wW, const float *wN, const float *wE,
                                            const float *wS, int nrows, int ncols, int iter, float omega, int mode, int col0)
{
    const char *who = "pdeip_oflow_sor_llin4_dev";
    RC(check_dims(who, nrows, ncols, 1));
    RC(check_mode(who, mode));
    hipStream_t s = static_cast<hipStream_t>(stream);
    SweepPlanes<ModelLlin4> P{};
    P.it_out[0] = const_cast<float *>(dU);
    P.it_out[1] = const_cast<float *>(dV);
    P.ro[0] = U;
    P.ro[1] = V;
    const float *cf[9] = {M, Cu, Cv, Du, Dv, wW, wN, wE, wS}; // Du,Dv: raw planes in the divisor slots
    for (int f = 0; f < 9; f++) P.cf[f] = cf[f];
    float *const dst[2] = {dU_out, dV_out};
    RC(run_sweeps<ModelLlin4>(s, P, nrows, ncols, 1, iter, omega, mode, col0, dst));
    return PDEIP_OK;
}
extern "C" int pdeip_oflow_sor_llin4_dev(void *stream, const float *U, const float *V, float *dU, float *dV,
                                         const float *M, const float *Cu, const float *Cv, const float *Du,
                                         const float *Dv, const float *wW, const float *wN, const float *wE,
                                         const float *wS, int nrows, int ncols, int iter, float omega,
                                         int mode, int col0)
{
    return pdeip_oflow_sor_llin4_dev_to(stream, U, V, dU, dV, dU, dV, M, Cu, Cv, Du, Dv, wW, wN, wE, wS, nrows, ncols, iter, omega, mode, col0);
}

extern "C" int pdeip_disp_sor_llin4_dev_to(void *stream, const float *U, const float *dU, float *dU_out, const float *Cu,
                                           const float *Du, const float *wW, const float *wN, const float *wE, const float *wS,
                                           int nrows, int ncols, int iter, float omega, int mode, int col0)
{
    const char *who = "pdeip_disp_sor_llin4_dev";
    RC(check_dims(who, nrows, ncols, 1));
    RC(check_mode(who, mode));
    hipStream_t s = static_cast<hipStream_t>(stream);
    SweepPlanes<ModelDisp4> P{};
    P.it_out[0] = const_cast<float *>(dU);
    P.ro[0] = U;
    const float *cf[6] = {Cu, Du, wW, wN, wE, wS}; // Cu,Du: raw planes in the dividend/divisor slots
    for (int f = 0; f < 6; f++) P.cf[f] = cf[f];
    float *const dst[1] = {dU_out};
    RC(run_sweeps<ModelDisp4>(s, P, nrows, ncols, 1, iter, omega, mode, col0, dst));
    return PDEIP_OK;
}
extern "C" int pdeip_disp_sor_llin4_dev(void *stream, const float *U, float *dU, const float *Cu,
                                        const float *Du, const float *wW, const float *wN, const float *wE,
                                        const float *wS, int nrows, int ncols, int iter, float omega,
                                        int mode, int col0)
{
    return pdeip_disp_sor_llin4_dev_to(stream, U, dU, dU, Cu, Du, wW, wN, wE, wS, nrows, ncols, iter, omega, mode, col0);
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
        launches += tls.last_launches;
    }
    tls.last_launches = launches;
    return PDEIP_OK;
}

extern "C" int pdeip_pde_sor4_dev_to(void *stream, const float *X, float *X_out, const float *TRACE, const float *B, const float *wW,
                                     const float *wN, const float *wE, const float *wS, int nrows, int ncols, int nframes, int iter,
                                     float omega, int mode, int col0)
{
    const char *who = "pdeip_pde_sor4_dev";
    RC(check_dims(who, nrows, ncols, nframes));
    RC(check_mode(who, mode));
    hipStream_t s = static_cast<hipStream_t>(stream);
    SweepPlanes<ModelPde4> P{};
    P.it_out[0] = const_cast<float *>(X);
    const float *cf[6] = {B, TRACE, wW, wN, wE, wS}; // B,TRACE: raw planes in the derived slots
    for (int f = 0; f < 6; f++) P.cf[f] = cf[f];
    float *const dst[1] = {X_out};
    RC(run_sweeps<ModelPde4>(s, P, nrows, ncols, nframes, iter, omega, mode, col0, dst));
    return PDEIP_OK;
}
extern "C" int pdeip_pde_sor4_dev(void *stream, float *X, const float *TRACE, const float *B, const float *wW,
                                  const float *wN, const float *wE, const float *wS, int nrows, int ncols,
                                  int nframes, int iter, float omega, int mode, int col0)
{
    return pdeip_pde_sor4_dev_to(stream, X, X, TRACE, B, wW, wN, wE, wS, nrows, ncols, nframes, iter, omega, mode, col0);
}

