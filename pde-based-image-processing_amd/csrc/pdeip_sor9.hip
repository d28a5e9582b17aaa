// pdeip_sor9.hip -- libpdeip.so: point SOR, 9-point model (PDEsolver8): launch logic and the *_dev entry point.
//
// Build (build.py): hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -c, one object per translation unit.
// -ffp-contract=off is part of the parity contract: the reference is plain C built without FMA.
#include "pdeip_ctx.hpp"

#include "pdeip_models.hpp"
#include "pdeip_pointwise.hpp"
#include "pdeip_sor_pde8.hpp"
#include "pdeip_sor_pde8_persist.hpp"
#include "pdeip_persist_host.hpp"
#include <vector>
#include "pdeip_sor_rb.hpp"

using namespace pdeip;

extern "C" int pdeip_pde_sor8_dev(void *stream, float *X, const float *TRACE, const float *B, const float *wW,
                                  const float *wNW, const float *wN, const float *wNE, const float *wE,
                                  const float *wSE, const float *wS, const float *wSW, int nrows, int ncols,
                                  int nframes, int iter, float omega, int mode, int col0)
{
    const char *who = "pdeip_pde_sor8_dev";
    RC(check_dims(who, nrows, ncols, nframes));
    RC(check_mode(who, mode));
    hipStream_t s = static_cast<hipStream_t>(stream);
    tls.last_launches = 0;
    if (iter <= 0) return PDEIP_OK;
    const size_t n = (size_t)nrows * ncols, nf = n * (size_t)nframes;
    float *bt, *inv, *scratch = nullptr;
    RC(ws_get(WS_AUX0, nf * sizeof(float), &bt));
    RC(ws_get(WS_AUX1, nf * sizeof(float), &inv));
    if (mode == PDEIP_MODE_EXACT_ORDER) {
        Pde8Planes P{};
        P.x = X;
        const float *cf[ModelPde8::NCF] = {bt, inv, wW, wNW, wN, wNE, wE, wSE, wS, wSW};
        for (int f = 0; f < ModelPde8::NCF; f++) P.cf[f] = cf[f];
        RC(ws_get(WS_PING, pde8_exact_scratch_floats(nrows, ncols, nframes, iter) * sizeof(float), &scratch));
        // One launch per call (progress counters) or one per front?  PDEIP_PDE8_PERSIST = 0 keeps the launch-per-front form.
        const int nstrips = (ncols - 2 + 63) / 64;
        const bool persist = env_int("PDEIP_PDE8_PERSIST", 1) != 0;
        const size_t pack_frame_bytes = pde8_pack_floats(nrows, ncols, 1) * sizeof(float);
        if (persist && nstrips <= 0xffff && iter <= 0x7fff && n * sizeof(float) < 0xffff0000ull && pack_frame_bytes < 0xffff0000ull && ncols <= 65535) {
            // pre-pass: B_temp / INV_TRACE and the eight weights of a pixel side by side (k_pde8_pack)
            float *pack = nullptr;
            RC(ws_get(WS_PACK, pack_frame_bytes * nframes, &pack));
            const int nbk = pde8_pack_blocks(nrows);
            hipLaunchKernelGGL(k_pde8_pack, dim3((unsigned)((nbk + 127) / 128), (unsigned)ncols, (unsigned)nframes), dim3(128), 0, s, pack, TRACE, B, wW,
                               wNW, wN, wNE, wE, wSE, wS, wSW, nrows, ncols, n);
            tls.last_launches++;
            // schedule table, control block, mailbox: one 8-byte {value, tag} word per (frame, sweep, strip, step of the walk)
            PersistCtl ctl{};
            RC(persist_prepare(s, nstrips, iter, nframes, (size_t)nframes * iter * nstrips * (size_t)pde8_persist_chunks(nrows) * EX_CH * sizeof(unsigned long long), &ctl));
            SweepTimer timer(s);
            const int nl = pde8_run_exact_persist(s, P, pack, scratch, ctl, nrows, ncols, nframes, iter, omega);
            if (nl < 0) return PDEIP_ERR_DEVICE;
            timer.stop(1);
            tls.last_launches += nl;
            HIPCHK(hipGetLastError());
            return PDEIP_OK;
        }
        hipLaunchKernelGGL(k_pde8_divisors, pixel_grid(nrows, ncols, nframes), dim3(256), 0, s, bt, inv, TRACE, B, wW, wNW, wN, wNE, wE, wSE, wS, wSW, nrows, ncols, n);
        tls.last_launches++;
        SweepTimer timer(s);
        const int nl = pde8_run_exact(s, P, scratch, nrows, ncols, nframes, iter, omega);
        if (nl < 0) return PDEIP_ERR_DEVICE; // LDS opt-in refused (message set by ensure_lds)
        timer.stop(nl);
        tls.last_launches += nl;
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
        // resident waves of the fused kernel (see pick_rb2_tj)
        const int slots = resident_waves(reinterpret_cast<const void *>(&k_pde8_colour2<true, false>), 64 * RB_WAVES_PER_BLOCK, RB_WAVES_PER_BLOCK);
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
        tls.last_launches++;
    }
    timer.stop(nlaunch);
    if (flips & 1) RC(copy_d2d(s, X, scratch, nf));
    HIPCHK(hipGetLastError());
    return PDEIP_OK;
}

#ifdef PDEIP_P8_STAMPS
extern "C" int pdeip_debug_read_p8_stamps(unsigned long long *out)
{
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_p8_stamps), 4096 * sizeof(unsigned long long)));
    return PDEIP_OK;
}
#endif
