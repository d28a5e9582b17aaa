// pdeip_multi.hip -- one frame across the devices of pdeip_set_devices(), inside one process (host entry points).
//
// The reference's callers are single-threaded MATLAB sessions: if more than one GPU is to work on a frame, the split has to
// happen behind the C-ABI.  A red-black point-SOR call of `iter` sweeps moves information by two columns per sweep (5-point:
// red from old neighbours, black from red; 9-point four-colour: even columns from old odd ones, odd columns from the new
// even ones), so a slab of consecutive MATLAB columns that carries H = 2*iter extra columns on each cut side can run the
// WHOLE call with no communication and still produce bit-exact values in the columns it owns (the halo's outer columns go
// stale; they are never copied back).  The planes are column-major, so a slab is one contiguous byte range of every host
// plane: each device copies its slab (+ halo) up, relaxes it with the ordinary single-device kernels (`col0` keeps the
// colour parity global) and copies its owned columns down.  Nothing is exchanged between devices: for a host-pointer call
// the upload IS the halo refresh.  What the split buys a MEX caller is mostly PCIe: the call is bound by moving 13-17
// planes over the link, and every device has its own link.
//
// One host thread per device (pageable-memory copies block the calling thread, so the devices would otherwise take turns);
// several slabs on one device (PDEIP_VIRTUAL_SLABS, a test knob) run one after the other in that device's thread.  The
// workers call the ordinary *_dev entry points concurrently: what those touch besides their device's state is per thread
// (ThreadState: error text, launch counter) or atomic / locked (workspace generation, profiling event slots).
// Exact-order calls and line relaxation do not decompose (their dependency front crosses the frame): they run on the first
// device of the group.  No slab narrower than its halo: the group is cut down until every slab is at least H + 1 wide.
#include "pdeip_ctx.hpp"

#include <thread>

using namespace pdeip;

namespace pdeip {

constexpr int OVERLAP_SLABS = 6;
struct SlabPlan {
    int nslabs = 1, halo = 0, nrows = 0;
    bool overlap = false;
    int c0[MAX_DEVICES], c1[MAX_DEVICES], lo[MAX_DEVICES], hi[MAX_DEVICES], dev[MAX_DEVICES];
};

// How many slabs the group wants for a frame of `ncols` columns and a halo of `halo`; 1 = no split.
int multi_plan(int ncols, int halo, SlabPlan *plan)
{
    read_env_once();
    int want = g.ngroup;
    const int virt = env_int("PDEIP_VIRTUAL_SLABS", 0); // testing: that many slabs, dealt round-robin over the group
    if (virt > want) want = virt;
    // One device and a large frame: the call is a PCIe transfer (13-17 planes up, 1-2 down) with a short kernel in between.
    // Cut into OVERLAP_SLABS column slabs on two worker threads with a stream each, the upload of one slab runs beside the
    // sweeps and the download of another (the link is full duplex): PDEIP_HOST_OVERLAP=0 keeps the single serial pass.
    plan->overlap = false;
    if (want == 1 && virt <= 0 && env_int("PDEIP_HOST_OVERLAP", 1) != 0 && (long)ncols * plan->nrows >= (1L << 21) && ncols / OVERLAP_SLABS >= 4 * (halo + 1)) {
        want = OVERLAP_SLABS;
        plan->overlap = true;
    }
    if (want > MAX_DEVICES) want = MAX_DEVICES;
    while (want > 1 && ncols / want < halo + 1) want--; // every slab wider than its halo
    plan->nslabs = want;
    plan->halo = halo;
    const int base = ncols / want, rem = ncols % want;
    int c = 0;
    for (int k = 0; k < want; k++) {
        const int w = base + (k < rem ? 1 : 0);
        plan->c0[k] = c;
        plan->c1[k] = c + w;
        plan->lo[k] = c - halo > 0 ? c - halo : 0;
        plan->hi[k] = c + w + halo < ncols ? c + w + halo : ncols;
        plan->dev[k] = g.group[k % g.ngroup];
        c += w;
    }
    return want;
}

// One solver call on one slab (MultiCall: pdeip_ctx.hpp).  Runs in the calling thread on the current device.
static int slab_run(const MultiCall &mc, const SlabPlan &plan, int k, hipStream_t st)
{
    const int lo = plan.lo[k], hi = plan.hi[k], ncl = hi - lo, nrows = mc.nrows, F = mc.frames;
    const size_t full = (size_t)nrows * mc.ncols, slab = (size_t)nrows * ncl;
    const int nplanes = 2 * mc.n_it + mc.n_ro + mc.n_cf;
    float *base = nullptr;
    RC(ws_get(WS_ARENA, pad4(slab * F) * nplanes * sizeof(float), &base));
    size_t used = 0;
    auto take = [&]() {
        float *p = base + used;
        used += pad4(slab * F);
        return p;
    };
    auto up = [&](float *dst, const float *src) -> int { // columns [lo, hi) of every frame
        for (int f = 0; f < F; f++)
            HIPCHK(hipMemcpyAsync(dst + (size_t)f * slab, src + (size_t)f * full + (size_t)lo * nrows, slab * sizeof(float), hipMemcpyHostToDevice, st));
        return PDEIP_OK;
    };
    float *d_in[2] = {nullptr, nullptr}, *d_out[2] = {nullptr, nullptr}, *d_ro[2] = {nullptr, nullptr}, *d_cf[11] = {};
    for (int p = 0; p < mc.n_it; p++) {
        d_in[p] = take();
        d_out[p] = take();
        RC(up(d_in[p], mc.it_in[p]));
    }
    for (int p = 0; p < mc.n_ro; p++) {
        d_ro[p] = take();
        RC(up(d_ro[p], mc.ro[p]));
    }
    for (int p = 0; p < mc.n_cf; p++) {
        d_cf[p] = take();
        RC(up(d_cf[p], mc.cf[p]));
    }
    const int mode = PDEIP_MODE_RED_BLACK, col0 = lo;
    switch (mc.kind) {
    case 0:
        RC(pdeip_oflow_sor_elin4_dev_to(st, d_in[0], d_in[1], d_out[0], d_out[1], d_cf[0], d_cf[1], d_cf[2], d_cf[3], d_cf[4], d_cf[5], d_cf[6],
                                        d_cf[7], d_cf[8], nrows, ncl, mc.iter, mc.omega, mode, col0));
        break;
    case 1:
        RC(pdeip_oflow_sor_llin4_dev_to(st, d_ro[0], d_ro[1], d_in[0], d_in[1], d_out[0], d_out[1], d_cf[0], d_cf[1], d_cf[2], d_cf[3], d_cf[4],
                                        d_cf[5], d_cf[6], d_cf[7], d_cf[8], nrows, ncl, mc.iter, mc.omega, mode, col0));
        break;
    case 2:
        RC(pdeip_disp_sor_llin4_dev_to(st, d_ro[0], d_in[0], d_out[0], d_cf[0], d_cf[1], d_cf[2], d_cf[3], d_cf[4], d_cf[5], nrows, ncl, mc.iter,
                                       mc.omega, mode, col0));
        break;
    case 3:
        RC(pdeip_pde_sor4_dev_to(st, d_in[0], d_out[0], d_cf[0], d_cf[1], d_cf[2], d_cf[3], d_cf[4], d_cf[5], nrows, ncl, F, mc.iter, mc.omega,
                                 mode, col0));
        break;
    case 4: // the 9-point solver relaxes in place: on a copy
        RC(copy_d2d(st, d_out[0], d_in[0], slab * F));
        RC(pdeip_pde_sor8_dev(st, d_out[0], d_cf[0], d_cf[1], d_cf[2], d_cf[3], d_cf[4], d_cf[5], d_cf[6], d_cf[7], d_cf[8], d_cf[9], nrows, ncl, F,
                              mc.iter, mc.omega, mode, col0));
        break;
    default:
        return set_err(PDEIP_ERR_ARG, "multi-device call: unknown solver kind %d", mc.kind);
    }
    // the columns this slab owns, every frame
    const int own0 = plan.c0[k] - lo, nown = plan.c1[k] - plan.c0[k];
    for (int p = 0; p < mc.n_it; p++)
        for (int f = 0; f < F; f++)
            HIPCHK(hipMemcpyAsync(mc.it_out[p] + (size_t)f * full + (size_t)plan.c0[k] * nrows, d_out[p] + (size_t)f * slab + (size_t)own0 * nrows,
                                  (size_t)nown * nrows * sizeof(float), hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st)); // the arena is reused by this worker's next slab; the caller's planes are pageable
    return PDEIP_OK;
}

// Splits one red-black solver call over the device group.  Returns PDEIP_OK and sets *handled when it ran the call;
// *handled = 0 means "not split: use the single-device path" (one device, or slabs would be narrower than their halo).
int multi_sor(const MultiCall &mc, int *handled)
{
    *handled = 0;
    if (mc.iter <= 0) return PDEIP_OK;
    SlabPlan plan;
    plan.nrows = mc.nrows;
    if (multi_plan(mc.ncols, 2 * mc.iter, &plan) <= 1) return PDEIP_OK;
    *handled = 1;
    // one thread per distinct device; each walks its slabs in order.  PDEIP_VIRTUAL_THREADS=n (testing, one-device group
    // only): the slabs are dealt over n worker threads on that device instead, each with a device-state slot of its own
    // (workspace, caches) -- the threaded path with everything it shares, on a one-GPU box.
    int rc[MAX_DEVICES] = {}, launches[MAX_DEVICES] = {};
    char errs[MAX_DEVICES][256] = {};
    std::thread th[MAX_DEVICES];
    int nth = 0, devs[MAX_DEVICES], slot[MAX_DEVICES], owner[MAX_DEVICES];
    int vthreads = g.ngroup == 1 ? env_int("PDEIP_VIRTUAL_THREADS", plan.overlap ? 2 : 0) : 0;
    if (vthreads > plan.nslabs) vthreads = plan.nslabs;
    if (vthreads > MAX_DEVICES / 2) vthreads = MAX_DEVICES / 2;
    if (vthreads > 1) {
        nth = vthreads;
        for (int ti = 0; ti < nth; ti++) {
            devs[ti] = g.group[0];
            slot[ti] = ti == 0 ? -1 : MAX_DEVICES - ti; // the first worker keeps the device's own state
        }
        for (int k = 0; k < plan.nslabs; k++) owner[k] = k % nth;
    } else {
        for (int k = 0; k < plan.nslabs; k++) {
            int j = 0;
            while (j < nth && devs[j] != plan.dev[k]) j++;
            if (j == nth) {
                devs[nth] = plan.dev[k];
                slot[nth++] = -1;
            }
            owner[k] = j;
        }
    }
    auto work = [&](int ti) {
        if (hipSetDevice(devs[ti]) != hipSuccess) {
            rc[ti] = PDEIP_ERR_DEVICE;
            snprintf(errs[ti], sizeof errs[ti], "multi-device call: hipSetDevice(%d) failed", devs[ti]);
            return;
        }
        const int keep_slot = tls.dev_slot;
        tls.dev_slot = slot[ti];
        DeviceState *ds = cur_dev();
        ds->device = devs[ti];
        if (ds->worker_stream == nullptr && hipStreamCreateWithFlags(&ds->worker_stream, hipStreamNonBlocking) != hipSuccess) {
            rc[ti] = PDEIP_ERR_DEVICE;
            snprintf(errs[ti], sizeof errs[ti], "multi-device call: cannot create a stream on device %d", devs[ti]);
            tls.dev_slot = keep_slot;
            return;
        }
        for (int k = 0; k < plan.nslabs && rc[ti] == PDEIP_OK; k++)
            if (owner[k] == ti) {
                rc[ti] = slab_run(mc, plan, k, ds->worker_stream);
                launches[ti] += tls.last_launches;
                if (rc[ti] != PDEIP_OK) snprintf(errs[ti], sizeof errs[ti], "%s", tls.err); // this thread's own text
            }
        if (rc[ti] == PDEIP_OK && hipStreamSynchronize(ds->worker_stream) != hipSuccess) rc[ti] = PDEIP_ERR_DEVICE;
        tls.dev_slot = keep_slot;
    };
    if (nth == 1) work(0); // virtual slabs on one device: no thread needed
    else {
        for (int ti = 0; ti < nth; ti++) th[ti] = std::thread(work, ti);
        for (int ti = 0; ti < nth; ti++) th[ti].join();
    }
    (void)hipSetDevice(g.group[0]);
    tls.last_launches = 0;
    for (int ti = 0; ti < nth; ti++) tls.last_launches += launches[ti];
    for (int ti = 0; ti < nth; ti++)
        if (rc[ti] != PDEIP_OK) return set_err(rc[ti], "%s", errs[ti][0] ? errs[ti] : "multi-device call failed");
    return PDEIP_OK;
}

} // namespace pdeip
