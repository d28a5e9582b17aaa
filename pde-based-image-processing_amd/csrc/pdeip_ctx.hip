// pdeip_ctx.hip -- libpdeip.so: the process-wide context, per-device state and the library-state part of the
// C-ABI (include/pdeip.h).  The kernel families live in their own translation units (build.py).
#include "pdeip_ctx.hpp"

namespace pdeip {

Context g;
thread_local ThreadState tls;

int set_err(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(tls.err, sizeof tls.err, fmt, ap);
    va_end(ap);
    return code;
}

SweepTimer::SweepTimer(hipStream_t stream) : s(stream)
{
    if (!g.profile) return;
    std::lock_guard<std::mutex> lock(g.ev_mutex);
    if (g.n_ev >= Context::MAX_EV) return;
    if (g.n_ev == g.n_ev_created) {
        if (hipEventCreate(&g.ev[g.n_ev][0]) != hipSuccess || hipEventCreate(&g.ev[g.n_ev][1]) != hipSuccess) return;
        g.n_ev_created++;
    }
    slot = g.n_ev++;
    (void)hipEventRecord(g.ev[slot][0], s);
}
void SweepTimer::stop(int launches)
{
    if (slot < 0) return;
    (void)hipEventRecord(g.ev[slot][1], s);
    g.ev_launches[slot] = launches;
}

// The persistent exact-order kernel raises an abort word in ws[WS_CTL] when a bounded dependency wait times
// out.  Before that buffer is freed or replaced the word is read and latched on the host, so the failure
// survives until pdeip_persist_error() reports it.
static void latch_abort(DeviceState *d)
{
    if (!d->ws[WS_CTL] || !d->persist_used) return;
    unsigned word = 0; // word 0 of the control block is the abort word (run_sweeps never clears it)
    if (hipDeviceSynchronize() == hipSuccess && hipMemcpy(&word, d->ws[WS_CTL], sizeof word, hipMemcpyDeviceToHost) == hipSuccess && word != 0)
        d->abort_latched = true;
}

DeviceState *cur_dev()
{
    if (tls.dev_slot >= 0 && tls.dev_slot < MAX_DEVICES) return &g.devs[tls.dev_slot]; // its `device` was set by whoever chose the slot
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAX_DEVICES) dev = 0;
    g.devs[dev].device = dev;
    return &g.devs[dev];
}

int ws_get(int slot, size_t bytes, float **out)
{
    DeviceState *d = cur_dev();
    if (d->ws_bytes[slot] < bytes) {
        HIPCHK(hipDeviceSynchronize());
        if (slot == WS_CTL) latch_abort(d);
        if (slot == WS_ORDER) d->order_B = d->order_T = 0; // the cached schedule table goes with its buffer
        if (d->ws[slot]) {
            HIPCHK(hipFree(d->ws[slot]));
            g.ws_generation++;
        }
        d->ws[slot] = nullptr;
        d->ws_bytes[slot] = 0;
        hipError_t e = hipMalloc(&d->ws[slot], bytes);
        if (e != hipSuccess) return set_err(PDEIP_ERR_NOMEM, "hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
        d->ws_bytes[slot] = bytes;
        if (slot == WS_CTL || slot == WS_SMALL) HIPCHK(hipMemset(d->ws[slot], 0, bytes)); // the abort word / load counter start clear
    }
    *out = static_cast<float *>(d->ws[slot]);
    return PDEIP_OK;
}

namespace {
__global__ void k_copy_d2d(float *__restrict__ dst, const float *__restrict__ src, size_t n, int vec)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x, tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (vec) {
        const size_t n4 = n >> 2;
        const float4 *s4 = reinterpret_cast<const float4 *>(src);
        float4 *d4 = reinterpret_cast<float4 *>(dst);
        for (size_t i = tid; i < n4; i += stride) d4[i] = s4[i];
        for (size_t i = (n4 << 2) + tid; i < n; i += stride) dst[i] = src[i];
    } else {
        for (size_t i = tid; i < n; i += stride) dst[i] = src[i];
    }
}
} // namespace

int copy_d2d(hipStream_t s, float *dst, const float *src, size_t n)
{
    if (n == 0 || dst == src) return PDEIP_OK;
    const int vec = aligned16(dst) && aligned16(src) ? 1 : 0;
    const size_t work = vec ? (n + 3) / 4 : n;
    size_t blocks = (work + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16; // 16 workgroups of 256 threads per compute unit, grid-stride beyond
    hipLaunchKernelGGL(k_copy_d2d, dim3((unsigned)blocks), dim3(256), 0, s, dst, src, n, vec);
    HIPCHK(hipGetLastError());
    return PDEIP_OK;
}

int ensure_lds(const void *kernel, size_t bytes)
{
    DeviceState *d = cur_dev();
    auto it = d->lds_opt_in.find(kernel);
    if (it != d->lds_opt_in.end() && it->second >= bytes) return PDEIP_OK;
    HIPCHK(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    d->lds_opt_in[kernel] = bytes;
    return PDEIP_OK;
}

int resident_waves(const void *kernel, int block_threads, int waves_per_block)
{
    DeviceState *d = cur_dev();
    auto it = d->resident_waves.find(kernel);
    if (it != d->resident_waves.end()) return it->second;
    int blocks = 0, dev = 0;
    hipDeviceProp_t prop;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks, kernel, block_threads, 0) != hipSuccess) blocks = 1;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) prop.multiProcessorCount = 256;
    const int slots = (blocks > 0 ? blocks : 1) * waves_per_block * prop.multiProcessorCount;
    d->resident_waves[kernel] = slots;
    return slots;
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

int env_int(const char *name, int dflt)
{
    const char *s = getenv(name);
    return (s && *s) ? atoi(s) : dflt;
}

int check_deriv_dims(const char *who, int nrows, int ncols, int nframes)
{
    RC(check_dims(who, nrows, ncols, nframes));
    if (nrows < 4 || ncols < 4) return set_err(PDEIP_ERR_ARG, "%s: the 5-tap filters need at least 4x4 pixels (got %dx%d)", who, nrows, ncols);
    return PDEIP_OK;
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


// Environment knobs of an unchanged MATLAB session: read once, before the first call
// that needs them; explicit pdeip_set_mode / pdeip_set_device(s) calls made earlier win.
//   PDEIP_MODE     exact | red_black (or 0 | 1)     sweep ordering of the host entry points
//   PDEIP_DEVICE   n                                HIP device of the host entry points
//   PDEIP_DEVICES  a,b,c,...                        device group: red-black solver calls are split into column slabs
static bool mode_set_explicitly = false, devices_set_explicitly = false;

void read_env_once()
{
    if (g.env_read) return;
    g.env_read = true;
    const char *m = getenv("PDEIP_MODE");
    if (m && *m && !mode_set_explicitly) {
        if (!strcasecmp(m, "red_black") || !strcasecmp(m, "redblack") || !strcasecmp(m, "rb") || !strcmp(m, "1")) g.mode = PDEIP_MODE_RED_BLACK;
        else if (!strcasecmp(m, "exact") || !strcasecmp(m, "exact_order") || !strcmp(m, "0")) g.mode = PDEIP_MODE_EXACT_ORDER;
        else fprintf(stderr, "libpdeip: PDEIP_MODE=%s not understood (exact | red_black); keeping exact order\n", m);
    }
    if (devices_set_explicitly) return;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess) ndev = 0;
    const char *ds = getenv("PDEIP_DEVICES");
    const char *d1 = getenv("PDEIP_DEVICE");
    if (ds && *ds) {
        int n = 0;
        const char *p = ds;
        while (*p && n < MAX_DEVICES) {
            char *end = nullptr;
            const long v = strtol(p, &end, 10);
            if (end == p) break;
            if (v >= 0 && v < ndev) g.group[n++] = (int)v;
            else fprintf(stderr, "libpdeip: PDEIP_DEVICES names device %ld, have %d; ignored\n", v, ndev);
            p = (*end == ',') ? end + 1 : end;
        }
        if (n > 0) g.ngroup = n;
    } else if (d1 && *d1) {
        const int v = atoi(d1);
        if (v >= 0 && v < ndev) {
            g.group[0] = v;
            g.ngroup = 1;
        } else
            fprintf(stderr, "libpdeip: PDEIP_DEVICE=%d, have %d devices; ignored\n", v, ndev);
    }
}

int use_device(int device)
{
    if (device < 0 || device >= MAX_DEVICES) return set_err(PDEIP_ERR_ARG, "device %d out of range", device);
    HIPCHK(hipSetDevice(device));
    g.devs[device].device = device;
    return PDEIP_OK;
}

int use_device()
{
    read_env_once();
    return use_device(g.group[0]);
}

static void release_device(DeviceState *d)
{
    if (d->device < 0) return;
    if (hipSetDevice(d->device) != hipSuccess) return;
    latch_abort(d);
    g.ws_generation++;
    for (int s = 0; s < WS_NSLOT; s++) {
        if (d->ws[s]) (void)hipFree(d->ws[s]);
        d->ws[s] = nullptr;
        d->ws_bytes[s] = 0;
    }
    if (d->worker_stream) (void)hipStreamDestroy(d->worker_stream);
    d->worker_stream = nullptr;
    d->reset_caches();
}

} // namespace pdeip

using namespace pdeip;

// ------------------------------------------------------------------------------------------------
// library state
// ------------------------------------------------------------------------------------------------
extern "C" const char *pdeip_version(void) { return "pdeip-mi355x 0.2 (gfx950)"; }
extern "C" const char *pdeip_last_error(void) { return tls.err; }
extern "C" int pdeip_set_mode(int mode)
{
    RC(check_mode("pdeip_set_mode", mode));
    g.mode = mode;
    mode_set_explicitly = true;
    return PDEIP_OK;
}
extern "C" int pdeip_get_mode(void)
{
    read_env_once();
    return g.mode;
}
extern "C" int pdeip_set_devices(int n, const int *ids)
{
    int have = 0;
    HIPCHK(hipGetDeviceCount(&have));
    if (n < 1 || n > MAX_DEVICES || !ids) return set_err(PDEIP_ERR_ARG, "pdeip_set_devices: need 1..%d device ids", MAX_DEVICES);
    for (int k = 0; k < n; k++) {
        if (ids[k] < 0 || ids[k] >= have || ids[k] >= MAX_DEVICES)
            return set_err(PDEIP_ERR_ARG, "pdeip_set_devices: no device %d (have %d)", ids[k], have);
        for (int j = 0; j < k; j++)
            if (ids[j] == ids[k]) return set_err(PDEIP_ERR_ARG, "pdeip_set_devices: device %d named twice", ids[k]);
    }
    for (int k = 0; k < n; k++) g.group[k] = ids[k];
    g.ngroup = n;
    devices_set_explicitly = true;
    return PDEIP_OK;
}
extern "C" int pdeip_set_device(int device_id) { return pdeip_set_devices(1, &device_id); }
extern "C" int pdeip_get_devices(int *ids, int capacity)
{
    read_env_once();
    for (int k = 0; k < g.ngroup && k < capacity; k++)
        if (ids) ids[k] = g.group[k];
    return g.ngroup;
}
extern "C" int pdeip_release(void)
{
    int cur = 0;
    const bool have_cur = hipGetDevice(&cur) == hipSuccess;
    for (int k = 0; k < MAX_DEVICES; k++) release_device(&g.devs[k]);
    if (have_cur) (void)hipSetDevice(cur);
    return PDEIP_OK;
}
extern "C" int pdeip_last_launch_count(void) { return tls.last_launches; }
extern "C" int pdeip_workspace_generation(void) { return g.ws_generation; }
extern "C" int pdeip_persist_error(void)
{ // waits for the device(s), then reports whether a bounded spin of the persistent kernel timed out
    bool bad = false;
    int cur = 0;
    const bool have_cur = hipGetDevice(&cur) == hipSuccess;
    for (int k = 0; k < MAX_DEVICES; k++) {
        DeviceState *d = &g.devs[k];
        if (d->device < 0) continue;
        if (d->persist_used && d->ws[WS_CTL]) {
            HIPCHK(hipSetDevice(d->device));
            unsigned word = 0;
            HIPCHK(hipDeviceSynchronize());
            HIPCHK(hipMemcpy(&word, d->ws[WS_CTL], sizeof word, hipMemcpyDeviceToHost));
            if (word != 0) {
                bad = true;
                HIPCHK(hipMemset(d->ws[WS_CTL], 0, sizeof(unsigned))); // reported: clear it
            }
            d->persist_used = false;
        }
        if (d->abort_latched) bad = true;
        d->abort_latched = false;
    }
    if (have_cur) (void)hipSetDevice(cur);
    if (bad) return set_err(PDEIP_ERR_DEVICE, "persistent exact-order kernel: a dependency wait timed out (results are invalid)");
    return PDEIP_OK;
}
// Diagnostic: raises the abort word of the current device's control block as a timed-out dependency wait of a walker would (the
// walkers' waits then fall through: a call made with the word set drains at once and its results are invalid), so that the
// reporting path -- pdeip_persist_error(), the latch across a regrown control block, the host entry points' check -- can be
// tested without waiting half a second for a real time-out.
extern "C" int pdeip_debug_raise_abort(void)
{
    RC(use_device());
    float *ctl = nullptr;
    RC(ws_get(WS_CTL, 16, &ctl));
    const unsigned one = 1u;
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpy(ctl, &one, sizeof one, hipMemcpyHostToDevice));
    cur_dev()->persist_used = true;
    return PDEIP_OK;
}

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
