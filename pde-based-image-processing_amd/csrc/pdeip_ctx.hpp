// pdeip_ctx.hpp -- library-internal state shared by the translation units of libpdeip.so.
//
// libpdeip.so is built from several .hip files (one per kernel family, compiled in parallel: build.py);
// this header is what they share: the process-wide context, the per-device state (workspace cache,
// per-function launch attributes), error plumbing and the launch-side helpers.  Nothing here is part of
// the C-ABI (include/pdeip.h).
#pragma once
#include "../../include/pdeip.h"

#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <vector>

namespace pdeip {

enum { WS_AUX0 = 0, WS_AUX1, WS_PING, WS_ARENA, WS_CTL, WS_ORDER, WS_ALR, WS_ALR_T, WS_TV, WS_SMALL, WS_MAIL, WS_PACK, WS_DRIVER, WS_LEX, WS_NSLOT };

constexpr int MAX_DEVICES = 16;

// What the library keeps per HIP device: everything that lives in that device's memory or is a property of
// the code object loaded on it.  pdeip_release() / pdeip_set_device() reset it as a whole, so no cache can
// outlive the buffer it describes.
struct DeviceState {
    int device = -1;
    void *ws[WS_NSLOT] = {};
    size_t ws_bytes[WS_NSLOT] = {};
    int order_B = 0, order_T = 0, order_affine = -1; // shape (and kind) of the schedule table cached in ws[WS_ORDER]
    int num_cus = 0;                                 // compute units of this device (0: not asked yet)
    bool persist_used = false;    // a persistent launch happened since the last pdeip_persist_error()
    bool abort_latched = false;   // a persistent kernel's abort word was seen set before its buffer went away
    std::map<const void *, size_t> lds_opt_in; // kernel -> dynamic LDS bytes opted into (hipFuncSetAttribute is per device)
    std::map<const void *, int> resident_waves; // kernel -> waves the device holds at once (occupancy query)
    hipStream_t worker_stream = nullptr; // the stream a worker thread of a slab-split host call uses on this device (pdeip_multi.hip)
    void reset_caches()
    {
        order_B = order_T = 0;
        order_affine = -1;
        persist_used = false;
    }
};

// What a thread of the library keeps to itself.  The host entry points are called from one thread at a time, but a device
// group runs one worker thread per device through the ordinary *_dev entry points (pdeip_multi.hip): the error text, the
// launch counter and the device-state slot are per thread, so workers neither tear each other's messages nor lose counts.
struct ThreadState {
    char err[512] = "";
    int last_launches = 0;
    int dev_slot = -1; // >= 0: cur_dev() returns this slot of Context::devs instead of the current HIP device's (virtual-thread test mode)
};
extern thread_local ThreadState tls;

struct Context {
    int mode = PDEIP_MODE_EXACT_ORDER;
    std::atomic<int> ws_generation{0}; // bumped whenever a workspace buffer is freed or regrown: captured HIP graphs hold its pointers
    int rb_tj = 0;        // columns per red-black unit (0 = default)
    bool env_read = false; // PDEIP_MODE / PDEIP_DEVICE / PDEIP_DEVICES consulted
    // device group of the host entry points: group[0] is "the" device of single-device calls
    int ngroup = 1;
    int group[MAX_DEVICES] = {0};
    DeviceState devs[MAX_DEVICES];
    // sweep-kernel timing (pdeip_profile_*)
    bool profile = false;
    static constexpr int MAX_EV = 4096;
    hipEvent_t ev[MAX_EV][2];
    int ev_launches[MAX_EV];
    int n_ev = 0, n_ev_created = 0;
    std::mutex ev_mutex; // event slots are handed out to whichever thread asks
};
extern Context g;

int set_err(int code, const char *fmt, ...);

#define HIPCHK(expr)                                                                                       \
    do {                                                                                                   \
        hipError_t e_ = (expr);                                                                            \
        if (e_ != hipSuccess) return ::pdeip::set_err(PDEIP_ERR_DEVICE, "%s failed: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

#define RC(expr)             \
    do {                     \
        int rc_ = (expr);    \
        if (rc_) return rc_; \
    } while (0)

#define NONNULL(who, p)                                                                                \
    do {                                                                                               \
        if ((p) == nullptr) return ::pdeip::set_err(PDEIP_ERR_ARG, "%s: argument '%s' is NULL", who, #p); \
    } while (0)

// Records an event pair around a run of sweep launches when profiling is on.
struct SweepTimer {
    hipStream_t s;
    int slot = -1;
    explicit SweepTimer(hipStream_t stream);
    void stop(int launches);
};

// State of the device HIP is currently set to (the *_dev entry points work on whatever device the caller selected).
DeviceState *cur_dev();
// Grow-only device workspace of the current device.  Growing synchronises the device (the old buffer may be in use).
int ws_get(int slot, size_t bytes, float **out);
// Opt a kernel into `bytes` of dynamic LDS on the current device (needed above 64 KiB), once per device and size.
int ensure_lds(const void *kernel, size_t bytes);
// Waves of `kernel` the current device holds at once (blocks per CU x waves per block x CUs), cached per device.
int resident_waves(const void *kernel, int block_threads, int waves_per_block);

int check_dims(const char *who, int nrows, int ncols, int nframes);
int check_mode(const char *who, int mode);
int check_solver(const char *who, int solver);
inline bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }
int env_int(const char *name, int dflt);
int check_deriv_dims(const char *who, int nrows, int ncols, int nframes);
int pick_rb_tj(int nrows, int ncols); // columns per unit of the one-sweep red-black kernels
// Device-to-device copy of n floats on stream s by a kernel of the library's own (16 bytes per lane, grid-stride): the runtime's
// blit kernel behind hipMemcpyAsync moves a 33 MB plane in 81 us (0.8 TB/s); this one in ~13.  Capturable like any launch.
int copy_d2d(hipStream_t s, float *dst, const float *src, size_t n);
inline dim3 pixel_grid(int nrows, int ncols, int nz) { return dim3((unsigned)((nrows + 255) / 256), (unsigned)ncols, (unsigned)nz); }

// Makes group[0] (or `device`) the current HIP device; reads the environment knobs on first use.
int use_device();
int use_device(int device);
void read_env_once();

// ---- host staging (pdeip_host.hip) ---------------------------------------------------------------------
inline size_t pad4(size_t n) { return (n + 3) & ~(size_t)3; }

// ---- one solver call across the device group (pdeip_multi.hip) --------------------------------------------
// Planes are HOST pointers in the order of the corresponding *_dev_to entry point.
struct MultiCall {
    int kind;                     // 0 elin4, 1 llin4, 2 disp4, 3 pde4, 4 pde8
    int n_it, n_ro, n_cf, frames; // iterate planes, read-only neighbour planes, coefficient planes; frames per plane
    const float *it_in[2];
    float *it_out[2];
    const float *ro[2];
    const float *cf[11];
    int nrows, ncols, iter;
    float omega;
};
// Runs a red-black point-SOR call as column slabs over the devices of pdeip_set_devices().  *handled = 0: not split (one
// device, or the slabs would be narrower than their halo): the caller takes the single-device path.
int multi_sor(const MultiCall &mc, int *handled);

} // namespace pdeip
