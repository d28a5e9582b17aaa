// pdeip_walk5.hip -- libpdeip.so: the exact-order walkers of the 5-point models (k_sor_walk, pdeip_sor_walk.hpp) and their launch.
#include "pdeip_walk_host.hpp"

#include "pdeip_sor_walk.hpp"

namespace pdeip {

template <class Mdl> int walk_width(int nrows, int ncols, int nframes, int iter)
{
    (void)nrows; (void)ncols; (void)nframes; (void)iter;
    const int forced = env_int("PDEIP_WALK_W", 0);
    if (forced == 32 || forced == 48 || forced == 64) return forced;
    return 64;
}

namespace {
template <class Mdl, int NBUF, int W>
int launch_one(hipStream_t s, const SweepPlanes<Mdl> &P, const float *pack, const PersistCtl &ctl, int nrows, int ncols, int B, int iter, int NC,
               int nframes, float omega, size_t n)
{
    using WL = WalkLayout<Mdl, NBUF, W>;
    static_assert(WL::FITS, "the chunk buffers must fit the 160 KB of LDS");
    RC(ensure_lds(reinterpret_cast<const void *>(&k_sor_walk<Mdl, NBUF, W>), WL::LDS_BYTES));
    hipLaunchKernelGGL((k_sor_walk<Mdl, NBUF, W>), dim3((unsigned)(B * iter * nframes)), dim3(WL::THREADS), WL::LDS_BYTES, s, P, pack, ctl, nrows, ncols, B,
                       iter, NC, nframes, omega, n, env_int("PDEIP_WALK_TUNE", 0));
    return PDEIP_OK;
}
template <class Mdl, int W>
int launch_w(hipStream_t s, const SweepPlanes<Mdl> &P, const float *pack, const PersistCtl &ctl, int nrows, int ncols, int B, int iter, int NC,
             int nframes, float omega, size_t n)
{
    // three chunk buffers where they fit the LDS (two chunks in flight), else two
    constexpr int NB = WalkLayout<Mdl, 3, W>::FITS ? 3 : 2;
    if (NB == 3 && env_int("PDEIP_WALK_NBUF", 3) < 3) return launch_one<Mdl, 2, W>(s, P, pack, ctl, nrows, ncols, B, iter, NC, nframes, omega, n);
    return launch_one<Mdl, NB, W>(s, P, pack, ctl, nrows, ncols, B, iter, NC, nframes, omega, n);
}
} // namespace

template <class Mdl>
int walk_launch(hipStream_t s, const SweepPlanes<Mdl> &P, const float *pack, const PersistCtl &ctl, int nrows, int ncols, int B, int iter, int NC,
                int nframes, float omega, size_t n, int W)
{
    if (W == 32) return launch_w<Mdl, 32>(s, P, pack, ctl, nrows, ncols, B, iter, NC, nframes, omega, n);
    if (W == 48) return launch_w<Mdl, 48>(s, P, pack, ctl, nrows, ncols, B, iter, NC, nframes, omega, n);
    return launch_w<Mdl, 64>(s, P, pack, ctl, nrows, ncols, B, iter, NC, nframes, omega, n);
}

#define PDEIP_WALK_INSTANCE(M)                                                                                                              \
    template int walk_width<M>(int, int, int, int);                                                                                          \
    template int walk_launch<M>(hipStream_t, const SweepPlanes<M> &, const float *, const PersistCtl &, int, int, int, int, int, int, float, size_t, int);
PDEIP_WALK_INSTANCE(ModelElin4)
PDEIP_WALK_INSTANCE(ModelLlin4)
PDEIP_WALK_INSTANCE(ModelDisp4)
PDEIP_WALK_INSTANCE(ModelDispSym4)
PDEIP_WALK_INSTANCE(ModelPde4)

} // namespace pdeip

#ifdef PDEIP_P8_STAMPS // diagnostic build only (tools/walk2_stamps.py)
extern "C" int pdeip_debug_read_walk2_stamps(unsigned long long *out)
{
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpyFromSymbol(out, HIP_SYMBOL(pdeip::g_p8_stamps), 4096 * sizeof(unsigned long long)));
    return PDEIP_OK;
}
extern "C" int pdeip_debug_read_walk_trace(unsigned long long *out)
{
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpyFromSymbol(out, HIP_SYMBOL(pdeip::g_walk_trace), 128 * 8 * 16 * sizeof(unsigned long long)));
    return PDEIP_OK;
}
#endif
