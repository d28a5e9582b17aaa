// pdeip_walk_host.hpp -- what the exact-order launch logic (pdeip_sor5.hip) needs to know about the walkers (pdeip_sor_walk.hpp),
// whose kernels are compiled in a translation unit of their own (pdeip_walk5.hip: they are large and build.py compiles units in parallel).
#pragma once
#include "pdeip_ctx.hpp"
#include "pdeip_models.hpp"
#include "pdeip_sor_exact.hpp"

namespace pdeip {

// Columns per strip for a call.  PDEIP_WALK_W = 32 | 48 | 64 forces it.
template <class Mdl> int walk_width(int nrows, int ncols, int nframes, int iter);
// Launches k_sor_walk<Mdl, NBUF, W> on `s` (B = strips of W columns; ctl from persist_prepare with that B).
template <class Mdl>
int walk_launch(hipStream_t s, const SweepPlanes<Mdl> &P, const float *pack, const PersistCtl &ctl, int nrows, int ncols, int B, int iter, int NC,
                int nframes, float omega, size_t frame_stride, int W);

} // namespace pdeip
