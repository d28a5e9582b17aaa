// pdeip_sor_small.hpp -- red-black sweeps of a small frame with the iterate resident in LDS (gfx950).
//
// The coarse scales of the drivers' pyramids (270x480 and below in the 4K multigrid run: 276 of its 324 solver calls) are
// bound by what a launch costs, not by what it computes: the marching kernels need 2-4 launches of 20-50 us per call there,
// each a short chain of dependent global loads.  Here every global value is read exactly once, all loads of a workgroup
// are in flight together, and the sweeps run out of LDS:
//   * a workgroup owns a slab of W consecutive MATLAB columns of the frame (the whole frame when it fits) plus a halo of
//     2 columns per sweep on each cut side -- a red-black sweep moves information by two columns, so after `sweeps`
//     sweeps the owned columns are exact although the halo's outer columns have gone stale (the same argument as the
//     multi-device slabs, pdeip_multi.hip); nothing is exchanged between workgroups;
//   * the iterate fields and the read-only neighbour fields of the late-linearisation models live in LDS as ONE vector per
//     pixel (u,v | du,dv,U,V | dU,U | x), split by row parity: pixel (i,j) sits at [(i&1)][j][i>>1].  The pixels of one
//     colour in one column are then contiguous, so consecutive lanes read consecutive vectors: every neighbour is one
//     conflict-free ds_read_b64/b128 instead of NIT+NRO strided dword reads;
//   * every thread owns up to Q pixels of each colour (slot = thread + THREADS q of the colour's column-major enumeration)
//     and keeps their coefficients -- divisors derived once, as the reference's first sweep does -- in registers for the
//     whole launch, so a half-sweep is five LDS reads, Mdl::update(), one LDS write per pixel; one workgroup barrier per
//     half-sweep; the border replicate (rows, then columns: opticalflowSolvers.c:161-179) in LDS after every sweep;
//   * in place with several slabs: a workgroup may store its owned columns only after every workgroup has loaded its
//     halo -- a global arrival counter, bumped after the load phase and polled (bounded) before the store phase; a second
//     counter of workgroups that got past the poll lets the last one reset both, so the pair is zero again when the next
//     launch starts (no host-side epoch: a captured HIP graph replays it unchanged).  The grid is at most a couple of
//     hundred workgroups, all resident.  Out of place, or a single slab, needs no gate.
// Same per-pixel arithmetic as every other ordering (Mdl::update): bit-identical to the marching kernels.
#pragma once
#include "pdeip_models.hpp"

namespace pdeip {

constexpr int SMALL_THREADS_MAX = 1024;
constexpr int SMALL_MAX_SLABS = 128;   // workgroups per frame set: all co-resident (256 CUs, one workgroup each)
constexpr int SMALL_MAX_SWEEPS = 4;    // sweeps per launch when the frame is cut into slabs (halo 2 x sweeps)
constexpr size_t SMALL_LDS_CAP = (size_t)152 * 1024;

struct SmallPlan {
    bool ok = false;
    int W = 0, nslabs = 0; // owned columns per slab, slabs per frame
    size_t lds = 0;
};

template <class Mdl> struct SmallLayout {
    static constexpr int NV = Mdl::NIT + Mdl::NRO;   // floats per pixel vector
    static constexpr int VW = NV == 3 ? 4 : NV;      // padded to a power of two
    // threads per workgroup and pixels of one colour per thread: 2 Q NCF coefficients stay in registers, and a workgroup of 16
    // waves caps a wave at 128 VGPRs, one of 12 waves at 168 -- the four-field model needs the latter to reach 3840 slots
    static constexpr int THREADS = (NV >= 4) ? 768 : 1024;
    static constexpr int Q = (NV >= 4) ? 5 : 4;
    static_assert(VW == 1 || VW == 2 || VW == 4, "pixel vector width");
    __host__ __device__ static int pr(int nrows) { return (nrows - 1) / 2; }   // slots per column of one colour
    __host__ __device__ static int hr(int nrows) { return (nrows + 1) / 2; }   // rows per parity plane
    static size_t lds_bytes(int nrows, int ncl) { return (size_t)2 * ncl * hr(nrows) * VW * sizeof(float); }
    // widest slab (local columns, halo included) that q pixels per thread and colour cover and LDS holds
    static int max_local_cols(int nrows, int q)
    {
        const long by_slots = (long)THREADS * q / pr(nrows) + 2;
        const long by_lds = (long)(SMALL_LDS_CAP / ((size_t)2 * hr(nrows) * VW * sizeof(float)));
        return (int)(by_slots < by_lds ? by_slots : by_lds);
    }
    // How to cut a frame for launches of `sweeps` sweeps.  qpref: pixels per thread and colour to aim for when cutting
    // (the fewer, the shorter a workgroup runs; the more slabs, the more redundant halo work).
    static SmallPlan plan(int nrows, int ncols, int sweeps, int qpref)
    {
        SmallPlan p;
        if (nrows < 3 || ncols < 3) return p;
        if (ncols <= max_local_cols(nrows, qpref < Q ? qpref : Q) || (ncols <= max_local_cols(nrows, Q) && sweeps > SMALL_MAX_SWEEPS)) {
            p.ok = true;
            p.W = ncols;
            p.nslabs = 1;
            p.lds = lds_bytes(nrows, ncols);
            return p;
        }
        if (sweeps > SMALL_MAX_SWEEPS) return p;
        const int H = 2 * sweeps;
        for (int q = (qpref < Q ? qpref : Q); q <= Q; q++) {
            const int w = max_local_cols(nrows, q) - 2 * H - 1; // - 1: a slab of one border column also holds its inner neighbour
            // worth cutting when a slab owns at least as many columns as one of its halos (q < Q), or at all (q == Q)
            if (w >= (q < Q ? H : 4) && (ncols + w - 1) / w <= SMALL_MAX_SLABS) {
                p.nslabs = (ncols + w - 1) / w;
                p.W = (ncols + p.nslabs - 1) / p.nslabs; // even slabs
                p.nslabs = (ncols + p.W - 1) / p.W;
                const int widest = p.W + 2 * H + 1 < ncols ? p.W + 2 * H + 1 : ncols;
                p.lds = lds_bytes(nrows, widest);
                p.ok = true;
                return p;
            }
        }
        if (ncols <= max_local_cols(nrows, Q)) { // does not cut well, but fits one workgroup
            p.ok = true;
            p.W = ncols;
            p.nslabs = 1;
            p.lds = lds_bytes(nrows, ncols);
        }
        return p;
    }
};

template <int VW> struct SmallVec;
template <> struct SmallVec<1> { typedef float type; };
template <> struct SmallVec<2> { typedef float2 type; };
template <> struct SmallVec<4> { typedef float4 type; };

template <int VW> __device__ __forceinline__ void small_read(float (&d)[VW], const float *lds, int pix)
{
    typedef typename SmallVec<VW>::type V;
    const V v = *reinterpret_cast<const V *>(lds + (size_t)pix * VW);
    if constexpr (VW == 1) d[0] = v;
    if constexpr (VW == 2) { d[0] = v.x; d[1] = v.y; }
    if constexpr (VW == 4) { d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w; }
}
// the first N floats of a pixel vector
template <int VW, int N> __device__ __forceinline__ void small_write(float *lds, int pix, const float (&d)[N])
{
    float *p = lds + (size_t)pix * VW;
    if constexpr (N == 1) p[0] = d[0];
    if constexpr (N == 2) *reinterpret_cast<float2 *>(p) = make_float2(d[0], d[1]);
    if constexpr (N == 4) *reinterpret_cast<float4 *>(p) = make_float4(d[0], d[1], d[2], d[3]);
}

template <class Mdl>
__global__ void __launch_bounds__((SmallLayout<Mdl>::THREADS))
k_sor_small(SweepPlanes<Mdl> P, int nrows, int ncols, int sweeps, float omega, int col0, size_t frame_stride, int W, unsigned *sync,
            unsigned *abort_flag)
{
    using L = SmallLayout<Mdl>;
    constexpr int NIT = Mdl::NIT, NRO = Mdl::NRO, NRO1 = at_least_one<NRO>::value, NCF = Mdl::NCF, Q = L::Q, VW = L::VW, SMALL_THREADS = L::THREADS;
    extern __shared__ __attribute__((aligned(16))) float small_lds[]; // [2 parities][ncl][HR] pixel vectors of VW floats
    const int tid = threadIdx.x;
    const size_t fo = (size_t)blockIdx.y * frame_stride;
    const float om1 = 1.0f - omega;
    // my slab: owned columns [c0, c1), held columns [lo, hi)
    const int H = 2 * sweeps;
    const int c0 = blockIdx.x * W, c1 = (c0 + W < ncols) ? c0 + W : ncols;
    // a border column is the replicate of its inner neighbour: a slab that owns one has to hold that neighbour exactly too
    const int c0e = c0 < ncols - 2 ? c0 : ncols - 2, c1e = c1 > 2 ? c1 : 2;
    const int lo = (gridDim.x == 1 || c0e - H < 0) ? 0 : c0e - H, hi = (gridDim.x == 1 || c1e + H > ncols) ? ncols : c1e + H;
    const int ncl = hi - lo, HR = L::hr(nrows), PLANE = ncl * HR;
    auto pix = [&](int i, int jl) { return ((i & 1) * ncl + jl) * HR + (i >> 1); };

    { // load phase: columns lo..hi-1 of every field, coalesced along the rows
        int jl = tid / nrows, i = tid - jl * nrows;
        const int dj = SMALL_THREADS / nrows, di = SMALL_THREADS - dj * nrows;
        while (jl < ncl) {
            const size_t gidx = fo + (size_t)(lo + jl) * nrows + i;
            float v[VW];
#pragma unroll
            for (int f = 0; f < VW; f++) v[f] = 0.0f;
#pragma unroll
            for (int f = 0; f < NIT; f++) v[f] = P.it_in[f][gidx];
#pragma unroll
            for (int f = 0; f < NRO; f++) v[NIT + f] = P.ro[f][gidx];
            small_write<VW, VW>(small_lds, pix(i, jl), v);
            i += di;
            jl += dj;
            if (i >= nrows) {
                i -= nrows;
                jl++;
            }
        }
    }

    // my pixels: colour c, slot s -> local column jl = 1 + s / PR, row i = first row of colour c in that column + 2 (s % PR)
    const int PR = L::pr(nrows);
    int pos[2][Q];    // 2 * pixel index + row parity, or -1
    float cf[2][Q][NCF];
#pragma unroll
    for (int c = 0; c < 2; c++)
#pragma unroll
        for (int q = 0; q < Q; q++) {
            const int s = tid + SMALL_THREADS * q;
            const int jl = 1 + s / PR, k = s - (jl - 1) * PR;
            const int i = 1 + ((1 + jl + lo + col0 + c) & 1) + 2 * k;
            const bool ok = (jl <= ncl - 2) && (i <= nrows - 2);
            pos[c][q] = ok ? 2 * pix(i, jl) + (i & 1) : -1;
            const size_t gidx = fo + (ok ? (size_t)(lo + jl) * nrows + i : 0);
            float kk[NCF];
#pragma unroll
            for (int f = 0; f < NCF; f++) kk[f] = P.cf[f][gidx];
            Mdl::derive(kk); // the divisor planes of the reference's first sweep (:111-127), once per launch
#pragma unroll
            for (int f = 0; f < NCF; f++) cf[c][q][f] = kk[f];
        }
    __syncthreads();
    const bool gated = (sync != nullptr);
    if (gated && tid == 0) __hip_atomic_fetch_add(sync, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); // my loads have returned

    const bool west_edge = (lo == 0), east_edge = (hi == ncols);
    for (int sweep = 0; sweep < sweeps; sweep++) {
#pragma unroll
        for (int c = 0; c < 2; c++) {
#pragma unroll
            for (int q = 0; q < Q; q++) {
                const int pp = pos[c][q];
                if (pp >= 0) {
                    const int oc = pp >> 1;
                    // row above: the other parity plane, one slot up for an even row; row below: the slot after it
                    const int on = (pp & 1) ? oc - PLANE : oc + PLANE - 1;
                    float vc[VW], vw[VW], ve[VW], vn[VW], vs[VW];
                    small_read<VW>(vc, small_lds, oc);
                    small_read<VW>(vw, small_lds, oc - HR);
                    small_read<VW>(ve, small_lds, oc + HR);
                    small_read<VW>(vn, small_lds, on);
                    small_read<VW>(vs, small_lds, on + 1);
                    float cc[NIT], w[NIT], e[NIT], n[NIT], s[NIT];
                    float rc[NRO1], rw[NRO1], re[NRO1], rn[NRO1], rs[NRO1];
#pragma unroll
                    for (int f = 0; f < NIT; f++) {
                        cc[f] = vc[f];
                        w[f] = vw[f];
                        e[f] = ve[f];
                        n[f] = vn[f];
                        s[f] = vs[f];
                    }
#pragma unroll
                    for (int f = 0; f < NRO1; f++) {
                        const int g = (NRO > 0) ? NIT + f : 0;
                        rc[f] = (NRO > 0) ? vc[g] : 0.0f;
                        rw[f] = (NRO > 0) ? vw[g] : 0.0f;
                        re[f] = (NRO > 0) ? ve[g] : 0.0f;
                        rn[f] = (NRO > 0) ? vn[g] : 0.0f;
                        rs[f] = (NRO > 0) ? vs[g] : 0.0f;
                    }
                    Mdl::update(cc, w, e, n, s, rc, rw, re, rn, rs, cf[c][q], omega, om1);
                    small_write<VW, NIT>(small_lds, oc, cc);
                }
            }
            __syncthreads();
        }
        // border replicate: rows first, then columns (:161-179)
        for (int jl = tid; jl < ncl; jl += SMALL_THREADS) {
            float t[VW], b[VW];
            small_read<VW>(t, small_lds, pix(1, jl));
            small_read<VW>(b, small_lds, pix(nrows - 2, jl));
            float tt[NIT], bb[NIT];
#pragma unroll
            for (int f = 0; f < NIT; f++) {
                tt[f] = t[f];
                bb[f] = b[f];
            }
            small_write<VW, NIT>(small_lds, pix(0, jl), tt);
            small_write<VW, NIT>(small_lds, pix(nrows - 1, jl), bb);
        }
        __syncthreads();
        if (west_edge || east_edge) {
            for (int i = tid; i < nrows; i += SMALL_THREADS) {
                float t[VW], tt[NIT];
                if (west_edge) {
                    small_read<VW>(t, small_lds, pix(i, 1));
#pragma unroll
                    for (int f = 0; f < NIT; f++) tt[f] = t[f];
                    small_write<VW, NIT>(small_lds, pix(i, 0), tt);
                }
                if (east_edge) {
                    small_read<VW>(t, small_lds, pix(i, ncl - 2));
#pragma unroll
                    for (int f = 0; f < NIT; f++) tt[f] = t[f];
                    small_write<VW, NIT>(small_lds, pix(i, ncl - 1), tt);
                }
            }
        }
        __syncthreads();
    }

    if (gated) { // in place: nobody stores before everybody has loaded.  sync[0]: arrivals, sync[1]: departures
        if (tid == 0) {
            const unsigned nwg = gridDim.x * gridDim.y;
            const unsigned long long t0 = __builtin_amdgcn_s_memrealtime(); // 100 MHz
            while (__hip_atomic_load(sync, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < nwg) {
                if (__hip_atomic_load(abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) break;
                if (__builtin_amdgcn_s_memrealtime() - t0 > 50000000ull) { // 0.5 s: a workgroup never started; report, do not hang
                    __hip_atomic_store(abort_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    break;
                }
                __builtin_amdgcn_s_sleep(8);
            }
            // everybody increments `departures` exactly once; whoever makes it nwg is the last to look at `arrivals`
            if (__hip_atomic_fetch_add(sync + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == nwg - 1) {
                __hip_atomic_store(sync, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(sync + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        __syncthreads();
    }
    { // store phase: the owned columns, every row
        const int nown = c1 - c0, jl0 = c0 - lo;
        int jj = tid / nrows, i = tid - jj * nrows;
        const int dj = SMALL_THREADS / nrows, di = SMALL_THREADS - dj * nrows;
        while (jj < nown) {
            const size_t gidx = fo + (size_t)(c0 + jj) * nrows + i;
            float v[VW];
            small_read<VW>(v, small_lds, pix(i, jl0 + jj));
#pragma unroll
            for (int f = 0; f < NIT; f++) P.it_out[f][gidx] = v[f];
            i += di;
            jj += dj;
            if (i >= nrows) {
                i -= nrows;
                jj++;
            }
        }
    }
}

} // namespace pdeip
