// pdeip_sor_walk.hpp -- the reference's lexicographic Gauss-Seidel order in one launch: k_sor_walk<Mdl, NBUF> (round 3).
//
// Same wavefront, same dependency protocol and same arithmetic as k_sor_exact_persist (pdeip_sor_exact.hpp: a workgroup owns
// one (strip of 64 columns, sweep), lane l relaxes row 1 + 16c + q - l at step q of chunk c; progress counters towards the
// previous sweep, a tagged mailbox towards the west strip of the same sweep) -- what changed is how a chunk reaches LDS.
// Round 2's stamps had the walk paced by the loader wave: 49 register-staged loads and 86 ds_write_b128 per chunk behind a
// dependency poll, one chunk in flight.  Here
//   * the LOADER moves nothing through registers: every piece of a chunk is one LDS-DMA instruction (buffer_load ... lds,
//     1 KiB per piece).  The LDS destination of a piece is lane-linear, so the XOR swizzle that makes the compute lanes'
//     ds_read_b128 conflict-free is applied on the SOURCE side: lane L of piece d fetches the 16 bytes that belong at granule
//     64 d + L of the chunk image.  Per-lane source offsets are computed once per walk and advanced by a constant per chunk.
//   * NBUF chunk buffers: with three, the chunk two ahead is in flight while the next one lands (counted vmcnt).
//   * the dependency poll left the loader: a POLLER wave refreshes the previous sweep's progress in an LDS word once per
//     interval (and takes the west edge from the mailbox); the loader reads that word and only spins when it really lacks a
//     dependency.  The STORER drains and publishes on its own vmcnt queue.
// Chunk image (per buffer): NF field images of 64 columns x 16 rows (granule (c, q) = rows 4q..4q+3 of column c at granule
// 4c + (q ^ s(c)), s(c) = (c >> 2) & 3), the packed coefficients (PackLayout: granule g of column c at c CS + (g ^ s(c))), and per
// field 16 west + 16 east edge values.  Results leave through two out images of the same field layout.
#pragma once
#include "pdeip_sor_exact.hpp"

namespace pdeip {

template <class Mdl, int NBUF_, int W_ = 64> struct WalkLayout {
    static constexpr int NIT = Mdl::NIT, NRO = Mdl::NRO, NF = NIT + NRO, NCF = Mdl::NCF, NBUF = NBUF_;
    static constexpr int W = W_;                               // columns of a strip (lanes W .. 63 idle): 64, 48 or 32
    static constexpr int NFP = W / 16;                         // pieces of one field's chunk image
    static constexpr int FIELD = W * EX_CH;                    // floats of one field's chunk image
    static constexpr int RUN = PackLayout<Mdl>::RUN, CS = PackLayout<Mdl>::CS;
    static constexpr int PACK = W * CS * 4;                    // floats of the packed image
    static constexpr int NPK = W * CS / 64;                    // its pieces (W CS granules / 64 lanes)
    static constexpr int EDGE = NF * 32;                       // per field: west 16 | east 16
    static constexpr int BUF = NF * FIELD + PACK + EDGE;       // floats per chunk buffer
    static constexpr int OUTB = NIT * FIELD;
    static constexpr int CTRL = 16 + 2 * NIT * 16;             // words: 0 item, 1 frame, 2 taken, 4 / 5 progress of (b,t-1) / (b+1,t-1) as the poller last saw it; 16..: its mailbox staging
    static constexpr size_t LDS_BYTES = (size_t)(NBUF * BUF + 2 * OUTB + CTRL) * sizeof(float);
    static constexpr int PIECES = NF * NFP + NPK + NF;         // LDS-DMA instructions per chunk
    static constexpr int NCOMP = NIT == 2 ? 2 : 1;             // compute waves (one per field of the coupled models)
#ifndef PDEIP_WALK_LOADERS
#define PDEIP_WALK_LOADERS 1
#endif
    static constexpr int NLOAD = PDEIP_WALK_LOADERS;           // loader waves: loader w issues the pieces i of a chunk with i % NLOAD == w (its own vmcnt queue)
    static constexpr int THREADS = 64 * (NCOMP + 2 + NLOAD);   // + storer, poller, loader(s) -- the extra loader is the last wave
    static constexpr int pieces_of(int w) { return (PIECES - w + NLOAD - 1) / NLOAD; }
    static constexpr bool FITS = LDS_BYTES <= 160 * 1024 && (NBUF < 3 || pieces_of(0) <= 63);
    static_assert(BUF % 4 == 0 && FIELD % 256 == 0 && (W * CS) % 64 == 0 && W % 16 == 0 && W <= 64, "16-byte granules, whole pieces");
};

#define WALK_LDS(p) ((__attribute__((address_space(3))) void *)(p))

#ifdef PDEIP_P8_STAMPS // diagnostic build only (tools/walk2_stamps.py): five roles per walker, eight slots of four words each
static __device__ unsigned long long g_walk_trace[128 * 8 * 16]; // [walker][role][interval < 16]: {busy cycles : 24 | barrier exit, 10 ns ticks since the loader's start : 40}
#define WKS_WRITE                                                                                                                  \
    if (lane == 0 && tk < 128) {                                                                                                  \
        g_p8_stamps[(tk * 8 + role) * 4 + 0] = s_busy_;                                                                            \
        g_p8_stamps[(tk * 8 + role) * 4 + 1] = (role == R_LOAD) ? s_r0_ : __builtin_amdgcn_s_memtime() - s_t0_;                    \
        g_p8_stamps[(tk * 8 + role) * 4 + 2] = __builtin_amdgcn_s_memrealtime() - s_r0_;                                          \
        g_p8_stamps[(tk * 8 + role) * 4 + 3] = (unsigned long long)(b | (t << 16));                                               \
    }
#define WKS_BEGIN P8S_BEGIN
#define WKS_END(k)                                                                                                                 \
    do {                                                                                                                           \
        const unsigned long long now_ = __builtin_amdgcn_s_memtime();                                                              \
        s_busy_ += now_ - s_i0_;                                                                                                   \
        if (lane == 0 && tk < 128 && (k) < 16)                                                                                     \
            g_walk_trace[(tk * 8 + role) * 16 + (k)] = ((now_ - s_i0_) & 0xffffffull) | (__builtin_amdgcn_s_memrealtime() << 24);  \
    } while (0)
#else
#define WKS_WRITE
#define WKS_BEGIN
#define WKS_END(k)
#endif

// One LDS-DMA piece: 64 lanes x 16 bytes from rs + voff (per lane, range-checked) to lds + 16 lane (buffer_load_dwordx4 ... lds).
// The host pass of hipcc does not know the builtin and silently drops every kernel that names it, stub included: device pass only.
template <int AUX> __device__ __forceinline__ void walk_dma16(__amdgpu_buffer_rsrc_t rs, float *lds, unsigned voff)
{
#if defined(__HIP_DEVICE_COMPILE__)
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, WALK_LDS(lds), 16, voff, 0, 0, AUX);
#else
    (void)rs; (void)lds; (void)voff;
#endif
}

template <class Mdl, int NBUF, int W = 64>
__global__ void __launch_bounds__((WalkLayout<Mdl, NBUF, W>::THREADS))
k_sor_walk(SweepPlanes<Mdl> P, const float *pack, PersistCtl ctl, int nrows, int ncols, int B, int T, int NC, int nframes, float omega,
           size_t frame_stride, int tune)
{
    // tune (PDEIP_WALK_TUNE, experiments): per role a pause of (nibble) x 64 cycles at the top of every interval --
    // bits 0-3 loader, 4-7 storer, 8-11 poller, 12-15 compute
    auto pause = [&](int shift) __attribute__((always_inline)) {
        for (int n = (tune >> shift) & 15; n > 0; n--) __builtin_amdgcn_s_sleep(1);
    };
    using L = WalkLayout<Mdl, NBUF, W>;
    constexpr int NFP = L::NFP;
    constexpr int NIT = L::NIT, NRO = L::NRO, NRO1 = at_least_one<NRO>::value, NF = L::NF, NCF = L::NCF, CS = L::CS;
    constexpr int D = NBUF - 1; // chunks in flight ahead of the one being relaxed
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *const outb_base = smem + NBUF * L::BUF;
    unsigned *const s_ctl = reinterpret_cast<unsigned *>(outb_base + 2 * L::OUTB);

    const int lane = threadIdx.x & 63;
    // wave -> role.  A workgroup's waves go to the SIMDs in cyclic order, so waves 0 and 4 share one: the two light ones.
    //   NIT = 2: 0 storer, 1 compute (field 0), 2 loader, 3 compute (field 1), 4 poller;   NIT = 1: 0 storer, 1 compute, 2 loader, 3 poller
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    enum { R_STORE = 0, R_COMP0 = 1, R_LOAD = 2, R_COMP1 = 3, R_POLL = 4 };
    const int nbase = (NIT == 2) ? 5 : 4; // waves of the one-loader form; wave nbase is the second loader
    const int ldw = wave >= nbase ? 1 : 0; // which loader (of a loader wave)
    const int role = wave >= nbase ? (int)R_LOAD : ((NIT == 2) ? wave : (wave == 3 ? (int)R_POLL : wave));

    if (threadIdx.x == 0) {
        unsigned item = 0, fr = 0;
        s_ctl[2] = persist_take_item(ctl, nframes, &item, &fr) ? 1u : 0u;
        s_ctl[0] = item;
        s_ctl[1] = fr;
        s_ctl[4] = s_ctl[5] = 0u;
    }
    __syncthreads();
    if (s_ctl[2] == 0u) return;
    const int frame = (int)s_ctl[1];
    const unsigned tk = s_ctl[0] * (unsigned)nframes + s_ctl[1];
    (void)tk;
    const int packed = ctl.order[16 + s_ctl[0]];
    const int b = packed & 0xffff, t = packed >> 16;
    const size_t fo = (size_t)frame * frame_stride;
    unsigned *prog_mine = ctl.progress + ((size_t)frame * T + t) * B + b;
    const unsigned *prog_prev = (t > 0) ? prog_mine - B : nullptr;
    const unsigned *prog_east = (t > 0 && b + 1 < B) ? prog_mine - B + 1 : nullptr;

    // every plane through a buffer descriptor: an access outside the plane reads nothing / writes nothing
    __amdgpu_buffer_rsrc_t rs[NF];
    const unsigned plane_bytes = (unsigned)((size_t)nrows * ncols * sizeof(float));
#pragma unroll
    for (int f = 0; f < NIT; f++) rs[f] = __builtin_amdgcn_make_buffer_rsrc(P.it_out[f] + fo, 0, plane_bytes, 0x00020000);
#pragma unroll
    for (int f = 0; f < NRO; f++) rs[NIT + f] = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(P.ro[f]) + fo, 0, plane_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_pack =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(pack) + fo * NCF, 0, (unsigned)((size_t)nrows * ncols * NCF * sizeof(float)), 0x00020000);

    const int jbase = 1 + W * b;
    auto crow = [&](int i) { return i < 0 ? 0 : (i > nrows - 1 ? nrows - 1 : i); };
    auto boff = [&](int jj, int row) { return (unsigned)(((long)jj * nrows + row) * 4); };
    const bool west_by_mail = (b > 0);

    if (role == R_LOAD) {
        // ======================================= loader wave ==========================================
        // Byte offsets of this lane's 16 bytes in every piece of chunk 0; a chunk further down adds a constant.  Rows above the
        // frame or below it fall into a neighbouring column or outside the plane: valid to fetch (or dropped by the range
        // check), and never used -- those rows belong to steps that relax nothing (pdeip_sor_exact.hpp, "Memory").
        unsigned vf[NFP], vp[L::NPK], ve;
#pragma unroll
        for (int d = 0; d < NFP; d++) {
            const int G = 64 * d + lane, c = G >> 2, q = (G & 3) ^ ((c >> 2) & 3);
            int jj = jbase + c;
            jj = jj < ncols - 1 ? jj : ncols - 1;
            vf[d] = boff(jj, 2 - c + 4 * q); // south rows: row of lane c at step 4q of chunk 0 is 1 - c + 4q, its south neighbour one further
        }
#pragma unroll
        for (int d = 0; d < L::NPK; d++) {
            const int G = 64 * d + lane, c = G / CS, gp = G - c * CS, gsrc = gp ^ ((c >> 2) & 3);
            int jj = jbase + c;
            jj = jj < ncols - 1 ? jj : ncols - 1;
            vp[d] = (unsigned)((((long)jj * nrows + (1 - c)) * NCF) * 4) + 16u * (unsigned)gsrc;
        }
        {
            const int which = (lane >> 2) & 1; // lanes 0-3: west column, centre rows of lane 0; lanes 4-7: east column, centre rows of lane 63
            const int ecol = which ? (jbase + W < ncols - 1 ? jbase + W : ncols - 1) : jbase - 1;
            ve = boff(ecol, (which ? 1 - (W - 1) : 1) + 4 * (lane & 3));
        }
        const bool edge_all = lane < 8, edge_east = lane >= 4 && lane < 8;
        auto mine = [&](int i) { return L::NLOAD == 1 || (i % L::NLOAD) == ldw; }; // piece i of the chunk is this loader's
        auto issue = [&](int buf) __attribute__((always_inline)) {
            float *base = smem + buf * L::BUF;
#pragma unroll
            for (int f = 0; f < NF; f++)
#pragma unroll
                for (int d = 0; d < NFP; d++)
                    if (mine(f * NFP + d)) {
                        if (f < NIT) walk_dma16<16>(rs[f], base + f * L::FIELD + d * 256, vf[d]); // sc1
                        else walk_dma16<0>(rs[f], base + f * L::FIELD + d * 256, vf[d]);
                    }
#pragma unroll
            for (int d = 0; d < L::NPK; d++)
                if (mine(NF * NFP + d)) walk_dma16<0>(rs_pack, base + NF * L::FIELD + d * 256, vp[d]);
            // edges: the east column (old values) of every field; the west column of the read-only fields, and of the iterate
            // fields only in the first strip (the frame's border column) -- a later strip's comes by mail (poller wave)
#pragma unroll
            for (int f = 0; f < NF; f++) {
                float *edst = base + NF * L::FIELD + L::PACK + f * 32;
                if (mine(NF * NFP + L::NPK + f)) {
                    if ((f < NIT && west_by_mail) ? edge_east : edge_all) { // a lane mask, never empty: one instruction per field and chunk
                        if (f < NIT) walk_dma16<16>(rs[f], edst, ve); // sc1
                        else walk_dma16<0>(rs[f], edst, ve);
                    }
                }
            }
#pragma unroll
            for (int d = 0; d < NFP; d++) vf[d] += 16u * 4u;
#pragma unroll
            for (int d = 0; d < L::NPK; d++) vp[d] += 16u * NCF * 4u;
            ve += 16u * 4u;
        };
        // dependencies of chunk c: progress(b, t-1) >= c+2, progress(b+1, t-1) >= c-2 (pdeip_sor_exact.hpp); known values are
        // refreshed by the poller wave through s_ctl[4..5]; only a real shortfall makes this wave poll (and drain) itself
        unsigned known_prev = prog_prev ? 0u : 0xffffffffu, known_east = prog_east ? 0u : 0xffffffffu;
        auto deps = [&](int c) __attribute__((always_inline)) {
            const int np = c + 2, ne = c + 2 - W / 16; // the east strip's lane 0 relaxed the rows my last lane needs W/16 - 1 chunks up
            const unsigned need_prev = (unsigned)(np < NC ? np : NC), need_east = (unsigned)(ne < 0 ? 0 : (ne < NC ? ne : NC));
            if (known_prev >= need_prev && known_east >= need_east) return;
            const unsigned kp = s_ctl[4], ke = s_ctl[5]; // what the poller saw last interval
            if (prog_prev && kp > known_prev) known_prev = kp;
            if (prog_east && ke > known_east) known_east = ke;
            if (known_prev >= need_prev && known_east >= need_east) return;
            const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
            for (;;) {
                if (prog_prev) known_prev = __hip_atomic_load(prog_prev, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (prog_east) known_east = __hip_atomic_load(prog_east, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (known_prev >= need_prev && known_east >= need_east) return;
                if (__hip_atomic_load(ctl.abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return;
                if (__builtin_amdgcn_s_memrealtime() - t0 > 50000000ull) { // 0.5 s: drain the grid, the host reports it
                    __hip_atomic_store(ctl.abort_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    return;
                }
                __builtin_amdgcn_s_sleep(2);
            }
        };
        // all of this loader's pieces but those of the newest chunk have landed (the count is an immediate: one wait per loader index)
        static_assert(L::NLOAD == 1 || L::NLOAD == 2, "one or two loader waves");
        auto one_chunk_in_flight = [&]() __attribute__((always_inline)) {
            if (ldw == 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(L::pieces_of(0)) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(L::pieces_of(L::NLOAD > 1 ? 1 : 0)) : "memory");
        };
        deps(0);
        issue(0);
        lds_barrier(); // A: chunk 0's dependencies hold -- the compute waves read their start state
        if (D >= 2 && NC > 1) {
            deps(1);
            issue(1 % NBUF);
            one_chunk_in_flight();
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        lds_barrier(); // 0: chunk 0 is in buffer 0
        P8S_DECL;
        int nb = (D - 1) % NBUF; // stepped at the top of an interval: buffer of chunk k + D
        for (int k = 0; k < NC; k++) {
            WKS_BEGIN;
            pause(0);
            nb = nb + 1 == NBUF ? 0 : nb + 1; // (k + D) % NBUF
            if (k + D < NC) {
                deps(k + D);
                issue(nb);
                if (D >= 2) one_chunk_in_flight(); // chunk k+1 has landed
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            WKS_END(k);
            lds_barrier();
        }
        WKS_WRITE;
        return;
    }

    if (role == R_STORE) {
        // ======================================= storer wave ==========================================
        // relaxed chunk c: LDS -> global, write-through (sc1); lane L of read g holds granule 64 g + L of the out image.
        // Every chunk goes out as one of three FIXED numbers of wave-wide store instructions, so the waits below can be
        // counted.  Piece g holds columns 16g .. 16g+15, whose lanes start 16g .. 16g+15 rows above lane 0:
        //   all rows of the chunk inner rows                  NS  = 4 NIT   16-byte stores;
        //   top of the strip (chunks 0..3, bottom far away)   NST = 8 NIT:  pieces left of the diagonal (g < c) whole, the
        //       piece on it (g = c) as one 16-byte store for the lanes whose four rows are all inner rows + four 4-byte
        //       stores for the lanes with a mixed quad, pieces right of it (g > c) hold no inner row; padded to a fixed count;
        //   anything else (the bottom of the frame)           NSD = 20 NIT: every piece in the five-store form.
        // A lane that has nothing to store gets an offset the range check drops.
        constexpr int NS = NIT * NFP, NST = NIT * (NFP + 4), NSD = NIT * NFP * 5;
        constexpr unsigned DROP = 0xfffffff0u;
        auto store_out = [&](int c) __attribute__((always_inline)) -> int { // returns the number of store instructions
            const float *outb = outb_base + (c & 1) * L::OUTB;
            const int i00 = 1 + EX_CH * c;
            const bool all_valid = (i00 - (W - 1) >= 1) && (i00 + EX_CH - 1 <= nrows - 2); // every row of the chunk is an inner row
            const bool top = !all_valid && c < NFP && (i00 + EX_CH - 1 <= nrows - 2);     // only the frame's top is in reach
            auto piece = [&](int f, int g, bool five) __attribute__((always_inline)) {
                const int G = 64 * g + lane, col = G >> 2, q = (G & 3) ^ ((col >> 2) & 3);
                const int jj = jbase + col;
                const int row = i00 - col + 4 * q;
                const float4 v = *reinterpret_cast<const float4 *>(&outb[f * L::FIELD + 4 * G]);
                v4u_t u;
                u.x = __float_as_uint(v.x); u.y = __float_as_uint(v.y); u.z = __float_as_uint(v.z); u.w = __float_as_uint(v.w);
                const bool col_in = jj <= ncols - 2;
                if (!five) {
                    __builtin_amdgcn_raw_buffer_store_b128(u, rs[f], col_in ? boff(jj, row) : DROP, 0, 16);
                } else {
                    const bool whole = col_in && row >= 1 && row + 3 <= nrows - 2;
                    __builtin_amdgcn_raw_buffer_store_b128(u, rs[f], whole ? boff(jj, row) : DROP, 0, 16);
                    const unsigned vv[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
                    for (int e = 0; e < 4; e++) {
                        const bool one = col_in && !whole && row + e >= 1 && row + e <= nrows - 2;
                        __builtin_amdgcn_raw_buffer_store_b32(vv[e], rs[f], one ? boff(jj, row + e) : DROP, 0, 16);
                    }
                }
            };
            if (all_valid) {
#pragma unroll
                for (int f = 0; f < NIT; f++)
#pragma unroll
                    for (int g = 0; g < NFP; g++) piece(f, g, false);
                return NS;
            }
            if (top) {
#pragma unroll
                for (int f = 0; f < NIT; f++) {
#pragma unroll
                    for (int g = 0; g < NFP - 1; g++) { // pieces left of the diagonal, or a dropped store in their place
                        if (g < c) piece(f, g, false);
                        else __builtin_amdgcn_raw_buffer_store_b32(0u, rs[f], DROP, 0, 16);
                    }
                    if (c == 0) piece(f, 0, true);
                    else if (c == 1 || NFP < 3) piece(f, 1, true);
                    else if (c == 2 || NFP < 4) piece(f, 2, true);
                    else piece(f, NFP - 1, true);
                }
                return NST;
            }
#pragma unroll
            for (int f = 0; f < NIT; f++)
#pragma unroll
                for (int g = 0; g < NFP; g++) piece(f, g, true);
            return NSD;
        };
        // progress = c+1 once every store of chunks <= c has left (Guideline 16, R1: drain, then the flag).  A write-through
        // store is acknowledged 1-3 us after it was issued, and this wave meets the others at a barrier per chunk: waiting
        // for its own stores every interval made it the slowest wave of the walk, at the top of a strip above all -- and
        // the first five chunks of a strip are what its east neighbour's start waits for (stamps, round 3).  So the counter
        // trails: after the stores of chunk k-1 have been ISSUED, a counted wait leaves the youngest m <= 3 chunks in flight
        // (their stores + one counter store per interval: vmcnt counts stores in issue order) and publishes the chunk
        // before them.  s_waitcnt takes an immediate: the count allowed is rounded down to a ladder of immediates.
        int published = 0;
        int n1 = 64, n2 = 64; // store instructions of chunks k-2, k-3 (64: none in flight that could be counted)
        // one counter store per call, whether the value moves or not: the counted waits rely on it
        auto publish_to = [&](int upto) __attribute__((always_inline)) { // progress = upto (chunks 0 .. upto-1 are complete)
            if (upto > published) published = upto;
            if (lane == 0) __hip_atomic_store(prog_mine, (unsigned)published, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        };
        auto wait_all_but = [&](int allowed) __attribute__((always_inline)) { // at most `allowed` of this wave's stores still in flight
            if (allowed >= 58) asm volatile("s_waitcnt vmcnt(58)" ::: "memory");
            else if (allowed >= 50) asm volatile("s_waitcnt vmcnt(50)" ::: "memory");
            else if (allowed >= 42) asm volatile("s_waitcnt vmcnt(42)" ::: "memory");
            else if (allowed >= 34) asm volatile("s_waitcnt vmcnt(34)" ::: "memory");
            else if (allowed >= 26) asm volatile("s_waitcnt vmcnt(26)" ::: "memory");
            else if (allowed >= 20) asm volatile("s_waitcnt vmcnt(20)" ::: "memory");
            else if (allowed >= 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
            else if (allowed >= 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
            else if (allowed >= 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else if (allowed >= 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        };
        lds_barrier(); // A
        lds_barrier(); // 0
        P8S_DECL;
        for (int k = 0; k < NC; k++) {
            WKS_BEGIN;
            pause(4);
            if (k >= 1) {
                const int n0 = store_out(k - 1);
                // youngest first: stores(k-1) | counter, stores(k-2) | counter, stores(k-3) | counter, stores(k-4) ...
                if (k >= 3 && n0 + n1 + n2 + 2 <= 63) { // chunks k-1, k-2, k-3 stay in flight
                    wait_all_but(n0 + n1 + n2 + 2);
                    publish_to(k - 3);
                } else if (k >= 2 && n0 + n1 + 1 <= 63) {
                    wait_all_but(n0 + n1 + 1);
                    publish_to(k - 2);
                } else if (n0 <= 63) {
                    wait_all_but(n0);
                    publish_to(k - 1);
                } else {
                    wait_all_but(0);
                    publish_to(k);
                }
                n2 = n1;
                n1 = n0;
            }
            WKS_END(k);
            lds_barrier();
        }
        store_out(NC - 1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        publish_to(NC);
        WKS_WRITE;
        return;
    }

    if (role == R_POLL) {
        // ======================================= poller wave ==========================================
        // West edge of chunk c: the west strip's compute waves leave lane 63's 16 results of a chunk as one 128-byte line of
        // {value, tag} words per field.  This wave fetches the lines of the next chunk by LDS-DMA (8 lanes x 16 bytes per field,
        // write-through data read with sc1) into a staging area, and -- split phase -- looks at them one interval later: if
        // every tag is set the values go to the west slot of the chunk's buffer.  In the steady state (west strip more than
        // a round trip ahead) it never waits and the barrier never waits for it; only a chunk that is needed NOW and has not
        // arrived is fetched in a loop.  The same way lanes 0 / 1 fetch the previous sweep's two progress counters for the
        // loader (s_ctl[4..5]).  No load of this wave has a register destination: nothing the compiler could wait for early.
        const int mf = lane >> 4, r16 = lane & 15;
        const bool polls = (b > 0) && (lane < 16 * NIT);
        const size_t mpitch = (size_t)NC * EX_CH;
        unsigned long long *const mail_west = ctl.mail + ((((size_t)frame * T + t) * B + (b > 0 ? b - 1 : 0)) * NIT) * mpitch;
        const __amdgpu_buffer_rsrc_t rs_mw = __builtin_amdgcn_make_buffer_rsrc(mail_west, 0, (unsigned)((size_t)NIT * mpitch * 8), 0x00020000);
        unsigned long long *const stagew = reinterpret_cast<unsigned long long *>(s_ctl + 16); // [NIT][16] words
        float *const edge0 = smem + NF * L::FIELD + L::PACK;
        const unsigned *my_prog = lane == 0 ? prog_prev : (lane == 1 ? prog_east : nullptr);
        (void)my_prog;
        int have = 0, pend_c = -1; // chunks 0 .. have-1 have been handed over; pend_c: the chunk the pending fetch was issued for
        auto fetch = [&](int c) __attribute__((always_inline)) { // chunk c's lines (c < NC) and the progress counters -> staging
            pend_c = c;
            if (b > 0 && c < NC && lane < 8) {
#pragma unroll
                for (int f = 0; f < NIT; f++) walk_dma16<16>(rs_mw, reinterpret_cast<float *>(stagew + f * 16), (unsigned)(((size_t)f * mpitch + (size_t)EX_CH * c) * 8) + 16u * lane);
            }
#if defined(__HIP_DEVICE_COMPILE__)
            if (my_prog != nullptr) __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)my_prog, WALK_LDS(s_ctl + 4), 4, 0, 16);
#endif
        };
        auto landed = [&]() __attribute__((always_inline)) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); };
        // looks at the staged lines of chunk c; hands them over if they are complete
        auto look = [&](int c) __attribute__((always_inline)) -> bool {
            const bool want = polls && (1 + EX_CH * c + r16) <= nrows - 2;
            const unsigned long long w = want ? stagew[lane] : (1ull << 32); // lane = 16 f + row
            if (!__all((unsigned)(w >> 32) != 0u)) return false;
            if (polls) (edge0 + (c % NBUF) * L::BUF)[mf * 32 + r16] = want ? __uint_as_float((unsigned)w) : 0.0f;
            have = c + 1;
            return true;
        };
        auto take_now = [&](int c) __attribute__((always_inline)) { // chunk c is needed before the next barrier
            if (b == 0) { have = c + 1; return; }
            const unsigned long long t0 = __builtin_amdgcn_s_memrealtime(); // 100 MHz
            for (;;) {
                fetch(c);
                landed();
                if (look(c)) break;
                if (__hip_atomic_load(ctl.abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) { have = c + 1; break; }
                if (__builtin_amdgcn_s_memrealtime() - t0 > 50000000ull) { // 0.5 s: drain the grid, the host reports it
                    __hip_atomic_store(ctl.abort_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    have = c + 1;
                    break;
                }
                __builtin_amdgcn_s_sleep(1);
            }
            pend_c = -1;
        };
        // one interval: the edge of chunk `need` must be in LDS before the barrier; the chunk after it may be asked for
        auto interval = [&](int need, int ahead) __attribute__((always_inline)) {
            if (pend_c >= 0) { // last interval's fetch has had a whole interval to land
                landed();
                if (pend_c == have && have < NC) look(pend_c);
                pend_c = -1;
            }
            while (have <= need && have < NC) take_now(have);
            fetch(have <= ahead ? have : NC); // past NC: only the progress counters
        };
        lds_barrier(); // A
        interval(0, 1);
        lds_barrier(); // 0
        P8S_DECL;
        for (int k = 0; k < NC; k++) {
            WKS_BEGIN;
            pause(8);
            interval(k + 1, k + 2);
            WKS_END(k);
            lds_barrier();
        }
        landed(); // nothing may land in LDS after the workgroup has gone
        WKS_WRITE;
        return;
    }

    // mailbox rows of this strip (written by its compute waves): [NIT][NC * 16] words
    unsigned long long *const mail_mine = ctl.mail + ((((size_t)frame * T + t) * B + b) * NIT) * ((size_t)NC * EX_CH);
    // ======================================== compute wave(s) ==========================================
    const int j = jbase + lane;
    const bool lane_in = W == 64 || lane < W; // lanes W .. 63 of a narrower strip relax nothing
    const bool col_ok = lane_in && j <= ncols - 2;
    const int jc = j < ncols - 1 ? j : ncols - 1;
    const float om1 = 1.0f - omega;
    const bool first_sweep = (t == 0);
    const bool has_east = (b + 1 < B); // my last column is the next strip's west column
    const int i0w63 = 1 - (W - 1);     // the last lane's row at step 0 of chunk 0
    const __amdgpu_buffer_rsrc_t rs_mail = __builtin_amdgcn_make_buffer_rsrc(mail_mine, 0, (unsigned)((size_t)NIT * NC * EX_CH * 8), 0x00020000);
    const int i0 = 1 - lane; // row of this lane at step 0 of chunk 0
    const int lw = lane_in ? lane : W - 1; // the lane whose LDS image an idle lane reads (stays inside the buffers)
    const int sw = (lw >> 2) & 3;

    auto compute_wave = [&](auto f0_tag, auto nfw_tag) __attribute__((always_inline)) {
    // this wave relaxes the fields [F0, F0 + NFW); of the other fields it only follows the centre value (the coupling term)
    constexpr int F0 = decltype(f0_tag)::value, NFW = decltype(nfw_tag)::value;
    auto mine = [](int f) { return f >= F0 && f < F0 + NFW; };
    lds_barrier(); // A: the loader has passed the dependency wait of chunk 0
    // state at step 0: the centre value of row 1 - lane (only lane 0's is ever used: the others pick theirs up from the staged
    // rows before they relax anything) and the frame's top border cell of this column (the north tap of row 1 in sweep 0)
    float prev[NIT], cen[NIT], topb[NIT], rcen[NRO1], rnorth[NRO1];
#pragma unroll
    for (int f = 0; f < NIT; f++) {
        cen[f] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs[f], boff(jc, crow(i0)), 0, 16));
        topb[f] = prev[f] = 0.0f;
        if (mine(f)) topb[f] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs[f], boff(jc, 0), 0, 16));
    }
#pragma unroll
    for (int f = 0; f < NRO1; f++) {
        rcen[f] = (NRO > 0) ? __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs[(NRO > 0 ? NIT + f : 0)], boff(jc, crow(i0)), 0, 0)) : 0.0f;
        rnorth[f] = (NRO > 0) ? __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs[(NRO > 0 ? NIT + f : 0)], boff(jc, crow(i0 - 1)), 0, 0)) : 0.0f;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    // Two dry passes before the walk: a strip's first chunks are what its east neighbour's start waits for, and the first
    // execution of a chunk body (5-10 KB of straight-line code) ran 1-2 us slower than the later ones -- instruction fetch.
    // While this wave waits for the west strip's first values anyway, it runs the top-of-strip body and the inner body once
    // on whatever buffer 0 holds (results go to an out image nobody reads yet, the state is put back afterwards).
    float keep_prev[NIT], keep_cen[NIT], keep_rcen[NRO1], keep_rnorth[NRO1];
#pragma unroll
    for (int f = 0; f < NIT; f++) { keep_prev[f] = prev[f]; keep_cen[f] = cen[f]; }
#pragma unroll
    for (int f = 0; f < NRO1; f++) { keep_rcen[f] = rcen[f]; keep_rnorth[f] = rnorth[f]; }
    const int k_inner = (EX_CH * NFP + EX_CH <= nrows - 3) ? NFP : 0; // a chunk whose rows are all inner rows, if the frame has one

    P8S_DECL;
    int bi = 0; // k % NBUF
    for (int kk = -2; kk < NC; kk++) {
        const bool dry = kk < 0;
        const int k = dry ? (kk == -2 ? 0 : k_inner) : kk;
        if (!dry) { WKS_BEGIN; pause(12); }
        const float *stage = smem + bi * L::BUF, *edge = stage + NF * L::FIELD + L::PACK;
        const float4 *cimg = reinterpret_cast<const float4 *>(stage + NF * L::FIELD) + lw * CS;
        float *outb = outb_base + (k & 1) * L::OUTB;
        if (!dry) bi = bi + 1 == NBUF ? 0 : bi + 1;
        const int i00 = 1 + EX_CH * k;
        auto relax_chunk = [&](auto rows_tag, auto cols_tag) __attribute__((always_inline)) {
        // rows: 2 = every lane's row is an inner row at every step and no tap is a border row; 1 = the top of the strip (rows above
        // the frame are possible, the bottom is out of reach: a lane only has to know whether it has started, and its first
        // pixel takes the top border cell as north tap); 0 = anything.  cols: 1 = the strip holds neither column 1 nor
        // ncols-2 and is complete; 0 = anything.  The first chunks of a strip are what its east neighbour's start waits
        // for, so they get a path of their own instead of the generic one (40 instead of 24 instructions per step).
        constexpr int ROWS = decltype(rows_tag)::value;
        constexpr bool ROWS_IN = ROWS == 2, ROWS_TOP = ROWS == 1, COLS_IN = decltype(cols_tag)::value == 1, INTERIOR = ROWS_IN && COLS_IN;
#ifndef PDEIP_WALK_UNROLL_MQ
#define PDEIP_WALK_UNROLL_MQ 1
#endif
        // a rolled loop over the four 4-step groups of a chunk: a quarter of the code (the kernel's five bodies x two compute waves
        // were 90 KB unrolled, more than the instruction cache two compute units share, and its run time moved by 15-20 % with
        // the placement of unrelated code)
#pragma unroll PDEIP_WALK_UNROLL_MQ
        for (int mq = 0; mq < EX_CH / 4; mq++) {
            float4 s4[NF], e4[NF], res[NIT];
            float cflat[4 * NCF]; // coefficients of the four rows of this group, [row][f]: granules NCF mq .. NCF mq + NCF - 1 of my run
#pragma unroll
            for (int jq = 0; jq < NCF; jq++) {
                const float4 v = cimg[(NCF * mq + jq) ^ sw];
                cflat[4 * jq] = v.x; cflat[4 * jq + 1] = v.y; cflat[4 * jq + 2] = v.z; cflat[4 * jq + 3] = v.w;
            }
#pragma unroll
            for (int f = 0; f < NF; f++) {
                s4[f] = *reinterpret_cast<const float4 *>(&stage[f * L::FIELD + 4 * (4 * lw + (mq ^ sw))]);
                e4[f] = *reinterpret_cast<const float4 *>(&edge[f * 32 + (lane >= W - 1 ? 16 : 0) + 4 * mq]);
            }
#pragma unroll
            for (int x = 0; x < 4; x++) {
                const int q = EX_CH * k + 4 * mq + x;
                const int i = i0 + q;
                const bool row_ok = ROWS_IN || (ROWS_TOP ? (i >= 1) : ((i >= 1) && (i <= nrows - 2)));
                const bool active = INTERIOR || ((COLS_IN || col_ok) && row_ok);
                auto el = [&](const float4 &v) { return x == 0 ? v.x : (x == 1 ? v.y : (x == 2 ? v.z : v.w)); };
                float c[NIT], w[NIT], e[NIT], n[NIT], s[NIT], kk[NCF];
                float rsouth[NRO1], reast[NRO1], rwest[NRO1];
#pragma unroll
                for (int f = 0; f < NIT; f++) {
                    const float sraw = el(s4[f]);
                    c[f] = cen[f];
                    if (mine(f)) {
                        float eraw = dpp_from_upper_lane(sraw, el(e4[f]));
                        if (W < 64 && lane == W - 1) eraw = el(e4[f]); // the strip's last lane: the east column, not an idle lane's value
                        const float wnew = dpp_from_lower_lane(prev[f], el(e4[f]));
                        float nv = prev[f];
                        if (!ROWS_IN && i == 1) nv = first_sweep ? topb[f] : cen[f];
                        n[f] = nv;
                        s[f] = (ROWS == 0 && i + 1 == nrows - 1 && !first_sweep) ? cen[f] : sraw;
                        e[f] = (!COLS_IN && j + 1 == ncols - 1 && !first_sweep) ? cen[f] : eraw;
                        w[f] = (!COLS_IN && j - 1 == 0 && !first_sweep) ? cen[f] : wnew;
                    } else {
                        n[f] = s[f] = e[f] = w[f] = 0.0f; // the other field's neighbours feed nothing this wave keeps
                    }
                    cen[f] = sraw;
                }
#pragma unroll
                for (int f = 0; f < NRO1; f++) {
                    if (NRO > 0) {
                        rsouth[f] = el(s4[(NRO > 0 ? NIT + f : 0)]);
                        reast[f] = dpp_from_upper_lane(rsouth[f], el(e4[(NRO > 0 ? NIT + f : 0)]));
                        if (W < 64 && lane == W - 1) reast[f] = el(e4[(NRO > 0 ? NIT + f : 0)]);
                        rwest[f] = dpp_from_lower_lane(rnorth[f], el(e4[(NRO > 0 ? NIT + f : 0)]));
                    } else {
                        rsouth[f] = reast[f] = rwest[f] = 0.0f;
                    }
                }
#pragma unroll
                for (int f = 0; f < NCF; f++) kk[f] = cflat[NCF * x + f];
                Mdl::update(c, w, e, n, s, rcen, rwest, reast, rnorth, rsouth, kk, omega, om1);
#pragma unroll
                for (int f = F0; f < F0 + NFW; f++) {
                    if (active) prev[f] = c[f];
                    const float r = c[f];
                    if (x == 0) res[f].x = r; else if (x == 1) res[f].y = r; else if (x == 2) res[f].z = r; else res[f].w = r;
                }
#pragma unroll
                for (int f = 0; f < NRO1; f++) {
                    rnorth[f] = rcen[f];
                    rcen[f] = rsouth[f];
                }
            }
#pragma unroll
            for (int f = F0; f < F0 + NFW; f++)
                if (lane_in) *reinterpret_cast<float4 *>(&outb[f * L::FIELD + 4 * (4 * lane + (mq ^ sw))]) = res[f];
        }
        };
        {
            const int lo_row = i00 - (W - 1), hi_row = i00 + EX_CH - 1;
            const bool rows_in = (lo_row >= 2) && (hi_row <= nrows - 3), rows_top = hi_row <= nrows - 3, cols_in = (jbase >= 2) && (jbase + W - 1 <= ncols - 3);
            using I0 = std::integral_constant<int, 0>;
            using I1 = std::integral_constant<int, 1>;
            using I2 = std::integral_constant<int, 2>;
            if (rows_in && cols_in) relax_chunk(I2{}, I1{});
            else if (rows_in) relax_chunk(I2{}, I0{});
            else if (rows_top && cols_in) relax_chunk(I1{}, I1{});
            else if (rows_top) relax_chunk(I1{}, I0{});
            else relax_chunk(I0{}, I0{});
        }
        if (has_east && !dry) {
            // Mailbox: lane 63's 16 results of this chunk (rows i0 + 16k .. + 15 of my last column) go out as ONE 128-byte line of
            // self-validating {value, tag} words per field -- lanes 0..15 pick them up from the out image this wave just wrote
            // (LDS operations of one wave complete in order) and store 8 bytes each.
            typedef unsigned int v2u_t __attribute__((ext_vector_type(2)));
            const int r = (i0w63 + EX_CH * k) + (lane & 15); // row of word (lane & 15)
            if (lane < 16 && r >= 1 && r <= nrows - 2) {
#pragma unroll
                for (int f = F0; f < F0 + NFW; f++) {
                    v2u_t wv;
                    wv.x = __float_as_uint(outb[f * L::FIELD + 4 * (4 * (W - 1) + ((lane >> 2) ^ 3)) + (lane & 3)]); // column W-1: s(W-1) = 3
                    wv.y = 1u;
                    __builtin_amdgcn_raw_buffer_store_b64(wv, rs_mail, (unsigned)(((size_t)f * NC * EX_CH + r - 1) * 8), 0, 16); // sc1
                }
            }
        }
        if (dry) {
            if (kk == -1) { // back to the state of step 0, then barrier 0: chunk 0 is in buffer 0
#pragma unroll
                for (int f = 0; f < NIT; f++) { prev[f] = keep_prev[f]; cen[f] = keep_cen[f]; }
#pragma unroll
                for (int f = 0; f < NRO1; f++) { rcen[f] = keep_rcen[f]; rnorth[f] = keep_rnorth[f]; }
                lds_barrier();
            }
            continue;
        }
        WKS_END(k);
        lds_barrier();
    }
    WKS_WRITE;
    };
    if constexpr (NIT == 2) {
        if (role == R_COMP0) compute_wave(std::integral_constant<int, 0>{}, std::integral_constant<int, 1>{});
        else compute_wave(std::integral_constant<int, 1>{}, std::integral_constant<int, 1>{});
    } else {
        compute_wave(std::integral_constant<int, 0>{}, std::integral_constant<int, NIT>{});
    }
}

} // namespace pdeip
