// pdeip_sor_pde8_persist.hpp -- GS_SOR_8_2d (pdeSolvers.c:184-262) in the reference's lexicographic order, ONE launch per call.
//
// k_pde8_exact (pdeip_sor_pde8.hpp) runs one launch per front m = a + 3b + 4t of 64-step tiles: 225 launches for a 4K frame at
// iter = 4, each a cold start (launch gap, first fetch, store tail), and the tile grid makes a strip trail its west neighbour
// by 192 rows where the stencil needs 128.  Here a workgroup owns one (strip b, sweep t) and walks down the strip chunk by
// chunk (16 steps), as k_sor_exact_persist does for the 5-point models; the front ordering is replaced by progress counters:
//   chunk c of (b,t) needs   progress[b][t-1]   >= c+2   own columns, 20 staged rows of sweep t-1
//                            progress[b+1][t-1] >= c-6   east column of sweep t-1 (and: that strip has read what I overwrite)
//                            the west strip's east column of sweep t down to row 16c+15: its chunk c+8 -- by mailbox, see below
// (derivation in DESIGN.md 5.3b; the two counters are checked before the chunk is FETCHED, two chunks ahead of its relaxation).
// West edge: lane 63's 16 results of a chunk (a border row's handed-on value included) leave as one 128-byte line of
// {value, tag} words, written through by the compute wave itself; the east strip's storer polls the 18 words the next chunk
// needs and puts them into that chunk's LDS edge slot.  One round trip, and the bulk fetch does not wait for the west strip
// at all: a strip trails its neighbour by ~10 chunks instead of 14 (through the plane: store, drain, publish, poll, fetch two
// chunks ahead).  The first strip's west column is the image border: its loader stages it from the plane / the ring.
//
// Geometry: lane l owns column jbase + l and relaxes row 16c - 1 - 2l + q at step q of chunk c (two rows of skew per lane: the
// south-west tap (i+1, j-1) is the west lane's result of the step before).  Every lane starts above the image and falls
// through: a cell that is not relaxed (a border cell, a row outside the image) hands its staged value on unchanged, so the
// north tap of row 1 and the west lane's taps of rows 0 and nrows-1 are the border cells without any start-up state.
// Border cells hold the previous sweep's replicate (pdeSolvers.c:249-262 runs after the sweep): sweep t records its ring
// (top/bottom/left/right) in a side array of its own, sweep t+1's loader substitutes ring values while staging.
//
// Hand-off as in k_sor_exact_persist: the iterate and the ring are stored write-through (sc1), drained, then the counter is
// published with a relaxed agent-scope store; consumers poll relaxed and read with sc1 loads.  Tickets map to (b,t) in an
// order in which every dependency has a smaller ticket; every spin is bounded and raises the sticky abort word.
#pragma once
#include "pdeip_sor_pde8.hpp"

namespace pdeip {

constexpr int P8P_THREADS = 256; // compute, loader, storer, west-edge poller: one wave per SIMD

// Packed coefficients.  The walk's pace is its loader, and the loader's cost is the number of cache lines its loads touch: ten
// coefficient planes give a (column, chunk) ten 64-byte pieces in ten places.  The pre-pass that derives B_temp / INV_TRACE
// (pdeSolvers.c:190-206) therefore writes all ten coefficients of a pixel side by side, in blocks of two rows:
//   pack[col][pb][f][2]   pb = (row + 1) / 2, element (row + 1) & 1   (rows -1 and nrows.. are padding)
// so the 16 rows a lane starts at row 16c - 1 - 2l (row + 1 even) are 640 contiguous, 16-byte aligned bytes.
__host__ __device__ inline int pde8_pack_blocks(int nrows) { return (nrows + 2) / 2; }
inline size_t pde8_pack_floats(int nrows, int ncols, int nframes) { return (size_t)pde8_pack_blocks(nrows) * ncols * 2 * ModelPde8::NCF * nframes; }

static __global__ void k_pde8_pack(float *pack, const float *TRACE, const float *B, const float *wW, const float *wNW, const float *wN,
                                   const float *wNE, const float *wE, const float *wSE, const float *wS, const float *wSW, int nrows,
                                   int ncols, size_t frame_stride)
{
    constexpr int NCF = ModelPde8::NCF;
    const int nb = pde8_pack_blocks(nrows);
    // last column first: see k_pack_coefficients
    const int pb = blockIdx.x * blockDim.x + threadIdx.x, col = ncols - 1 - (int)blockIdx.y;
    if (pb >= nb) return;
    const size_t fo = (size_t)blockIdx.z * frame_stride;
    float v[NCF][2];
#pragma unroll
    for (int r2 = 0; r2 < 2; r2++) {
        const int row = 2 * pb + r2 - 1;
        const bool in = row >= 0 && row < nrows;
        const size_t p = fo + (size_t)col * nrows + (in ? row : 0);
        const float tr = TRACE[p];
        float t = wE[p] + wW[p];
        t += wS[p] + wN[p];
        t += wSW[p] + wNW[p];
        t += wSE[p] + wNE[p];
        const bool ok = !is_nan(tr);
        const float c[NCF] = {ok ? B[p] : 0.0f, ok ? 1.0f / tr : 1.0f / t, wW[p], wNW[p], wN[p], wNE[p], wE[p], wSE[p], wS[p], wSW[p]};
#pragma unroll
        for (int f = 0; f < NCF; f++) v[f][r2] = in ? c[f] : 0.0f;
    }
    float4 *dst = reinterpret_cast<float4 *>(pack + (((size_t)blockIdx.z * ncols + col) * nb + pb) * (2 * NCF));
#pragma unroll
    for (int q = 0; q < NCF / 2; q++) dst[q] = make_float4(v[2 * q][0], v[2 * q][1], v[2 * q + 1][0], v[2 * q + 1][1]);
}

// LDS image of a chunk's coefficients: the packed run of every column as it lies in memory, 40 granules of 16 bytes; granule q of
// column c at granule c * P8_CS + (q ^ ((c >> 2) & 3)) (pdeip_sor_exact.hpp, PackLayout: 16 compute lanes read granule q of 16
// columns from 16 different bank groups).
constexpr int P8_RUN = 40, P8_CS = 44;
static_assert(64 * P8_CS * 4 <= Pde8Layout::CST, "the packed image fits the coefficient region");

inline int pde8_persist_chunks(int nrows) { return (nrows + 127 + EX_CH - 1) / EX_CH; } // lane 63 reaches row nrows-1 (handed on to the east strip)

__global__ void __launch_bounds__(P8P_THREADS)
k_pde8_exact_persist(Pde8Planes P, const float *pack, float *side, PersistCtl ctl, int nrows, int ncols, int B, int T, int NC, int nframes, float omega,
                     size_t frame_stride)
{
    using L = Pde8Layout;
    constexpr int NCF = ModelPde8::NCF;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *outb_base = smem + 2 * L::BUF;
    unsigned *s_ticket = reinterpret_cast<unsigned *>(smem + 2 * L::BUF + 2 * L::OUTB); // behind the 16-byte aligned region

    const int lane = threadIdx.x & 63;
    const int role = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6); // 0 compute, 1 loader, 2 storer, 3 west edge
    if (threadIdx.x == 0) {
        unsigned item = 0, fr = 0;
        s_ticket[2] = persist_take_item(ctl, nframes, &item, &fr) ? 1u : 0u;
        s_ticket[0] = item;
        s_ticket[1] = fr;
    }
    __syncthreads();
    if (s_ticket[2] == 0u) return;
    const int frame = (int)s_ticket[1];
    const unsigned tk = s_ticket[0] * (unsigned)nframes + s_ticket[1]; // dense id of (item, frame): diagnostics only
    (void)tk;
    const int packed = ctl.order[16 + s_ticket[0]];
    const int b = packed & 0xffff, t = packed >> 16;
    const size_t fo = (size_t)frame * frame_stride;
    unsigned *prog_mine = ctl.progress + ((size_t)frame * T + t) * B + b;
    const unsigned *prog_prev = (t > 0) ? prog_mine - B : nullptr;
    const unsigned *prog_east = (t > 0 && b + 1 < B) ? prog_mine - B + 1 : nullptr;

    // every plane through a range-checked buffer descriptor (an access outside the plane reads 0 / writes nothing)
    const unsigned plane_bytes = (unsigned)((size_t)nrows * ncols * sizeof(float));
    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(P.x + fo, 0, plane_bytes, 0x00020000);
    const int nb = pde8_pack_blocks(nrows);
    const unsigned pack_bytes = (unsigned)((size_t)ncols * nb * 2 * NCF * sizeof(float));
    const __amdgpu_buffer_rsrc_t rs_pack =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(pack) + (size_t)frame * ncols * nb * 2 * NCF, 0, pack_bytes, 0x00020000);
    // border ring after sweep t-1 (read) and after sweep t (written): top[ncols] bot[ncols] left[nrows] right[nrows]
    const size_t sstride = pde8_side_stride(nrows, ncols);
    float *ring_w = side + ((size_t)frame * T + t) * sstride;
    const unsigned ring_bytes = (unsigned)(sstride * sizeof(float));
    const __amdgpu_buffer_rsrc_t rs_ring_w = __builtin_amdgcn_make_buffer_rsrc(ring_w, 0, ring_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_ring_r = __builtin_amdgcn_make_buffer_rsrc(t > 0 ? ring_w - sstride : ring_w, 0, ring_bytes, 0x00020000);

    const int jbase = 1 + 64 * b;
    auto ccol = [&](int jj) { return jj < 0 ? 0 : (jj > ncols - 1 ? ncols - 1 : jj); };
    auto crow = [&](int i) { return i < 0 ? 0 : (i > nrows - 1 ? nrows - 1 : i); };
    auto boff = [&](int jj, int row) { return (unsigned)(((long)jj * nrows + row) * 4); }; // negative -> out of range -> 0
    auto row0 = [&](int c) { return EX_CH * c - 1; };                                       // lane 0's row at step 0 of chunk c
    const int lcol = lane >> 2, lrq = lane & 3;
    auto as_f4u = [](v4u_t v, f4u &o) {
        o.v[0] = __uint_as_float(v.x); o.v[1] = __uint_as_float(v.y); o.v[2] = __uint_as_float(v.z); o.v[3] = __uint_as_float(v.w);
    };

    if (role == 1 || role == 3) {
        // ============================ loader wave, west-edge wave ==============================
        // The loader's pace is the walk's pace, so the wave that only polls the mailbox takes half of the coefficient loads: the
        // loader stages columns 0-31 of the packed coefficients and everything of the iterate, the west-edge wave columns 32-63
        // (its mailbox polls queue behind its loads -- vmcnt counts in order -- which costs the hand-off less than it saves).
        const int gc0 = (role == 1) ? 0 : 2;
        const bool xs = (role == 1);
        f4u cA[5][4], cB[5][4], xA[6], xB[6];
        const int l8 = lane >> 3, part = lane & 7; // coefficient loads: eight lanes to a column, 128 contiguous bytes per instruction
        const unsigned *my_ptr = lane == 1 ? prog_prev : (lane == 2 ? prog_east : nullptr); // the west strip: by mailbox (storer)
        auto wait_deps = [&](int c) __attribute__((always_inline)) {
            const int need = lane == 1 ? c + 2 : c - 6;
            persist_wait3(my_ptr, (unsigned)(need < 0 ? 0 : (need < NC ? need : NC)), ctl.abort_flag);
        };
        // the replicate of sweep t-1 for a border cell of the image (t > 0)
        auto ring_cell = [&](int ii, int jj) __attribute__((always_inline)) -> float {
            int idx;
            if (ii == 0) idx = jj;
            else if (ii == nrows - 1) idx = ncols + jj;
            else if (jj == 0) idx = 2 * ncols + ii;
            else idx = 2 * ncols + nrows + ii;
            return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs_ring_r, (unsigned)idx * 4u, 0, 16));
        };
        // replace the border cells among rows row..row+3 of column jj by the previous sweep's ring values
        auto patch = [&](f4u &v, int row, int jj) __attribute__((always_inline)) {
            if (jj < 0 || jj > ncols - 1) return;
            const bool colb = (jj == 0) || (jj == ncols - 1);
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const int ii = row + e;
                if (ii >= 0 && ii <= nrows - 1 && (colb || ii == 0 || ii == nrows - 1)) v.v[e] = ring_cell(ii, jj);
            }
        };
        auto fetch = [&](int c, f4u (&cpre)[5][4], f4u (&xpre)[6]) __attribute__((always_inline)) {
            const int r0 = row0(c);
#pragma unroll
            for (int gg = 0; gg < 4; gg++) {
                const int col = 8 * (2 * gc0 + gg) + l8;
                const int jj = ccol(jbase + col);
                // coefficients: the eight lanes of a column read 128 contiguous bytes of its 640-byte run per instruction
                const unsigned run = (unsigned)((((long)jj * nb + ((r0 - P8_SKEW * col + 1) >> 1)) * (2 * NCF)) * 4) + 16u * (unsigned)part;
#pragma unroll
                for (int k = 0; k < 5; k++) as_f4u(__builtin_amdgcn_raw_buffer_load_b128(rs_pack, run + 128u * k, 0, 0), cpre[k][gg]);
            }
            if (!xs) return;
            // does a staged window hold border cells of sweep t-1's ring?  rows r0-128 .. r0+19, columns jbase-1 .. jbase+64
            const bool ringed = (t > 0) && ((r0 - 128 <= 0 && r0 + 19 >= 0) || (r0 - 128 <= nrows - 1 && r0 + 19 >= nrows - 1) ||
                                            jbase - 1 == 0 || jbase + 64 >= ncols - 1);
#pragma unroll
            for (int g = 0; g < 4; g++) {
                const int col = 16 * g + lcol;
                const int jj = ccol(jbase + col);
                const int row = r0 - P8_SKEW * col + 4 * lrq;
                as_f4u(__builtin_amdgcn_raw_buffer_load_b128(rs_x, boff(jj, row), 0, 16), xpre[g]);
            }
            const int row4 = r0 - P8_SKEW * lane + 16; // fifth quad (rows +16..+19) of every own column: lane -> column
            as_f4u(__builtin_amdgcn_raw_buffer_load_b128(rs_x, boff(ccol(jbase + lane), row4), 0, 16), xpre[4]);
            // lanes 0-4: east edge column (as lane 64); lanes 5-9: west edge column, rows r0-1 ...; the others repeat
            const int w = lane % 10;
            const bool west = w >= 5;
            const int ej = west ? jbase - 1 : jbase + 64;
            const int erow = west ? r0 - 1 + 4 * (w - 5) : r0 - P8_SKEW * 64 + 4 * w;
            // row by row, clamped: the first strip's west column is column 0, where a negative row is a negative offset (the whole
            // vector would read as out of range, valid rows included)
            if (r0 - 128 >= 0 && r0 + 18 <= nrows - 1) { // every edge row lies in the plane: one vector
                as_f4u(__builtin_amdgcn_raw_buffer_load_b128(rs_x, boff(ccol(ej), erow), 0, 16), xpre[5]);
            } else {
#pragma unroll
                for (int e = 0; e < 4; e++)
                    xpre[5].v[e] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs_x, boff(ccol(ej), crow(erow + e)), 0, 16));
            }
            if (ringed) {
#pragma unroll
                for (int g = 0; g < 4; g++) patch(xpre[g], r0 - P8_SKEW * (16 * g + lcol) + 4 * lrq, jbase + 16 * g + lcol);
                patch(xpre[4], row4, jbase + lane);
                patch(xpre[5], erow, ej);
            }
        };
        auto stash = [&](const f4u (&cpre)[5][4], const f4u (&xpre)[6], int buf) __attribute__((always_inline)) {
            float *cst = smem + buf * L::BUF, *xst = cst + L::CST, *wed = xst + L::XST;
            auto put = [&](float *dst, const f4u &v) { *reinterpret_cast<float4 *>(dst) = make_float4(v.v[0], v.v[1], v.v[2], v.v[3]); };
            float4 *cimg = reinterpret_cast<float4 *>(cst);
#pragma unroll
            for (int gg = 0; gg < 4; gg++) {
                const int col = 8 * (2 * gc0 + gg) + l8;
#pragma unroll
                for (int k = 0; k < 5; k++)
                    cimg[col * P8_CS + ((8 * k + part) ^ ((col >> 2) & 3))] =
                        make_float4(cpre[k][gg].v[0], cpre[k][gg].v[1], cpre[k][gg].v[2], cpre[k][gg].v[3]);
            }
            if (!xs) return;
#pragma unroll
            for (int g = 0; g < 4; g++) put(&xst[(16 * g + lcol) * EX_STR + 4 * lrq], xpre[g]);
            put(&xst[lane * EX_STR + 16], xpre[4]);
            if (lane < 5) put(&xst[64 * EX_STR + 4 * lane], xpre[5]);
            else if (lane < 10 && b == 0) put(&wed[4 * (lane - 5)], xpre[5]); // a later strip's west column: the mailbox
        };
        // chunk c is fetched two barriers before it is relaxed and stashed one barrier before (two register sets)
        // Per interval: look at the dependency poll issued one interval ago, issue the loads of the chunk two ahead, THEN stash
        // the chunk that was fetched one interval ago (its loads are older than the ones just issued and vmcnt counts in order,
        // so they have had a whole interval to land; stashing first and fetching second left them only the barrier and showed
        // their full latency in every interval -- the walk's pace), and issue the next poll.
        auto poll_issue = [&]() __attribute__((always_inline)) -> unsigned {
            return (t > 0 && my_ptr != nullptr) ? __hip_atomic_load(my_ptr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0xffffffffu;
        };
        auto poll_finish = [&](int c, unsigned seen) __attribute__((always_inline)) {
            const int need = lane == 1 ? c + 2 : c - 6;
            const unsigned un = (unsigned)(need < 0 ? 0 : (need < NC ? need : NC));
            if (__all(my_ptr == nullptr || seen >= un)) return;
            persist_wait3(my_ptr, un, ctl.abort_flag); // not there yet: the bounded spin
        };
        // West edge of chunk c: lane r polls the west strip's mailbox word of row 16c - 2 + r (r < 18) until its tag is set and puts
        // the value into the chunk's LDS edge slot.  Not the storer's job: behind its write-through stores a poll would wait for
        // their drain first, a round trip in every interval.
        const size_t mpitch = (size_t)NC * EX_CH;
        const unsigned long long *mail_w = ctl.mail + (((size_t)frame * T + t) * B + (b > 0 ? b - 1 : 0)) * mpitch;
        auto take = [&](int c) __attribute__((always_inline)) {
            if (b > 0 && c < NC) {
                const int row = row0(c) - 1 + lane;
                const bool want = lane < 18 && row >= 0 && row <= nrows - 1;
                float v = 0.0f;
                bool ok = !want;
                const unsigned long long t0 = __builtin_amdgcn_s_memrealtime(); // 100 MHz
                for (;;) {
                    if (!ok) {
                        const unsigned long long wd = __hip_atomic_load(mail_w + row + 127, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if ((unsigned)(wd >> 32) != 0u) {
                            v = __uint_as_float((unsigned)wd);
                            ok = true;
                        }
                    }
                    if (__all(ok)) break;
                    if (__hip_atomic_load(ctl.abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) break;
                    if (__builtin_amdgcn_s_memrealtime() - t0 > 50000000ull) { // 0.5 s: drain the grid, the host reports it
                        __hip_atomic_store(ctl.abort_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        break;
                    }
                    __builtin_amdgcn_s_sleep(1);
                }
                if (lane < 18) (smem + (c & 1) * L::BUF + L::CST + L::XST)[lane] = v;
            }
        };
        wait_deps(0);
        fetch(0, cA, xA);
        stash(cA, xA, 0);
        if (NC > 1) {
            wait_deps(1);
            fetch(1, cB, xB);
        }
        if (!xs) take(0);
        unsigned seen = poll_issue();
        lds_barrier(); // chunk 0 is in buffer 0
        P8S_DECL;
        for (int c = 0; c < NC; c += 2) {
            // ---- while chunk c (buffer 0) is relaxed ----
            P8S_BEGIN;
            if (c + 2 < NC) {
                poll_finish(c + 2, seen);
                fetch(c + 2, cA, xA);
            }
            if (c + 1 < NC) stash(cB, xB, 1);
            if (!xs) take(c + 1);
            seen = poll_issue();
            P8S_END;
            lds_barrier();
            if (c + 1 >= NC) break;
            // ---- while chunk c+1 (buffer 1) is relaxed ----
            P8S_BEGIN;
            if (c + 3 < NC) {
                poll_finish(c + 3, seen);
                fetch(c + 3, cB, xB);
            }
            if (c + 2 < NC) stash(cA, xA, 0);
            if (!xs) take(c + 2);
            seen = poll_issue();
            P8S_END;
            lds_barrier();
        }
        P8S_WRITE;
        return;
    }

    if (role == 2) {
        // ================================ storer wave =========================================
        auto store_out = [&](int c) __attribute__((always_inline)) {
            const float *outb = outb_base + (c & 1) * L::OUTB;
            const int r0 = row0(c);
            const bool all_valid = (r0 - 126 >= 1) && (r0 + EX_CH - 1 <= nrows - 2); // every row an inner row: whole vectors for the inner columns
            // does the chunk relax a pixel next to the image border?
            const bool ringed = (r0 - 126 <= 1 && r0 + EX_CH - 1 >= 1) || (r0 - 126 <= nrows - 2 && r0 + EX_CH - 1 >= nrows - 2) ||
                                jbase == 1 || jbase + 63 >= ncols - 2;
#pragma unroll
            for (int g = 0; g < 4; g++) {
                const int col = 16 * g + lcol;
                const int jj = jbase + col;
                const int row = r0 - P8_SKEW * col + 4 * lrq;
                const float4 v = *reinterpret_cast<const float4 *>(&outb[col * EX_STR + 4 * lrq]);
                if (all_valid) {
                    if (jj <= ncols - 2) {
                        v4u_t u;
                        u.x = __float_as_uint(v.x); u.y = __float_as_uint(v.y); u.z = __float_as_uint(v.z); u.w = __float_as_uint(v.w);
                        __builtin_amdgcn_raw_buffer_store_b128(u, rs_x, boff(jj, row), 0, 16);
                    }
                } else if (jj <= ncols - 2) {
                    const float vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                    for (int e = 0; e < 4; e++)
                        if (row + e >= 1 && row + e <= nrows - 2)
                            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(vv[e]), rs_x, boff(jj, row + e), 0, 16);
                }
                // border ring after this sweep: nearest-interior replicate (pdeSolvers.c:249-262).  Written here, not by the compute
                // wave: these stores and their drain would sit in every chunk of the first strip, which paces all the others.
                if (ringed && jj <= ncols - 2) {
                    const float vv[4] = {v.x, v.y, v.z, v.w};
                    const int top = 0, bot = ncols, left = 2 * ncols, right = 2 * ncols + nrows;
#pragma unroll
                    for (int e = 0; e < 4; e++) {
                        const int i = row + e;
                        if (i < 1 || i > nrows - 2) continue;
                        const unsigned uv = __float_as_uint(vv[e]);
                        auto put = [&](int idx) { __builtin_amdgcn_raw_buffer_store_b32(uv, rs_ring_w, (unsigned)idx * 4u, 0, 16); };
                        if (i == 1) {
                            put(top + jj);
                            if (jj == 1) { put(top); put(left); }
                            if (jj == ncols - 2) { put(top + ncols - 1); put(right); }
                        }
                        if (i == nrows - 2) {
                            put(bot + jj);
                            if (jj == 1) { put(bot); put(left + nrows - 1); }
                            if (jj == ncols - 2) { put(bot + ncols - 1); put(right + nrows - 1); }
                        }
                        if (jj == 1) put(left + i);
                        if (jj == ncols - 2) put(right + i);
                    }
                }
            }
        };
        // progress = c+1 once every store of chunk c has left (the compute wave drained its ring stores before the barrier)
        auto publish = [&](int c) __attribute__((always_inline)) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (lane == 0) __hip_atomic_store(prog_mine, (unsigned)(c + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        };
        lds_barrier(); // chunk 0 is in buffer 0
        // While chunk k is relaxed: publish chunk k-2 -- its stores have had a whole interval to drain, the wait costs nothing --
        // then write chunk k-1 back.  (Publishing k-1 right behind its own stores put a write-through round trip, 2-3 us at 4K,
        // into every interval of this wave, and that was the pace of the whole walk.  The counter only orders sweep t+1 behind
        // sweep t: one chunk later there is harmless; the strip to the east does not wait for it at all.)
        P8S_DECL;
        for (int k = 0; k < NC; k++) {
            P8S_BEGIN;
            if (k >= 2) publish(k - 2);
            if (k >= 1) store_out(k - 1);
            P8S_END;
            lds_barrier();
        }
        if (NC >= 2) publish(NC - 2);
        store_out(NC - 1);
        publish(NC - 1);
        P8S_WRITE;
        return;
    }

    // ================================== compute wave ===========================================
    const int j = jbase + lane;
    const bool col_ok = j <= ncols - 2;
    const float om1 = 1.0f - omega;
    float prev = 0.0f, w = 0.0f, nw = 0.0f;
    const bool has_east = (b + 1 < B); // my last column is the next strip's west column
    unsigned long long *const mail_mine = ctl.mail + (((size_t)frame * T + t) * B + b) * ((size_t)NC * EX_CH);
    const __amdgpu_buffer_rsrc_t rs_mail = __builtin_amdgcn_make_buffer_rsrc(mail_mine, 0, (unsigned)((size_t)NC * EX_CH * 8), 0x00020000);
    lds_barrier(); // chunk 0 is in buffer 0

    P8S_DECL;
    for (int k = 0; k < NC; k++) {
        P8S_BEGIN;
        const float *cst = smem + (k & 1) * L::BUF, *xst = cst + L::CST, *wed = xst + L::XST;
        float *outb = outb_base + (k & 1) * L::OUTB;
        const int i0 = row0(k) - P8_SKEW * lane; // my row at step 0 of this chunk
        auto relax_chunk = [&](auto interior_tag) __attribute__((always_inline)) {
            constexpr bool INTERIOR = decltype(interior_tag)::value;
#pragma unroll
            for (int mq = 0; mq < EX_CH / 4; mq++) {
                float4 res;
                float xo[8], xe[8], we[8];
                // granules 10 mq .. 10 mq + 9 of my run: two blocks of two rows, granule jq = coefficients 2 (jq % 5), + 1 of block jq / 5
                float cflat[40];
                {
                    const float4 *cimg = reinterpret_cast<const float4 *>(cst) + lane * P8_CS;
#pragma unroll
                    for (int jq = 0; jq < 10; jq++) {
                        const float4 v = cimg[(10 * mq + jq) ^ ((lane >> 2) & 3)];
                        cflat[4 * jq] = v.x; cflat[4 * jq + 1] = v.y; cflat[4 * jq + 2] = v.z; cflat[4 * jq + 3] = v.w;
                    }
                }
                {
                    const float4 a0 = *reinterpret_cast<const float4 *>(&xst[lane * EX_STR + 4 * mq]);
                    const float4 a1 = *reinterpret_cast<const float4 *>(&xst[lane * EX_STR + 4 * mq + 4]);
                    const float4 b0 = *reinterpret_cast<const float4 *>(&xst[(lane + 1) * EX_STR + 4 * mq]);
                    const float4 b1 = *reinterpret_cast<const float4 *>(&xst[(lane + 1) * EX_STR + 4 * mq + 4]);
                    const float4 c0 = *reinterpret_cast<const float4 *>(&wed[4 * mq]);
                    const float4 c1 = *reinterpret_cast<const float4 *>(&wed[4 * mq + 4]);
                    xo[0] = a0.x; xo[1] = a0.y; xo[2] = a0.z; xo[3] = a0.w; xo[4] = a1.x; xo[5] = a1.y; xo[6] = a1.z; xo[7] = a1.w;
                    xe[0] = b0.x; xe[1] = b0.y; xe[2] = b0.z; xe[3] = b0.w; xe[4] = b1.x; xe[5] = b1.y; xe[6] = b1.z; xe[7] = b1.w;
                    we[0] = c0.x; we[1] = c0.y; we[2] = c0.z; we[3] = c0.w; we[4] = c1.x; we[5] = c1.y; we[6] = c1.z; we[7] = c1.w;
                }
#pragma unroll
                for (int xq = 0; xq < 4; xq++) {
                    const int i = i0 + 4 * mq + xq;
                    const bool active = INTERIOR || (col_ok && (i >= 1) && (i <= nrows - 2));
                    // south-west (new): lane l-1 relaxed (i+1, j-1) one step ago; lane 0 reads the west edge column
                    const float sw = dpp_from_lower_lane(prev, we[xq + 2]);
                    float kk[NCF];
#pragma unroll
                    for (int f = 0; f < NCF; f++) kk[f] = cflat[(5 * (xq >> 1) + (f >> 1)) * 4 + 2 * (f & 1) + (xq & 1)];
                    // update(xc, xW, xE, xN, xS, xNW, xNE, xSW, xSE)
                    const float v = ModelPde8::update(xo[xq], w, xe[xq + 2], prev, xo[xq + 1], nw, xe[xq + 1], sw, xe[xq + 3], kk, omega, om1);
                    prev = active ? v : xo[xq]; // a cell that is not relaxed hands its value on: border rows feed the taps of rows 1 / nrows-2
                    if (xq == 0) res.x = prev; else if (xq == 1) res.y = prev; else if (xq == 2) res.z = prev; else res.w = prev;
                    nw = w;
                    w = sw;
                }
                *reinterpret_cast<float4 *>(&outb[lane * EX_STR + 4 * mq]) = res;
            }
        };
        {
            const int lo_row = row0(k) - 126, hi_row = row0(k) + EX_CH - 1;
            const bool interior = (lo_row >= 2) && (hi_row <= nrows - 3) && (jbase >= 2) && (jbase + 63 <= ncols - 3);
            if (interior) relax_chunk(std::true_type{});
            else relax_chunk(std::false_type{});
        }
        if (has_east && lane < 16) {
            // mailbox: lane 63's 16 values of this chunk (rows 16k - 127 .. 16k - 112 of my last column) as ONE 128-byte line of
            // self-validating words -- lanes 0..15 pick them up from the out buffer this wave just wrote (LDS operations of one wave
            // complete in order).  A full-line store needs no read of the line it replaces.
            typedef unsigned int v2u_t __attribute__((ext_vector_type(2)));
            v2u_t wv;
            wv.x = __float_as_uint(outb[63 * EX_STR + lane]);
            wv.y = 1u;
            __builtin_amdgcn_raw_buffer_store_b64(wv, rs_mail, (unsigned)((EX_CH * k + lane) * 8), 0, 16); // sc1
        }
        P8S_END;
        lds_barrier();
    }
    P8S_WRITE;
}

// One launch + the final border replicate.  Returns the number of launches, or -1 (message set).
inline int pde8_run_exact_persist(hipStream_t s, Pde8Planes P, const float *pack, float *side, PersistCtl ctl, int nrows, int ncols, int nframes, int iter,
                                  float omega)
{
    const size_t n = (size_t)nrows * ncols;
    const int B = (ncols - 2 + 63) / 64;
    const int NC = pde8_persist_chunks(nrows);
    constexpr size_t lds = Pde8Layout::LDS_BYTES + 16;
    if (ensure_lds(reinterpret_cast<const void *>(&k_pde8_exact_persist), lds) != PDEIP_OK) return -1;
    hipLaunchKernelGGL(k_pde8_exact_persist, dim3((unsigned)(B * iter * nframes)), dim3(P8P_THREADS), lds, s, P, pack, side, ctl, nrows, ncols, B, iter,
                       NC, nframes, omega, n);
    const int nb = 2 * ncols + 2 * (nrows - 2);
    hipLaunchKernelGGL(k_fill_borders, dim3((nb + 255) / 256, nframes, 1), dim3(256), 0, s, P.x, P.x, 1, nrows, ncols, n);
    return 2;
}

} // namespace pdeip
