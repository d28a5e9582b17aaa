// pdeip_sor_exact.hpp -- the reference's lexicographic Gauss-Seidel order on the GPU (gfx950).
//
// The reference sweeps j (MATLAB column) outer, i (row) inner, in place
// (opticalflowSolvers.c:76-79): pixel (i,j) of sweep t needs the NEW values of (i-1,j) and
// (i,j-1) and the OLD (sweep t-1) values of (i+1,j), (i,j+1) and of itself.  Those are the
// only dependencies, so every pixel on a hyperplane i + j + 2t = const is independent and
// several sweeps can be in flight at once.  Results are bit-identical to the serial loop.
//
// Tile wavefront.  The (row, column, sweep) space is cut into tiles:
//   * a strip b is 64 adjacent interior columns, one per lane (lane l <-> column 1+64b+l);
//   * a tile (a,b,t) is R=64 steps of strip b in sweep t; at step q lane l relaxes row
//     1 + a*R + q - l, so a wave is a diagonal front: the value lane l-1 produced one step
//     earlier is lane l's west neighbour (one DPP wave shift), the lane's own previous value
//     is its north neighbour, and its south/east neighbours are still sweep t-1 in memory.
//   * tile (a,b,t) depends only on tiles with a smaller m = a + 2b + 3t, and nothing it
//     overwrites is still needed by a tile with the same m (DESIGN.md "exact order").
//     One launch relaxes every tile of one m; launches are stream-ordered, so there is no
//     inter-workgroup synchronisation inside a launch and nothing that can spin.
//
// Memory.  Everything a tile reads was final before its launch, so it is all prefetchable.  A
// tile runs as four chunks of 16 steps.  The parallelogram of each plane that a chunk touches
// (64 columns x 16 rows, row offset = -lane) is fetched with coalesced 16-byte loads (lane ->
// 4 consecutive rows of one column, 16 columns per instruction) into registers while the
// previous chunk is being relaxed, then transposed through LDS ([plane][column][20-float row],
// conflict-free ds_read_b128 of 4 steps at a time).  Results go back the same way: LDS, then
// coalesced 16-byte stores.  The naive form (each lane streaming its own column) issues 64
// cache lines per load instruction and ran 15x slower.
//
// Borders.  The reference replicates the border after every sweep (:161-179).  For a 5-point
// stencil an interior pixel only ever reads the border cell next to itself, whose value
// after sweep t-1 is that pixel's own sweep t-1 value; in sweep 0 it is the caller's border
// cell.  So border cells are read from memory in sweep 0, substituted by the pixel's own
// old value in later sweeps, never written during the sweeps, and filled once at the end.
#pragma once
#include <type_traits>

#include "pdeip_models.hpp"

namespace pdeip {

constexpr int EX_R = 64;   // steps per tile; the dependency analysis needs EX_R >= 63
constexpr int EX_CH = 16;  // steps per chunk
constexpr int EX_STR = 20; // LDS floats per (plane, column) row: 16 + 4 pad -> conflict-free b128 reads

// 16-byte global access at 4-byte alignment (rows of a column start anywhere).  A vector typedef with
// reduced alignment keeps the access one global_load/store_dwordx4; a packed struct gets split into
// four dword accesses by SROA.
typedef float v4f_a4 __attribute__((ext_vector_type(4), aligned(4)));
struct f4u {
    float v[4];
    __device__ __forceinline__ void load(const float *p)
    {
        const v4f_a4 t = *reinterpret_cast<const v4f_a4 *>(p);
        v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
    }
    __device__ __forceinline__ void store(float *p) const
    {
        v4f_a4 t;
        t.x = v[0]; t.y = v[1]; t.z = v[2]; t.w = v[3];
        *reinterpret_cast<v4f_a4 *>(p) = t;
    }
};

__device__ __forceinline__ float dpp_from_lower_lane(float v, float lane0_value)
{ // lane l <- lane l-1 (v_mov_b32_dpp wave_shr:1); lane 0 keeps lane0_value
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(lane0_value), __float_as_int(v), 0x138, 0xf, 0xf, false));
}
__device__ __forceinline__ float dpp_from_upper_lane(float v, float lane63_value)
{ // lane l <- lane l+1 (wave_shl:1); lane 63 keeps lane63_value
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(lane63_value), __float_as_int(v), 0x130, 0xf, 0xf, false));
}

#ifdef PDEIP_EXACT_STAMPS // diagnostic build only (tools/): phase stamps of one tile, never in the product
__device__ unsigned long long g_exact_stamps[64];
#define EX_STAMP(n)                                                                                   \
    do {                                                                                              \
        if (stamp_tile) {                                                                             \
            unsigned long long t_;                                                                    \
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory"); \
            if (lane == 0) g_exact_stamps[n] = t_;                                                    \
        }                                                                                             \
    } while (0)
#else
#define EX_STAMP(n)
#endif

template <class Mdl> struct ExactLayout {
    static constexpr int NIT = Mdl::NIT, NRO = Mdl::NRO, NF = NIT + NRO, NCF = Mdl::NCF, NP = NF + NCF;
    static constexpr int STAGE = NP * 64 * EX_STR;  // chunk of every plane
    static constexpr int EDGE = 2 * NF * EX_CH;     // west column of lane 0, east column of lane 63
    static constexpr int BUF = STAGE + EDGE;        // one chunk buffer; two of them (double buffering)
    static constexpr int OUTB = NIT * 64 * EX_STR;  // relaxed values of one chunk; two of them as well
    static constexpr size_t LDS_BYTES = (size_t)(2 * BUF + 2 * OUTB) * sizeof(float);
};

// Workgroup barrier that does NOT drain outstanding global loads (a __syncthreads() would add
// s_waitcnt vmcnt(0) and stall the loader's prefetch): LDS traffic only.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// One workgroup = one tile = two waves.  Wave 1 (the mover) streams the tile's chunks global ->
// registers -> LDS (double-buffered, loads of chunk k+2 in flight while chunk k is relaxed) and writes
// the relaxed chunk k-1 back LDS -> global; wave 0 only relaxes.
template <class Mdl>
__global__ void __launch_bounds__(128)
k_sor_exact(SweepPlanes<Mdl> P, int nrows, int ncols, int A, int B, int T, int m, float omega,
            size_t frame_stride)
{
    using L = ExactLayout<Mdl>;
    constexpr int NIT = L::NIT, NRO = L::NRO, NRO1 = at_least_one<NRO>::value, NF = L::NF, NCF = L::NCF, NP = L::NP;
    constexpr int NCHUNK = EX_R / EX_CH;
    static_assert(NCHUNK == 4, "the loader/compute schedule below is written for four chunks");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *outb_base = smem + 2 * L::BUF;

    const int lane = threadIdx.x & 63;
    const bool loader = (threadIdx.x >> 6) == 1; // wave-uniform
    const int b = blockIdx.x % B, t = blockIdx.x / B;
    const int a = m - 2 * b - 3 * t;
    if (a < 0 || a >= A) return; // both waves leave together
    const size_t fo = (size_t)blockIdx.y * frame_stride;
#ifdef PDEIP_EXACT_STAMPS
    const bool stamp_tile = (a == 10 && b == 10 && t == 0) && !loader;
#endif
    EX_STAMP(0);

    // planes in staging order: iterate fields, read-only neighbour fields, coefficients
    const float *pl[NP];
    float *it[NIT];
#pragma unroll
    for (int f = 0; f < NIT; f++) {
        it[f] = P.it_out[f] + fo;
        pl[f] = it[f];
    }
#pragma unroll
    for (int f = 0; f < NRO; f++) pl[NIT + f] = P.ro[f] + fo;
#pragma unroll
    for (int f = 0; f < NCF; f++) pl[NF + f] = P.cf[f] + fo;

    const int jbase = 1 + 64 * b;                 // column of lane 0
    const int i00 = 1 + a * EX_R;                 // row of lane 0 at step 0
    auto crow = [&](int i) { return i < 0 ? 0 : (i > nrows - 1 ? nrows - 1 : i); };
    // loader/storer geometry: instruction g covers columns 16g..16g+15 x 16 rows; this lane takes 4
    // consecutive rows (quad lrq) of column 16g+lcol: four consecutive lanes cover one column's 64
    // contiguous bytes (fewer cache-line accesses per instruction than lane -> column; measured).
    const int lcol = lane >> 2, lrq = lane & 3;

    if (loader) {
        // ================================ loader wave ==========================================
        f4u preA[NP][4], preB[(NP <= 11) ? NP : 1][4], epreA[NF], epreB[NF];
        auto fetch_rows = [&](int k, f4u (&pre)[NP][4], f4u (&epre)[NF], auto inside_tag) __attribute__((always_inline)) {
            constexpr bool INSIDE = decltype(inside_tag)::value;
#pragma unroll
            for (int p = 0; p < NP; p++) {
#pragma unroll
                for (int g = 0; g < 4; g++) {
                    const int col = 16 * g + lcol;
                    int jj = jbase + col;
                    jj = jj < ncols - 1 ? jj : ncols - 1;
                    // centre rows for coefficients, south rows (one further) for the neighbour fields
                    const int row = i00 - col + EX_CH * k + (p < NF ? 1 : 0) + 4 * lrq;
                    const float *src = pl[p] + (size_t)jj * nrows;
                    if (INSIDE) {
                        pre[p][g].load(src + row);
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; e++) pre[p][g].v[e] = src[crow(row + e)];
                    }
                }
            }
            // edge columns: lanes 0-3 fetch the west column of lane 0 (centre rows of lane 0), lanes 4-7 the
            // east column of lane 63 (centre rows of lane 63); the other lanes repeat them (same addresses).
            // Always element-wise with clamped rows: column 0 / ncols-1 sit at the ends of the buffer.
            const int which = (lane >> 2) & 1;
            const int ecol = which ? (jbase + 64 < ncols - 1 ? jbase + 64 : ncols - 1) : jbase - 1;
            const int erow = (which ? i00 - 63 : i00) + EX_CH * k + 4 * (lane & 3);
#pragma unroll
            for (int f = 0; f < NF; f++) {
                const float *src = pl[f] + (size_t)ecol * nrows;
#pragma unroll
                for (int e = 0; e < 4; e++) epre[f].v[e] = src[crow(erow + e)];
            }
        };
        // The planes are column-major and contiguous and every staged column is >= 1, so a row index
        // below 0 or above nrows-1 lands in a neighbouring column: a valid address whose value is never
        // used (those rows belong to steps that relax nothing) -- as long as the access stays inside the
        // plane.  The flat index grows with the column, so the last staged column of the last chunk
        // bounds it.  Decided once per tile, so each path has a fixed number of loads in flight and the
        // compiler's wait counts stay exact (a per-chunk branch made it drain everything).
        const int jmax = jbase + 63 < ncols - 1 ? jbase + 63 : ncols - 1;
        const bool tile_inside = (long)jmax * nrows + (i00 + EX_R + 4) < (long)nrows * ncols;
        auto stash = [&](const f4u (&pre)[NP][4], const f4u (&epre)[NF], int buf) __attribute__((always_inline)) {
            float *stage = smem + buf * L::BUF, *edge = stage + L::STAGE;
#pragma unroll
            for (int p = 0; p < NP; p++)
#pragma unroll
                for (int g = 0; g < 4; g++) {
                    const int col = 16 * g + lcol;
                    *reinterpret_cast<float4 *>(&stage[(p * 64 + col) * EX_STR + 4 * lrq]) =
                        make_float4(pre[p][g].v[0], pre[p][g].v[1], pre[p][g].v[2], pre[p][g].v[3]);
                }
            if (lane < 8) {
                const int which = (lane >> 2) & 1;
#pragma unroll
                for (int f = 0; f < NF; f++)
                    *reinterpret_cast<float4 *>(&edge[(which * NF + f) * EX_CH + 4 * (lane & 3)]) =
                        make_float4(epre[f].v[0], epre[f].v[1], epre[f].v[2], epre[f].v[3]);
            }
        };
        // relaxed chunk k: LDS -> global, coalesced; only interior pixels are written
        auto store_out = [&](int k) __attribute__((always_inline)) {
            const float *outb = outb_base + (k & 1) * L::OUTB;
            const int lo_row = i00 - 63 + EX_CH * k, hi_row = i00 + EX_CH * k + EX_CH - 1;
            const bool all_valid = (lo_row >= 1) && (hi_row <= nrows - 2) && (jbase + 63 <= ncols - 2);
#pragma unroll
            for (int f = 0; f < NIT; f++)
#pragma unroll
                for (int g = 0; g < 4; g++) {
                    const int col = 16 * g + lcol;
                    const int jj = jbase + col;
                    const int row = i00 - col + EX_CH * k + 4 * lrq;
                    const float4 v = *reinterpret_cast<const float4 *>(&outb[(f * 64 + col) * EX_STR + 4 * lrq]);
                    float *dst = it[f] + (size_t)(jj < ncols - 1 ? jj : ncols - 1) * nrows;
                    if (all_valid) {
                        f4u o;
                        o.v[0] = v.x; o.v[1] = v.y; o.v[2] = v.z; o.v[3] = v.w;
                        o.store(dst + row);
                    } else if (jj <= ncols - 2) {
                        if (row >= 1 && row <= nrows - 2) dst[row] = v.x;
                        if (row + 1 >= 1 && row + 1 <= nrows - 2) dst[row + 1] = v.y;
                        if (row + 2 >= 1 && row + 2 <= nrows - 2) dst[row + 2] = v.z;
                        if (row + 3 >= 1 && row + 3 <= nrows - 2) dst[row + 3] = v.w;
                    }
                }
        };
        // schedule: chunk c is loaded two barriers before it is relaxed, stashed one barrier before, and
        // written back one barrier after
        // Two register sets (chunk k+2 in flight while chunk k+1 is stashed) when they fit the 512-register
        // file; the 13-plane late-linearization model runs with one set (one chunk of look-ahead).
        constexpr bool TWO_SETS = (NP <= 11);
        auto run = [&](auto inside_tag) __attribute__((always_inline)) {
            if constexpr (TWO_SETS) {
                fetch_rows(0, preA, epreA, inside_tag);
                stash(preA, epreA, 0);
                fetch_rows(1, preB, epreB, inside_tag);
                lds_barrier(); // #0: buffer 0 holds chunk 0
                fetch_rows(2, preA, epreA, inside_tag);
                stash(preB, epreB, 1);
                lds_barrier(); // #1: buffer 1 holds chunk 1; chunk 0 has been relaxed
                fetch_rows(3, preB, epreB, inside_tag);
                stash(preA, epreA, 0);
                store_out(0);
                lds_barrier(); // #2
                stash(preB, epreB, 1);
                store_out(1);
                lds_barrier(); // #3
                store_out(2);
                lds_barrier(); // #4
                store_out(3);
            } else {
                fetch_rows(0, preA, epreA, inside_tag);
                stash(preA, epreA, 0);
                fetch_rows(1, preA, epreA, inside_tag);
                lds_barrier(); // #0
                stash(preA, epreA, 1);
                fetch_rows(2, preA, epreA, inside_tag);
                lds_barrier(); // #1
                store_out(0);
                stash(preA, epreA, 0);
                fetch_rows(3, preA, epreA, inside_tag);
                lds_barrier(); // #2
                store_out(1);
                stash(preA, epreA, 1);
                lds_barrier(); // #3
                store_out(2);
                lds_barrier(); // #4
                store_out(3);
            }
        };
        if (tile_inside) run(std::true_type{});
        else run(std::false_type{});
        return;
    }

    // ================================== compute wave ===========================================
    const int j = jbase + lane;                   // this lane's column
    const bool col_ok = j <= ncols - 2;           // interior column: relaxed
    const int jc = j < ncols - 1 ? j : ncols - 1; // clamped (valid address, value unused)
    const size_t cb = (size_t)jc * nrows;
    const float om1 = 1.0f - omega;
    const bool first_sweep = (t == 0);
    const int i0 = i00 - lane;                    // row of this lane at step 0

    // ---- per-lane state at step 0 (scattered loads, once per tile) --------------------------------
    float prev[NIT], cen[NIT], north0[NIT], west0[NIT], topb[NIT], rcen[NRO1], rnorth[NRO1];
#pragma unroll
    for (int f = 0; f < NIT; f++) {
        cen[f] = it[f][cb + crow(i0)];
        north0[f] = it[f][cb + crow(i0 - 1)];            // relaxed by tile a-1 (an earlier launch)
        west0[f] = it[f][cb - nrows + crow(i0)];         // column j-1 >= 0; relaxed by an earlier launch
        topb[f] = it[f][cb];                             // border row 0 (used in sweep 0 only)
        prev[f] = 0.0f;
    }
#pragma unroll
    for (int f = 0; f < NRO1; f++) {
        rcen[f] = (NRO > 0) ? pl[(NRO > 0 ? NIT + f : 0)][cb + crow(i0)] : 0.0f;
        rnorth[f] = (NRO > 0) ? pl[(NRO > 0 ? NIT + f : 0)][cb + crow(i0 - 1)] : 0.0f;
    }
    EX_STAMP(1);
    lds_barrier(); // #0
    EX_STAMP(2);

    for (int k = 0; k < NCHUNK; k++) {
        const float *stage = smem + (k & 1) * L::BUF, *edge = stage + L::STAGE;
        float *outb = outb_base + (k & 1) * L::OUTB;
        EX_STAMP(3 + 4 * k);
        // INTERIOR chunk: every lane relaxes an interior pixel at every step and none of them touches the
        // image border, so the activity masks and border substitutions below fold away (most chunks).
        auto relax_chunk = [&](auto interior_tag) __attribute__((always_inline)) {
        constexpr bool INTERIOR = decltype(interior_tag)::value;
#pragma unroll
        for (int mq = 0; mq < EX_CH / 4; mq++) {
            float4 ck[NCF], s4[NF], e4[NF], res[NIT];
#pragma unroll
            for (int f = 0; f < NCF; f++) ck[f] = *reinterpret_cast<const float4 *>(&stage[((NF + f) * 64 + lane) * EX_STR + 4 * mq]);
#pragma unroll
            for (int f = 0; f < NF; f++) {
                s4[f] = *reinterpret_cast<const float4 *>(&stage[(f * 64 + lane) * EX_STR + 4 * mq]);
                e4[f] = *reinterpret_cast<const float4 *>(&edge[((lane == 63 ? 1 : 0) * NF + f) * EX_CH + 4 * mq]);
            }
#pragma unroll
            for (int x = 0; x < 4; x++) {
                const int q = EX_CH * k + 4 * mq + x;
                const int i = i0 + q;
                const bool row_ok = INTERIOR || ((i >= 1) && (i <= nrows - 2));
                const bool active = INTERIOR || (col_ok && row_ok);
                auto el = [&](const float4 &v) { return x == 0 ? v.x : (x == 1 ? v.y : (x == 2 ? v.z : v.w)); };

                float c[NIT], w[NIT], e[NIT], n[NIT], s[NIT], kk[NCF];
                float rsouth[NRO1], reast[NRO1], rwest[NRO1];
#pragma unroll
                for (int f = 0; f < NIT; f++) {
                    const float sraw = el(s4[f]);
                    // east: lane l+1 sits one row higher, its south value is (i, j+1); lane 63 reads the edge column
                    const float eraw = dpp_from_upper_lane(sraw, el(e4[f]));
                    // west: lane l-1 relaxed (i, j-1) one step ago; lane 0 reads the edge column (earlier launch)
                    float wnew = dpp_from_lower_lane(prev[f], el(e4[f]));
                    if (q == 0 && lane != 0) wnew = west0[f];
                    float nv = (q == 0) ? north0[f] : prev[f];
                    if (!INTERIOR && i == 1) nv = first_sweep ? topb[f] : cen[f];
                    c[f] = cen[f];
                    n[f] = nv;
                    s[f] = (!INTERIOR && i + 1 == nrows - 1 && !first_sweep) ? cen[f] : sraw;
                    e[f] = (!INTERIOR && j + 1 == ncols - 1 && !first_sweep) ? cen[f] : eraw;
                    w[f] = (!INTERIOR && j - 1 == 0 && !first_sweep) ? cen[f] : wnew;
                    cen[f] = sraw; // next step's centre
                }
#pragma unroll
                for (int f = 0; f < NRO1; f++) {
                    if (NRO > 0) {
                        rsouth[f] = el(s4[(NRO > 0 ? NIT + f : 0)]);
                        reast[f] = dpp_from_upper_lane(rsouth[f], el(e4[(NRO > 0 ? NIT + f : 0)]));
                        rwest[f] = dpp_from_lower_lane(rnorth[f], el(e4[(NRO > 0 ? NIT + f : 0)])); // lane l-1's north is (i, j-1)
                    } else {
                        rsouth[f] = reast[f] = rwest[f] = 0.0f;
                    }
                }
#pragma unroll
                for (int f = 0; f < NCF; f++) kk[f] = el(ck[f]);
                Mdl::update(c, w, e, n, s, rcen, rwest, reast, rnorth, rsouth, kk, omega, om1);
#pragma unroll
                for (int f = 0; f < NIT; f++) {
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
            for (int f = 0; f < NIT; f++) *reinterpret_cast<float4 *>(&outb[(f * 64 + lane) * EX_STR + 4 * mq]) = res[f];
        }
        };
        {
            const int lo_row = i00 - 63 + EX_CH * k, hi_row = i00 + EX_CH * k + EX_CH - 1;
            const bool interior = (lo_row >= 2) && (hi_row <= nrows - 3) && (jbase >= 2) && (jbase + 63 <= ncols - 3);
            if (interior) relax_chunk(std::true_type{});
            else relax_chunk(std::false_type{});
        }
        EX_STAMP(4 + 4 * k);
        EX_STAMP(5 + 4 * k);
        lds_barrier(); // #k+1: this buffer may be refilled, the next one is ready
        EX_STAMP(6 + 4 * k);
    }
}

// =================================================================================================
// Persistent form: one launch per call.  A workgroup owns one (strip b, sweep t) and walks down the
// strip chunk by chunk (same mover/compute pair and LDS layout as k_sor_exact, no tile boundaries, no
// cold start); the launch-per-front ordering is replaced by progress counters:
//   chunk c of (b,t) needs   the west strip's east column of sweep t at the 16 rows lane 0 relaxes in chunk c
//                            progress[b][t-1]   >= c+2   (own columns' south rows of sweep t-1)
//                            progress[b+1][t-1] >= c-2   (east column of sweep t-1)
// These conditions also cover the write-after-read hazards.
// West edge.  The one value a strip needs from its west neighbour of the SAME sweep is that neighbour's last column.  Routing
// it through the iterate plane (store, drain, publish a counter, poll, load: four round trips, and the whole chunk fetch
// gated on it two chunks ahead) made every strip trail its neighbour by ~12 chunk times where the dependency needs 5.  It
// travels by a mailbox instead: the compute wave's lane 63 stores every result as one 8-byte {value, tag} word (write-
// through, no drain: the tag makes the word self-validating), and a wave of the east strip polls the 16 words of the
// next chunk and puts them into the LDS edge slot -- one round trip.  (That wave is the storer: its stores of the previous chunk
// are issued first, the polls queue behind them, and the first poll to return also proves the stores drained -- the publish needs
// no wait of its own.  A fifth wave would halve every wave's register budget; the loader holds a whole chunk in registers.)  The chunk fetch
// itself (old values and coefficients) no longer waits for the west strip at all.
// Hand-off (cdna_hip_programming.md Guideline 16, recipe R1): the iterate is stored write-through
// (buffer_store ... sc1) by the mover wave only, which drains (s_waitcnt vmcnt(0)) and then publishes
// the counter with one relaxed agent-scope atomic store; a consumer's mover polls relaxed and reads
// the iterate with sc1 buffer loads only.  Buffer addressing also gives free bounds clamping, so
// there is no element-wise slow path here.
// Liveness: workgroups take a ticket at start and tickets are mapped to (b,t) in an order in which
// every dependency has a smaller ticket, so whatever the dispatch order a running workgroup only ever
// waits for workgroups that are running or finished.  Every spin is bounded; a timeout raises the
// abort word, after which all waits fall through and the grid drains (the host reports the error).
#ifdef PDEIP_P8_STAMPS // diagnostic build only (tools/p8_stamps.py, tools/walk_stamps.py): how long each wave of a walker works per interval; never in the product
static __device__ unsigned long long g_p8_stamps[4096]; // one copy per translation unit (5-point / 9-point), each with its reader
#define P8S_DECL unsigned long long s_busy_ = 0, s_i0_ = 0; const unsigned long long s_t0_ = __builtin_amdgcn_s_memtime(), s_r0_ = __builtin_amdgcn_s_memrealtime()
#define P8S_BEGIN s_i0_ = __builtin_amdgcn_s_memtime()
#define P8S_END s_busy_ += __builtin_amdgcn_s_memtime() - s_i0_
#define P8S_WRITE                                                                                                                  \
    if (lane == 0 && tk < 256) {                                                                                                  \
        g_p8_stamps[(tk * 4 + role) * 4 + 0] = s_busy_;                                                                            \
        g_p8_stamps[(tk * 4 + role) * 4 + 1] = (role == 1) ? s_r0_ : __builtin_amdgcn_s_memtime() - s_t0_; /* loader: when the walk began, 100 MHz */ \
        g_p8_stamps[(tk * 4 + role) * 4 + 2] = __builtin_amdgcn_s_memrealtime() - s_r0_;                                          \
        g_p8_stamps[(tk * 4 + role) * 4 + 3] = (unsigned long long)(b | (t << 16));                                               \
    }
#else
#define P8S_DECL
#define P8S_BEGIN
#define P8S_END
#define P8S_WRITE
#endif

// Packed coefficients (persistent form).  The walk's pace is its loader, and the loader's cost is the cache lines its loads
// touch: NCF coefficient planes give a (column, chunk) NCF 64-byte pieces in NCF places.  The pre-pass that
// derives the divisor planes (k_derive's job) therefore writes all NCF coefficients of a pixel side by side,
//   pack[col][row][NCF],
// so the 16 rows of a column's chunk are NCF*64 contiguous bytes (4-byte aligned: the lanes are skewed by one row).  The chunk's
// LDS image is that run as it lies in memory, one run per column: 16-byte granule q of column c at granule
// c * CS + (q ^ ((c >> 2) & 3)), CS = 4 NCF rounded up to an odd multiple of 4 -- the loader's four lanes of a column write 64
// contiguous bytes, the compute lanes (one column each) read granule q of 16 columns from 16 different bank groups.
template <class Mdl> struct PackLayout {
    static constexpr int NCF = Mdl::NCF;
    static constexpr int RUN = 4 * NCF;                            // granules per column and chunk
    static constexpr int CS = (RUN % 8 == 4) ? RUN : RUN + 4;      // column stride in granules: 4 x odd
    static_assert(64 * CS * 4 <= NCF * 64 * EX_STR, "the packed image fits the coefficient planes' LDS region");
};
template <class Mdl>
__global__ void k_pack_coefficients(SweepPlanes<Mdl> P, float *pack, int nrows, int ncols, size_t frame_stride)
{
    // last column first: the walkers start at column 1 right after this kernel, and what was written last is what the
    // Infinity Cache still holds when the packed planes (299 MB at 4K) are larger than it (measured: +4 % for the 9-point walk
    // at 4K, nothing either way for the 5-point ones)
    const int i = blockIdx.x * blockDim.x + threadIdx.x, j = ncols - 1 - (int)blockIdx.y;
    if (i >= nrows) return;
    const size_t pos = (size_t)j * nrows + i;
    const size_t q = pos + (size_t)blockIdx.z * frame_stride;
    float k[Mdl::NCF];
#pragma unroll
    for (int f = 0; f < Mdl::NCF; f++) k[f] = __builtin_nontemporal_load(P.cf[f] + q); // read once: leave the Infinity Cache to the packed copy
    Mdl::derive(k);
    float *dst = pack + q * Mdl::NCF;
#pragma unroll
    for (int f = 0; f < Mdl::NCF; f++) dst[f] = k[f];
}

struct PersistCtl {
    unsigned *ticket;    // [8] one per XCD list
    unsigned *abort_flag; // [1]
    unsigned *progress;  // [nframes][T][B] chunks completed and visible
    const int *order;    // schedule table: [0..8] list offsets, [16..] items b | (t << 16) (pdeip_persist_host.hpp)
    unsigned long long *mail; // [nframes][T][B][NIT][nrows] east-column results of a strip, {value, tag}: see "West edge" below
};

// One item per workgroup (thread 0): the next one of the list of the XCD this workgroup runs on, or of the next list that still has
// one (pdeip_persist_host.hpp).  Returns the item's index in the table and the frame, or false (cannot happen: one item per workgroup).
__device__ __forceinline__ bool persist_take_item(const PersistCtl &ctl, int nframes, unsigned *item, unsigned *frame)
{
    const unsigned xcc = (unsigned)__builtin_amdgcn_s_getreg((3 << 11) | 20) & 7u; // HW_REG_XCC_ID, bits 3:0
    for (unsigned a = 0; a < 8; a++) {
        const unsigned x = (xcc + a) & 7u;
        const unsigned first = (unsigned)ctl.order[x], len = ((unsigned)ctl.order[x + 1] - first) * (unsigned)nframes;
        if (len == 0) continue;
        const unsigned idx = atomicAdd(ctl.ticket + x, 1u);
        if (idx < len) {
            *item = first + idx / (unsigned)nframes;
            *frame = idx % (unsigned)nframes;
            return true;
        }
    }
    return false;
}

// Wave-wide wait: lane k polls counter ptrs[k] (k < 3) until it reaches need[k]; the three polls are ONE
// load instruction per round (one write-through latency instead of three).  Bounded: a timeout raises
// the abort word and every later wait falls through.
__device__ __forceinline__ void persist_wait3(const unsigned *my_ptr, unsigned my_need, unsigned *abort_flag)
{
    const bool polls = (my_ptr != nullptr) && (my_need != 0);
    bool ok = !polls || (__hip_atomic_load(my_ptr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= my_need);
    if (__all(ok)) return;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime(); // 100 MHz
    for (;;) {
        __builtin_amdgcn_s_sleep(4);
        if (!ok) ok = __hip_atomic_load(my_ptr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= my_need;
        if (__all(ok)) return;
        if (__hip_atomic_load(abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return;
        if (__builtin_amdgcn_s_memrealtime() - t0 > 50000000ull) { // 0.5 s: something is wrong; drain the grid
            __hip_atomic_store(abort_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return;
        }
    }
}

typedef unsigned int v4u_t __attribute__((ext_vector_type(4)));

// compute wave(s), loader wave, storer wave; the coupled two-field models relax with two compute waves, one per field
template <class Mdl> constexpr int exp_threads() { return Mdl::NIT == 2 ? 256 : 192; }

template <class Mdl>
__global__ void __launch_bounds__((exp_threads<Mdl>()))
k_sor_exact_persist(SweepPlanes<Mdl> P, const float *pack, PersistCtl ctl, int nrows, int ncols, int B, int T, int NC, int nframes,
                    float omega, size_t frame_stride)
{
    using L = ExactLayout<Mdl>;
    constexpr int NIT = L::NIT, NRO = L::NRO, NRO1 = at_least_one<NRO>::value, NF = L::NF, NCF = L::NCF, NP = L::NP;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *outb_base = smem + 2 * L::BUF;
    // the ticket word lives at the END of the dynamic region: a static __shared__ would sit in front of it
    // and knock the 16-byte alignment of every ds_read_b128 (cdna_hip_programming.md Guideline 17)
    unsigned *s_ticket = reinterpret_cast<unsigned *>(smem + 2 * L::BUF + 2 * L::OUTB);

    const int lane = threadIdx.x & 63;
    // Roles.  The walker's pace is set by whichever wave is slowest per chunk.  A wave that both polls the neighbours'
    // counters (a write-through round trip of ~1 us) and drains its own write-through stores before publishing (another one)
    // is slower than the wave that relaxes, so the two round trips belong to two waves with their own vmcnt queues: the
    // LOADER polls, fetches and stashes; the STORER writes the relaxed chunk back and publishes the counter.  The coupled
    // models update u from the OLD v of the same pixel and vice versa (opticalflowSolvers.c:129-149), so their two fields are
    // relaxed by two COMPUTE waves that never exchange anything: half the instructions per step each.
    const int role = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6); // 0 compute (field 0 / all), 1 loader, 2 storer + west edge, 3 compute (field 1)
    const bool mover = role == 1;
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

    // every plane through a buffer descriptor (range-checked: an out-of-plane row reads 0 / writes nothing);
    // iterate fields with aux = 16 (sc1), coefficients and read-only fields with the default policy
    __amdgpu_buffer_rsrc_t rs[NP];
    const unsigned plane_bytes = (unsigned)((size_t)nrows * ncols * sizeof(float));
#pragma unroll
    for (int f = 0; f < NIT; f++) rs[f] = __builtin_amdgcn_make_buffer_rsrc(P.it_out[f] + fo, 0, plane_bytes, 0x00020000);
#pragma unroll
    for (int f = 0; f < NRO; f++) rs[NIT + f] = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(P.ro[f]) + fo, 0, plane_bytes, 0x00020000);
    // the coefficients: packed per pixel (see PackLayout); rs[NF..] stay unused here
    using PL = PackLayout<Mdl>;
    const __amdgpu_buffer_rsrc_t rs_pack =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(pack) + fo * NCF, 0, (unsigned)((size_t)nrows * ncols * NCF * sizeof(float)), 0x00020000);

    const int jbase = 1 + 64 * b;
    auto crow = [&](int i) { return i < 0 ? 0 : (i > nrows - 1 ? nrows - 1 : i); };
    auto boff = [&](int jj, int row) { return (unsigned)(((long)jj * nrows + row) * 4); }; // byte offset; >= 0 for jj >= 1
    const int lcol = lane >> 2, lrq = lane & 3;
    auto as_f4u = [](v4u_t v, f4u &o) {
        o.v[0] = __uint_as_float(v.x); o.v[1] = __uint_as_float(v.y); o.v[2] = __uint_as_float(v.z); o.v[3] = __uint_as_float(v.w);
    };

    if (mover) {
        // ================================ mover wave ==========================================
        f4u preA[NP][4], preB[(NP <= 11) ? NP : 1][4], epreA[NF], epreB[NF];
        constexpr bool TWO_SETS = (NP <= 11);
        // lane 1 watches this strip's previous sweep, lane 2 the east strip's previous sweep
        const unsigned *my_ptr = lane == 1 ? prog_prev : (lane == 2 ? prog_east : nullptr); // the west strip: role 4
        auto wait_deps = [&](int c) __attribute__((always_inline)) {
            const int need = lane == 0 ? c + 5 : (lane == 1 ? c + 2 : c - 2);
            persist_wait3(my_ptr, (unsigned)(need < 0 ? 0 : (need < NC ? need : NC)), ctl.abort_flag);
        };
        const bool west_by_mail = (b > 0);
        auto fetch = [&](int c, f4u (&pre)[NP][4], f4u (&epre)[NF]) __attribute__((always_inline)) {
            const int i00 = 1 + EX_CH * c;
#pragma unroll
            for (int p = 0; p < NF; p++) {
#pragma unroll
                for (int g = 0; g < 4; g++) {
                    const int col = 16 * g + lcol;
                    int jj = jbase + col;
                    jj = jj < ncols - 1 ? jj : ncols - 1;
                    const int row = i00 - col + 1 + 4 * lrq;
                    const v4u_t v = (p < NIT) ? __builtin_amdgcn_raw_buffer_load_b128(rs[p], boff(jj, row), 0, 16)
                                              : __builtin_amdgcn_raw_buffer_load_b128(rs[p], boff(jj, row), 0, 0);
                    as_f4u(v, pre[p][g]);
                }
            }
            // coefficients: the column's run of 16 rows x NCF floats, the four lanes of a column 64 contiguous bytes per instruction
#pragma unroll
            for (int g = 0; g < 4; g++) {
                const int col = 16 * g + lcol;
                int jj = jbase + col;
                jj = jj < ncols - 1 ? jj : ncols - 1;
                const unsigned run = (unsigned)((((long)jj * nrows + (i00 - col)) * NCF) * 4) + 16u * (unsigned)lrq;
#pragma unroll
                for (int k = 0; k < NCF; k++) as_f4u(__builtin_amdgcn_raw_buffer_load_b128(rs_pack, run + 64u * k, 0, 0), pre[NF + k][g]);
            }
            // edge columns: east (old values) of every field; west of the read-only fields, and of the iterate fields only for
            // the first strip (the frame's border column) -- a later strip's west iterate column is the mailbox's (role 4)
            const int which = (lane >> 2) & 1;
            const int ecol = which ? (jbase + 64 < ncols - 1 ? jbase + 64 : ncols - 1) : jbase - 1;
            const int erow = (which ? i00 - 63 : i00) + 4 * (lane & 3);
            if (i00 - 63 >= 0 && i00 + EX_CH - 1 <= nrows - 1) { // every edge row lies in the plane: one vector per field (no clamping)
#pragma unroll
                for (int f = 0; f < NF; f++) {
                    const v4u_t v = (f < NIT) ? __builtin_amdgcn_raw_buffer_load_b128(rs[f], boff(ecol, erow), 0, 16)
                                              : __builtin_amdgcn_raw_buffer_load_b128(rs[f], boff(ecol, erow), 0, 0);
                    as_f4u(v, epre[f]);
                }
            } else {
#pragma unroll
                for (int f = 0; f < NF; f++)
#pragma unroll
                    for (int e = 0; e < 4; e++) {
                        const unsigned u = (f < NIT) ? __builtin_amdgcn_raw_buffer_load_b32(rs[f], boff(ecol, crow(erow + e)), 0, 16)
                                                     : __builtin_amdgcn_raw_buffer_load_b32(rs[f], boff(ecol, crow(erow + e)), 0, 0);
                        epre[f].v[e] = __uint_as_float(u);
                    }
            }
        };
        auto stash = [&](const f4u (&pre)[NP][4], const f4u (&epre)[NF], int buf) __attribute__((always_inline)) {
            float *stage = smem + buf * L::BUF, *edge = stage + L::STAGE;
#pragma unroll
            for (int p = 0; p < NF; p++)
#pragma unroll
                for (int g = 0; g < 4; g++) {
                    const int col = 16 * g + lcol;
                    *reinterpret_cast<float4 *>(&stage[(p * 64 + col) * EX_STR + 4 * lrq]) =
                        make_float4(pre[p][g].v[0], pre[p][g].v[1], pre[p][g].v[2], pre[p][g].v[3]);
                }
            {
                float4 *cimg = reinterpret_cast<float4 *>(stage + NF * 64 * EX_STR); // the packed image (PackLayout)
#pragma unroll
                for (int g = 0; g < 4; g++) {
                    const int col = 16 * g + lcol;
#pragma unroll
                    for (int k = 0; k < NCF; k++)
                        cimg[col * PL::CS + ((4 * k + lrq) ^ ((col >> 2) & 3))] =
                            make_float4(pre[NF + k][g].v[0], pre[NF + k][g].v[1], pre[NF + k][g].v[2], pre[NF + k][g].v[3]);
                }
            }
            if (lane < 8) {
                const int which = (lane >> 2) & 1;
#pragma unroll
                for (int f = 0; f < NF; f++)
                    if (which == 1 || f >= NIT || !west_by_mail)
                        *reinterpret_cast<float4 *>(&edge[(which * NF + f) * EX_CH + 4 * (lane & 3)]) =
                            make_float4(epre[f].v[0], epre[f].v[1], epre[f].v[2], epre[f].v[3]);
            }
        };
        // chunk c is fetched two barriers before it is relaxed and stashed one barrier before.  Per interval: issue the
        // dependency poll of the chunk two ahead (one load, three counters), stash the chunk that has landed while the poll is
        // in flight, then look at the poll and issue the loads.
        auto poll_issue = [&]() __attribute__((always_inline)) -> unsigned {
            return my_ptr != nullptr ? __hip_atomic_load(my_ptr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0xffffffffu;
        };
        auto poll_finish = [&](int c, unsigned seen) __attribute__((always_inline)) {
            const int need = lane == 0 ? c + 5 : (lane == 1 ? c + 2 : c - 2);
            const unsigned un = (unsigned)(need < 0 ? 0 : (need < NC ? need : NC));
            if (__all(my_ptr == nullptr || seen >= un)) return;
            persist_wait3(my_ptr, un, ctl.abort_flag); // not there yet: the bounded spin
        };
        // Measured and dropped: a fetch distance of three chunks through two register sets (two intervals for a chunk to arrive)
        // with the mailbox polled by this wave, and a publish on a counted wait one interval late in the storer -- no gain at 4K
        // (neither the load latency nor the store drain sets the pace there), 20 % slower at iter = 20 on small frames.
        wait_deps(0);
        fetch(0, preA, epreA);
        stash(preA, epreA, 0);
        if (NC > 1) {
            wait_deps(1);
            if constexpr (TWO_SETS) fetch(1, preB, epreB);
            else fetch(1, preA, epreA);
        }
        lds_barrier(); // chunk 0 is in buffer 0
        P8S_DECL;
        for (int c = 0; c < NC; c += 2) {
            // ---- while chunk c (buffer 0) is relaxed ----
            P8S_BEGIN;
            {
                const bool pf = c + 2 < NC;
                const unsigned seen = pf ? poll_issue() : 0u;
                if (c + 1 < NC) {
                    if constexpr (TWO_SETS) stash(preB, epreB, 1);
                    else stash(preA, epreA, 1);
                }
                if (pf) { poll_finish(c + 2, seen); fetch(c + 2, preA, epreA); }
            }
            P8S_END;
            lds_barrier();
            if (c + 1 >= NC) break;
            // ---- while chunk c+1 (buffer 1) is relaxed ----
            P8S_BEGIN;
            {
                const bool pf = c + 3 < NC;
                const unsigned seen = pf ? poll_issue() : 0u;
                if (c + 2 < NC) stash(preA, epreA, 0);
                if (pf) {
                    poll_finish(c + 3, seen);
                    if constexpr (TWO_SETS) fetch(c + 3, preB, epreB);
                    else fetch(c + 3, preA, epreA);
                }
            }
            P8S_END;
            lds_barrier();
        }
        P8S_WRITE;
        return;
    }

    if (role == 2) {
        // ================================ storer wave =========================================
        // relaxed chunk c: LDS -> global, write-through (sc1)
        auto store_out = [&](int c) __attribute__((always_inline)) {
            const float *outb = outb_base + (c & 1) * L::OUTB;
            const int i00 = 1 + EX_CH * c;
            const int lo_row = i00 - 63, hi_row = i00 + EX_CH - 1;
            const bool all_valid = (lo_row >= 1) && (hi_row <= nrows - 2); // every row of the chunk is an inner row: whole vectors, for the
            // inner columns (the last strip may be partial -- stored element by element its walk, which ends the call, was 30 % slower)
#pragma unroll
            for (int f = 0; f < NIT; f++)
#pragma unroll
                for (int g = 0; g < 4; g++) {
                    const int col = 16 * g + lcol;
                    const int jj = jbase + col;
                    const int row = i00 - col + 4 * lrq;
                    const float4 v = *reinterpret_cast<const float4 *>(&outb[(f * 64 + col) * EX_STR + 4 * lrq]);
                    if (all_valid) {
                        if (jj > ncols - 2) continue;
                        v4u_t u;
                        u.x = __float_as_uint(v.x); u.y = __float_as_uint(v.y); u.z = __float_as_uint(v.z); u.w = __float_as_uint(v.w);
                        __builtin_amdgcn_raw_buffer_store_b128(u, rs[f], boff(jj, row), 0, 16);
                    } else if (jj <= ncols - 2) {
                        const float vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                        for (int e = 0; e < 4; e++)
                            if (row + e >= 1 && row + e <= nrows - 2)
                                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(vv[e]), rs[f], boff(jj, row + e), 0, 16);
                    }
                }
        };
        // publish progress = c+1 once every store of this wave has left (recipe R1: drain, then the flag)
        auto publish = [&](int c) __attribute__((always_inline)) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (lane == 0) __hip_atomic_store(prog_mine, (unsigned)(c + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        };
        // West edge of chunk c (see the header): lane = 16 f + row polls the west strip's mailbox word of that row until its tag is
        // set and hands the value to the compute waves through the edge slot of the chunk's LDS buffer.  The polls queue behind the
        // stores of the previous chunk, so once the first one has returned those stores have drained: publish `pub` then.
        const int mf = lane >> 4, r16 = lane & 15;
        const bool polls = (b > 0) && (lane < 16 * NIT);
        // word of row r of field f of a strip: [(f * NC) * 16 + r - 1]: the 16 words of a chunk are one aligned 128-byte line
        const size_t mpitch = (size_t)NC * EX_CH;
        const unsigned long long *mail_w = ctl.mail + ((((size_t)frame * T + t) * B + (b > 0 ? b - 1 : 0)) * NIT + (polls ? mf : 0)) * mpitch;
        auto take = [&](int c, int pub) __attribute__((always_inline)) {
            if (b > 0 && c < NC) { // the first strip's west column is the frame border: the loader stages it
                const int row = 1 + EX_CH * c + r16;
                const bool want = polls && row <= nrows - 2;
                float v = 0.0f;
                bool ok = !want;
                const unsigned long long t0 = __builtin_amdgcn_s_memrealtime(); // 100 MHz
                for (;;) {
                    if (!ok) {
                        const unsigned long long w = __hip_atomic_load(mail_w + row - 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if ((unsigned)(w >> 32) != 0u) {
                            v = __uint_as_float((unsigned)w);
                            ok = true;
                        }
                    }
                    if (pub >= 0) {
                        publish(pub);
                        pub = -1;
                    }
                    if (__all(ok)) break;
                    if (__hip_atomic_load(ctl.abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) break;
                    if (__builtin_amdgcn_s_memrealtime() - t0 > 50000000ull) { // 0.5 s: drain the grid, the host reports it
                        __hip_atomic_store(ctl.abort_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        break;
                    }
                    __builtin_amdgcn_s_sleep(1);
                }
                if (polls) (smem + (c & 1) * L::BUF + L::STAGE)[mf * EX_CH + r16] = v; // edge[(which = 0) * NF + f][row]
            }
            if (pub >= 0) publish(pub);
        };
        take(0, -1);
        lds_barrier(); // chunk 0 is in buffer 0
        P8S_DECL;
        for (int k = 0; k < NC; k++) { // while chunk k is relaxed: write chunk k-1 back, fetch the west values of chunk k+1
            P8S_BEGIN;
            if (k >= 1) store_out(k - 1);
            take(k + 1, k >= 1 ? k - 1 : -1);
            P8S_END;
            lds_barrier();
        }
        store_out(NC - 1);
        publish(NC - 1);
        P8S_WRITE;
        return;
    }

    // mailbox rows of this strip (written by its compute waves) and of its west neighbour (read by role 4): [NIT][nrows] words
    unsigned long long *const mail_mine = ctl.mail + ((((size_t)frame * T + t) * B + b) * NIT) * ((size_t)NC * EX_CH);
    // ================================== compute wave ===========================================
    const int j = jbase + lane;
    const bool col_ok = j <= ncols - 2;
    const int jc = j < ncols - 1 ? j : ncols - 1;
    const float om1 = 1.0f - omega;
    const bool first_sweep = (t == 0);
    const bool has_east = (b + 1 < B); // my last column is the next strip's west column
    const int i0w63 = 1 - 63;          // lane 63's row at step 0 of chunk 0
    const __amdgpu_buffer_rsrc_t rs_mail = __builtin_amdgcn_make_buffer_rsrc(mail_mine, 0, (unsigned)((size_t)NIT * NC * EX_CH * 8), 0x00020000);
    const int i0 = 1 - lane; // row of this lane at step 0 of chunk 0

    auto compute_wave = [&](auto f0_tag, auto nfw_tag) __attribute__((always_inline)) {
    // this wave relaxes the fields [F0, F0 + NFW); of the other fields it only follows the centre value (the coupling term)
    constexpr int F0 = decltype(f0_tag)::value, NFW = decltype(nfw_tag)::value;
    auto mine = [](int f) { return f >= F0 && f < F0 + NFW; };
    lds_barrier(); // the mover has passed the dependency waits of chunks 0 and 1
    // state at step 0 (rows <= 1): sc1 loads, after the waits above
    float prev[NIT], cen[NIT], north0[NIT], west0[NIT], topb[NIT], rcen[NRO1], rnorth[NRO1];
#pragma unroll
    for (int f = 0; f < NIT; f++) {
        cen[f] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs[f], boff(jc, crow(i0)), 0, 16));
        north0[f] = west0[f] = topb[f] = prev[f] = 0.0f;
        if (mine(f)) {
            north0[f] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs[f], boff(jc, crow(i0 - 1)), 0, 16));
            west0[f] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs[f], boff(jc - 1, crow(i0)), 0, 16));
            topb[f] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs[f], boff(jc, 0), 0, 16));
        }
    }
#pragma unroll
    for (int f = 0; f < NRO1; f++) {
        rcen[f] = (NRO > 0) ? __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs[(NRO > 0 ? NIT + f : 0)], boff(jc, crow(i0)), 0, 0)) : 0.0f;
        rnorth[f] = (NRO > 0) ? __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs[(NRO > 0 ? NIT + f : 0)], boff(jc, crow(i0 - 1)), 0, 0)) : 0.0f;
    }

    P8S_DECL;
    for (int k = 0; k < NC; k++) {
        P8S_BEGIN;
        const float *stage = smem + (k & 1) * L::BUF, *edge = stage + L::STAGE;
        float *outb = outb_base + (k & 1) * L::OUTB;
        const int i00 = 1 + EX_CH * k;
        auto relax_chunk = [&](auto interior_tag) __attribute__((always_inline)) {
        // 2: every lane relaxes an inner pixel at every step and no tap is a border cell; 1: the same for the rows, but the strip
        // holds column 1 or ncols-2 or is partial (the first strip heads the whole walk, the last one ends it: they should not
        // pay for row tests they never need); 0: anything
        constexpr bool INTERIOR = decltype(interior_tag)::value == 2, ROWS_IN = decltype(interior_tag)::value >= 1;
#pragma unroll
        for (int mq = 0; mq < EX_CH / 4; mq++) {
            float4 s4[NF], e4[NF], res[NIT];
            float cflat[4 * NCF]; // coefficients of the four rows of this group, [row][f]: granules NCF mq .. NCF mq + NCF - 1 of my run
            {
                const float4 *cimg = reinterpret_cast<const float4 *>(stage + NF * 64 * EX_STR) + lane * PL::CS;
#pragma unroll
                for (int jq = 0; jq < NCF; jq++) {
                    const float4 v = cimg[(NCF * mq + jq) ^ ((lane >> 2) & 3)];
                    cflat[4 * jq] = v.x; cflat[4 * jq + 1] = v.y; cflat[4 * jq + 2] = v.z; cflat[4 * jq + 3] = v.w;
                }
            }
#pragma unroll
            for (int f = 0; f < NF; f++) {
                s4[f] = *reinterpret_cast<const float4 *>(&stage[(f * 64 + lane) * EX_STR + 4 * mq]);
                e4[f] = *reinterpret_cast<const float4 *>(&edge[((lane == 63 ? 1 : 0) * NF + f) * EX_CH + 4 * mq]);
            }
#pragma unroll
            for (int x = 0; x < 4; x++) {
                const int q = EX_CH * k + 4 * mq + x;
                const int i = i0 + q;
                const bool row_ok = ROWS_IN || ((i >= 1) && (i <= nrows - 2));
                const bool active = INTERIOR || (col_ok && row_ok);
                auto el = [&](const float4 &v) { return x == 0 ? v.x : (x == 1 ? v.y : (x == 2 ? v.z : v.w)); };
                float c[NIT], w[NIT], e[NIT], n[NIT], s[NIT], kk[NCF];
                float rsouth[NRO1], reast[NRO1], rwest[NRO1];
#pragma unroll
                for (int f = 0; f < NIT; f++) {
                    const float sraw = el(s4[f]);
                    c[f] = cen[f];
                    if (mine(f)) {
                        const float eraw = dpp_from_upper_lane(sraw, el(e4[f]));
                        float wnew = dpp_from_lower_lane(prev[f], el(e4[f]));
                        if (q == 0 && lane != 0) wnew = west0[f];
                        float nv = (q == 0) ? north0[f] : prev[f];
                        if (!ROWS_IN && i == 1) nv = first_sweep ? topb[f] : cen[f];
                        n[f] = nv;
                        s[f] = (!ROWS_IN && i + 1 == nrows - 1 && !first_sweep) ? cen[f] : sraw;
                        e[f] = (!INTERIOR && j + 1 == ncols - 1 && !first_sweep) ? cen[f] : eraw;
                        w[f] = (!INTERIOR && j - 1 == 0 && !first_sweep) ? cen[f] : wnew;
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
            for (int f = F0; f < F0 + NFW; f++) *reinterpret_cast<float4 *>(&outb[(f * 64 + lane) * EX_STR + 4 * mq]) = res[f];
        }
        };
        {
            const int lo_row = i00 - 63, hi_row = i00 + EX_CH - 1;
            const bool rows_in = (lo_row >= 2) && (hi_row <= nrows - 3), cols_in = (jbase >= 2) && (jbase + 63 <= ncols - 3);
            if (rows_in && cols_in) relax_chunk(std::integral_constant<int, 2>{});
            else if (rows_in) relax_chunk(std::integral_constant<int, 1>{});
            else relax_chunk(std::integral_constant<int, 0>{});
        }
        if (has_east) {
            // Mailbox: lane 63's 16 results of this chunk (rows i0 + 16k .. + 15 of my last column) go out as ONE 128-byte line of
            // self-validating {value, tag} words per field -- lanes 0..15 pick them up from the out buffer this wave just wrote
            // (LDS operations of one wave complete in order) and store 8 bytes each.  A full-line store needs no read of the
            // line it replaces; the piecemeal form (lane 63 storing four words at a time) stalled on exactly that once the
            // planes had pushed the mailbox out of the Infinity Cache.
            typedef unsigned int v2u_t __attribute__((ext_vector_type(2)));
            const int r = (i0w63 + EX_CH * k) + (lane & 15); // row of word (lane & 15)
            if (lane < 16 && r >= 1 && r <= nrows - 2) {
#pragma unroll
                for (int f = F0; f < F0 + NFW; f++) {
                    v2u_t wv;
                    wv.x = __float_as_uint(outb[(f * 64 + 63) * EX_STR + lane]);
                    wv.y = 1u;
                    __builtin_amdgcn_raw_buffer_store_b64(wv, rs_mail, (unsigned)(((size_t)f * NC * EX_CH + r - 1) * 8), 0, 16); // sc1
                }
            }
        }
        P8S_END;
        lds_barrier();
    }
    P8S_WRITE;
    };
    if constexpr (NIT == 2) {
        if (role == 0) compute_wave(std::integral_constant<int, 0>{}, std::integral_constant<int, 1>{});
        else compute_wave(std::integral_constant<int, 1>{}, std::integral_constant<int, 1>{});
    } else {
        compute_wave(std::integral_constant<int, 0>{}, std::integral_constant<int, NIT>{});
    }
}

// Final border replicate of the iterate (rows first, then columns; opticalflowSolvers.c:161-179):
// border cell <- nearest interior pixel.  Reads interior cells only, writes border cells only.
static __global__ void k_fill_borders(float *p0, float *p1, int nfields, int nrows, int ncols,
                               size_t frame_stride)
{
    const int n = 2 * ncols + 2 * (nrows - 2);
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n) return;
    int i, j;
    if (idx < ncols) { i = 0; j = idx; }
    else if (idx < 2 * ncols) { i = nrows - 1; j = idx - ncols; }
    else if (idx < 2 * ncols + nrows - 2) { i = 1 + idx - 2 * ncols; j = 0; }
    else { i = 1 + idx - 2 * ncols - (nrows - 2); j = ncols - 1; }
    const int si = i < 1 ? 1 : (i > nrows - 2 ? nrows - 2 : i);
    const int sj = j < 1 ? 1 : (j > ncols - 2 ? ncols - 2 : j);
    const size_t fo = (size_t)blockIdx.y * frame_stride;
    float *p = (blockIdx.z == 0 ? p0 : p1) + fo;
    if ((int)blockIdx.z < nfields) p[(size_t)j * nrows + i] = p[(size_t)sj * nrows + si];
}

} // namespace pdeip
