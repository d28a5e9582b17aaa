// pdeip_sor_rbp.hpp -- S red-black sweeps per launch as a wave pipeline through LDS (gfx950).
//
// Why.  One red-black sweep moves every plane through HBM once (52 B per pixel for the Horn-Schunck model);
// k_sor_rb<TWO> fuses two sweeps in the registers of one wave and is then bound by its own instruction
// stream at one wave per SIMD (359 VGPRs), and a third and fourth sweep do not fit a register file.  The
// solver calls of every driver run iter = 4 sweeps, which share all coefficient planes: read once, a call
// needs 13 planes of traffic instead of 4 x 13.
//
// How.  A workgroup owns a unit of 240 rows x TJ columns and marches along the columns like k_sor_rb, but
// the S sweeps are S WAVES, one per SIMD, each a plain single-sweep march (red on column x, black on x-1):
//
//     loader wave  --LDS-DMA-->  K ring  (coefficient / read-only columns, read by every sweep wave)
//                  --LDS-DMA-->  O ring  (old iterate columns)  -> wave 0 (sweep 1) -> H ring 0 -> wave 1 -> ... -> wave S-1 -> HBM
//
// Wave s trails wave s-1 by three columns: in step t it reads the column its predecessor finished in step
// t-1, so one workgroup barrier per step is the only synchronisation.  The loader wave issues
// global_load_lds_dwordx4 (no registers, 1 KiB per instruction: a column of 256 rows is contiguous in the
// MATLAB layout) P steps ahead and retires them with a counted s_waitcnt vmcnt; nothing else in the kernel
// reads global memory.  Each sweep wave keeps three-column windows in registers as k_sor_rb does.
//
// Exactness.  Per pixel this is Mdl::update() through rb_phase() like every other ordering; the border
// replicate the reference does between sweeps (rows, then columns: opticalflowSolvers.c:161-179) is reproduced
// as in rb_march2: rows in the finished column before it is handed on, "column 0 = column 1" substituted where
// the next sweep reads it.  Bit-identical to S launches of k_sor_rb (tests/test_gpu_parity.py runs through it).
//
// Halo.  A half-sweep invalidates one halo row and one halo column per side: 2S columns per side are loaded
// and 8 rows (two lanes) per side are recomputed, so a wave owns 240 rows; redundancy (TJ + 4S)/TJ x 256/240.
#pragma once
#include "pdeip_sor_rb.hpp"

namespace pdeip {

constexpr int RBP_OWN_ROWS = 240; // 60 storing lanes x 4 rows; lanes 0,1 and 62,63 are halo lanes (8 half-sweeps)

template <class Mdl, int S> struct RbpLayout {
    static constexpr int NIT = Mdl::NIT, NRO = Mdl::NRO, NCF = Mdl::NCF;
    static constexpr int NRING = NCF + NRO;            // planes every sweep wave reads: coefficients, then read-only fields
    static constexpr int COL = 256;                    // floats per plane column (64 lanes x 4 rows)
    static constexpr int GROUP = NRING + NIT;          // LDS-DMA instructions per column
    // K-ring columns for a DMA lead of p steps.  Column group g lands before step g; wave s reads it into registers (as "column
    // x+1", one step ahead of its use) in step g + 3s -- sweep 0 of a first launch writes the derived planes back into the slot in
    // step g + 1 --; the slot is refilled by group g + NK, issued in step g + NK - p, which has to be a LATER step than wave S-1's
    // read in step g + 3(S-1): NK >= p + 3(S-1) + 1.  (Until the coefficients were read a step ahead this was one column more;
    // the column saved is what lets the lead grow from four to five for the coupled models.)
#ifndef RBP_NK_SLACK
#define RBP_NK_SLACK 1
#endif
    static constexpr int nk_for(int p) { return p + 3 * (S - 1) + RBP_NK_SLACK; }
    // The read-only fields (late-linearisation models) live in a ring of their own: they ride one column ahead of the
    // coefficients (the red half reads them at x+1) into register windows, so wave s reads group g once, in step g + 3s;
    // the last reader is wave S-1 in step g + 3(S-1): one column fewer than the coefficient ring needs.
    static constexpr int nq_for(int p) { return NRO > 0 ? p + 3 * (S - 1) + 1 : 0; }
    static constexpr size_t lds_for(int p)
    {
        return (size_t)(nk_for(p) * NCF * COL + nq_for(p) * NRO * COL + (p + 1) * NIT * COL + (S - 1) * 2 * NIT * COL) * sizeof(float);
    }
    // DMA lead in steps: what is in flight per CU (lead x GROUP KiB) has to cover an HBM round trip at the rate the
    // kernel consumes it, so take the longest lead the 160 KiB of LDS and the 6-bit vmcnt counter allow (at most 8)
    static constexpr int pick_lead()
    {
        int best = 0;
        for (int p = 1; p <= 8; p++)
            if (lds_for(p) <= (size_t)160 * 1024 && (p - 1) * GROUP <= 63) best = p;
        return best;
    }
#ifdef RBP_LEAD
    static constexpr int P = RBP_LEAD;
#else
    static constexpr int P = pick_lead();
#endif
    static constexpr int NK = nk_for(P);
    static constexpr int NQ = nq_for(P) > 0 ? nq_for(P) : 1;
    static constexpr int NO = P + 1;                   // O-ring columns
    static constexpr int K_FLOATS = NK * NCF * COL;
    static constexpr int Q_FLOATS = nq_for(P) * NRO * COL;
    static constexpr int O_FLOATS = NO * NIT * COL;
    static constexpr int H_FLOATS = (S - 1) * 2 * NIT * COL;
    static constexpr size_t LDS_BYTES = (size_t)(K_FLOATS + Q_FLOATS + O_FLOATS + H_FLOATS) * sizeof(float);
    static constexpr int NW = (NIT == 2) ? 2 : 1;      // waves per sweep: the two fields of a coupled model are relaxed by two waves
#ifndef RBP_LOADERS
#define RBP_LOADERS 1
#endif
    static constexpr int NLOAD = RBP_LOADERS;          // loader waves: loader w issues the pieces i of a column group with i % NLOAD == w
    static constexpr int THREADS = 64 * (S * NW + NLOAD); // sweep waves + the loader wave(s)
    static constexpr int pieces_of(int w) { return (GROUP - w + NLOAD - 1) / NLOAD; }
    static constexpr int HALO = 2 * S;                 // columns per side
    static constexpr bool FITS = P >= 2 && LDS_BYTES <= 160 * 1024 && (P - 1) * GROUP <= 63;
    // wave S-1 finishes column j1-1 in step TJ + 5S - 2 (two warm-up steps, 2S halo columns, three columns of lag per sweep)
    __host__ __device__ static constexpr int nsteps(int tj) { return tj + 5 * S - 1; }
};

// one column of one plane: 64 lanes x 16 bytes, contiguous
// (explicit LDS address space: with generic pointers the compiler merged the hand-off write of sweeps 0..S-2 with the global
// store of sweep S-1 into ONE flat store behind a pointer select -- an LDS write through the flat path, counted on vmcnt too)
typedef float rbp_f4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) rbp_f4 rbp_lds_f4;
__device__ __forceinline__ void rbp_lds_read(float (&d)[4], const float *col, int lane)
{
    const rbp_f4 t = *(const rbp_lds_f4 *)(col + 4 * lane);
    d[0] = t.x; d[1] = t.y; d[2] = t.z; d[3] = t.w;
}
__device__ __forceinline__ void rbp_lds_write(float *col, int lane, const float (&d)[4])
{
    const rbp_f4 t = {d[0], d[1], d[2], d[3]};
    *(rbp_lds_f4 *)(col + 4 * lane) = t;
}

// rows first (opticalflowSolvers.c:161-170): border row 0 <- row 1, border row nrows-1 <- row nrows-2, in the lane that holds them.
// The empty asm makes the four values opaque: without it the compiler turns the first select into a load from a SELECTED
// ADDRESS of d[], which keeps the whole array in scratch memory.
__device__ __forceinline__ void rbp_replicate_rows(float (&d)[4], bool top_lane, bool bot_lane)
{
    float a0 = d[0], a1 = d[1], a2 = d[2], a3 = d[3];
    asm("" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));
    d[0] = top_lane ? a1 : a0;
    d[3] = bot_lane ? a2 : a3;
}

// address-space casts for __builtin_amdgcn_global_load_lds (global source per lane, LDS destination = wave-uniform base + 16 x lane)
#ifndef RBP_AUX_COEF
#define RBP_AUX_COEF 0 /* default cache policy; nt (2) measured the same: the kernel is bound by instruction issue, not by the DMA */
#endif
#define RBP_LDS(p) ((__attribute__((address_space(3))) void *)(p))
#define RBP_GLB(p) ((const __attribute__((address_space(1))) void *)(p))

#ifdef PDEIP_RBP_STAMPS // diagnostic build only (tools/rbp_stamps.py): where one sweep wave's step goes; never in the product
__device__ unsigned long long g_rbp_stamps[256];
#define RBP_STAMP_T0 60
#define RBP_STAMP(n)                                                                                              \
    do {                                                                                                          \
        if (stamp_on) {                                                                                           \
            unsigned long long t_;                                                                                \
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                            \
            if (lane == 0) g_rbp_stamps[(t - RBP_STAMP_T0) * 8 + (n)] = t_;                                        \
        }                                                                                                         \
    } while (0)
#else
#define RBP_STAMP(n)
#endif

// LDS traffic only: the loader's DMA stays in flight across it (a __syncthreads() would drain vmcnt)
#ifdef RBP_NO_BARRIER /* timing experiment only (results are garbage): what the step costs without the workgroup-wide meeting */
__device__ __forceinline__ void rbp_barrier() { asm volatile("" ::: "memory"); }
#else
__device__ __forceinline__ void rbp_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
#endif

// One colour of one column, branch-free, for the fields [F0, F0+NF) of the model: OUT <- CEN with elements {E0, E0+2}
// relaxed where the row is an interior pixel (ok[], fixed per lane for the whole march).  XC holds the centre values of ALL
// fields (the coupled models update u from the old v of the same pixel and vice versa); neighbours are needed of the own
// fields only.  The arithmetic is Mdl::update(), as in every other ordering; what the other fields' half of it would
// produce is unused here and drops out at compile time.  OUT and CEN are distinct register sets, so the window the result
// goes into never has to be copied first.
template <class Mdl, int F0, int NF, int E0>
__device__ __forceinline__ void rbp_phase(float (&OUT)[NF][4], const float (&C)[NF][4], const float (&W)[NF][4], const float (&E)[NF][4],
                                          const float (&XC)[Mdl::NIT][4], const float (&rC)[at_least_one<Mdl::NRO>::value][4],
                                          const float (&rW)[at_least_one<Mdl::NRO>::value][4],
                                          const float (&rE)[at_least_one<Mdl::NRO>::value][4], const float (&cf)[Mdl::NCF][4],
                                          const bool (&ok)[4], float omega, float om1)
{
    constexpr int NIT = Mdl::NIT, NRO = Mdl::NRO, NRO1 = at_least_one<NRO>::value;
    float edge[NF], redge[NRO1]; // the neighbouring lane's adjacent row
#pragma unroll
    for (int f = 0; f < NF; f++) edge[f] = (E0 == 0) ? lane_above(C[f][3]) : lane_below(C[f][0]);
#pragma unroll
    for (int f = 0; f < NRO1; f++) redge[f] = (NRO == 0) ? 0.0f : ((E0 == 0) ? lane_above(rC[f][3]) : lane_below(rC[f][0]));
#pragma unroll
    for (int e = 0; e < 4; e++) {
        if ((e & 1) != E0) {
#pragma unroll
            for (int f = 0; f < NF; f++) OUT[f][e] = C[f][e];
            continue;
        }
        float c[NIT], w[NIT], ea[NIT], n[NIT], s[NIT];
        float rc[NRO1], rw[NRO1], re[NRO1], rn[NRO1], rs[NRO1], k[Mdl::NCF];
#pragma unroll
        for (int f = 0; f < NIT; f++) {
            c[f] = XC[f][e];
            w[f] = ea[f] = n[f] = s[f] = 0.0f;
        }
#pragma unroll
        for (int f = 0; f < NF; f++) {
            c[F0 + f] = C[f][e];
            w[F0 + f] = W[f][e];
            ea[F0 + f] = E[f][e];
            n[F0 + f] = (e == 0) ? edge[f] : C[f][e == 0 ? 0 : e - 1];
            s[F0 + f] = (e == 3) ? edge[f] : C[f][e == 3 ? 3 : e + 1];
        }
#pragma unroll
        for (int f = 0; f < NRO1; f++) {
            rc[f] = rC[f][e];
            rw[f] = rW[f][e];
            re[f] = rE[f][e];
            rn[f] = (e == 0) ? redge[f] : rC[f][e == 0 ? 0 : e - 1];
            rs[f] = (e == 3) ? redge[f] : rC[f][e == 3 ? 3 : e + 1];
        }
#pragma unroll
        for (int f = 0; f < Mdl::NCF; f++) k[f] = cf[f][e];
        Mdl::update(c, w, ea, n, s, rc, rw, re, rn, rs, k, omega, om1);
#pragma unroll
        for (int f = 0; f < NF; f++) OUT[f][e] = ok[e] ? c[F0 + f] : C[f][e];
    }
}

// The march of one sweep wave: sweep s of the launch, fields [F0, F0+NF) of the model.
template <class Mdl, int S, bool FIRST, int F0, int NF>
__device__ __forceinline__ void rbp_sweep_wave(const SweepPlanes<Mdl> &P, float *dout0, float *dout1, float *Kring, float *Qring, float *Oring, float *Hring,
                                               int s, int lane, int r, int nrows, int ncols, int j0, int j1, int xbase, int nsteps,
                                               float omega, int col0, size_t fo, int mirror)
{
    using L = RbpLayout<Mdl, S>;
    constexpr int NIT = L::NIT, NRO = L::NRO, NRO1 = at_least_one<NRO>::value, NCF = L::NCF, COL = L::COL;
    const float om1 = 1.0f - omega;
    const bool store_lane = (lane >= 2) && (lane <= 61);
    auto inner = [&](int col) { return col >= 1 && col <= ncols - 2; };
    auto cmap = [&](int col) { return mirror ? ncols - 1 - col : col; }; // march column -> column in memory (see k_sor_rbp)
    // per-lane row predicates, fixed for the whole march: which of the lane's four rows are interior pixels, and whether
    // the lane holds the top / bottom border row (nrows is a multiple of 4 here, so they are elements 0 and 3)
    bool ok[4];
#pragma unroll
    for (int e = 0; e < 4; e++) ok[e] = (r + e >= 1) && (r + e <= nrows - 2);
    const bool top_lane = (r == 0), bot_lane = (r == nrows - 4);

    // Three-column windows.  The step below is instantiated three times with the roles rotated, so no window ever moves:
    // O*: the previous sweep's result at x-1, x, x+1, ALL fields (centre values of the other fields feed the coupling term);
    // R*: own fields after this sweep's red half at x-2, x-1, x.
    float O0[NIT][4], O1[NIT][4], O2[NIT][4], R0[NF][4], R1[NF][4], R2[NF][4];
    float Q2[NRO1][4], Q1[NRO1][4], Q0[NRO1][4], Qn[NRO1][4]; // read-only fields at x-2, x-1, x, x+1 (moved: four columns, few planes)
    // coefficients of columns x-1, x and x+1, roles rotating: each column is read from the K ring once -- one step AHEAD of its
    // use (round 3): a step then waits for the two planes its predecessor handed over, not for nine more (the K column of
    // group t has landed when step t begins: the loader brings it in with the O column the same step reads)
    float KA[NCF][4], KB[NCF][4], KC[NCF][4];
#pragma unroll
    for (int f = 0; f < NCF; f++)
#pragma unroll
        for (int e = 0; e < 4; e++) KA[f][e] = KB[f][e] = KC[f][e] = 0.0f;
#pragma unroll
    for (int e = 0; e < 4; e++) {
#pragma unroll
        for (int f = 0; f < NIT; f++) O0[f][e] = O1[f][e] = O2[f][e] = 0.0f;
#pragma unroll
        for (int f = 0; f < NF; f++) R0[f][e] = R1[f][e] = R2[f][e] = 0.0f;
#pragma unroll
        for (int f = 0; f < NRO1; f++) Q2[f][e] = Q1[f][e] = Q0[f][e] = Qn[f][e] = 0.0f;
    }
    // ring positions: K ring slot of column x (group x - xbase - 1 = t - 3s - 1; a wave that is still in front of the first
    // fetched column reads some slot whose content it never uses), O ring slot of column x+1 (sweep 0), H ring parity
    int ki = ((-3 * s - 1) % L::NK + L::NK) % L::NK, qi = ((-3 * s) % L::NQ + L::NQ) % L::NQ, oi = 0, hp = 1;

    auto step = [&](int t, float (&Om)[NIT][4], float (&Oc)[NIT][4], float (&Op)[NIT][4], float (&Rpp)[NF][4], float (&Rp)[NF][4],
                    float (&Rc)[NF][4], const float (&Kp)[NCF][4], float (&Kc)[NCF][4], float (&Kn)[NCF][4]) __attribute__((always_inline)) {
        const int x = xbase + t - 3 * s; // this wave's red column; black on x-1
#ifdef PDEIP_RBP_STAMPS
        const bool stamp_on = (blockIdx.x == 100) && (s == RBP_STAMP_SWEEP) && (F0 == 0) && (t >= RBP_STAMP_T0) && (t < RBP_STAMP_T0 + 24);
#endif
        RBP_STAMP(0);
        // ---- take in column x+1 of the previous sweep's result, and the coefficients of columns x and x-1 ----
        float *const ks = Kring + (size_t)ki * NCF * COL; // slot of column x (Kc, read one step ago -- before the loop for step 0; column x-1 is Kp)
        {
            const float *src = (s == 0) ? Oring + (size_t)oi * NIT * COL : Hring + (size_t)((s - 1) * 2 + hp) * NIT * COL;
#pragma unroll
            for (int f = 0; f < NIT; f++) rbp_lds_read(Op[f], src + f * COL, lane);
#pragma unroll
            for (int f = 0; f < NRO1; f++)
                if (NRO > 0) {
#pragma unroll
                    for (int e = 0; e < 4; e++) {
                        Q2[f][e] = Q1[f][e];
                        Q1[f][e] = Q0[f][e];
                        Q0[f][e] = Qn[f][e];
                    }
                    // read-only fields ride one column ahead of the coefficients (the red half reads them at x+1)
                    rbp_lds_read(Qn[f], Qring + ((size_t)qi * NRO + (NRO > 0 ? f : 0)) * COL, lane);
                }
        }
        const int p = (x + col0) & 1;
#ifdef PDEIP_RBP_STAMPS
        if (stamp_on) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
        RBP_STAMP(1); // the step's LDS reads have landed

        if (FIRST && s == 0) {
            // Sweep 1 of a call builds the divisor planes as it goes (opticalflowSolvers.c:111-127): column x was just read
            // raw; derive it, put the derived planes back into its K-ring slot (this wave reads them again as column x-1 in the
            // next step, the later sweeps three steps from now) and store them if launches follow that need them.  A wave
            // that owns one field of a coupled model derives that field's divisor.
            // The reciprocals are this wave's heaviest arithmetic (the whole workgroup waits for it at the barrier): v_rcp_f32 + one
            // Newton step where every denominator of the wave is in the range in which that equals the IEEE quotient bit for
            // bit (RcpFast, pdeip_models.hpp), the division itself otherwise.
            constexpr int WHICH = (NF == NIT) ? 3 : (1 << F0);
            // the raw coefficients of row e; a mirrored launch has wW and wE swapped, and a derive() that adds them in a fixed order
            // (disparitySolvers.c:94-113) gets them back in their memory roles
            auto raw = [&](float (&k)[NCF], int e) __attribute__((always_inline)) {
#pragma unroll
                for (int f = 0; f < NCF; f++) k[f] = Kc[f][e];
                if (!Mdl::DERIVE_WE_SYMMETRIC) {
                    const float w = k[Mdl::cWW], ea = k[Mdl::cWE];
                    k[Mdl::cWW] = mirror ? ea : w;
                    k[Mdl::cWE] = mirror ? w : ea;
                }
            };
            bool in_range = true;
#pragma unroll
            for (int e = 0; e < 4; e++) {
                float k[NCF];
                raw(k, e);
                Mdl::template derive<WHICH>(k, RcpRange{in_range});
            }
            float D[2][4]; // the fast form unconditionally (three instructions a value, harmless outside its range) ...
#pragma unroll
            for (int e = 0; e < 4; e++) {
                float k[NCF];
                raw(k, e);
#ifdef RBP_IEEE_DIV /* A/B aid: the division always */
                Mdl::template derive<WHICH>(k, RcpIeee());
#else
                Mdl::template derive<WHICH>(k, RcpFast());
#endif
                D[0][e] = k[Mdl::D0];
                D[1][e] = k[Mdl::D1];
            }
            if (__builtin_expect(__builtin_amdgcn_ballot_w64(!in_range) != 0, 0)) { // ... and the division for the whole wave in the rare case
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    float k[NCF];
                    raw(k, e);
                    Mdl::template derive<WHICH>(k, RcpIeee());
                    D[0][e] = k[Mdl::D0];
                    D[1][e] = k[Mdl::D1];
                }
            }
#pragma unroll
            for (int e = 0; e < 4; e++) {
                Kc[Mdl::D0][e] = D[0][e];
                Kc[Mdl::D1][e] = D[1][e];
            }
            const bool d0_mine = (NF == NIT) || F0 == 0, d1_mine = (NF == NIT) || F0 == 1;
            if (d0_mine) rbp_lds_write(ks + Mdl::D0 * COL, lane, Kc[Mdl::D0]);
            if (d1_mine) rbp_lds_write(ks + Mdl::D1 * COL, lane, Kc[Mdl::D1]);
            if (dout0 != nullptr && store_lane && x >= j0 && x < j1) {
                if (d0_mine) rb_store4<true>(Kc[Mdl::D0], dout0 + fo, cmap(x), r, nrows);
                if (d1_mine) rb_store4<true>(Kc[Mdl::D1], dout1 + fo, cmap(x), r, nrows);
            }
        }

#define PDEIP_RBP_PHASE(OUT, CEN, WEST, EAST, XCEN, QC, QW, QE, KK)                                               \
    do {                                                                                                          \
        if (p == 0) rbp_phase<Mdl, F0, NF, 0>(OUT, CEN, WEST, EAST, XCEN, QC, QW, QE, KK, ok, omega, om1);        \
        else        rbp_phase<Mdl, F0, NF, 1>(OUT, CEN, WEST, EAST, XCEN, QC, QW, QE, KK, ok, omega, om1);        \
    } while (0)
        // the own fields of the O windows
        float OmF[NF][4], OcF[NF][4], OpF[NF][4];
#pragma unroll
        for (int f = 0; f < NF; f++)
#pragma unroll
            for (int e = 0; e < 4; e++) {
                OmF[f][e] = Om[F0 + f][e];
                OcF[f][e] = Oc[F0 + f][e];
                OpF[f][e] = Op[F0 + f][e];
            }

        // Coefficients of column x+1, for the next step: read BETWEEN the two half-sweeps.  At the top of the step they sat between
        // the barrier and the first multiply (the compiler waits for every outstanding LDS read there); behind both half-sweeps they
        // sat in front of the hand-off write, whose acknowledgement the barrier waits for.  Here the black half hides them.
        auto read_next_coefficients = [&]() __attribute__((always_inline)) {
            const float *kn = Kring + (size_t)(ki + 1 == L::NK ? 0 : ki + 1) * NCF * COL;
#pragma unroll
            for (int f = 0; f < NCF; f++) rbp_lds_read(Kn[f], kn + f * COL, lane);
        };
        const int xb = x - 1;
        float F[NF][4];
        if (x >= 3 && x <= ncols - 3) {
            // ---- the common case: neither half-sweep touches or reads a border column; one parity branch for both ----
            // red on column x: Rc <- Oc with the red pixels relaxed; black on column x-1: F <- Rp with the black pixels relaxed
            // (the other fields' centre values at the black pixels are still the previous sweep's: the red half left them alone)
            if (p == 0) {
                rbp_phase<Mdl, F0, NF, 0>(Rc, OcF, OmF, OpF, Oc, Q0, Q1, Qn, Kc, ok, omega, om1);
                read_next_coefficients();
                rbp_phase<Mdl, F0, NF, 0>(F, Rp, Rpp, Rc, Om, Q1, Q2, Q0, Kp, ok, omega, om1);
            } else {
                rbp_phase<Mdl, F0, NF, 1>(Rc, OcF, OmF, OpF, Oc, Q0, Q1, Qn, Kc, ok, omega, om1);
                read_next_coefficients();
                rbp_phase<Mdl, F0, NF, 1>(F, Rp, Rpp, Rc, Om, Q1, Q2, Q0, Kp, ok, omega, om1);
            }
#pragma unroll
            for (int f = 0; f < NF; f++) { // rows first (:161-170): border row 0 <- row 1, border row nrows-1 <- row nrows-2
                rbp_replicate_rows(F[f], top_lane, bot_lane);
            }
        } else {
            // ---- columns at the image border (and the clamped columns outside it) ----
            read_next_coefficients();
            if (inner(x)) {
                // sweeps after the first see a border column as the replicate of its inner neighbour after the previous sweep
                // (:172-179), i.e. as this column itself; the first sweep of a launch reads the stored border
                const bool wb = (s > 0) && !inner(x - 1), eb = (s > 0) && !inner(x + 1);
                float Wv[NF][4], Ev[NF][4];
#pragma unroll
                for (int f = 0; f < NF; f++)
#pragma unroll
                    for (int e = 0; e < 4; e++) {
                        Wv[f][e] = wb ? OcF[f][e] : OmF[f][e];
                        Ev[f][e] = eb ? OcF[f][e] : OpF[f][e];
                    }
                PDEIP_RBP_PHASE(Rc, OcF, Wv, Ev, Oc, Q0, Q1, Qn, Kc);
            } else {
#pragma unroll
                for (int f = 0; f < NF; f++)
#pragma unroll
                    for (int e = 0; e < 4; e++) Rc[f][e] = OcF[f][e];
            }
            if (inner(xb)) {
                // a border column kept its replicate through this sweep's red half: the previous sweep's result of column xb (= Om)
                const bool wb = (s > 0) && !inner(xb - 1), eb = (s > 0) && !inner(xb + 1);
                float Wv[NF][4], Ev[NF][4];
#pragma unroll
                for (int f = 0; f < NF; f++)
#pragma unroll
                    for (int e = 0; e < 4; e++) {
                        Wv[f][e] = wb ? OmF[f][e] : Rpp[f][e];
                        Ev[f][e] = eb ? OmF[f][e] : Rc[f][e];
                    }
                PDEIP_RBP_PHASE(F, Rp, Wv, Ev, Om, Q1, Q2, Q0, Kp);
#pragma unroll
                for (int f = 0; f < NF; f++) {
                    rbp_replicate_rows(F[f], top_lane, bot_lane);
                }
            } else {
#pragma unroll
                for (int f = 0; f < NF; f++)
#pragma unroll
                    for (int e = 0; e < 4; e++) F[f][e] = Rp[f][e];
            }
        }
#undef PDEIP_RBP_PHASE
        RBP_STAMP(2); // both half-sweeps issued
        if (s < S - 1) {
            float *dst = Hring + (size_t)(s * 2 + (hp ^ 1)) * NIT * COL;
#pragma unroll
            for (int f = 0; f < NF; f++) rbp_lds_write(dst + (F0 + f) * COL, lane, F[f]);
        } else if (store_lane && xb >= j0 && xb < j1 && inner(xb)) {
#pragma unroll
            for (int f = 0; f < NF; f++) {
                float *out = P.it_out[F0 + f] + fo;
                rb_store4<true>(F[f], out, cmap(xb), r, nrows);
                if (xb == 1) rb_store4<true>(F[f], out, cmap(0), r, nrows); // then columns (:172-179)
                if (xb == ncols - 2) rb_store4<true>(F[f], out, cmap(ncols - 1), r, nrows);
            }
        }
        ki = (ki + 1 == L::NK) ? 0 : ki + 1;
        qi = (qi + 1 == L::NQ) ? 0 : qi + 1;
        oi = (oi + 1 == L::NO) ? 0 : oi + 1;
        hp ^= 1;
        RBP_STAMP(3); // hand-off written / stores issued, bookkeeping done
#ifdef PDEIP_RBP_STAMPS
        if (stamp_on) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
        RBP_STAMP(4); // LDS writes acknowledged
        rbp_barrier();
        RBP_STAMP(5); // through the barrier
    };

    rbp_barrier(); // group 0 is in LDS
#pragma unroll
    for (int f = 0; f < NCF; f++) rbp_lds_read(KB[f], Kring + ((size_t)ki * NCF + f) * COL, lane); // "column x" of step 0
    // the O/R windows and the coefficient triple (columns x-1, x, x+1) rotate with period 3: three steps per trip
    for (int t = 0; t < nsteps; t += 3) {
        step(t, O0, O1, O2, R0, R1, R2, KA, KB, KC);
        if (t + 1 < nsteps) step(t + 1, O1, O2, O0, R1, R2, R0, KB, KC, KA);
        if (t + 2 < nsteps) step(t + 2, O2, O0, O1, R2, R0, R1, KC, KA, KB);
    }
}

// Mirrored units (`mirror_mode` 1: the odd strips, 2: every strip, 0: none) march their strip from its last column to its first.
// A red-black half-sweep does not depend on the order its pixels are visited in, so a mirrored unit is the same code on mirrored
// column addresses (loader, stores) with the planes wW and wE exchanged and the colour of column 0 adjusted: the two products
// W*wW and E*wE reach the same addition in the other order, which IEEE addition does not notice (a derive() that adds wE and
// wW in a fixed order gets them back in their memory roles).  Why: strip b reads the halo columns it shares with strip b+1 at the
// END of a forward march and strip b+1 reads them at the START of one; with alternate strips mirrored both neighbours are at
// their common edge at the same time, and with the strip-major unit order mostly behind the same L2.
template <class Mdl, int S, bool FIRST>
__global__ void __launch_bounds__((RbpLayout<Mdl, S>::THREADS))
k_sor_rbp(SweepPlanes<Mdl> P, float *dout0, float *dout1, int nrows, int ncols, int TJ, int ntiles_r, int nunits, float omega,
          int col0, size_t frame_stride, int mirror_mode)
{
    using L = RbpLayout<Mdl, S>;
    constexpr int NIT = L::NIT, NRO = L::NRO, NCF = L::NCF, COL = L::COL, NW = L::NW;
    extern __shared__ __attribute__((aligned(16))) float rbp_lds[];
    float *const Kring = rbp_lds;
    float *const Qring = Kring + L::K_FLOATS;
    float *const Oring = Qring + L::Q_FLOATS;
    float *const Hring = Oring + L::O_FLOATS;

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6); // 0 .. S*NW-1: sweep waves; S*NW ..: loader(s)
    int unit = blockIdx.x;
    { // XCD-aware order: contiguous unit ranges per XCD (neighbouring strips share halo columns through one L2)
        const int nb = gridDim.x, per = nb >> 3;
        if (unit < (per << 3)) unit = (unit & 7) * per + (unit >> 3);
    }
    if (unit >= nunits) return;
    const size_t fo = (size_t)blockIdx.y * frame_stride;
    const int nstrips = nunits / ntiles_r;
    // Strip-major: the row tiles of one strip are consecutive units, i.e. (with the ranges above) run on ONE XCD and march in step.
    // A column piece of 256 rows starts 8 rows above the tile (32 bytes into a 128-byte line: nine lines for eight lines' worth)
    // and overlaps its vertical neighbours' by 16 rows; with the neighbours behind the same L2 those lines are fetched once.
    // (Horizontal neighbours share halo COLUMNS, but one reads them at the end of its march and the other at the start.)
#ifndef RBP_STRIP_MAJOR
#define RBP_STRIP_MAJOR 1
#endif
    const int a = RBP_STRIP_MAJOR ? unit % ntiles_r : unit / nstrips, b = RBP_STRIP_MAJOR ? unit / ntiles_r : unit % nstrips;
    const int r = a * RBP_OWN_ROWS - 8 + 4 * lane;
    int j0 = b * TJ, j1 = (j0 + TJ < ncols) ? j0 + TJ : ncols;
    const int mirror = mirror_mode == 2 ? 1 : (mirror_mode == 1 ? (b & 1) : 0);
    if (mirror) { // the strip, the colour of column 0 and the west / east planes in march coordinates (column x <-> ncols - 1 - x)
        const int t0 = ncols - j1;
        j1 = ncols - j0;
        j0 = t0;
        col0 += ncols - 1;
        const float *w = P.cf[Mdl::cWW];
        P.cf[Mdl::cWW] = P.cf[Mdl::cWE];
        P.cf[Mdl::cWE] = w;
    }
    const int nsteps = L::nsteps(TJ);
    // red column of sweep 0 in step t: x0(t) = xbase + t; sweep s: x0(t) - 3s.  Two warm-up steps fill the windows.
    const int xbase = j0 - L::HALO + 1 - 2;

    if (wave >= S * NW) {
        // =========================================== loader wave(s) ==========================================
        // group g = column xbase + 1 + g, consumed by sweep 0 in step g
        const int lw = wave - S * NW; // which loader
        const int rr = r < 0 ? 0 : (r > nrows - 4 ? nrows - 4 : r);
        int kslot = 0, qslot = 0, oslot = 0;
        auto issue = [&](int g) __attribute__((always_inline)) {
            const int y = xbase + 1 + g;
            const int cc = y < 0 ? 0 : (y > ncols - 1 ? ncols - 1 : y);
            const size_t off = fo + (size_t)(mirror ? ncols - 1 - cc : cc) * nrows + rr;
            float *kdst = Kring + (size_t)kslot * NCF * COL, *qdst = Qring + (size_t)qslot * NRO * COL;
#pragma unroll
            for (int f = 0; f < NCF; f++)
                if (L::NLOAD == 1 || f % L::NLOAD == lw)
                    __builtin_amdgcn_global_load_lds(RBP_GLB(P.cf[f] + off), RBP_LDS(kdst + f * COL), 16, 0, RBP_AUX_COEF);
#pragma unroll
            for (int f = 0; f < NRO; f++)
                if (L::NLOAD == 1 || (NCF + f) % L::NLOAD == lw)
                    __builtin_amdgcn_global_load_lds(RBP_GLB(P.ro[f] + off), RBP_LDS(qdst + f * COL), 16, 0, 0);
            float *odst = Oring + (size_t)oslot * NIT * COL;
#pragma unroll
            for (int f = 0; f < NIT; f++)
                if (L::NLOAD == 1 || (NCF + NRO + f) % L::NLOAD == lw)
                    __builtin_amdgcn_global_load_lds(RBP_GLB(P.it_in[f] + off), RBP_LDS(odst + f * COL), 16, 0, 0);
            kslot = (kslot + 1 == L::NK) ? 0 : kslot + 1;
            qslot = (qslot + 1 == L::NQ) ? 0 : qslot + 1;
            oslot = (oslot + 1 == L::NO) ? 0 : oslot + 1;
        };
        // all but the newest P-1 groups of THIS loader's pieces have landed (the count is an immediate: one wait per loader index)
        auto landed = [&]() __attribute__((always_inline)) {
            if (lw == 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((L::P - 1) * L::pieces_of(0)) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0)" ::"n"((L::P - 1) * L::pieces_of(L::NLOAD > 1 ? 1 : 0)) : "memory");
        };
        static_assert(L::NLOAD == 1 || L::NLOAD == 2, "one or two loader waves");
        for (int g = 0; g < L::P; g++) issue(g);
        landed(); // group 0 has landed
        rbp_barrier();
#ifndef RBP_LOADER_SLEEP
#define RBP_LOADER_SLEEP 0 /* A/B aid: s_sleep units (64 cycles) between the barrier and the step's DMA issue */
#endif
        for (int t = 0; t < nsteps; t++) {
#ifdef PDEIP_RBP_STAMPS
            const bool stamp_on = (blockIdx.x == 100) && (RBP_STAMP_SWEEP == 9) && (lw == 0) && (t >= RBP_STAMP_T0) && (t < RBP_STAMP_T0 + 24);
#endif
            RBP_STAMP(0);
            if (RBP_LOADER_SLEEP > 0) __builtin_amdgcn_s_sleep(RBP_LOADER_SLEEP);
            if (t + L::P < nsteps) { // group nsteps-1 is the last one a sweep wave reads (as "column x+1" of its last step)
                issue(t + L::P);
                RBP_STAMP(1); // the step's DMA instructions issued
                landed(); // group t+1 has landed
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // the tail: nothing new to fetch, everything issued has landed
            }
            RBP_STAMP(2);
            RBP_STAMP(3);
            RBP_STAMP(4);
            rbp_barrier();
            RBP_STAMP(5);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // nothing may land in LDS after the workgroup has gone
        return;
    }

    // ============================================ sweep waves ===============================================
    // Coupled two-field models run two waves per sweep, one per field (u is updated from the OLD v of the same pixel and vice
    // versa -- opticalflowSolvers.c:129-149 -- so the two halves of a half-sweep are independent); the workgroup then has two
    // waves per SIMD, which is what keeps the VALUs busy while a wave waits for LDS or the barrier.
    // Which wave plays which sweep: waves w, w+4, w+8 of a workgroup share a SIMD (the hardware deals them out cyclically).  Sweep 0
    // of a call's first launch is the heavy one (two exact divisions per pixel for the divisor planes), so it is paired with ONE
    // other sweep wave, and the loader (the last wave) sits with two plain ones.
#ifndef RBP_ROLE_REMAP
#define RBP_ROLE_REMAP 1
#endif
    if constexpr (NW == 2) {
        // wave 0..7 -> (sweep, field): (1,0) (0,0) (0,1) (1,1) (3,0) (2,0) (2,1) (3,1); SIMD classes {0,4,L} {1,5} {2,6} {3,7}
        const int q = wave & 3;
        const int s = (RBP_ROLE_REMAP && S == 4) ? ((wave >> 2) << 1) | ((q == 0 || q == 3) ? 1 : 0) : wave >> 1;
        const int fld = (RBP_ROLE_REMAP && S == 4) ? (wave >> 1) & 1 : wave & 1;
        if (fld == 0) rbp_sweep_wave<Mdl, S, FIRST, 0, 1>(P, dout0, dout1, Kring, Qring, Oring, Hring, s, lane, r, nrows, ncols, j0, j1, xbase, nsteps, omega, col0, fo, mirror);
        else rbp_sweep_wave<Mdl, S, FIRST, 1, 1>(P, dout0, dout1, Kring, Qring, Oring, Hring, s, lane, r, nrows, ncols, j0, j1, xbase, nsteps, omega, col0, fo, mirror);
    } else {
        const int s = (RBP_ROLE_REMAP && wave < 2) ? wave ^ 1 : wave; // the loader (wave S) shares a SIMD with wave 0: sweep 1, not sweep 0
        rbp_sweep_wave<Mdl, S, FIRST, 0, NIT>(P, dout0, dout1, Kring, Qring, Oring, Hring, s, lane, r, nrows, ncols, j0, j1, xbase, nsteps, omega, col0, fo, mirror);
    }
}

} // namespace pdeip
