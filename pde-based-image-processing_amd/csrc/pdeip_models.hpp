// pdeip_models.hpp -- per-pixel relaxation arithmetic of each point solver.
//
// One "model" per reference solver.  Both sweep orderings (exact-order tile wavefront and
// red-black register marching) call the same Model::update(), so the two kernels cannot
// drift apart arithmetically.  Every expression keeps the reference's association; the
// translation unit is compiled with -ffp-contract=off so no FMA is formed, and HIP's
// default correctly-rounded f32 divide/sqrt matches x86 SSE2.
//
// A model describes:
//   NIT  iterate fields (updated in place; neighbour access)          e.g. U,V / dU,dV / dU / X
//   NRO  read-only fields that also need neighbour access             e.g. U,V of the llin solvers
//   NCF  coefficient planes read at the centre pixel only
// Divisor planes are precomputed by a prologue kernel exactly as the reference builds
// them during its first sweep (opticalflowSolvers.c:111-127 etc.).
#pragma once
#include <hip/hip_runtime.h>

namespace pdeip {

__device__ __forceinline__ bool is_nan(float x) { return x != x; }

template <int N> struct at_least_one { static constexpr int value = N > 0 ? N : 1; };

// Reciprocals of the divisor planes (Model::derive).  RcpIeee is the reference's `1.0f / d` (HIP's correctly rounded division:
// two v_div_scale, v_rcp, six fused steps, v_div_fmas, v_div_fixup).  RcpFast = v_rcp_f32 + ONE Newton step with two explicit
// FMAs: bit-identical to the IEEE quotient for every normal d whose reciprocal is normal (exponent field 1..252, either sign) --
// all 4 227 858 432 such inputs compared on gfx950 (tools/rcp_probe.hip; tests/test_gpu_rcp.py repeats it through the library).
// RcpRange only records whether its arguments are inside that range (NaN, zero, infinities and denormals are not); a caller
// runs derive() with it first and takes RcpIeee for the whole wave when any lane is outside.
struct RcpIeee {
    __device__ __forceinline__ float operator()(float d) const { return 1.0f / d; }
};
struct RcpFast {
    __device__ __forceinline__ float operator()(float d) const
    {
        const float r = __builtin_amdgcn_rcpf(d);
        const float e = __builtin_fmaf(-d, r, 1.0f);
        return __builtin_fmaf(e, r, r);
    }
};
struct RcpRange {
    bool &ok;
    __device__ __forceinline__ float operator()(float d) const
    {
        const float a = __builtin_fabsf(d);
        ok = ok && (a >= 0x1p-126f) && (a < 0x1p126f);
        return d;
    }
};

// Value of the neighbouring lane in one VALU instruction (DPP wave_shr:1 / wave_shl:1 across the whole
// 64-lane wave); the first / last lane keeps its own value, like __shfl_up(v, 1) / __shfl_down(v, 1) --
// which compile to an LDS-crossbar ds_bpermute with a round trip that nothing hides at one wave per SIMD.
__device__ __forceinline__ float lane_above(float v) // lane l <- lane l-1
{
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), 0x138, 0xf, 0xf, false));
}
__device__ __forceinline__ float lane_below(float v) // lane l <- lane l+1
{
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), 0x130, 0xf, 0xf, false));
}

// ---- Horn-Schunck / early linearization: GS_SOR_elin4_2d (opticalflowSolvers.c:41-186)
struct ModelElin4 {
    static constexpr int NIT = 2, NRO = 0, NCF = 9;
    enum { cM = 0, cCu, cCv, cDivU, cDivV, cWW, cWN, cWE, cWS };
    static constexpr int D0 = cDivU, D1 = cDivV; // derived planes; their slots hold Du, Dv before derive()
    static constexpr bool DERIVE_WE_SYMMETRIC = true; // derive() gives the same bits with wW and wE exchanged ((wW + wE) + ...)
    // opticalflowSolvers.c:111-127: the divisors the reference builds during its first sweep
    // WHICH: bit 0 = the divisor of u, bit 1 = the divisor of v (a wave that relaxes one field derives that field's only)
    template <int WHICH = 3, class R = RcpIeee> __device__ __forceinline__ static void derive(float (&k)[9], R rcp = R())
    {
        float t1 = k[cWW] + k[cWE];
        const float t2 = k[cWN] + k[cWS];
        t1 += t2;
        const float du = k[cDivU], dv = k[cDivV];
        if (WHICH & 1) k[cDivU] = rcp(is_nan(du) ? t1 : t1 + du);
        if (WHICH & 2) k[cDivV] = rcp(is_nan(dv) ? t1 : t1 + dv);
    }
    __device__ __forceinline__ static void update(float (&c)[2], const float (&W)[2],
                                                  const float (&E)[2], const float (&N)[2],
                                                  const float (&S)[2], const float (&)[1],
                                                  const float (&)[1], const float (&)[1],
                                                  const float (&)[1], const float (&)[1],
                                                  const float (&cf)[9], float omega, float om1)
    {
        // :89-108
        float nbU = W[0] * cf[cWW];
        float t1 = E[0] * cf[cWE];
        nbU += t1;
        float t2 = N[0] * cf[cWN];
        float t3 = S[0] * cf[cWS];
        t2 += t3;
        nbU += t2;
        float nbV = W[1] * cf[cWW];
        t1 = E[1] * cf[cWE];
        nbV += t1;
        t2 = N[1] * cf[cWN];
        t3 = S[1] * cf[cWS];
        t2 += t3;
        nbV += t2;
        // :129-149 (both use the pre-update centre values)
        float a = nbU + cf[cCu];
        float b = cf[cM] * c[1];
        a = a - b;
        float Unew = is_nan(cf[cCu]) ? nbU * cf[cDivU] : a * cf[cDivU];
        a = nbV + cf[cCv];
        b = cf[cM] * c[0];
        a = a - b;
        float Vnew = is_nan(cf[cCv]) ? nbV * cf[cDivV] : a * cf[cDivV];
        // :151-152
        t1 = om1 * c[0];
        t2 = omega * Unew;
        c[0] = t1 + t2;
        t1 = om1 * c[1];
        t2 = omega * Vnew;
        c[1] = t1 + t2;
    }
};

// ---- late linearization: GS_SOR_llin4_2d (:504-680) == GS_SOR_llin8_2d point path (:1487-1667)
struct ModelLlin4 {
    static constexpr int NIT = 2, NRO = 2, NCF = 9;
    enum { cM = 0, cCu, cCv, cDivU, cDivV, cWW, cWN, cWE, cWS };
    static constexpr int D0 = cDivU, D1 = cDivV;
    static constexpr bool DERIVE_WE_SYMMETRIC = true;
    template <int WHICH = 3, class R = RcpIeee> __device__ __forceinline__ static void derive(float (&k)[9], R rcp = R()) { ModelElin4::derive<WHICH, R>(k, rcp); } // :606-622
    __device__ __forceinline__ static float neigh(float dW, float dE, float dN, float dS, float uW,
                                                  float uE, float uN, float uS, float uc,
                                                  const float (&cf)[9])
    {
        // :563-580
        float a = dW + uW, b = dE + uE, c = dN + uN, d = dS + uS;
        a -= uc;
        b -= uc;
        c -= uc;
        d -= uc;
        a *= cf[cWW];
        b *= cf[cWE];
        c *= cf[cWN];
        d *= cf[cWS];
        a += b;
        c += d;
        a += c;
        return a;
    }
    __device__ __forceinline__ static void update(float (&c)[2], const float (&W)[2],
                                                  const float (&E)[2], const float (&N)[2],
                                                  const float (&S)[2], const float (&rc)[2],
                                                  const float (&rW)[2], const float (&rE)[2],
                                                  const float (&rN)[2], const float (&rS)[2],
                                                  const float (&cf)[9], float omega, float om1)
    {
        float nbU = neigh(W[0], E[0], N[0], S[0], rW[0], rE[0], rN[0], rS[0], rc[0], cf);
        float nbV = neigh(W[1], E[1], N[1], S[1], rW[1], rE[1], rN[1], rS[1], rc[1], cf);
        // :624-644
        float a = nbU + cf[cCu];
        float b = cf[cM] * c[1];
        a -= b;
        float dUnew = is_nan(cf[cCu]) ? nbU * cf[cDivU] : a * cf[cDivU];
        a = nbV + cf[cCv];
        b = cf[cM] * c[0];
        a -= b;
        float dVnew = is_nan(cf[cCv]) ? nbV * cf[cDivV] : a * cf[cDivV];
        // :646-647
        float t1 = om1 * c[0];
        float t2 = omega * dUnew;
        c[0] = t1 + t2;
        t1 = om1 * c[1];
        t2 = omega * dVnew;
        c[1] = t1 + t2;
    }
};

// ---- disparity: GS_SOR_llin4_2d (disparitySolvers.c:41-144)
struct ModelDisp4 {
    static constexpr int NIT = 1, NRO = 1, NCF = 6;
    enum { cDividend = 0, cDiv, cWW, cWN, cWE, cWS };
    static constexpr int D0 = cDividend, D1 = cDiv; // slots hold Cu, Du before derive()
    static constexpr bool DERIVE_WE_SYMMETRIC = false; // ((Du + wE) + wW) + ...: the order of wE and wW matters
    // disparitySolvers.c:94-113
    template <int WHICH = 3, class R = RcpIeee> __device__ __forceinline__ static void derive(float (&k)[6], R rcp = R())
    {
        const float cu = k[cDividend];
        const bool ok = !is_nan(cu);
        float t = ok ? k[cDiv] + k[cWE] : k[cWE];
        t = t + k[cWW];
        t = t + k[cWS];
        t = t + k[cWN];
        k[cDividend] = ok ? cu : 0.0f;
        k[cDiv] = rcp(t);
    }
    __device__ __forceinline__ static float neigh(float dW, float dE, float dN, float dS, float uW,
                                                  float uE, float uN, float uS, float uc,
                                                  const float (&cf)[6])
    {
        // :89-92, summed left to right in the order E, W, S, N
        float e = ((uE + dE) - uc) * cf[cWE];
        float w = ((uW + dW) - uc) * cf[cWW];
        float s = ((uS + dS) - uc) * cf[cWS];
        float n = ((uN + dN) - uc) * cf[cWN];
        float r = e + w;
        r = r + s;
        r = r + n;
        return r;
    }
    __device__ __forceinline__ static void update(float (&c)[1], const float (&W)[1],
                                                  const float (&E)[1], const float (&N)[1],
                                                  const float (&S)[1], const float (&rc)[1],
                                                  const float (&rW)[1], const float (&rE)[1],
                                                  const float (&rN)[1], const float (&rS)[1],
                                                  const float (&cf)[6], float omega, float om1)
    {
        float nb = neigh(W[0], E[0], N[0], S[0], rW[0], rE[0], rN[0], rS[0], rc[0], cf);
        // :116-118
        float A = om1 * c[0];
        float B = omega * (nb + cf[cDividend]);
        B = B * cf[cDiv];
        c[0] = A + B;
    }
};

// ---- symmetric stereo: GS_SOR_llinsym4_2d (disparitySolvers.c:301-460) relaxes two disparity fields that do not
// read each other; per field it is the solver above except for the association of the last line:
// omega * ((nb + dividend) * div) instead of (omega * (nb + dividend)) * div (:425-429 vs :116-118).
struct ModelDispSym4 : ModelDisp4 {
    __device__ __forceinline__ static void update(float (&c)[1], const float (&W)[1],
                                                  const float (&E)[1], const float (&N)[1],
                                                  const float (&S)[1], const float (&rc)[1],
                                                  const float (&rW)[1], const float (&rE)[1],
                                                  const float (&rN)[1], const float (&rS)[1],
                                                  const float (&cf)[6], float omega, float om1)
    {
        const float nb = neigh(W[0], E[0], N[0], S[0], rW[0], rE[0], rN[0], rS[0], rc[0], cf);
        float approx = nb + cf[cDividend];
        approx = approx * cf[cDiv];
        const float A = om1 * c[0];
        const float B = omega * approx;
        c[0] = A + B;
    }
};

// ---- scalar PDE, 4 neighbours: GS_SOR_4_2d (pdeSolvers.c:44-146)
struct ModelPde4 {
    static constexpr int NIT = 1, NRO = 0, NCF = 6;
    enum { cB = 0, cInv, cWW, cWN, cWE, cWS };
    static constexpr int D0 = cB, D1 = cInv; // slots hold B, TRACE before derive()
    static constexpr bool DERIVE_WE_SYMMETRIC = true;
    // pdeSolvers.c:99-115
    template <int WHICH = 3, class R = RcpIeee> __device__ __forceinline__ static void derive(float (&k)[6], R rcp = R())
    {
        const float tr = k[cInv];
        float t = k[cWE] + k[cWW];
        t += k[cWS] + k[cWN];
        const bool ok = !is_nan(tr);
        k[cInv] = rcp(ok ? tr : t);
        k[cB] = ok ? k[cB] : 0.0f;
    }
    __device__ __forceinline__ static void update(float (&c)[1], const float (&W)[1],
                                                  const float (&E)[1], const float (&N)[1],
                                                  const float (&S)[1], const float (&)[1],
                                                  const float (&)[1], const float (&)[1],
                                                  const float (&)[1], const float (&)[1],
                                                  const float (&cf)[6], float omega, float om1)
    {
        // :94-97
        float nb = E[0] * cf[cWE] + W[0] * cf[cWW];
        nb += S[0] * cf[cWS] + N[0] * cf[cWN];
        // :117-118
        float x = om1 * c[0];
        float t = omega * (cf[cB] + nb);
        t = t * cf[cInv];
        c[0] = x + t;
    }
};

// ---- scalar PDE, 8 neighbours: GS_SOR_8_2d (pdeSolvers.c:153-268); own kernels (9-point)
struct ModelPde8 {
    enum { cB = 0, cInv, cWW, cWNW, cWN, cWNE, cWE, cWSE, cWS, cWSW, NCF };
    __device__ __forceinline__ static float update(float xc, float xW, float xE, float xN, float xS,
                                                   float xNW, float xNE, float xSW, float xSE,
                                                   const float (&cf)[NCF], float omega, float om1)
    {
        // :208-215
        float nb = xE * cf[cWE] + xW * cf[cWW];
        nb += xS * cf[cWS] + xN * cf[cWN];
        nb += xSW * cf[cWSW] + xNW * cf[cWNW];
        nb += xSE * cf[cWSE] + xNE * cf[cWNE];
        // :239-240
        float x = om1 * xc;
        float t = omega * (cf[cB] + nb);
        t = t * cf[cInv];
        return x + t;
    }
};

// Plane pointers handed to a sweep kernel.  `it_in`/`it_out` are the same buffers for
// the in-place exact-order kernel and the ping-pong pair for the red-black kernel.
template <class Mdl> struct SweepPlanes {
    const float *it_in[Mdl::NIT];
    float *it_out[Mdl::NIT];
    const float *ro[at_least_one<Mdl::NRO>::value];
    const float *cf[Mdl::NCF];
};

} // namespace pdeip
