// pdeip_drivers.hip -- libpdeip.so: the eight drivers runme.m calls, as host-pointer entry points that keep a whole coarse-to-fine
// run resident on the device.
//
//   [U V] = FlowEminND_llin_2D_v10(Iin, channels, fstTerm, sndTerm, ...)   matlab/optical_flow/FlowEminND_llin_2D_v10.m:52-369       pdeip_flow_nd_llin
//   [U V] = FlowEminAD_llin_2D_v10(...)                                     matlab/optical_flow/FlowEminAD_llin_2D_v10.m:52-383       pdeip_flow_ad_llin
//   [U V] = FlowEminHS_elin_2D_v10(Iin, channels, ...)                      matlab/optical_flow/FlowEminHS_elin_2D_v10.m:52-200       pdeip_flow_hs_elin
//   [U V] = FlowEminNDFASFMG_elin_2D_v10(Iin, channels, ...)                matlab/optical_flow/FlowEminNDFASFMG_elin_2D_v10.m:53-273 pdeip_flow_fas_fmg_elin
//   U     = DispEminND_llin_2D(Il, Ir, fstTerm, sndTerm, ...)              matlab/disparity/DispEminND_llin_2D.m:51-316              pdeip_disp_nd_llin
//   U     = DispEminND_llin_sym_2D(Il, Ir, ...)                            matlab/disparity/DispEminND_llin_sym_2D.m:51-275          pdeip_disp_nd_llin_sym
//   Iout  = TVdenoise8(I_in, ...) / TVdenoise4(I_in, ...)                  matlab/denoising/TVdenoise{8,4}.m                         pdeip_tvdenoise8 / 4
//
// A MATLAB session that only swaps the MEX gateways pays 13-17 planes of PCIe traffic per solver call (7.8 ms per 4K call,
// INTEGRATION.md); calling these instead moves the frames up once and the result down once.  The level loops below are host
// control flow around the library's own device entry points -- the same `_dev` stage kernels, in the same order and with
// the same arguments, as the Python drivers (drivers.py / flow_level.py / fas.py / pyramid.py), which the tests compare with bit
// for bit.  The pyramid's imresize / imfilter / fspecial are the definitions of pyramid.py (csrc/pdeip_pyr.hpp), there
// being no Image Processing Toolbox to compare with.  Ordering and solver as everywhere: pdeip_set_mode / PDEIP_MODE,
// param.solver.
#include "pdeip_ctx.hpp"

#include <cmath>
#include <utility>
#include <vector>

using namespace pdeip;

namespace {

__global__ void k_scale(float *out, const float *in, size_t n, float s, int divide)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = divide ? in[i] / s : in[i] * s; // a true division where MATLAB divides (Iin ./ 255)
}

__global__ void k_fill(float *out, size_t n, float v)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = v;
}

// One run: a stream, a device arena and a dry mode.  The arena is one cached workspace buffer per device, bump-allocated;
// a run is played twice -- dry (nothing is launched: the sizes only depend on the frame's dimensions) to learn how much it
// needs, then for real.
struct Run {
    hipStream_t s = nullptr;
    bool dry = true;
    int rc = PDEIP_OK, mode = PDEIP_MODE_EXACT_ORDER;
    char *base = nullptr;
    size_t used = 0, peak = 0;
    void *take(size_t bytes)
    {
        const size_t at = (used + 255) & ~(size_t)255;
        used = at + bytes;
        if (used > peak) peak = used;
        return base + at; // dry: base is null and nothing dereferences it
    }
    float *planes(int nrows, int ncols, int frames = 1) { return static_cast<float *>(take(sizeof(float) * (size_t)nrows * ncols * frames)); }
    double *dplane(int nrows, int ncols) { return static_cast<double *>(take(sizeof(double) * (size_t)nrows * ncols)); }
    size_t mark() const { return used; }
    void release(size_t m) { used = m; }
};
#define DO(R, expr)                                                \
    do {                                                           \
        if (!(R).dry && (R).rc == PDEIP_OK) (R).rc = (expr);       \
    } while (0)
#define DOHIP(R, expr)                                                                                                      \
    do {                                                                                                                    \
        if (!(R).dry && (R).rc == PDEIP_OK && (expr) != hipSuccess) (R).rc = set_err(PDEIP_ERR_DEVICE, "%s failed", #expr); \
    } while (0)

void scale(Run &R, float *out, const float *in, size_t n, float s, bool divide)
{
    if (R.dry || R.rc != PDEIP_OK) return;
    hipLaunchKernelGGL(k_scale, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, R.s, out, in, n, s, divide ? 1 : 0);
}

struct Params { // the drivers' parameter structs, defaults filled in
    double alpha, omega, gammaS, b1, b2, scl_factor;
    int firstLoop, secondLoop, iter, solver, scales;
};
Params merge(const pdeip_driver_params *u, const Params &d)
{
    Params p = d;
    if (u == nullptr) return p;
    auto D = [](double v, double dflt) { return (v > 0.0) ? v : dflt; }; // <= 0 or NaN: the driver's default
    auto I = [](int v, int dflt) { return v > 0 ? v : dflt; };
    p.alpha = D(u->alpha, d.alpha);
    p.omega = D(u->omega, d.omega);
    p.gammaS = D(u->gammaS, d.gammaS);
    p.b1 = D(u->b1, d.b1);
    p.b2 = D(u->b2, d.b2);
    p.scl_factor = D(u->scl_factor, d.scl_factor);
    p.firstLoop = I(u->firstLoop, d.firstLoop);
    p.secondLoop = I(u->secondLoop, d.secondLoop);
    p.iter = I(u->iter, d.iter);
    p.solver = I(u->solver, d.solver);
    p.scales = I(u->scales, d.scales);
    return p;
}

// fspecial('gaussian', [size size], sigma) as pyramid.py states it (row-major doubles), before the division by the sum
std::vector<double> gaussian(int size, double sigma)
{
    const int r = size / 2;
    std::vector<double> g((size_t)size * size);
    for (int a = -r; a <= r; a++)
        for (int b = -r; b <= r; b++) g[(size_t)(a + r) * size + (b + r)] = std::exp(-((double)(a * a) + (double)(b * b)) / (2.0 * sigma * sigma));
    return g;
}
// numpy.ndarray.sum() of a contiguous float64 array of n < 128 elements: eight running sums r[0..7] over i = 0, 8, 16, ... then
// ((r0 + r1) + (r2 + r3)) + ((r4 + r5) + (r6 + r7)), then the tail added one by one (numpy/core/src/umath/loops_utils.h.src,
// pairwise_sum).  The Gaussian mask is divided by this sum; pyramid.py (numpy) is the definition the tests compare with.
double numpy_sum(const std::vector<double> &a)
{
    const size_t n = a.size();
    if (n < 8) {
        double r = 0.0; // numpy: res = 0.; res += a[i] (with a -0.0 start that does not matter for positive terms)
        for (size_t i = 0; i < n; i++) r += a[i];
        return r;
    }
    double r[8];
    for (int k = 0; k < 8; k++) r[k] = a[k];
    size_t i = 8;
    for (; i + 8 <= n; i += 8)
        for (int k = 0; k < 8; k++) r[k] += a[i + k];
    double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    for (; i < n; i++) res += a[i];
    return res;
}
std::vector<double> gaussian_mask(int size, double sigma)
{
    std::vector<double> g = gaussian(size, sigma);
    const double s = numpy_sum(g);
    for (double &v : g) v /= s;
    return g;
}

// imresize(A, [out_rows out_cols], 'bilinear') of a DOUBLE array on the host, as pyramid.resize(..., out_dtype=float64) does it:
// triangle kernel at MATLAB's pixel-centre convention, stretched by 1/scale when shrinking, taps normalised, rows first, then
// columns, every sum left to right.  A: column-major [nrows x ncols].  (The a-priori fields only: a few small planes per run.)
struct Taps {
    int T = 0;
    std::vector<int> idx;   // [n_out][T], clamped
    std::vector<double> w;  // [n_out][T], normalised
};
Taps resize_taps(int n_in, int n_out)
{
    Taps t;
    const double scale = (double)n_out / (double)n_in;
    const double stretch = scale >= 1.0 ? 1.0 : 1.0 / scale;
    const double width = 1.0 * stretch;
    t.T = (int)std::ceil(2.0 * width) + 2;
    t.idx.resize((size_t)n_out * t.T);
    t.w.resize((size_t)n_out * t.T);
    for (int o = 0; o < n_out; o++) {
        const double x = ((double)o + 0.5) / scale - 0.5;
        const long first = (long)std::floor(x - width);
        double total = 0.0;
        for (int k = 0; k < t.T; k++) {
            const long id = first + k;
            const double a = std::fabs(((double)id - x) / stretch);
            const double wk = (1.0 - a) > 0.0 ? (1.0 - a) : 0.0;
            t.w[(size_t)o * t.T + k] = wk;
            total = k == 0 ? wk : total + wk;
            t.idx[(size_t)o * t.T + k] = (int)(id < 0 ? 0 : (id > n_in - 1 ? n_in - 1 : id));
        }
        for (int k = 0; k < t.T; k++) t.w[(size_t)o * t.T + k] /= total;
    }
    return t;
}
std::vector<double> resize_double(const std::vector<double> &A, int nrows, int ncols, int out_rows, int out_cols)
{
    const Taps r = resize_taps(nrows, out_rows), c = resize_taps(ncols, out_cols);
    std::vector<double> T1((size_t)out_rows * ncols), out((size_t)out_rows * out_cols);
    for (int j = 0; j < ncols; j++)
        for (int o = 0; o < out_rows; o++) {
            double acc = r.w[(size_t)o * r.T] * A[(size_t)j * nrows + r.idx[(size_t)o * r.T]];
            for (int k = 1; k < r.T; k++) acc = acc + r.w[(size_t)o * r.T + k] * A[(size_t)j * nrows + r.idx[(size_t)o * r.T + k]];
            T1[(size_t)j * out_rows + o] = acc;
        }
    for (int o2 = 0; o2 < out_cols; o2++)
        for (int o = 0; o < out_rows; o++) {
            double acc = c.w[(size_t)o2 * c.T] * T1[(size_t)c.idx[(size_t)o2 * c.T] * out_rows + o];
            for (int k = 1; k < c.T; k++) acc = acc + c.w[(size_t)o2 * c.T + k] * T1[(size_t)c.idx[(size_t)o2 * c.T + k] * out_rows + o];
            out[(size_t)o2 * out_rows + o] = acc;
        }
    return out;
}

struct Level {
    int nr = 0, nc = 0;
    float *I0 = nullptr, *I1 = nullptr; // [C] planes each
    double *Us = nullptr, *Vs = nullptr; // the a-priori fields of the scale on the device (or null)
};

// The image pyramid of both frames (:100-127 / DispEminND_llin_2D.m:77-104): resize to ceil(size * scl_factor), the level
// just left is smoothed after it has been resized; downscaling stops at min_size (then the last scale is smoothed too) or at
// param.scales (then it is not: the loop `for scl=2:param.scales` simply ends).
std::vector<Level> build_pyramid(Run &R, float *f0, float *f1, int nrows, int ncols, int C, const Params &p, int min_size, const std::vector<double> &G, int gsize,
                                 bool smooth_last = true)
{
    std::vector<Level> L(1);
    L[0].nr = nrows;
    L[0].nc = ncols;
    L[0].I0 = f0;
    L[0].I1 = f1;
    while ((int)L.size() < p.scales) {
        const Level cur = L.back();
        Level nx;
        nx.nr = (int)std::ceil(cur.nr * p.scl_factor);
        nx.nc = (int)std::ceil(cur.nc * p.scl_factor);
        nx.I0 = R.planes(nx.nr, nx.nc, C);
        nx.I1 = R.planes(nx.nr, nx.nc, C);
        DO(R, pdeip_pyr_resize_dev(R.s, cur.I0, cur.nr, cur.nc, C, nx.nr, nx.nc, 0, nx.I0));
        DO(R, pdeip_pyr_resize_dev(R.s, cur.I1, cur.nr, cur.nc, C, nx.nr, nx.nc, 0, nx.I1));
        float *s0 = R.planes(cur.nr, cur.nc, C), *s1 = R.planes(cur.nr, cur.nc, C);
        DO(R, pdeip_pyr_smooth_dev(R.s, cur.I0, cur.nr, cur.nc, C, G.data(), gsize, s0));
        DO(R, pdeip_pyr_smooth_dev(R.s, cur.I1, cur.nr, cur.nc, C, G.data(), gsize, s1));
        L.back().I0 = s0;
        L.back().I1 = s1;
        L.push_back(nx);
        if (nx.nr <= min_size || nx.nc <= min_size) {
            if (!smooth_last) break; // DispEminND_llin_sym_2D.m:94-98 leaves its coarsest scale as resized
            float *t0 = R.planes(nx.nr, nx.nc, C), *t1 = R.planes(nx.nr, nx.nc, C);
            DO(R, pdeip_pyr_smooth_dev(R.s, nx.I0, nx.nr, nx.nc, C, G.data(), gsize, t0));
            DO(R, pdeip_pyr_smooth_dev(R.s, nx.I1, nx.nr, nx.nc, C, G.data(), gsize, t1));
            L.back().I0 = t0;
            L.back().I1 = t1;
            break;
        }
    }
    return L;
}

// param.Us / param.Vs down the pyramid (:162-184): NaN -> 0, USap{scl} = imresize(USap{scl-1} .* scl_factor, scl_factor), doubles;
// every scale's field goes to the device as a double plane.  The coarsest one also starts the flow (:178, :183) -- MATLAB
// keeps that double array through the first firstLoop of the coarsest scale (`u_double` below); here, as in drivers.py, `start`
// receives its float32 rounding.
void apriori_pyramid(Run &R, const double *Us_host, std::vector<Level> &L, double scl_factor, bool second, float *start)
{
    if (Us_host == nullptr) return;
    std::vector<double> cur;
    for (size_t s = 0; s < L.size(); s++) {
        double *d = R.dplane(L[s].nr, L[s].nc);
        (second ? L[s].Vs : L[s].Us) = d;
        if (R.dry || R.rc != PDEIP_OK) continue;
        if (s == 0) {
            cur.assign(Us_host, Us_host + (size_t)L[0].nr * L[0].nc);
            for (double &v : cur)
                if (v != v) v = 0.0;
        } else {
            std::vector<double> scaled(cur.size());
            for (size_t i = 0; i < cur.size(); i++) scaled[i] = cur[i] * scl_factor;
            cur = resize_double(scaled, L[s - 1].nr, L[s - 1].nc, L[s].nr, L[s].nc);
        }
        // `cur` is rewritten for the next scale: the copy must have left it
        DOHIP(R, hipMemcpyAsync(d, cur.data(), cur.size() * sizeof(double), hipMemcpyHostToDevice, R.s));
        DOHIP(R, hipStreamSynchronize(R.s));
    }
    if (R.dry || R.rc != PDEIP_OK) return;
    std::vector<float> f(cur.size());
    for (size_t i = 0; i < cur.size(); i++) f[i] = (float)cur[i];
    DOHIP(R, hipMemcpyAsync(start, f.data(), f.size() * sizeof(float), hipMemcpyHostToDevice, R.s));
    DOHIP(R, hipStreamSynchronize(R.s));
}

// One scale of the flow driver: firstLoop x [warp, derivatives, secondLoop x (robust assembly + OPdiffWeights, a-priori terms,
// Oflow_sor_llin4_2d), median] -- FlowEminND_llin_2D_v10.m:208-356, flow_level.py FlowLlinLevel.run.  The flow enters in
// (U, V) and leaves in the pair returned through them; (Ua, Va) is the other plane pair of the ping-pong.
void flow_level(Run &R, const Params &p, const Level &lv, const float *I1t0, const float *I1t1, int C1, const float *I2t0, const float *I2t1, int C2,
                bool gradmag, float *&U, float *&V, float *&Ua, float *&Va, double as_diff, bool u_double)
{
    const int nr = lv.nr, nc = lv.nc;
    const size_t n = (size_t)nr * nc;
    const size_t m = R.mark();
    float *w1 = R.planes(nr, nc, C1), *d1[3], *w2 = nullptr, *d2[5] = {};
    for (auto &q : d1) q = R.planes(nr, nc, C1);
    const int nd2 = C2 > 0 ? (gradmag ? 5 : 3) : 0;
    if (C2 > 0) {
        w2 = R.planes(nr, nc, C2);
        for (int k = 0; k < nd2; k++) d2[k] = R.planes(nr, nc, C2);
    }
    float *coef[9]; // MGd, CuGd, CvGd, DuGd, DvGd, wW, wN, wE, wS
    for (auto &q : coef) q = R.planes(nr, nc);
    float *dUV = R.planes(nr, nc, 2), *dU = dUV, *dV = dUV + n;
    for (int first = 0; first < p.firstLoop; first++) {
        DO(R, pdeip_flow_warp_dev(R.s, U, V, I1t1, C1, C2 > 0 ? I2t1 : nullptr, C2, nr, nc, w1, w2));
        DO(R, pdeip_fst_derivatives5_dev(R.s, I1t0, w1, nr, nc, C1, d1[0], d1[1], d1[2]));
        if (C2 > 0) {
            if (gradmag) DO(R, pdeip_snd_derivatives5_dev(R.s, I2t0, w2, nr, nc, C2, d2[0], d2[1], d2[2], d2[3], d2[4]));
            else DO(R, pdeip_fst_derivatives5_dev(R.s, I2t0, w2, nr, nc, C2, d2[0], d2[1], d2[2]));
        }
        DOHIP(R, hipMemsetAsync(dUV, 0, 2 * n * sizeof(float), R.s));
        for (int k = 0; k < p.secondLoop; k++) {
            // weights come back in wW wN wS wE order (OPdiffWeights, :389-433)
            DO(R, pdeip_flow_assemble_weights_dev(R.s, d1[0], d1[1], d1[2], C1, (float)p.b1, d2[0], d2[1], d2[2], d2[3], d2[4], C2, (float)p.b2, U, V, dU, dV,
                                                  (float)p.alpha, nr, nc, coef[0], coef[1], coef[2], coef[3], coef[4], coef[5], coef[6], coef[8], coef[7]));
            if (lv.Us) DO(R, pdeip_flow_apriori_dev(R.s, lv.Us, U, dU, p.gammaS, p.alpha, as_diff, u_double && first == 0, k == 0, nr, nc, coef[1], coef[3]));
            if (lv.Vs) DO(R, pdeip_flow_apriori_dev(R.s, lv.Vs, V, dV, p.gammaS, p.alpha, as_diff, u_double && first == 0, k == 0, nr, nc, coef[2], coef[4]));
            if (p.solver == PDEIP_SOLVER_SOR)
                DO(R, pdeip_oflow_sor_llin4_dev(R.s, U, V, dU, dV, coef[0], coef[1], coef[2], coef[3], coef[4], coef[5], coef[6], coef[7], coef[8], nr, nc, p.iter,
                                                (float)p.omega, R.mode, 0));
            else
                DO(R, pdeip_oflow_alr_llin4_dev(R.s, U, V, dU, dV, coef[0], coef[1], coef[2], coef[3], coef[4], coef[5], coef[6], coef[7], coef[8], nr, nc, p.iter,
                                                (float)p.omega, R.mode));
        }
        DO(R, pdeip_median3_pair_dev(R.s, U, dU, V, dV, nr, nc, Ua, Va));
        std::swap(U, Ua);
        std::swap(V, Va);
    }
    R.release(m);
}

// The anisotropic-diffusion twin: FlowEminAD_llin_2D_v10.m:198-366, flow_level.py FlowAdLevel.run -- eight ADdiffWeights from the
// image (`flow_diffusion` false: once per level, from frame 0 of the scale) or from U+dU+V+dV (true: every inner iteration), and
// Oflow_sor_llin8_2d, whose point solver runs the 4-neighbour arithmetic on W, N, E, S (opticalflowSolvers.c:1487).
void flow_ad_level(Run &R, const Params &p, const Level &lv, int C, const float *I1t0, const float *I1t1, int C1, const float *I2t0, const float *I2t1, int C2,
                   bool gradmag, double quantile, bool flow_diffusion, float *&U, float *&V, float *&Ua, float *&Va, double as_diff, bool u_double)
{
    const int nr = lv.nr, nc = lv.nc;
    const size_t n = (size_t)nr * nc;
    const size_t m = R.mark();
    float *X = R.planes(nr, nc), *Y = R.planes(nr, nc), *S = R.planes(nr, nc);
    float *w1 = R.planes(nr, nc, C1), *d1[3], *w2 = nullptr, *d2[5] = {};
    for (auto &q : d1) q = R.planes(nr, nc, C1);
    if (C2 > 0) {
        w2 = R.planes(nr, nc, C2);
        for (int k = 0; k < (gradmag ? 5 : 3); k++) d2[k] = R.planes(nr, nc, C2);
    }
    float *coef[5], *w8[8], *dU = R.planes(nr, nc), *dV = R.planes(nr, nc); // MGd, CuGd, CvGd, DuGd, DvGd | wW, wNW, wN, wNE, wE, wSE, wS, wSW
    for (auto &q : coef) q = R.planes(nr, nc);
    for (auto &q : w8) q = R.planes(nr, nc);
    if (!flow_diffusion) DO(R, pdeip_ad_weights_dev(R.s, lv.I0, nr, nc, C, quantile, w8[0], w8[1], w8[2], w8[3], w8[4], w8[5], w8[6], w8[7]));
    for (int first = 0; first < p.firstLoop; first++) {
        DO(R, pdeip_flow_coords_dev(R.s, U, V, nr, nc, X, Y));
        DO(R, pdeip_warp_bilinear_dev(R.s, I1t1, X, Y, nr, nc, C1, w1));
        DO(R, pdeip_fst_derivatives5_dev(R.s, I1t0, w1, nr, nc, C1, d1[0], d1[1], d1[2]));
        if (C2 > 0) {
            DO(R, pdeip_warp_bilinear_dev(R.s, I2t1, X, Y, nr, nc, C2, w2));
            if (gradmag) DO(R, pdeip_snd_derivatives5_dev(R.s, I2t0, w2, nr, nc, C2, d2[0], d2[1], d2[2], d2[3], d2[4]));
            else DO(R, pdeip_fst_derivatives5_dev(R.s, I2t0, w2, nr, nc, C2, d2[0], d2[1], d2[2]));
        }
        DOHIP(R, hipMemsetAsync(dU, 0, n * sizeof(float), R.s));
        DOHIP(R, hipMemsetAsync(dV, 0, n * sizeof(float), R.s));
        for (int k = 0; k < p.secondLoop; k++) {
            if (C2 > 0 && gradmag)
                DO(R, pdeip_flow_assemble_gradmag_dev(R.s, d1[0], d1[1], d1[2], C1, (float)p.b1, d2[0], d2[1], d2[2], d2[3], d2[4], C2, (float)p.b2, dU, dV, (float)p.alpha,
                                                      nr, nc, coef[0], coef[1], coef[2], coef[3], coef[4]));
            else
                DO(R, pdeip_flow_assemble_dev(R.s, d1[0], d1[1], d1[2], C1, (float)p.b1, C2 > 0 ? d2[0] : nullptr, C2 > 0 ? d2[1] : nullptr, C2 > 0 ? d2[2] : nullptr, C2,
                                              C2 > 0 ? (float)p.b2 : 0.0f, dU, dV, (float)p.alpha, nr, nc, coef[0], coef[1], coef[2], coef[3], coef[4]));
            if (lv.Us) DO(R, pdeip_flow_apriori_dev(R.s, lv.Us, U, dU, p.gammaS, p.alpha, as_diff, u_double && first == 0, k == 0, nr, nc, coef[1], coef[3]));
            if (lv.Vs) DO(R, pdeip_flow_apriori_dev(R.s, lv.Vs, V, dV, p.gammaS, p.alpha, as_diff, u_double && first == 0, k == 0, nr, nc, coef[2], coef[4]));
            if (flow_diffusion) { // U+dU+V+dV, left to right
                DO(R, pdeip_add_dev(R.s, U, dU, nr, nc, S));
                DO(R, pdeip_add_dev(R.s, S, V, nr, nc, S));
                DO(R, pdeip_add_dev(R.s, S, dV, nr, nc, S));
                DO(R, pdeip_ad_weights_dev(R.s, S, nr, nc, 1, quantile, w8[0], w8[1], w8[2], w8[3], w8[4], w8[5], w8[6], w8[7]));
            }
            if (p.solver == PDEIP_SOLVER_SOR)
                DO(R, pdeip_oflow_sor_llin4_dev(R.s, U, V, dU, dV, coef[0], coef[1], coef[2], coef[3], coef[4], w8[0], w8[2], w8[4], w8[6], nr, nc, p.iter, (float)p.omega,
                                                R.mode, 0));
            else
                DO(R, pdeip_oflow_alr_llin8_dev(R.s, U, V, dU, dV, coef[0], coef[1], coef[2], coef[3], coef[4], w8[0], w8[1], w8[2], w8[3], w8[4], w8[5], w8[6], w8[7], nr, nc,
                                                p.iter, (float)p.omega, R.mode));
        }
        DO(R, pdeip_median3_dev(R.s, U, dU, nr, nc, Ua));
        DO(R, pdeip_median3_dev(R.s, V, dV, nr, nc, Va));
        std::swap(U, Ua);
        std::swap(V, Va);
    }
    R.release(m);
}

// The stereo twin: DispEminND_llin_2D.m:202-316, flow_level.py DispLlinLevel.run
void disp_level(Run &R, const Params &p, const Level &lv, const float *I1t0, const float *I1t1, int C1, const float *I2t0, const float *I2t1, int C2,
                bool gradmag, float *&U, float *&Ua, double as_diff, bool u_double)
{
    const int nr = lv.nr, nc = lv.nc;
    const size_t n = (size_t)nr * nc;
    const size_t m = R.mark();
    float *w1 = R.planes(nr, nc, C1), *d1[3], *w2 = nullptr, *d2[5] = {};
    for (auto &q : d1) q = R.planes(nr, nc, C1);
    if (C2 > 0) {
        w2 = R.planes(nr, nc, C2);
        for (int k = 0; k < (gradmag ? 5 : 3); k++) d2[k] = R.planes(nr, nc, C2);
    }
    float *CuGd = R.planes(nr, nc), *DuGd = R.planes(nr, nc), *S = R.planes(nr, nc), *w[4], *dU = R.planes(nr, nc);
    for (auto &q : w) q = R.planes(nr, nc); // wW, wN, wE, wS
    for (int first = 0; first < p.firstLoop; first++) {
        DO(R, pdeip_flow_warp_dev(R.s, U, nullptr, I1t1, C1, C2 > 0 ? I2t1 : nullptr, C2, nr, nc, w1, w2)); // along x only
        DO(R, pdeip_fst_derivatives5_dev(R.s, I1t0, w1, nr, nc, C1, d1[0], d1[1], d1[2]));
        if (C2 > 0) {
            if (gradmag) DO(R, pdeip_snd_derivatives5_dev(R.s, I2t0, w2, nr, nc, C2, d2[0], d2[1], d2[2], d2[3], d2[4]));
            else DO(R, pdeip_fst_derivatives5_dev(R.s, I2t0, w2, nr, nc, C2, d2[0], d2[1], d2[2]));
        }
        DOHIP(R, hipMemsetAsync(dU, 0, n * sizeof(float), R.s));
        for (int k = 0; k < p.secondLoop; k++) {
            if (C2 > 0 && gradmag) // (Ixt, Iyt, Ixx, Ixy)
                DO(R, pdeip_disp_assemble_gradmag_dev(R.s, d1[0], d1[1], C1, (float)p.b1, d2[0], d2[1], d2[2], d2[4], C2, (float)p.b2, dU, (float)p.alpha, nr, nc, CuGd, DuGd));
            else
                DO(R, pdeip_disp_assemble_dev(R.s, d1[0], d1[1], C1, (float)p.b1, C2 > 0 ? d2[0] : nullptr, C2 > 0 ? d2[1] : nullptr, C2, C2 > 0 ? (float)p.b2 : 0.0f, dU,
                                              (float)p.alpha, nr, nc, CuGd, DuGd));
            if (lv.Us) DO(R, pdeip_disp_apriori_dev(R.s, lv.Us, U, dU, p.gammaS, p.alpha, as_diff, u_double && first == 0, k == 0, nr, nc, CuGd, DuGd));
            DO(R, pdeip_add_dev(R.s, U, dU, nr, nc, S));
            DO(R, pdeip_diffweights6_dev(R.s, S, nr, nc, 1, 0.00001f, w[0], w[1], w[2], w[3]));
            if (p.solver == PDEIP_SOLVER_SOR) DO(R, pdeip_disp_sor_llin4_dev(R.s, U, dU, CuGd, DuGd, w[0], w[1], w[2], w[3], nr, nc, p.iter, (float)p.omega, R.mode, 0));
            else DO(R, pdeip_disp_alr_llin4_dev(R.s, U, dU, CuGd, DuGd, w[0], w[1], w[2], w[3], nr, nc, p.iter, (float)p.omega, R.mode));
        }
        DO(R, pdeip_median3_dev(R.s, U, dU, nr, nc, Ua));
        std::swap(U, Ua);
    }
    R.release(m);
}

// First / second constancy images of a scale (:133-170)
struct Terms {
    const float *a0, *a1, *b0, *b1;
    int C1, C2;
};
Terms terms(Run &R, const Level &lv, int C, int fst, int snd)
{
    Terms t{lv.I0, lv.I1, nullptr, nullptr, C, 0};
    if (fst == PDEIP_TERM_GRAD) {
        float *g0 = R.planes(lv.nr, lv.nc, 2 * C), *g1 = R.planes(lv.nr, lv.nc, 2 * C);
        DO(R, pdeip_rgb2grad_dev(R.s, lv.I0, lv.nr, lv.nc, C, g0));
        DO(R, pdeip_rgb2grad_dev(R.s, lv.I1, lv.nr, lv.nc, C, g1));
        t.a0 = g0;
        t.a1 = g1;
        t.C1 = 2 * C;
    }
    if (snd != PDEIP_TERM_NONE) {
        t.b0 = lv.I0;
        t.b1 = lv.I1;
        t.C2 = C;
    }
    return t;
}

int check_terms(const char *who, int fst, int snd)
{
    if (fst != PDEIP_TERM_RGB && fst != PDEIP_TERM_GRAD) return set_err(PDEIP_ERR_ARG, "%s: No such fstTerm", who);
    if (snd != PDEIP_TERM_NONE && snd != PDEIP_TERM_RGB && snd != PDEIP_TERM_GRADMAG) return set_err(PDEIP_ERR_ARG, "%s: No such sndTerm", who);
    return PDEIP_OK;
}

// the body of pdeip_flow_nd_llin for one Run (dry or real)
struct AdOptions { // quantile < 0: the isotropic driver
    double quantile = -1.0;
    bool flow_diffusion = false;
};
void flow_nd(Run &R, const float *Iin, int nrows, int ncols, int C, int fst, int snd, const Params &p, const double *Us, const double *Vs, float *U_out, float *V_out,
             const AdOptions ad = AdOptions())
{
    const size_t n = (size_t)nrows * ncols;
    float *up = R.planes(nrows, ncols, 2 * C), *fr = R.planes(nrows, ncols, 2 * C);
    DOHIP(R, hipMemcpyAsync(up, Iin, 2 * C * n * sizeof(float), hipMemcpyHostToDevice, R.s));
    scale(R, fr, up, 2 * C * n, 255.0f, true); // Iin = single(Iin) ./ 255
    const std::vector<double> G = gaussian_mask(5, 1.25);
    std::vector<Level> L = build_pyramid(R, fr, fr + C * n, nrows, ncols, C, p, 20, G, 5);
    const Level top = L.back();
    float *U = R.planes(top.nr, top.nc), *V = R.planes(top.nr, top.nc), *Ua = R.planes(top.nr, top.nc), *Va = R.planes(top.nr, top.nc);
    DOHIP(R, hipMemsetAsync(U, 0, (size_t)top.nr * top.nc * sizeof(float), R.s)); // "zero flow" unless an a-priori field starts it
    DOHIP(R, hipMemsetAsync(V, 0, (size_t)top.nr * top.nc * sizeof(float), R.s));
    apriori_pyramid(R, Us, L, p.scl_factor, false, U);
    apriori_pyramid(R, Vs, L, p.scl_factor, true, V);
    const float inv = (float)(1.0 / p.scl_factor);
    for (int s = (int)L.size() - 1; s >= 0; s--) {
        const size_t m = R.mark();
        const Terms t = terms(R, L[s], C, fst, snd);
        const double as_diff = 2.0 * std::pow(p.scl_factor, (double)s);
        const bool coarsest_apriori = s == (int)L.size() - 1 && (Us != nullptr || Vs != nullptr);
        if (ad.quantile < 0.0) flow_level(R, p, L[s], t.a0, t.a1, t.C1, t.b0, t.b1, t.C2, snd == PDEIP_TERM_GRADMAG, U, V, Ua, Va, as_diff, coarsest_apriori);
        else flow_ad_level(R, p, L[s], C, t.a0, t.a1, t.C1, t.b0, t.b1, t.C2, snd == PDEIP_TERM_GRADMAG, ad.quantile, ad.flow_diffusion, U, V, Ua, Va, as_diff, coarsest_apriori);
        if (s > 0) { // U = imresize(U .* (1/scl_factor), size of the finer scale, 'triangle')
            const Level &f = L[s - 1];
            scale(R, Ua, U, (size_t)L[s].nr * L[s].nc, inv, false);
            scale(R, Va, V, (size_t)L[s].nr * L[s].nc, inv, false);
            R.release(m); // the finer planes may reuse what this scale's terms held; U, V, Ua, Va lie below the mark
            float *Un = R.planes(f.nr, f.nc), *Vn = R.planes(f.nr, f.nc), *Una = R.planes(f.nr, f.nc), *Vna = R.planes(f.nr, f.nc);
            DO(R, pdeip_pyr_resize_dev(R.s, Ua, L[s].nr, L[s].nc, 1, f.nr, f.nc, 0, Un));
            DO(R, pdeip_pyr_resize_dev(R.s, Va, L[s].nr, L[s].nc, 1, f.nr, f.nc, 0, Vn));
            U = Un;
            V = Vn;
            Ua = Una;
            Va = Vna;
        }
    }
    DOHIP(R, hipMemcpyAsync(U_out, U, n * sizeof(float), hipMemcpyDeviceToHost, R.s));
    DOHIP(R, hipMemcpyAsync(V_out, V, n * sizeof(float), hipMemcpyDeviceToHost, R.s));
    DOHIP(R, hipStreamSynchronize(R.s));
}

void disp_nd(Run &R, const float *Il, const float *Ir, int nrows, int ncols, int C, int fst, int snd, const Params &p, const double *Us, float *U_out)
{
    const size_t n = (size_t)nrows * ncols;
    float *up = R.planes(nrows, ncols, 2 * C), *fr = R.planes(nrows, ncols, 2 * C);
    DOHIP(R, hipMemcpyAsync(up, Il, C * n * sizeof(float), hipMemcpyHostToDevice, R.s));
    DOHIP(R, hipMemcpyAsync(up + C * n, Ir, C * n * sizeof(float), hipMemcpyHostToDevice, R.s));
    scale(R, fr, up, 2 * C * n, 255.0f, true);
    const std::vector<double> G = gaussian_mask(5, 1.25);
    std::vector<Level> L = build_pyramid(R, fr, fr + C * n, nrows, ncols, C, p, 10, G, 5);
    const Level top = L.back();
    float *U = R.planes(top.nr, top.nc), *Ua = R.planes(top.nr, top.nc);
    DOHIP(R, hipMemsetAsync(U, 0, (size_t)top.nr * top.nc * sizeof(float), R.s));
    apriori_pyramid(R, Us, L, p.scl_factor, false, U);
    const float inv = (float)(1.0 / p.scl_factor);
    for (int s = (int)L.size() - 1; s >= 0; s--) {
        const size_t m = R.mark();
        const Terms t = terms(R, L[s], C, fst, snd);
        disp_level(R, p, L[s], t.a0, t.a1, t.C1, t.b0, t.b1, t.C2, snd == PDEIP_TERM_GRADMAG, U, Ua, 1.75 * std::pow(p.scl_factor, (double)s),
                   s == (int)L.size() - 1 && Us != nullptr);
        if (s > 0) {
            const Level &f = L[s - 1];
            scale(R, Ua, U, (size_t)L[s].nr * L[s].nc, inv, false);
            R.release(m);
            float *Un = R.planes(f.nr, f.nc), *Una = R.planes(f.nr, f.nc);
            DO(R, pdeip_pyr_resize_dev(R.s, Ua, L[s].nr, L[s].nc, 1, f.nr, f.nc, 0, Un));
            U = Un;
            Ua = Una;
        }
    }
    DOHIP(R, hipMemcpyAsync(U_out, U, n * sizeof(float), hipMemcpyDeviceToHost, R.s));
    DOHIP(R, hipStreamSynchronize(R.s));
}

// DispEminND_llin_sym_2D.m:51-275 (runme.m:28): symmetric stereo -- both views' disparities at once, each warped into the other for
// the symmetry term.  flow_level.py DispSymLevel.run / drivers.py DispEminND_llin_sym_2D are the statement the tests compare with
// (same `_dev` stages, same order).  No division by 255 in this driver (:81-82), a 3 x 3 Gaussian, coarsest scale unsmoothed.
struct SymParams {
    double alpha, beta, omega, b1, b2, scl_factor;
    int firstLoop, secondLoop, iter, solver;
};
void sym_level(Run &R, const SymParams &p, const Level &lv, int C, float *(&U)[2], double sr_diff)
{
    const int nr = lv.nr, nc = lv.nc;
    const size_t n = (size_t)nr * nc;
    const double kS = C * p.beta / p.alpha, sr2 = std::pow(sr_diff, 2.0);
    const float *I[2] = {lv.I0, lv.I1};
    float *zero = R.planes(nr, nc), *X = R.planes(nr, nc), *Y = R.planes(nr, nc), *S = R.planes(nr, nc);
    float *warped[2], *der[2][8], *CuG[2], *DuG[2], *w[2][4], *Un[2], *dU[2];
    double *Uw[2], *sym[2][4];
    for (int v = 0; v < 2; v++) {
        warped[v] = R.planes(nr, nc, C);
        for (auto &q : der[v]) q = R.planes(nr, nc, C);
        CuG[v] = R.planes(nr, nc);
        DuG[v] = R.planes(nr, nc);
        for (auto &q : w[v]) q = R.planes(nr, nc);
        Un[v] = R.planes(nr, nc);
        dU[v] = R.planes(nr, nc);
        Uw[v] = R.dplane(nr, nc);
        for (auto &q : sym[v]) q = R.dplane(nr, nc);
    }
    DOHIP(R, hipMemsetAsync(zero, 0, n * sizeof(float), R.s));
    for (int fl = 0; fl < p.firstLoop; fl++) {
        for (int v = 0; v < 2; v++) { // view v: own image, the other view warped by U{v}
            DO(R, pdeip_flow_coords_dev(R.s, U[v], zero, nr, nc, X, Y));
            DO(R, pdeip_warp_bilinear_dev(R.s, I[1 - v], X, Y, nr, nc, C, warped[v]));
        }
        DO(R, pdeip_sym_warp_flow_dev(R.s, U[0], U[1], nr, nc, Uw[0]));
        DO(R, pdeip_sym_warp_flow_dev(R.s, U[1], U[0], nr, nc, Uw[1]));
        for (int v = 0; v < 2; v++) {
            DO(R, pdeip_fst_derivatives5_dev(R.s, I[v], warped[v], nr, nc, C, der[v][0], der[v][1], der[v][2]));
            DO(R, pdeip_snd_derivatives5_dev(R.s, I[v], warped[v], nr, nc, C, der[v][3], der[v][4], der[v][5], der[v][6], der[v][7]));
            DO(R, pdeip_sym_flow_terms_dev(R.s, U[v], Uw[1 - v], nr, nc, sym[v][0], sym[v][1], sym[v][2], sym[v][3]));
            DOHIP(R, hipMemsetAsync(dU[v], 0, n * sizeof(float), R.s));
        }
        for (int k = 0; k < p.secondLoop; k++) {
            for (int v = 0; v < 2; v++) {
                DO(R, pdeip_sym_assemble_dev(R.s, der[v][0], der[v][1], der[v][3], der[v][4], der[v][5], der[v][7], C, sym[v][0], sym[v][1], sym[v][2], sym[v][3], dU[v],
                                             (float)p.b1, (float)p.b2, (float)p.alpha, kS, sr2, k == 0 ? 1 : 0, nr, nc, CuG[v], DuG[v]));
                DO(R, pdeip_add_dev(R.s, U[v], dU[v], nr, nc, S));
                DO(R, pdeip_diffweights6_dev(R.s, S, nr, nc, 1, (float)0.00001, w[v][0], w[v][1], w[v][2], w[v][3]));
            }
            DO(R, pdeip_disp_sor_llin_sym4_dev(R.s, U[0], dU[0], CuG[0], DuG[0], w[0][0], w[0][1], w[0][2], w[0][3], U[1], dU[1], CuG[1], DuG[1], w[1][0], w[1][1],
                                               w[1][2], w[1][3], nr, nc, p.iter, (float)p.omega, p.solver, R.mode, 0));
        }
        for (int v = 0; v < 2; v++) {
            DO(R, pdeip_median3_dev(R.s, U[v], dU[v], nr, nc, Un[v]));
            std::swap(U[v], Un[v]);
        }
    }
}
void disp_sym(Run &R, const float *Il, const float *Ir, int nrows, int ncols, int C, const SymParams &p, float *U_out)
{
    const size_t n = (size_t)nrows * ncols;
    float *fr = R.planes(nrows, ncols, 2 * C);
    DOHIP(R, hipMemcpyAsync(fr, Il, C * n * sizeof(float), hipMemcpyHostToDevice, R.s));
    DOHIP(R, hipMemcpyAsync(fr + C * n, Ir, C * n * sizeof(float), hipMemcpyHostToDevice, R.s));
    Params pp{};
    pp.scl_factor = p.scl_factor;
    pp.scales = 0x7fffffff;
    const std::vector<double> G = gaussian_mask(3, 1.0);
    const std::vector<Level> L = build_pyramid(R, fr, fr + C * n, nrows, ncols, C, pp, 10, G, 3, false);
    const Level top = L.back();
    float *U[2] = {R.planes(top.nr, top.nc), R.planes(top.nr, top.nc)};
    for (auto *u : U) DOHIP(R, hipMemsetAsync(u, 0, (size_t)top.nr * top.nc * sizeof(float), R.s));
    const float inv = (float)(1.0 / p.scl_factor);
    for (int s = (int)L.size() - 1; s >= 0; s--) {
        // the level gets its own copies of the two fields (the Python level clones them): its ping-pong may end in either plane set
        float *W[2] = {R.planes(L[s].nr, L[s].nc), R.planes(L[s].nr, L[s].nc)};
        for (int v = 0; v < 2; v++) DO(R, copy_d2d(R.s, W[v], U[v], (size_t)L[s].nr * L[s].nc));
        sym_level(R, p, L[s], C, W, 2.0 * std::pow(1.0 / p.scl_factor, (double)(-s))); // srDiff = 2*(1/scl_factor)^-(scl-1), scl 1-based there
        U[0] = W[0];
        U[1] = W[1];
        if (s > 0) {
            const Level &f = L[s - 1];
            for (int v = 0; v < 2; v++) {
                float *sc = R.planes(L[s].nr, L[s].nc), *up = R.planes(f.nr, f.nc);
                scale(R, sc, U[v], (size_t)L[s].nr * L[s].nc, inv, false);
                DO(R, pdeip_pyr_resize_dev(R.s, sc, L[s].nr, L[s].nc, 1, f.nr, f.nc, 0, up));
                U[v] = up;
            }
        }
    }
    DOHIP(R, hipMemcpyAsync(U_out, U[0], n * sizeof(float), hipMemcpyDeviceToHost, R.s));
    DOHIP(R, hipMemcpyAsync(U_out + n, U[1], n * sizeof(float), hipMemcpyDeviceToHost, R.s));
    DOHIP(R, hipStreamSynchronize(R.s));
}

// FlowEminHS_elin_2D_v10.m:52-200 (runme.m:74): Horn-Schunck with early linearisation.  Per scale the data terms from the unwarped
// frames (k_hs_assemble), a constant diffusion weight alpha * channels and ONE Oflow_sor_elin4_2d call; between scales
// imresize(medfilt2(U .* (1/scl_factor)), 'OutputSize', ...) with imresize's default, bicubic, kernel (:189-190).  drivers.py
// FlowEminHS_elin_2D_v10 / flow_level.py FlowHsLevel are the statement the tests compare with.
void flow_hs(Run &R, const float *Iin, int nrows, int ncols, int C, const Params &p, float *U_out, float *V_out)
{
    const size_t n = (size_t)nrows * ncols;
    float *up = R.planes(nrows, ncols, 2 * C), *fr = R.planes(nrows, ncols, 2 * C);
    DOHIP(R, hipMemcpyAsync(up, Iin, 2 * C * n * sizeof(float), hipMemcpyHostToDevice, R.s));
    scale(R, fr, up, 2 * C * n, 255.0f, true);
    const std::vector<double> G = gaussian_mask(5, 1.25);
    const std::vector<Level> L = build_pyramid(R, fr, fr + C * n, nrows, ncols, C, p, 20, G, 5);
    const Level top = L.back();
    float *U = R.planes(top.nr, top.nc), *V = R.planes(top.nr, top.nc);
    DOHIP(R, hipMemsetAsync(U, 0, (size_t)top.nr * top.nc * sizeof(float), R.s));
    DOHIP(R, hipMemsetAsync(V, 0, (size_t)top.nr * top.nc * sizeof(float), R.s));
    const float inv = (float)(1.0 / p.scl_factor);
    for (int s = (int)L.size() - 1; s >= 0; s--) {
        const int nr = L[s].nr, nc = L[s].nc;
        const size_t m = (size_t)nr * nc;
        float *coef[5], *W = R.planes(nr, nc);
        for (auto &q : coef) q = R.planes(nr, nc);
        DO(R, pdeip_hs_assemble_dev(R.s, L[s].I0, L[s].I1, C, (float)p.b1, (float)p.b2, nr, nc, coef[0], coef[1], coef[2], coef[3], coef[4]));
        if (!R.dry && R.rc == PDEIP_OK) hipLaunchKernelGGL(k_fill, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, R.s, W, m, (float)(p.alpha * C)); // W = alpha*channels*ones (:121)
        if (p.iter > 0) {
            if (p.solver == PDEIP_SOLVER_SOR) DO(R, pdeip_oflow_sor_elin4_dev(R.s, U, V, coef[0], coef[1], coef[2], coef[3], coef[4], W, W, W, W, nr, nc, p.iter, (float)p.omega, R.mode, 0));
            else DO(R, pdeip_oflow_alr_elin4_dev(R.s, U, V, coef[0], coef[1], coef[2], coef[3], coef[4], W, W, W, W, nr, nc, p.iter, (float)p.omega, R.mode));
        }
        if (s > 0) {
            const Level &f = L[s - 1];
            float *sU = R.planes(nr, nc), *sV = R.planes(nr, nc), *mU = R.planes(nr, nc), *mV = R.planes(nr, nc);
            scale(R, sU, U, m, inv, false);
            scale(R, sV, V, m, inv, false);
            DO(R, pdeip_median3_dev(R.s, sU, nullptr, nr, nc, mU));
            DO(R, pdeip_median3_dev(R.s, sV, nullptr, nr, nc, mV));
            float *Un = R.planes(f.nr, f.nc), *Vn = R.planes(f.nr, f.nc);
            DO(R, pdeip_pyr_resize_dev(R.s, mU, nr, nc, 1, f.nr, f.nc, 1, Un));
            DO(R, pdeip_pyr_resize_dev(R.s, mV, nr, nc, 1, f.nr, f.nc, 1, Vn));
            U = Un;
            V = Vn;
        }
    }
    DOHIP(R, hipMemcpyAsync(U_out, U, n * sizeof(float), hipMemcpyDeviceToHost, R.s));
    DOHIP(R, hipMemcpyAsync(V_out, V, n * sizeof(float), hipMemcpyDeviceToHost, R.s));
    DOHIP(R, hipStreamSynchronize(R.s));
}

// FlowEminNDFASFMG_elin_2D_v10.m (runme.m:90): FAS full-multigrid flow with early linearisation.  Image pyramid by halving
// (:104-120), per-scale derivative planes and constants (:125-153), per scale, coarse to fine (:161-183), one FAS V- or W-cycle
// (FAS_CYCLE, :193-273) whose smoother (:367-464) is firstLoop x [robust data weights + OPdiffWeights, Oflow_sor_elin4_2d];
// residuals through the solver's residual operator, the coarse right-hand side through Oflow_lhs_elin4_2d, full-weighting
// restriction, bilinear prolongation of the correction, bicubic up-scaling of the flow between scales.  fas.py FasFmgFlow is the
// statement the tests compare with: the same `_dev` stages in the same order.
struct FasParams {
    double alpha, omega, b1, b2, scl_factor;
    int firstLoop, iter, solver, cycle_index, scales;
};
struct FasScale {
    int nr, nc;
    float *pl; // [13][C][nc][nr]: Idt, Idx, Idy, Idxx, Idyy, Idxy, Idxt, Idyt, M, Cu, Cv, Du, Dv
};
struct Fas {
    Run &R;
    const FasParams &p;
    int C;
    std::vector<FasScale> S;
    float *plane(int s, int k) const { return S[s].pl + (size_t)k * C * S[s].nr * S[s].nc; }
    void solve(int s, float *U, float *V, float *const (&c)[9])
    {
        if (p.solver == PDEIP_SOLVER_SOR) DO(R, pdeip_oflow_sor_elin4_dev(R.s, U, V, c[0], c[1], c[2], c[3], c[4], c[5], c[6], c[7], c[8], S[s].nr, S[s].nc, p.iter, (float)p.omega, R.mode, 0));
        else DO(R, pdeip_oflow_alr_elin4_dev(R.s, U, V, c[0], c[1], c[2], c[3], c[4], c[5], c[6], c[7], c[8], S[s].nr, S[s].nc, p.iter, (float)p.omega, R.mode));
    }
    // in place on U, V; RU, RV ([C] planes) receive the residuals when given
    void smooth(int s, float *U, float *V, const float *Cu, const float *Cv, float *RU, float *RV)
    {
        const int nr = S[s].nr, nc = S[s].nc;
        const size_t m = R.mark();
        float *coef[9]; // MGd CuGd CvGd DuGd DvGd wW wN wE wS
        for (auto &q : coef) q = R.planes(nr, nc);
        for (int it = 0; it < p.firstLoop; it++) { // robust data weights (:377-397) and OPdiffWeights(U, V) (:392): wW wN wS wE order
            DO(R, pdeip_fas_assemble_weights_dev(R.s, S[s].pl, Cu, Cv, U, V, nr, nc, C, (float)p.b1, (float)p.b2, (float)(C * p.alpha), coef[0], coef[1], coef[2], coef[3], coef[4],
                                                 coef[5], coef[6], coef[8], coef[7]));
            solve(s, U, V, coef);
        }
        if (RU != nullptr) {
            float *f[5];
            for (auto &q : f) q = R.planes(nr, nc, C);
            DO(R, pdeip_fas_assemble_dev(R.s, S[s].pl, Cu, Cv, U, V, nr, nc, C, (float)p.b1, (float)p.b2, (float)p.alpha, 1, f[0], f[1], f[2], f[3], f[4], nullptr));
            DO(R, pdeip_flow_opdiffweights_dev(R.s, U, V, nullptr, nullptr, nr, nc, coef[5], coef[6], coef[8], coef[7]));
            DO(R, pdeip_oflow_res_elin4_dev(R.s, RU, RV, U, V, f[0], f[1], f[2], f[3], f[4], coef[5], coef[6], coef[7], coef[8], nr, nc, C));
        }
        R.release(m);
    }
    // one V (cycle_index 1) or W (2) cycle at scale s on U, V (in place); Cu / Cv: the scale's own or the cycle's right-hand side
    void cycle(int s, float *U, float *V, const float *Cu, const float *Cv)
    {
        if (s == (int)S.size() - 1) {
            smooth(s, U, V, Cu, Cv, nullptr, nullptr);
            return;
        }
        const int nr = S[s].nr, nc = S[s].nc, cr = S[s + 1].nr, cc = S[s + 1].nc;
        const float sf = (float)p.scl_factor;
        for (int ci = 0; ci < p.cycle_index; ci++) {
            const size_t m = R.mark();
            float *RU = R.planes(nr, nc, C), *RV = R.planes(nr, nc, C);
            smooth(s, U, V, Cu, Cv, RU, RV);
            float *RUres = R.planes(cr, cc, C), *RVres = R.planes(cr, cc, C), *Ures = R.planes(cr, cc), *Vres = R.planes(cr, cc);
            DO(R, pdeip_fas_restrict_dev(R.s, RU, nr, nc, C, sf, RUres));
            DO(R, pdeip_fas_restrict_dev(R.s, RV, nr, nc, C, sf, RVres));
            DO(R, pdeip_fas_restrict_dev(R.s, U, nr, nc, 1, sf, Ures));
            DO(R, pdeip_fas_restrict_dev(R.s, V, nr, nc, 1, sf, Vres));
            float *MGd = R.planes(cr, cc, C), *DuGd = R.planes(cr, cc, C), *DvGd = R.planes(cr, cc, C), *gd = R.planes(cr, cc, C), *w[4];
            for (auto &q : w) q = R.planes(cr, cc); // wW wN wE wS
            DO(R, pdeip_fas_assemble_dev(R.s, S[s + 1].pl, nullptr, nullptr, Ures, Vres, cr, cc, C, (float)p.b1, (float)p.b2, (float)p.alpha, 1, MGd, nullptr, nullptr, DuGd, DvGd, gd));
            DO(R, pdeip_flow_opdiffweights_dev(R.s, Ures, Vres, nullptr, nullptr, cr, cc, w[0], w[1], w[3], w[2]));
            float *Au = R.planes(cr, cc, C), *Av = R.planes(cr, cc, C), *fu = R.planes(cr, cc, C), *fv = R.planes(cr, cc, C);
            DO(R, pdeip_oflow_lhs_elin4_dev(R.s, Au, Av, Ures, Vres, MGd, DuGd, DvGd, w[0], w[1], w[2], w[3], cr, cc, C));
            DO(R, pdeip_fas_rhs_dev(R.s, RUres, Au, gd, cr, cc, C, fu));
            DO(R, pdeip_fas_rhs_dev(R.s, RVres, Av, gd, cr, cc, C, fv));
            float *Uc = R.planes(cr, cc), *Vc = R.planes(cr, cc);
            DO(R, copy_d2d(R.s, Uc, Ures, (size_t)cr * cc));
            DO(R, copy_d2d(R.s, Vc, Vres, (size_t)cr * cc));
            cycle(s + 1, Uc, Vc, fu, fv);
            DO(R, pdeip_fas_prolong_add_dev(R.s, U, nr, nc, Uc, Ures, cr, cc, (float)(1.0 / p.scl_factor)));
            DO(R, pdeip_fas_prolong_add_dev(R.s, V, nr, nc, Vc, Vres, cr, cc, (float)(1.0 / p.scl_factor)));
            R.release(m);
        }
        smooth(s, U, V, Cu, Cv, nullptr, nullptr);
    }
};
void fas_fmg(Run &R, const float *Iin, int nrows, int ncols, int C, const FasParams &p, float *U_out, float *V_out)
{
    const size_t n = (size_t)nrows * ncols;
    float *up = R.planes(nrows, ncols, 2 * C);
    DOHIP(R, hipMemcpyAsync(up, Iin, 2 * C * n * sizeof(float), hipMemcpyHostToDevice, R.s)); // 0..255: this driver does not rescale
    // fspecial('gaussian', [5 5], 1) as fas.py states it (float64, values below eps * max zeroed, divided by the sum, single)
    std::vector<double> g = gaussian(5, 1.0);
    double gmax = 0.0;
    for (double v : g) gmax = v > gmax ? v : gmax;
    for (double &v : g)
        if (v < 2.220446049250313e-16 * gmax) v = 0.0;
    const double gsum = numpy_sum(g);
    float G[25];
    for (int i = 0; i < 25; i++) G[i] = (float)(g[i] / gsum);
    struct Img {
        int nr, nc;
        float *a, *b;
    };
    std::vector<Img> P(1);
    P[0] = {nrows, ncols, R.planes(nrows, ncols, C), R.planes(nrows, ncols, C)};
    DO(R, pdeip_fas_gauss5_dev(R.s, up, nrows, ncols, C, G, P[0].a));
    DO(R, pdeip_fas_gauss5_dev(R.s, up + C * n, nrows, ncols, C, G, P[0].b));
    while ((int)P.size() < p.scales) {
        const Img cur = P.back();
        Img nx{(cur.nr + 1) / 2, (cur.nc + 1) / 2, nullptr, nullptr};
        nx.a = R.planes(nx.nr, nx.nc, C);
        nx.b = R.planes(nx.nr, nx.nc, C);
        DO(R, pdeip_fas_down_dev(R.s, cur.a, cur.nr, cur.nc, C, nx.a));
        DO(R, pdeip_fas_down_dev(R.s, cur.b, cur.nr, cur.nc, C, nx.b));
        P.push_back(nx);
        if (nx.nr <= 10 || nx.nc <= 10) break;
    }
    Fas F{R, p, C, {}};
    for (const Img &im : P) {
        FasScale sc{im.nr, im.nc, R.planes(im.nr, im.nc, 13 * C)};
        DO(R, pdeip_fas_prepare_dev(R.s, im.a, im.b, im.nr, im.nc, C, (float)p.b1, (float)p.b2, sc.pl));
        F.S.push_back(sc);
    }
    const int last = (int)F.S.size() - 1;
    float *U = R.planes(F.S[last].nr, F.S[last].nc), *V = R.planes(F.S[last].nr, F.S[last].nc);
    DOHIP(R, hipMemsetAsync(U, 0, (size_t)F.S[last].nr * F.S[last].nc * sizeof(float), R.s));
    DOHIP(R, hipMemsetAsync(V, 0, (size_t)F.S[last].nr * F.S[last].nc * sizeof(float), R.s));
    for (int s = last; s >= 0; s--) {
        F.cycle(s, U, V, F.plane(s, 9), F.plane(s, 10)); // the scale's own Cu, Cv
        if (s > 0) {
            float *Un = R.planes(F.S[s - 1].nr, F.S[s - 1].nc), *Vn = R.planes(F.S[s - 1].nr, F.S[s - 1].nc);
            DO(R, pdeip_fas_upscale_dev(R.s, U, F.S[s].nr, F.S[s].nc, (float)(1.0 / p.scl_factor), F.S[s - 1].nr, F.S[s - 1].nc, Un));
            DO(R, pdeip_fas_upscale_dev(R.s, V, F.S[s].nr, F.S[s].nc, (float)(1.0 / p.scl_factor), F.S[s - 1].nr, F.S[s - 1].nc, Vn));
            U = Un;
            V = Vn;
        }
    }
    DOHIP(R, hipMemcpyAsync(U_out, U, n * sizeof(float), hipMemcpyDeviceToHost, R.s));
    DOHIP(R, hipMemcpyAsync(V_out, V, n * sizeof(float), hipMemcpyDeviceToHost, R.s));
    DOHIP(R, hipStreamSynchronize(R.s));
}

// TVdenoise8.m:36-111 / TVdenoise4.m:37-114: a short pyramid (down to scl x the frame), per scale the lagged-diffusivity loop --
// outer_iter + 1 times [diffusion weights of the current estimate, PsiData / TRACE / B, PDEsolver8 | PDEsolver4] --, the
// estimate resized up to the next finer scale.  drivers.py `_tv` / flow_level.py TvLevel, Tv4Level are the statement the tests
// compare with.  TVdenoise8 leaves its coarsest scale unsmoothed (the driver writes that result to a misspelt variable, :72).
struct TvParams {
    double alpha, omega, scl, scl_factor;
    int outer_iter, inner_iter, solver;
};
void tv_run(Run &R, const float *Iin, int nrows, int ncols, int F, const TvParams &p, bool eight, float *Iout_host)
{
    struct Lv {
        int nr, nc;
        float *I;
    };
    const size_t n0 = (size_t)nrows * ncols * F;
    std::vector<Lv> L(1);
    L[0] = {nrows, ncols, R.planes(nrows, ncols, F)};
    DOHIP(R, hipMemcpyAsync(L[0].I, Iin, n0 * sizeof(float), hipMemcpyHostToDevice, R.s));
    const int ds_rows = (int)std::ceil(nrows * p.scl), ds_cols = (int)std::ceil(ncols * p.scl);
    const int gsize = eight ? 5 : 7;
    const std::vector<double> G = gaussian_mask(gsize, eight ? 1.25 : 2.0);
    for (;;) {
        const Lv cur = L.back();
        Lv nx{(int)std::ceil(cur.nr * p.scl_factor), (int)std::ceil(cur.nc * p.scl_factor), nullptr};
        nx.I = R.planes(nx.nr, nx.nc, F);
        DO(R, pdeip_pyr_resize_dev(R.s, cur.I, cur.nr, cur.nc, F, nx.nr, nx.nc, 0, nx.I));
        float *sm = R.planes(cur.nr, cur.nc, F);
        DO(R, pdeip_pyr_smooth_dev(R.s, cur.I, cur.nr, cur.nc, F, G.data(), gsize, sm));
        L.back().I = sm;
        L.push_back(nx);
        // (a frame that no longer shrinks ends the pyramid too: the reference's loop would not end)
        if (nx.nr <= ds_rows || nx.nc <= ds_cols || (nx.nr == cur.nr && nx.nc == cur.nc)) {
            if (!eight) {
                float *t = R.planes(nx.nr, nx.nc, F);
                DO(R, pdeip_pyr_smooth_dev(R.s, nx.I, nx.nr, nx.nc, F, G.data(), gsize, t));
                L.back().I = t;
            }
            break;
        }
    }
    const float *est = L.back().I;
    int er = L.back().nr, ec = L.back().nc;
    for (int s = (int)L.size() - 1; s >= 0; s--) {
        const int nr = L[s].nr, nc = L[s].nc;
        float *X = R.planes(nr, nc, F), *TRACE = R.planes(nr, nc, F), *B = R.planes(nr, nc, F), *w[8];
        for (int k = 0; k < (eight ? 8 : 4); k++) w[k] = R.planes(nr, nc, F);
        DO(R, copy_d2d(R.s, X, est, (size_t)nr * nc * F));
        (void)er; (void)ec;
        for (int it = 0; it <= p.outer_iter; it++) { // for iter=0:param.outer_iter
            if (eight) {
                DO(R, pdeip_tv_assemble_dev(R.s, X, L[s].I, nr, nc, F, (float)p.alpha, TRACE, B, w[0], w[1], w[2], w[3], w[4], w[5], w[6], w[7]));
                if (p.solver == PDEIP_SOLVER_SOR) DO(R, pdeip_pde_sor8_dev(R.s, X, TRACE, B, w[0], w[1], w[2], w[3], w[4], w[5], w[6], w[7], nr, nc, F, p.inner_iter, (float)p.omega, R.mode, 0));
                else DO(R, pdeip_pde_alr8_dev(R.s, X, TRACE, B, w[0], w[1], w[2], w[3], w[4], w[5], w[6], w[7], nr, nc, F, p.inner_iter, (float)p.omega, R.mode));
            } else {
                DO(R, pdeip_tv4_assemble_dev(R.s, X, L[s].I, nr, nc, F, (float)p.alpha, TRACE, B, w[0], w[1], w[2], w[3]));
                if (p.solver == PDEIP_SOLVER_SOR) DO(R, pdeip_pde_sor4_dev(R.s, X, TRACE, B, w[0], w[1], w[2], w[3], nr, nc, F, p.inner_iter, (float)p.omega, R.mode, 0));
                else DO(R, pdeip_pde_alr4_dev(R.s, X, TRACE, B, w[0], w[1], w[2], w[3], nr, nc, F, p.inner_iter, (float)p.omega, R.mode));
            }
        }
        est = X;
        er = nr;
        ec = nc;
        if (s > 0) {
            float *up = R.planes(L[s - 1].nr, L[s - 1].nc, F);
            DO(R, pdeip_pyr_resize_dev(R.s, X, nr, nc, F, L[s - 1].nr, L[s - 1].nc, 0, up));
            est = up;
        }
    }
    DOHIP(R, hipMemcpyAsync(Iout_host, est, n0 * sizeof(float), hipMemcpyDeviceToHost, R.s));
    DOHIP(R, hipStreamSynchronize(R.s));
}

int tv_entry(const char *who, const float *Iin, int nrows, int ncols, int frames, const pdeip_tv_params *u, const TvParams &dflt, bool eight, float *Iout);

template <class Body> int play(const char *who, Body body)
{
    RC(use_device());
    Run plan;
    body(plan); // dry: sizes only
    float *arena = nullptr;
    RC(ws_get(WS_DRIVER, plan.peak + 256, &arena));
    Run R;
    R.dry = false;
    R.base = reinterpret_cast<char *>(arena);
    R.mode = g.mode;
    R.s = nullptr; // the host entry points use the device's default stream
    body(R);
    if (R.rc != PDEIP_OK) return R.rc;
    if (R.used > plan.peak + 256) return set_err(PDEIP_ERR_DEVICE, "%s: the run used more device memory than its plan", who);
    RC(pdeip_persist_error());
    return PDEIP_OK;
}

int tv_entry(const char *who, const float *Iin, int nrows, int ncols, int frames, const pdeip_tv_params *u, const TvParams &dflt, bool eight, float *Iout)
{
    NONNULL(who, Iin);
    NONNULL(who, Iout);
    RC(check_dims(who, nrows, ncols, frames));
    read_env_once();
    TvParams p = dflt;
    if (u != nullptr) { // <= 0 or NaN: the driver's default
        auto D = [](double v, double d) { return (v > 0.0) ? v : d; };
        auto I = [](int v, int d) { return v > 0 ? v : d; };
        p.alpha = D(u->alpha, dflt.alpha);
        p.omega = D(u->omega, dflt.omega);
        p.scl = D(u->scl, dflt.scl);
        p.scl_factor = D(u->scl_factor, dflt.scl_factor);
        p.outer_iter = I(u->outer_iter, dflt.outer_iter);
        p.inner_iter = I(u->inner_iter, dflt.inner_iter);
        p.solver = I(u->solver, dflt.solver);
    }
    RC(check_solver(who, p.solver));
    if (!(p.scl_factor < 1.0)) return set_err(PDEIP_ERR_ARG, "%s: scl_factor must be below 1", who);
    return play(who, [&](Run &R) { tv_run(R, Iin, nrows, ncols, frames, p, eight, Iout); });
}

} // namespace

extern "C" int pdeip_tvdenoise8(const float *Iin, int nrows, int ncols, int frames, const pdeip_tv_params *prm, float *Iout)
{
    return tv_entry("pdeip_tvdenoise8", Iin, nrows, ncols, frames, prm, TvParams{500.0, 1.75, 0.75, 0.75, 20, 4, PDEIP_SOLVER_ALR}, true, Iout); // TVdenoise8.m:36-44
}

extern "C" int pdeip_tvdenoise4(const float *Iin, int nrows, int ncols, int frames, const pdeip_tv_params *prm, float *Iout)
{
    return tv_entry("pdeip_tvdenoise4", Iin, nrows, ncols, frames, prm, TvParams{5.0, 1.75, 0.5, 0.75, 10, 5, PDEIP_SOLVER_ALR}, false, Iout); // TVdenoise4.m:37-45
}

extern "C" int pdeip_flow_nd_llin(const float *Iin, int nrows, int ncols, int channels, int fst_term, int snd_term, const pdeip_driver_params *prm,
                                  const double *Us, const double *Vs, float *U, float *V)
{
    const char *who = "pdeip_flow_nd_llin";
    NONNULL(who, Iin);
    NONNULL(who, U);
    NONNULL(who, V);
    RC(check_dims(who, nrows, ncols, channels));
    RC(check_terms(who, fst_term, snd_term));
    read_env_once();
    // FlowEminND_llin_2D_v10.m:52-67
    const Params dflt{0.042, 1.9, 0.01, 1.4843, 0.2915, 0.75, 4, 4, 4, PDEIP_SOLVER_ALR, 0x7fffffff};
    const Params p = merge(prm, dflt);
    RC(check_solver(who, p.solver));
    if (!(p.scl_factor < 1.0)) return set_err(PDEIP_ERR_ARG, "%s: scl_factor must be below 1", who);
    return play(who, [&](Run &R) { flow_nd(R, Iin, nrows, ncols, channels, fst_term, snd_term, p, Us, Vs, U, V); });
}

extern "C" int pdeip_flow_ad_llin(const float *Iin, int nrows, int ncols, int channels, int fst_term, int snd_term, const pdeip_driver_params *prm,
                                  double quantile, int flow_diffusion, const double *Us, const double *Vs, float *U, float *V)
{
    const char *who = "pdeip_flow_ad_llin";
    NONNULL(who, Iin);
    NONNULL(who, U);
    NONNULL(who, V);
    RC(check_dims(who, nrows, ncols, channels));
    RC(check_terms(who, fst_term, snd_term));
    read_env_once();
    // FlowEminAD_llin_2D_v10.m:52-70: the isotropic driver's defaults + quantile 0.9, diffusion 'image'
    const Params dflt{0.042, 1.9, 0.01, 1.4843, 0.2915, 0.75, 4, 4, 4, PDEIP_SOLVER_ALR, 0x7fffffff};
    const Params p = merge(prm, dflt);
    RC(check_solver(who, p.solver));
    if (!(p.scl_factor < 1.0)) return set_err(PDEIP_ERR_ARG, "%s: scl_factor must be below 1", who);
    AdOptions ad;
    ad.quantile = (quantile > 0.0) ? quantile : 0.9;
    if (ad.quantile > 1.0) return set_err(PDEIP_ERR_ARG, "%s: quantile must be in (0, 1]", who);
    ad.flow_diffusion = flow_diffusion != 0;
    return play(who, [&](Run &R) { flow_nd(R, Iin, nrows, ncols, channels, fst_term, snd_term, p, Us, Vs, U, V, ad); });
}

extern "C" int pdeip_flow_fas_fmg_elin(const float *Iin, int nrows, int ncols, int channels, const pdeip_fas_params *u, float *U, float *V)
{
    const char *who = "pdeip_flow_fas_fmg_elin";
    NONNULL(who, Iin);
    NONNULL(who, U);
    NONNULL(who, V);
    RC(check_dims(who, nrows, ncols, channels));
    read_env_once();
    FasParams p{0.035, 1.9, 0.03, 0.97, 0.5, 4, 4, PDEIP_SOLVER_ALR, 1, 0x7fffffff}; // FlowEminNDFASFMG_elin_2D_v10.m:53-69
    if (u != nullptr) { // <= 0 or NaN: the driver's default
        auto D = [](double v, double d) { return (v > 0.0) ? v : d; };
        auto I = [](int v, int d) { return v > 0 ? v : d; };
        p.alpha = D(u->alpha, p.alpha);
        p.omega = D(u->omega, p.omega);
        p.b1 = D(u->b1, p.b1);
        p.b2 = D(u->b2, p.b2);
        p.scl_factor = D(u->scl_factor, p.scl_factor);
        p.firstLoop = I(u->firstLoop, p.firstLoop);
        p.iter = I(u->iter, p.iter);
        p.solver = I(u->solver, p.solver);
        p.cycle_index = I(u->cycle_index, p.cycle_index);
        p.scales = I(u->scales, p.scales);
    }
    RC(check_solver(who, p.solver));
    return play(who, [&](Run &R) { fas_fmg(R, Iin, nrows, ncols, channels, p, U, V); });
}

extern "C" int pdeip_flow_hs_elin(const float *Iin, int nrows, int ncols, int channels, const pdeip_driver_params *prm, float *U, float *V)
{
    const char *who = "pdeip_flow_hs_elin";
    NONNULL(who, Iin);
    NONNULL(who, U);
    NONNULL(who, V);
    RC(check_dims(who, nrows, ncols, channels));
    read_env_once();
    // FlowEminHS_elin_2D_v10.m:52-62 (gammaS, firstLoop, secondLoop, scales: not parameters of this driver)
    const Params dflt{0.2, 1.9, 0.0, 0.25, 0.75, 0.75, 1, 1, 20, PDEIP_SOLVER_ALR, 0x7fffffff};
    Params p = merge(prm, dflt);
    p.scales = 0x7fffffff;
    RC(check_solver(who, p.solver));
    if (!(p.scl_factor < 1.0)) return set_err(PDEIP_ERR_ARG, "%s: scl_factor must be below 1", who);
    return play(who, [&](Run &R) { flow_hs(R, Iin, nrows, ncols, channels, p, U, V); });
}

extern "C" int pdeip_disp_nd_llin_sym(const float *Il, const float *Ir, int nrows, int ncols, int channels, const pdeip_sym_params *u, float *U)
{
    const char *who = "pdeip_disp_nd_llin_sym";
    NONNULL(who, Il);
    NONNULL(who, Ir);
    NONNULL(who, U);
    RC(check_dims(who, nrows, ncols, channels));
    read_env_once();
    SymParams p{0.035, 0.4, 1.9, 0.25, 0.72, 0.75, 3, 4, 4, PDEIP_SOLVER_ALR}; // DispEminND_llin_sym_2D.m:51-62
    if (u != nullptr) { // <= 0 or NaN: the driver's default
        auto D = [](double v, double d) { return (v > 0.0) ? v : d; };
        auto I = [](int v, int d) { return v > 0 ? v : d; };
        p.alpha = D(u->alpha, p.alpha);
        p.beta = D(u->beta, p.beta);
        p.omega = D(u->omega, p.omega);
        p.b1 = D(u->b1, p.b1);
        p.b2 = D(u->b2, p.b2);
        p.scl_factor = D(u->scl_factor, p.scl_factor);
        p.firstLoop = I(u->firstLoop, p.firstLoop);
        p.secondLoop = I(u->secondLoop, p.secondLoop);
        p.iter = I(u->iter, p.iter);
        p.solver = I(u->solver, p.solver);
    }
    RC(check_solver(who, p.solver));
    if (!(p.scl_factor < 1.0)) return set_err(PDEIP_ERR_ARG, "%s: scl_factor must be below 1", who);
    return play(who, [&](Run &R) { disp_sym(R, Il, Ir, nrows, ncols, channels, p, U); });
}

extern "C" int pdeip_disp_nd_llin(const float *Il, const float *Ir, int nrows, int ncols, int channels, int fst_term, int snd_term,
                                  const pdeip_driver_params *prm, const double *Us, float *U)
{
    const char *who = "pdeip_disp_nd_llin";
    NONNULL(who, Il);
    NONNULL(who, Ir);
    NONNULL(who, U);
    RC(check_dims(who, nrows, ncols, channels));
    RC(check_terms(who, fst_term, snd_term));
    read_env_once();
    // DispEminND_llin_2D.m:51-63
    const Params dflt{0.042, 1.9, 0.005, 1.48, 0.29, 0.75, 4, 6, 4, PDEIP_SOLVER_ALR, 0x7fffffff};
    const Params p = merge(prm, dflt);
    RC(check_solver(who, p.solver));
    if (!(p.scl_factor < 1.0)) return set_err(PDEIP_ERR_ARG, "%s: scl_factor must be below 1", who);
    return play(who, [&](Run &R) { disp_nd(R, Il, Ir, nrows, ncols, channels, fst_term, snd_term, p, Us, U); });
}
