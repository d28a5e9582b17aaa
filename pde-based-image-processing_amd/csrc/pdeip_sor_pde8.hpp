// pdeip_sor_pde8.hpp -- GS_SOR_8_2d (pdeSolvers.c:153-268): scalar SOR with a 9-point stencil.
//
// Two orderings, same ModelPde8::update():
//
//  * four-colour (RED_BLACK mode): colour = (i&1) | ((j+col0)&1)<<1, passes 0..3.  No two pixels
//    of one colour are neighbours in a 9-point stencil, so a pass is one in-place launch; the
//    border replicate (read by the next sweep's diagonal taps) is a fifth small launch.
//
//  * exact order: the reference's lexicographic order.  Pixel (i,j) needs the new values of its
//    N, NW, W, SW neighbours, so the independent fronts are i + 2j + 4t = const.  Same tile
//    wavefront as pdeip_sor_exact.hpp with a skew of two rows per lane: lane l relaxes row
//    1 + 64a + q - 2l of column 1 + 64b + l; tile (a,b,t) depends only on tiles with a smaller
//    m = a + 3b + 4t.  West values arrive through a 3-deep shift register fed by one wavefront
//    shuffle per step, east values through three shuffles of the neighbour's look-ahead queue.
//    Because diagonal taps read border cells that hold the PREVIOUS sweep's replicate, the
//    border ring of every sweep is kept in a small side array (2*(nrows+ncols) floats per
//    sweep and frame) written by the lane that relaxes the adjacent interior pixel.
#pragma once
#include "pdeip_models.hpp"
#include "pdeip_sor_exact.hpp"

namespace pdeip {

struct Pde8Planes {
    float *x;
    const float *cf[ModelPde8::NCF];
};

// ---------------------------------------------------------------------------------------------
// four-colour ordering
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
k_pde8_colour(Pde8Planes P, int nrows, int ncols, int colour, float omega, int col0, size_t frame_stride)
{
    // thread -> one pixel of this colour: rows i = 1 + (colour&1 ^ 1 ...) handled via parity
    const int pi = colour & 1, pj = (colour >> 1) & 1;
    // first interior row with i&1 == pi, first interior column with (j+col0)&1 == pj
    const int ifirst = (pi == 1) ? 1 : 2;
    const int jfirst = (((1 + col0) & 1) == pj) ? 1 : 2;
    const int i = ifirst + 2 * (int)(blockIdx.x * blockDim.x + threadIdx.x);
    const int j = jfirst + 2 * (int)blockIdx.y;
    if (i > nrows - 2 || j > ncols - 2) return;
    const size_t fo = (size_t)blockIdx.z * frame_stride;
    float *x = P.x + fo;
    const size_t pos = (size_t)j * nrows + i, wp = pos - nrows, ep = pos + nrows;
    float k[ModelPde8::NCF];
#pragma unroll
    for (int f = 0; f < ModelPde8::NCF; f++) k[f] = P.cf[f][fo + pos];
    x[pos] = ModelPde8::update(x[pos], x[wp], x[ep], x[pos - 1], x[pos + 1], x[wp - 1], x[ep - 1], x[wp + 1],
                               x[ep + 1], k, omega, 1.0f - omega);
}

inline int pde8_run_colour(hipStream_t s, Pde8Planes P, int nrows, int ncols, int nframes, int iter,
                           float omega, int col0)
{
    const size_t n = (size_t)nrows * ncols;
    const int hi = (nrows + 1) / 2, hj = (ncols + 1) / 2;
    const dim3 grid((unsigned)((hi + 255) / 256), (unsigned)hj, (unsigned)nframes);
    const int nb = 2 * ncols + 2 * (nrows - 2);
    int launches = 0;
    for (int it = 0; it < iter; it++) {
        for (int c = 0; c < 4; c++) {
            hipLaunchKernelGGL(k_pde8_colour, grid, dim3(256), 0, s, P, nrows, ncols, c, omega, col0, n);
            launches++;
        }
        hipLaunchKernelGGL(k_fill_borders, dim3((nb + 255) / 256, nframes, 1), dim3(256), 0, s, P.x, P.x, 1, nrows, ncols, n);
        launches++;
    }
    return launches;
}

// ---------------------------------------------------------------------------------------------
// exact (lexicographic) ordering
// ---------------------------------------------------------------------------------------------
constexpr int P8_R = 64;    // steps per tile
constexpr int P8_SKEW = 2;  // rows per lane
constexpr int P8_G = 3;     // m = a + P8_G*b + P8_H*t   (see header comment / DESIGN.md)
constexpr int P8_H = 4;

__host__ __device__ inline size_t pde8_side_stride(int nrows, int ncols) { return 2 * ((size_t)nrows + ncols); }
inline size_t pde8_exact_scratch_floats(int nrows, int ncols, int nframes, int iter)
{
    return pde8_side_stride(nrows, ncols) * (size_t)nframes * (size_t)(iter > 0 ? iter : 1);
}

__global__ void __launch_bounds__(64)
k_pde8_exact(Pde8Planes P, float *side, int nrows, int ncols, int A, int B, int T, int m, float omega,
             size_t frame_stride)
{
    const int lane = threadIdx.x;
    const int b = blockIdx.x % B, t = blockIdx.x / B;
    const int a = m - P8_G * b - P8_H * t;
    if (a < 0 || a >= A) return;
    const size_t fo = (size_t)blockIdx.y * frame_stride;
    float *x = P.x + fo;
    const size_t sstride = pde8_side_stride(nrows, ncols);
    // border ring after sweep t-1 (read) and after sweep t (written): top[ncols] bot[ncols] left[nrows] right[nrows]
    float *ring_w = side + ((size_t)blockIdx.y * T + t) * sstride;
    const float *ring_r = (t > 0) ? ring_w - sstride : nullptr;

    const int j = 1 + 64 * b + lane;
    const bool col_in = j <= ncols - 1, col_ok = j <= ncols - 2;
    const size_t cb = (size_t)j * nrows;
    const float om1 = 1.0f - omega;
    const int i0 = 1 + a * P8_R - P8_SKEW * lane;

    // value of cell (ii,jj) as the reference's sweep t sees a cell that sweep t has not (yet) relaxed:
    // border cells hold the replicate of sweep t-1 (the caller's cell in sweep 0)
    auto old_at = [&](int ii, int jj) -> float {
        if (ii < 0 || ii > nrows - 1 || jj < 0 || jj > ncols - 1) return 0.0f;
        if (ring_r != nullptr) {
            if (ii == 0) return ring_r[jj];
            if (ii == nrows - 1) return ring_r[ncols + jj];
            if (jj == 0) return ring_r[2 * ncols + ii];
            if (jj == ncols - 1) return ring_r[2 * ncols + nrows + ii];
        }
        return x[(size_t)jj * nrows + ii];
    };
    auto is_border = [&](int ii, int jj) -> bool { return ii <= 0 || ii >= nrows - 1 || jj <= 0 || jj >= ncols - 1; };

    // own column look-ahead queue of old values: rows i, i+1, i+2, i+3
    float o0 = col_in ? old_at(i0, j) : 0.0f, o1 = col_in ? old_at(i0 + 1, j) : 0.0f, o2 = col_in ? old_at(i0 + 2, j) : 0.0f;
    // west column shift register (new values): rows i-1, i (i+1 is fetched per step)
    float nw = 0.0f, w = 0.0f, prev = 0.0f;
    if (col_ok) {
        nw = old_at(i0 - 1, j - 1); // memory holds sweep-t values there already, or the border rule applies
        w = old_at(i0, j - 1);
    }

    for (int q = 0; q < P8_R; q++) {
        const int i = i0 + q;
        const bool row_ok = (i >= 1) && (i <= nrows - 2);
        const bool active = col_ok && row_ok;
        const float o3 = col_in ? old_at(i + 3, j) : 0.0f;

        // east column (old): lane l+1 is two rows higher, its queue rows +1,+2,+3 are my rows i-1,i,i+1
        float ne = __shfl_down(o1, 1), e = __shfl_down(o2, 1), se = __shfl_down(o3, 1);
        if (lane == 63) {
            ne = old_at(i - 1, j + 1);
            e = old_at(i, j + 1);
            se = old_at(i + 1, j + 1);
        }
        // south-west (new): lane l-1 relaxed (i+1, j-1) one step ago
        float sw = __shfl_up(prev, 1);
        if (lane == 0 || q == 0 || is_border(i + 1, j - 1)) sw = col_ok ? old_at(i + 1, j - 1) : 0.0f;

        if (active) {
            float n = (q == 0 || i == 1) ? old_at(i - 1, j) : prev;
            float k[ModelPde8::NCF];
#pragma unroll
            for (int f = 0; f < ModelPde8::NCF; f++) k[f] = P.cf[f][fo + cb + i];
            const float v = ModelPde8::update(o0, w, e, n, o1, nw, ne, sw, se, k, omega, om1);
            x[cb + i] = v;
            prev = v;
            // border ring after this sweep: nearest-interior replicate (pdeSolvers.c:249-262)
            float *top = ring_w, *bot = ring_w + ncols, *left = ring_w + 2 * ncols, *right = left + nrows;
            if (i == 1) {
                top[j] = v;
                if (j == 1) { top[0] = v; left[0] = v; }
                if (j == ncols - 2) { top[ncols - 1] = v; right[0] = v; }
            }
            if (i == nrows - 2) {
                bot[j] = v;
                if (j == 1) { bot[0] = v; left[nrows - 1] = v; }
                if (j == ncols - 2) { bot[ncols - 1] = v; right[nrows - 1] = v; }
            }
            if (j == 1) left[i] = v;
            if (j == ncols - 2) right[i] = v;
        }
        // advance one row
        o0 = o1; o1 = o2; o2 = o3;
        nw = w; w = sw;
    }
}

inline int pde8_run_exact(hipStream_t s, Pde8Planes P, float *side, int nrows, int ncols, int nframes,
                          int iter, float omega)
{
    const size_t n = (size_t)nrows * ncols;
    const int A = (nrows - 2 + P8_SKEW * 63 + P8_R - 1) / P8_R;
    const int B = (ncols - 2 + 63) / 64;
    const int last_m = (A - 1) + P8_G * (B - 1) + P8_H * (iter - 1);
    const dim3 grid((unsigned)(B * iter), (unsigned)nframes);
    int launches = 0;
    for (int m = 0; m <= last_m; m++) {
        hipLaunchKernelGGL(k_pde8_exact, grid, dim3(64), 0, s, P, side, nrows, ncols, A, B, iter, m, omega, n);
        launches++;
    }
    const int nb = 2 * ncols + 2 * (nrows - 2);
    hipLaunchKernelGGL(k_fill_borders, dim3((nb + 255) / 256, nframes, 1), dim3(256), 0, s, P.x, P.x, 1, nrows, ncols, n);
    return launches + 1;
}

} // namespace pdeip
