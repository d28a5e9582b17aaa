"""Coarse-to-fine pyramid around the resident levels of flow_level.py (host side, numpy).

The MATLAB drivers build their pyramid with Image Processing Toolbox calls (`imresize(...,'bilinear')`,
`fspecial('gaussian',[5 5],1.25)` + `imfilter(...,'replicate')`, FlowEminND_llin_2D_v10.m:100-127, :360-363).
There is no IPT here and nothing to compare with, so these are OUR definitions of those calls, written from
their documentation: bilinear = triangle kernel at MATLAB's pixel-centre convention, widened by 1/scale when
shrinking (IPT's antialiasing), output size ceil(size*scale).  They are host-side numpy on purpose: the
pyramid is built once per image pair, everything inside a level stays on the device.
"""
import math

import numpy as np


def gaussian5(sigma=1.25):
    """fspecial('gaussian', [5 5], sigma)"""
    ax = np.arange(-2, 3, dtype=np.float64)
    g = np.exp(-(ax[:, None] ** 2 + ax[None, :] ** 2) / (2.0 * sigma * sigma))
    return g / g.sum()


def gaussian(size, sigma):
    """fspecial('gaussian', [size size], sigma)"""
    r = size // 2
    ax = np.arange(-r, r + 1, dtype=np.float64)
    g = np.exp(-(ax[:, None] ** 2 + ax[None, :] ** 2) / (2.0 * sigma * sigma))
    return g / g.sum()


def smooth(I, G=None):
    """imfilter(I, G, 'replicate') for an odd square kernel (default: the 5x5, sigma 1.25 one), per channel."""
    G = gaussian5() if G is None else G
    r = G.shape[0] // 2
    I3 = I if I.ndim == 3 else I[:, :, None]
    P = np.pad(I3.astype(np.float64), ((r, r), (r, r), (0, 0)), mode="edge")
    out = np.zeros(I3.shape, dtype=np.float64)
    H, W = I3.shape[:2]
    for a in range(G.shape[0]):
        for b in range(G.shape[1]):
            out += G[a, b] * P[a:a + H, b:b + W, :]
    return out.astype(np.float32).reshape(I.shape)


def _cubic(t):
    """imresize's bicubic kernel (Keys, a = -0.5), support [-2, 2]."""
    t = abs(t)
    if t <= 1:
        return 1.5 * t ** 3 - 2.5 * t ** 2 + 1
    return -0.5 * t ** 3 + 2.5 * t ** 2 - 4 * t + 2 if t <= 2 else 0.0


def _resize_taps(n_in, n_out, method="bilinear"):
    """For every output index the source indices (clamped: replicate at the ends) and normalised weights of a triangle- or
    cubic-kernel resize at MATLAB's pixel-centre convention, the kernel stretched by 1/scale when shrinking (IPT's
    antialiasing).  Arrays [n_out, T]; unused taps have weight 0.  This tap list, summed left to right, IS our definition of
    imresize along one axis (the device kernel k_pyr_resize walks the same list)."""
    scale = n_out / n_in
    stretch = 1.0 if scale >= 1 else 1.0 / scale
    support = 1.0 if method == "bilinear" else 2.0
    width = support * stretch
    T = int(math.ceil(2 * width)) + 2
    x = (np.arange(n_out, dtype=np.float64) + 0.5) / scale - 0.5        # centre of each output pixel in input coordinates
    first = np.floor(x - width).astype(np.int64)
    idx = first[:, None] + np.arange(T, dtype=np.int64)[None, :]
    t = (idx.astype(np.float64) - x[:, None]) / stretch
    a = np.abs(t)
    if method == "bilinear":
        w = np.maximum(0.0, 1.0 - a)
    else:
        a2 = a * a
        a3 = a2 * a
        w = np.where(a <= 1.0, (1.5 * a3 - 2.5 * a2) + 1.0, np.where(a <= 2.0, ((-0.5 * a3 + 2.5 * a2) - 4.0 * a) + 2.0, 0.0))
    total = w[:, 0].copy()
    for k in range(1, T):
        total = total + w[:, k]
    return np.clip(idx, 0, n_in - 1), w / total[:, None]


def _resize_matrix(n_in, n_out, method="bilinear"):
    """The same as a dense [n_out, n_in] matrix (for tests)."""
    idx, w = _resize_taps(n_in, n_out, method)
    M = np.zeros((n_out, n_in), dtype=np.float64)
    for k in range(idx.shape[1]):
        np.add.at(M, (np.arange(n_out), idx[:, k]), w[:, k])
    return M


def resize(I, out_rows, out_cols, method="bilinear", out_dtype=np.float32):
    """imresize(I, [out_rows out_cols], method); method 'bilinear' or 'bicubic' (imresize's default).  Rows first, then
    columns, taps summed left to right in double, rounded to single once (out_dtype=np.float64: a double array stays double)."""
    I3 = (I if I.ndim == 3 else I[:, :, None]).astype(np.float64)
    ri, rw = _resize_taps(I3.shape[0], out_rows, method)
    ci, cw = _resize_taps(I3.shape[1], out_cols, method)
    T1 = rw[:, 0][:, None, None] * I3[ri[:, 0]]
    for k in range(1, ri.shape[1]):
        T1 = T1 + rw[:, k][:, None, None] * I3[ri[:, k]]
    out = cw[:, 0][None, :, None] * T1[:, ci[:, 0]]
    for k in range(1, ci.shape[1]):
        out = out + cw[:, k][None, :, None] * T1[:, ci[:, k]]
    return out.astype(out_dtype).reshape((out_rows, out_cols) + I.shape[2:])


def build(I0, I1, scl_factor=0.75, min_size=20, G=None):
    """Image pyramids of the two frames (finest first), as the drivers build them (:100-127)."""
    P0, P1 = [I0.astype(np.float32)], [I1.astype(np.float32)]
    while True:
        rows, cols = P0[-1].shape[:2]
        nr, nc = int(math.ceil(rows * scl_factor)), int(math.ceil(cols * scl_factor))
        P0.append(resize(P0[-1], nr, nc))
        P1.append(resize(P1[-1], nr, nc))
        P0[-2], P1[-2] = smooth(P0[-2], G), smooth(P1[-2], G)    # the level just left is smoothed after it has been resized
        if nr <= min_size or nc <= min_size:
            P0[-1], P1[-1] = smooth(P0[-1], G), smooth(P1[-1], G)
            return P0, P1


def build_dev(d0, d1, scl_factor=0.75, min_size=20, G=None, smooth_last=True, max_scales=None):
    """build() on the device: d0, d1 torch planes [C, ncols, nrows]; the same definitions, bit for bit (device.pyr_resize /
    pyr_smooth).  smooth_last=False: the symmetric stereo driver leaves its coarsest scale unsmoothed.  max_scales: param.scales
    of the drivers -- their loop `for scl=2:param.scales` simply ends there, so a scale reached by the limit stays unsmoothed
    (FlowEminND_llin_2D_v10.m:105-127)."""
    from . import device as dev
    G = gaussian5() if G is None else G
    P0, P1 = [d0], [d1]
    while max_scales is None or len(P0) < max_scales:
        ncols, nrows = P0[-1].shape[-2:]
        nr, nc = int(math.ceil(nrows * scl_factor)), int(math.ceil(ncols * scl_factor))
        P0.append(dev.pyr_resize(P0[-1], nr, nc))
        P1.append(dev.pyr_resize(P1[-1], nr, nc))
        P0[-2], P1[-2] = dev.pyr_smooth(P0[-2], G), dev.pyr_smooth(P1[-2], G)
        if nr <= min_size or nc <= min_size:
            if smooth_last:
                P0[-1], P1[-1] = dev.pyr_smooth(P0[-1], G), dev.pyr_smooth(P1[-1], G)
            return P0, P1
    return P0, P1


def median3(A):
    """medfilt2(A, [3 3], 'symmetric')"""
    P = np.pad(A.astype(np.float32), 1, mode="symmetric")
    rows, cols = A.shape
    return np.sort(np.stack([P[a:a + rows, b:b + cols] for a in range(3) for b in range(3)]), axis=0)[4]


def coarse_to_fine(P0, P1, run_level, scl_factor=0.75, median_before_resize=False):
    """`run_level(I0, I1, U, V) -> U, V` on MATLAB-shaped numpy arrays; returns the flow at the finest level.
    median_before_resize: FlowEminHS_elin_2D_v10.m:193-194 filters the up-scaled flow before resizing it;
    FlowEminND_llin_2D_v10.m:360-363 does not (its median is inside the level)."""
    U = np.zeros(P0[-1].shape[:2], dtype=np.float32)
    V = np.zeros_like(U)
    for scl in range(len(P0) - 1, -1, -1):
        U, V = run_level(P0[scl], P1[scl], U, V)
        if scl > 0:
            rows, cols = P0[scl - 1].shape[:2]
            U, V = U * np.float32(1.0 / scl_factor), V * np.float32(1.0 / scl_factor)
            if median_before_resize:
                U, V = median3(U), median3(V)
            U, V = resize(U, rows, cols), resize(V, rows, cols)
    return U, V
