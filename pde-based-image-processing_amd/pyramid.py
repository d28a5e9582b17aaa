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


def _resize_matrix(n_in, n_out, method="bilinear"):
    """[n_out, n_in] matrix (rows summing to one) of a triangle- or cubic-kernel resize, antialiased when shrinking."""
    scale = n_out / n_in
    stretch = 1.0 if scale >= 1 else 1.0 / scale
    support = 1.0 if method == "bilinear" else 2.0
    kern = (lambda t: max(0.0, 1.0 - abs(t))) if method == "bilinear" else _cubic
    width = support * stretch
    M = np.zeros((n_out, n_in), dtype=np.float64)
    for o in range(n_out):
        x = (o + 0.5) / scale - 0.5                       # centre of output pixel o in input coordinates
        lo, hi = int(math.floor(x - width)), int(math.ceil(x + width))
        for i in range(lo, hi + 1):
            w = kern((i - x) / stretch)
            if w != 0:
                M[o, min(max(i, 0), n_in - 1)] += w       # replicate at the ends
        M[o] /= M[o].sum()
    return M


def resize(I, out_rows, out_cols, method="bilinear"):
    """imresize(I, [out_rows out_cols], method); method 'bilinear' or 'bicubic' (imresize's default)"""
    I3 = I if I.ndim == 3 else I[:, :, None]
    R, C = _resize_matrix(I3.shape[0], out_rows, method), _resize_matrix(I3.shape[1], out_cols, method)
    out = np.einsum("or,rck->ock", R, I3.astype(np.float64))
    out = np.einsum("pc,ock->opk", C, out)
    return out.astype(np.float32).reshape((out_rows, out_cols) + I.shape[2:])


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
