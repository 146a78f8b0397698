"""TEST INFRASTRUCTURE -- CPU restatement (numpy, float32) of the reference's affine augmentation.  Only tests/, smoke() and the
bench's cpu_baseline leg may import this.

affine_nearest follows th_affine2d(mode='nearest', center=True) + th_nearest_interp2d
(/root/reference/src/dataloader/transforms/utils.py:67-147): output pixel (i, j) of every channel takes the input pixel at
round(clamp(A @ (i - ci, j - cj) + b + (ci, cj))) with ci = H/2 - 0.5, cj = W/2 - 0.5, all in float32, round half to even.
random_affine_matrix follows RandomAffine / Rotate / Translate / Shear / Zoom (transforms.py:23-103,285-333,445-500,592-622,
723-774): float32 3x3 matrices multiplied in the order rotation @ translation @ shear @ zoom; the draws come from python's
`random` in that order (one for the rotation, two for the translation, one for the shear, two for the zoom).
Pinned against the unmodified reference by tests/golden/affine.npz (oracle/make_golden_affine.py)."""
import math

import numpy as np


def affine_nearest(x, m):
    x = np.asarray(x, dtype=np.float32)
    m = np.asarray(m, dtype=np.float32)
    C, H, W = x.shape
    A, b = m[:2, :2], m[:2, 2]
    ci, cj = np.float32(H / 2.0 - 0.5), np.float32(W / 2.0 - 0.5)
    ii, jj = np.meshgrid(np.arange(H, dtype=np.float32), np.arange(W, dtype=np.float32), indexing="ij")
    ii, jj = ii - ci, jj - cj
    ni = (ii * A[0, 0] + jj * A[0, 1]) + b[0] + ci          # (utils.py:112: coords.bmm(A^T) + b, then un-centre :116-117)
    nj = (ii * A[1, 0] + jj * A[1, 1]) + b[1] + cj
    si = np.rint(np.clip(ni, 0, H - 1)).astype(np.int64)    # :131-132 clamp, then round (half to even)
    sj = np.rint(np.clip(nj, 0, W - 1)).astype(np.int64)
    return x[:, si, sj]


def rotation_matrix(deg):
    t = math.pi / 180 * deg
    return np.array([[math.cos(t), -math.sin(t), 0], [math.sin(t), math.cos(t), 0], [0, 0, 1]], dtype=np.float32)


def translation_matrix(frac_h, frac_w, H, W):
    return np.array([[1, 0, frac_h * H], [0, 1, frac_w * W], [0, 0, 1]], dtype=np.float32)


def shear_matrix(deg):
    t = (math.pi * deg) / 180
    return np.array([[1, -math.sin(t), 0], [0, math.cos(t), 0], [0, 0, 1]], dtype=np.float32)


def zoom_matrix(zx, zy):
    return np.array([[zx, 0, 0], [0, zy, 0], [0, 0, 1]], dtype=np.float32)


def random_affine_matrix(rng, H, W, rotation, translation, shear, zoom_range):
    """rng: a `random.Random`-like object (uniform).  translation: float or (h, w) fractions."""
    th, tw = translation if isinstance(translation, (tuple, list)) else (translation, translation)
    m = rotation_matrix(rng.uniform(-rotation, rotation))
    fh = rng.uniform(-th, th)
    fw = rng.uniform(-tw, tw)
    m = m @ translation_matrix(fh, fw, H, W)
    m = m @ shear_matrix(rng.uniform(-shear, shear))
    zx = rng.uniform(zoom_range[0], zoom_range[1])
    zy = rng.uniform(zoom_range[0], zoom_range[1])
    return (m @ zoom_matrix(zx, zy)).astype(np.float32)
