#!/usr/bin/env python
"""TEST INFRASTRUCTURE.  Generates tests/golden/affine.npz by importing the UNMODIFIED reference augmentation code
(/root/reference/src/dataloader/transforms/{utils,transforms}.py: th_affine2d, RandomAffine) in this container.
Fixtures: for a few seeds / shapes, the lazily composed 3x3 matrix of RandomAffine (python `random` seeded) and the
nearest-neighbour transform of an image-like float tensor and an instance-id map.  Run:  python oracle/make_golden_affine.py"""
import os
import random
import sys

import numpy as np
import torch

REF = "/root/reference/src/dataloader/transforms"
sys.path.insert(0, REF)
import transforms as T  # noqa: E402  (the reference module; py2-style `from utils import ...` resolves through sys.path)
import utils as U  # noqa: E402

CASES = [  # (seed, C, H, W, rotation, translation, shear, zoom)
    (1, 3, 16, 16, 10, 0.1, 0.1, 0.7),
    (2, 1, 24, 40, 25, (0.2, 0.05), 0.3, 0.9),
    (3, 3, 33, 17, 5, 0.0, 0.0, 1.0),
    (4, 2, 64, 64, 45, 0.15, 0.2, 0.5),
    (5, 1, 8, 8, 180, 0.3, 0.4, 0.6),
]
out = {}
for seed, C, H, W, rot, tr, sh, zoom in CASES:
    random.seed(seed)
    aug = T.RandomAffine(rotation_range=rot, translation_range=tr, shear_range=sh, zoom_range=(zoom, max(zoom * 2, 1.0)),
                         interp="nearest", lazy=True)
    rng = np.random.default_rng(seed)
    img = torch.from_numpy(rng.normal(0, 1, (C, H, W)).astype(np.float32))
    ids = torch.from_numpy(rng.integers(0, 7, (1, H, W)).astype(np.float32))
    m = aug(img)                                   # 3x3 float32, rotation @ translation @ shear @ zoom
    out["m%d" % seed] = m.numpy()
    out["img%d" % seed] = img.numpy()
    out["ids%d" % seed] = ids.numpy()
    out["img_t%d" % seed] = U.th_affine2d(img, m, mode="nearest").numpy()
    out["ids_t%d" % seed] = U.th_affine2d(ids, m, mode="nearest").numpy()
    out["args%d" % seed] = np.array([rot, tr[0] if isinstance(tr, tuple) else tr, tr[1] if isinstance(tr, tuple) else tr, sh, zoom],
                                    dtype=np.float64)
dst = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "affine.npz")
np.savez_compressed(dst, **out)
print("wrote", os.path.normpath(dst), {k: v.shape for k, v in out.items() if k.startswith("img_t")})
