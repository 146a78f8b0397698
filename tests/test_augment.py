"""Affine augmentation: the numpy oracle against fixtures captured from the unmodified reference
(tests/golden/affine.npz, oracle/make_golden_affine.py), and the HIP kernel + host mirror against the oracle."""
import os
import random

import numpy as np
import pytest
import torch

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "affine.npz")
SEEDS = [1, 2, 3, 4, 5]


def _zoom_range(z):
    return (z, max(z * 2, 1.0))


@pytest.mark.parametrize("seed", SEEDS)
def test_oracle_affine_matches_reference(seed):
    from oracle import affine_numpy as A
    g = np.load(GOLD)
    m = g["m%d" % seed]
    for k in ("img", "ids"):
        got = A.affine_nearest(g["%s%d" % (k, seed)], m)
        assert np.array_equal(got, g["%s_t%d" % (k, seed)]), "%s seed %d: %d pixels differ" % (
            k, seed, int((got != g["%s_t%d" % (k, seed)]).sum()))


@pytest.mark.parametrize("seed", SEEDS)
def test_oracle_random_matrix_matches_reference(seed):
    from oracle import affine_numpy as A
    g = np.load(GOLD)
    rot, th, tw, sh, zoom = [float(v) for v in g["args%d" % seed]]
    _, H, W = g["img%d" % seed].shape
    m = A.random_affine_matrix(random.Random(seed), H, W, rot, (th, tw), sh, _zoom_range(zoom))
    np.testing.assert_allclose(m, g["m%d" % seed], rtol=2e-6, atol=1e-6)


@pytest.mark.parametrize("seed", SEEDS)
def test_host_random_affine_matches_reference(seed):
    """the product's RandomAffine mirror draws the same numbers in the same order and composes the same float32 matrix"""
    from rsis_amd.dataloader.augment import RandomAffine
    g = np.load(GOLD)
    rot, th, tw, sh, zoom = [float(v) for v in g["args%d" % seed]]
    _, H, W = g["img%d" % seed].shape
    random.seed(seed)
    aug = RandomAffine(rotation_range=rot, translation_range=(th, tw), shear_range=sh, zoom_range=_zoom_range(zoom), interp="nearest")
    m = aug.matrix(H, W)
    np.testing.assert_allclose(m.numpy(), g["m%d" % seed], rtol=2e-6, atol=1e-6)


def test_affine_needs_the_gpu():
    from rsis_amd.dataloader.augment import affine_nearest
    with pytest.raises(Exception):
        affine_nearest(torch.zeros(1, 4, 4), torch.eye(3))


@pytest.mark.gpu
@pytest.mark.parametrize("seed", SEEDS)
def test_hip_affine_matches_reference_fixtures(seed):
    from rsis_amd.dataloader.augment import affine_nearest
    g = np.load(GOLD)
    m = torch.from_numpy(g["m%d" % seed])
    for k in ("img", "ids"):
        got = affine_nearest(torch.from_numpy(g["%s%d" % (k, seed)]).cuda(), m).cpu().numpy()
        assert np.array_equal(got, g["%s_t%d" % (k, seed)]), "%s seed %d" % (k, seed)


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(2, 3, 40, 56), (5, 1, 7, 9), (1, 21, 256, 256)])
def test_hip_affine_batch_matches_oracle(shape):
    """batched call (one matrix per sample) against the numpy oracle on seeded inputs"""
    from oracle import affine_numpy as A
    from rsis_amd.dataloader.augment import affine_nearest
    N, C, H, W = shape
    rng = np.random.default_rng(11)
    x = rng.normal(0, 1, shape).astype(np.float32)
    r = random.Random(7)
    ms = np.stack([A.random_affine_matrix(r, H, W, 30, 0.2, 0.2, (0.6, 1.3)) for _ in range(N)])
    got = affine_nearest(torch.from_numpy(x).cuda(), torch.from_numpy(ms)).cpu().numpy()
    for n in range(N):
        want = A.affine_nearest(x[n], ms[n])
        assert np.array_equal(got[n], want), "sample %d: %d pixels differ" % (n, int((got[n] != want).sum()))
