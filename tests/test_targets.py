"""Target-tensor construction against golden vectors produced by the reference's own
MyDataset.sequence_from_masks (tests/golden/targets.npz, oracle/make_golden_targets.py)."""
import os

import numpy as np
import pytest
import torch

from rsis_amd.dataloader import sequence_from_masks, targets_from_maps
from rsis_amd.utils.utils import batch_to_var

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "targets.npz")


def _cases():
    g = np.load(GOLD)
    return [(g["ins%d" % i], g["seg%d" % i], int(g["T%d" % i]), g["target%d" % i]) for i in range(int(g["n"]))]


def test_sequence_from_masks_matches_reference():
    for ins, seg, T, want in _cases():
        got = sequence_from_masks(ins, seg, T)
        assert got.shape == want.shape and got.dtype == np.float64
        assert np.array_equal(got, want)


def _check_device_targets(device):
    import argparse
    a = argparse.Namespace(use_gpu=False)
    for ins, seg, T, want in _cases():
        ym, yc, swm, swc = targets_from_maps(ins[None], seg[None], T, device=device)
        x = torch.zeros(1, 3, 2, 2)
        _x, rm, rc, rsm, rsc = batch_to_var(a, x, torch.from_numpy(want)[None])      # the reference's own split of the target
        assert torch.equal(ym.cpu(), rm) and torch.equal(yc.cpu(), rc)
        assert torch.equal(swm.cpu().double(), rsm.double()) and torch.equal(swc.cpu().double(), rsc.double())


def test_targets_from_maps_cpu_matches_reference():
    _check_device_targets("cpu")


@pytest.mark.gpu
def test_targets_from_maps_gpu_matches_reference():
    _check_device_targets("cuda")


def test_synthetic_targets_clamp_instances_to_gt_slots():
    """`-synthetic_instances` above `-gt_maxseqlen` must not overflow the target tensor (python -m rsis_amd.train -gt_maxseqlen 8)"""
    from rsis_amd.synthetic import synthetic_targets
    ym, yc, swm, swc = synthetic_targets(1, 2, 16, 16, gt_maxseqlen=5, n_inst=12, num_classes=7)
    assert ym.shape == (2, 5, 256) and float(swm.sum()) == 10 and float(swc.sum()) == 10


def test_leaves_loader_rank_sharding():
    """rsis_amd.dataloader.leaves.shard_batches: under one process per GPU every rank sees len(ds) // (B_rank * world) steps per epoch
    (the reference's len(ds) // B with B the global batch, train.py:46-49), the shards of a step are disjoint and their union is
    the global batch"""
    import random
    from rsis_amd.dataloader.leaves import shard_batches
    order = list(range(103))
    random.Random(5).shuffle(order)
    for world, bs in ((1, 4), (2, 4), (8, 2), (4, 32)):
        per_rank = [shard_batches(order, bs, r, world) for r in range(world)]
        steps = len(order) // (bs * world)
        assert all(len(p) == steps for p in per_rank)
        for k in range(steps):
            got = [i for p in per_rank for i in p[k]]
            assert all(len(p[k]) == bs for p in per_rank)
            assert sorted(got) == sorted(order[k * bs * world:(k + 1) * bs * world]) and len(set(got)) == len(got)
