"""CVPPP A1 leaves data layer (rsis_amd/dataloader/leaves.py; reference src/dataloader/leaves.py + dataset.py): host part on the CPU,
the device batch path and a `train.py` run of BASELINE configs[0]'s flag set on the GPU."""
import argparse
import os
import sys

import numpy as np
import pytest
import torch


def _args(d, **kw):
    a = argparse.Namespace(gt_maxseqlen=10, batch_size=4, leaves_dir=d, leaves_test_dir=d, rotation=10, translation=0.1, shear=0.1,
                           zoom=0.7)
    for k, v in kw.items():
        setattr(a, k, v)
    return a


def test_leaves_dataset_host_side(tmp_path):
    from rsis_amd.dataloader.leaves import LeavesDataset, synthesize_leaves_dir
    import random
    d = synthesize_leaves_dir(str(tmp_path / "A1"), n=100, size=(120, 136), seed=1)
    tr = LeavesDataset(_args(d), split="train", imsize=64)
    va = LeavesDataset(_args(d), split="val", imsize=64)
    assert len(tr) == 96 and len(va) == 4                      # leaves.py:72-94: the first 96 train, the rest validate
    assert tr.get_classes() == ["<eos>", "leaf"]
    img, ins, seg = tr.get_raw_sample(3)
    assert img.size == (136, 120) and ins.shape == (120, 136)
    assert set(np.unique(seg)) <= {0, 1} and ((seg > 0) == (ins > 0)).all()      # :105-106
    im, m = tr.host_item(3, random.Random(0))
    assert im.shape == (3, 64, 64) and im.dtype == np.uint8 and m.shape == (64, 64) and m.dtype == np.int32
    # the instance map went through a NEAREST resize: no new ids
    assert set(np.unique(m)) <= set(np.unique(ins))
    # square resize instead of scale + crop
    sq = LeavesDataset(_args(d), split="train", imsize=48, resize=True)
    im2, m2 = sq.host_item(0, random.Random(0))
    assert im2.shape == (3, 48, 48) and m2.shape == (48, 48)


def test_decoded_samples_are_cached_and_never_mutated(tmp_path, monkeypatch):
    """the deterministic part of a sample (decode + resizes) is kept after its first evaluation: later draws (flip, crop) give exactly what an
    uncached dataset gives, the cached arrays stay untouched, and RSIS_LOADER_CACHE_MB bounds / disables the cache"""
    from rsis_amd.dataloader.leaves import LeavesDataset, synthesize_leaves_dir
    import random
    d = synthesize_leaves_dir(str(tmp_path / "A1"), n=100, size=(120, 136), seed=2)
    cached = LeavesDataset(_args(d), split="train", imsize=64, augment=True)
    monkeypatch.setenv("RSIS_LOADER_CACHE_MB", "0")
    plain = LeavesDataset(_args(d), split="train", imsize=64, augment=True)
    assert plain._cache_limit == 0
    first = [cached.host_item(i, random.Random(100 + i)) for i in range(6)]
    snap = {i: (a.copy(), b.copy()) for i, (a, b) in cached._cache.items()}
    assert len(snap) == 6 and cached._cache_bytes == sum(a.nbytes + b.nbytes for a, b in snap.values())
    for rep in range(3):                                       # other random draws on cache hits
        for i in range(6):
            got, want = cached.host_item(i, random.Random(7 * rep + i)), plain.host_item(i, random.Random(7 * rep + i))
            assert (got[0] == want[0]).all() and (got[1] == want[1]).all()
            got[0][...] = 0                                     # the caller owns what it gets ...
    for i, (a, b) in cached._cache.items():                    # ... the cache is what was decoded
        assert (a == snap[i][0]).all() and (b == snap[i][1]).all()
    again = [cached.host_item(i, random.Random(100 + i)) for i in range(6)]
    assert all((p[0] == q[0]).all() and (p[1] == q[1]).all() for p, q in zip(first, again))
    assert not plain._cache
    monkeypatch.setenv("RSIS_LOADER_CACHE_MB", "0.03")          # 31 KB: room for one 64-pixel sample (3 x 64 x 72 + 64 x 72 bytes), not two
    small = LeavesDataset(_args(d), split="train", imsize=64)
    for i in range(4):
        small.host_item(i, random.Random(i))
    assert len(small._cache) == 1 and small._cache_bytes <= small._cache_limit


@pytest.mark.gpu
def test_device_loader_targets_equal_reference_sequence_from_masks(tmp_path):
    """without augmentation the device batch must be exactly batch_to_var(sequence_from_masks(host maps)) and the normalised image"""
    import random
    from rsis_amd.dataloader import sequence_from_masks
    from rsis_amd.dataloader.leaves import MEAN, STD, DeviceLoader, LeavesDataset, synthesize_leaves_dir
    d = synthesize_leaves_dir(str(tmp_path / "A1"), n=100, size=(96, 112), seed=2)
    ds = LeavesDataset(_args(d), split="train", imsize=64)
    dl = DeviceLoader(ds, 4, shuffle=False, num_workers=2, seed=5)
    assert len(dl) == 24
    # replay the loader's random stream to rebuild the first batch on the host
    rng = random.Random(5 * 1000003 + 1)       # DeviceLoader's per-sample stream of rank 0 (the shuffle order has its own, common to all ranks)
    seeds = [rng.getrandbits(32) for _ in range(4)]
    host = [ds.host_item(i, random.Random(s)) for i, s in zip(range(4), seeds)]
    x, y_mask, y_class, sw_mask, sw_class = next(iter(dl))
    assert x.shape == (4, 3, 64, 64) and y_mask.shape == (4, 10, 64 * 64)
    for b, (im, ins) in enumerate(host):
        want = ((torch.from_numpy(im.copy()).float() / 255.0) - torch.tensor(MEAN).view(3, 1, 1)) / torch.tensor(STD).view(3, 1, 1)
        assert float((x[b].cpu() - want).abs().max()) < 1e-6
        t = sequence_from_masks(ins, (ins > 0).astype(np.int64), 10)
        areas = t[:, :-3].sum(1)
        if len(set(areas[areas > 0])) == int((areas > 0).sum()):        # (equal-area instances: order is implementation-defined)
            assert np.array_equal(y_mask[b].cpu().numpy(), t[:, :-3].astype(np.float32))
        assert np.array_equal(y_class[b].cpu().numpy(), t[:, -3].astype(np.int64))
        assert np.array_equal(sw_mask[b].cpu().numpy(), t[:, -2].astype(np.float32))
        assert np.array_equal(sw_class[b].cpu().numpy(), t[:, -1].astype(np.float32))


@pytest.mark.gpu
def test_train_py_runs_configs0_flag_set_on_leaves(tmp_path):
    """BASELINE configs[0] (CVPPP A1, ResNet-101 encoder, T=16, batch 2, train.py) end to end on the device, augmentation on, one
    epoch of a synthesised A1 directory: the losses are finite and a checkpoint directory is written."""
    import subprocess
    from rsis_amd.dataloader.leaves import synthesize_leaves_dir
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    d = synthesize_leaves_dir(str(tmp_path / "A1"), n=104, size=(80, 96), seed=3)
    models = str(tmp_path / "models")
    cmd = [sys.executable, "-m", "rsis_amd.train", "-dataset", "leaves", "-leaves_dir", d, "-leaves_test_dir", d, "-imsize", "64",
           "-batch_size", "2", "-maxseqlen", "16", "-gt_maxseqlen", "20", "-num_classes", "2", "--augment", "--log_term", "-max_epoch", "1",
           "-print_every", "16", "-model_name", "leaves_smoke", "-models_root", models, "-num_workers", "2", "-hidden_size", "32"]
    r = subprocess.run(cmd, cwd=root, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "Epoch 0:" in r.stdout and "nan" not in r.stdout.lower()
    assert os.path.exists(os.path.join(models, "leaves_smoke", "encoder.pt"))
