"""Inference post-processing: run-length encoding and resize + threshold of predicted masks.
CPU: the numpy restatement (oracle/rle_numpy.py) against the golden vectors produced by the reference's own maskApi.c
(tests/golden/rle.npz, oracle/make_golden_rle.py) and, when the compiled reference library is present, against it directly.
GPU: librsis_hip.so (rsis_rle_encode / rsis_rle_to_string / rsis_mask_resize_threshold) against the oracle, bit-exact for
the integer / byte work."""
import os

import numpy as np
import pytest
import torch

from oracle import maskapi_ref, rle_numpy

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "rle.npz")


def _golden():
    g = np.load(GOLD)
    return [(g["mask%d" % i], g["counts%d" % i], g["string%d" % i].tobytes()) for i in range(int(g["n"]))]


def test_oracle_rle_matches_golden():
    for m, counts, s in _golden():
        c = rle_numpy.rle_counts(m)
        assert np.array_equal(c, counts)
        assert rle_numpy.rle_string(c) == s
        assert rle_numpy.rle_area(c) == int(m.sum())


@pytest.mark.skipif(not maskapi_ref.available(), reason="oracle/_ref/libmaskapi_ref.so not built (make -C oracle)")
def test_oracle_rle_matches_reference_library():
    rng = np.random.default_rng(7)
    for it in range(60):
        h, w = int(rng.integers(1, 50)), int(rng.integers(1, 50))
        blob = int(rng.integers(1, 6))
        m = np.kron((rng.random((-(-h // blob), -(-w // blob))) < rng.random()).astype(np.uint8), np.ones((blob, blob), np.uint8))[:h, :w]
        c, s = maskapi_ref.encode(m)
        c2 = rle_numpy.rle_counts(m)
        assert np.array_equal(c, c2) and s == rle_numpy.rle_string(c2)


def test_host_rle_to_string_matches_golden():
    """rsis_rle_to_string is a host function of the library: checked without a GPU"""
    import ctypes
    from rsis_amd._lib import lib
    L = lib()
    for _m, counts, s in _golden():
        c = np.ascontiguousarray(counts.astype(np.uint32))
        buf = ctypes.create_string_buffer(6 * len(c) + 8)
        n = L.rsis_rle_to_string(c.ctypes.data_as(ctypes.c_void_p), len(c), buf, len(buf))
        assert n == len(s) and buf.raw[:n] == s
    assert L.rsis_rle_to_string(c.ctypes.data_as(ctypes.c_void_p), len(c), buf, 2) == -1       # too small a buffer is an error


@pytest.mark.gpu
def test_device_rle_matches_oracle():
    from rsis_amd import eval_post
    from rsis_amd._lib import lib
    L = lib()
    rng = np.random.default_rng(11)
    cases = [m for m, _c, _s in _golden()]
    big = np.zeros((1024, 2048), np.uint8)
    big[100:900, 300:1800] = 1
    big[500, 1000] = 0
    cases += [big, np.ones((37, 1025), np.uint8), np.zeros((300, 17), np.uint8),
              (rng.random((257, 129)) < 0.5).astype(np.uint8)]      # worst case: a change at almost every pixel
    for m in cases:
        h, w = m.shape
        col = torch.from_numpy(np.ascontiguousarray(m.T.reshape(1, -1))).cuda()          # column-major, as the kernel expects
        dicts = eval_post._rle_dicts(L, col, 1, h * w, h, w)
        c = rle_numpy.rle_counts(m)
        assert dicts[0]["size"] == [h, w]
        assert dicts[0]["counts"] == rle_numpy.rle_string(c), (h, w)
    # several masks of one image in one launch
    ms = [(rng.random((40, 56)) < p).astype(np.uint8) for p in (0.0, 0.2, 0.9, 1.0)]
    col = torch.from_numpy(np.stack([m.T.reshape(-1) for m in ms])).cuda()
    dicts = eval_post._rle_dicts(L, col, len(ms), 40 * 56, 40, 56)
    for d, m in zip(dicts, ms):
        assert d["counts"] == rle_numpy.rle_string(rle_numpy.rle_counts(m))


@pytest.mark.gpu
@pytest.mark.parametrize("Hm,Wm,h,w", [(16, 16, 16, 16), (64, 64, 100, 132), (32, 48, 21, 35), (256, 256, 375, 500), (8, 8, 1, 1)])
def test_device_resize_threshold_matches_oracle(Hm, Wm, h, w):
    """resample + threshold + ignore + area + encoding against scipy.ndimage.zoom(order=1) in float64 (oracle restatement of
    eval.py:96-117); pixels whose interpolated value is within 1e-5 of the threshold may legitimately differ (fp32 vs fp64)"""
    from rsis_amd import eval_post
    rng = np.random.default_rng(3)
    n = 3
    prob = rng.random((n, Hm, Wm)).astype(np.float32)
    ignore = (rng.random((h, w)) < 0.1).astype(np.uint8)
    th = 0.5
    segs, areas, raws = eval_post.encode_masks(torch.from_numpy(prob).cuda(), h, w, th, ignore)
    from scipy.ndimage import zoom
    for k in range(n):
        seg, raw = rle_numpy.resize_threshold(prob[k], h, w, th, ignore)
        z = zoom(prob[k].astype(np.float64).reshape(Hm, Wm, 1), [float(h) / Hm, float(w) / Wm, 1], order=1).reshape(h, w)
        near = np.abs(z - th) < 1e-5
        if not near.any():
            assert segs[k]["counts"] == rle_numpy.rle_string(rle_numpy.rle_counts(seg))
            assert raws[k]["counts"] == rle_numpy.rle_string(rle_numpy.rle_counts(raw))
            assert int(areas[k]) == int(seg.sum())
        else:                                                       # decode ours and compare away from the threshold
            assert abs(int(areas[k]) - int(seg.sum())) <= int(near.sum())


@pytest.mark.gpu
def test_resize_mask_reference_signature():
    import argparse
    from rsis_amd import eval_post
    rng = np.random.default_rng(9)
    a = argparse.Namespace(mask_th=0.5, min_size=0.001)
    p = rng.random((64, 64)).astype(np.float32)
    seg, ok, raw = eval_post.resize_mask(a, p, 120, 90)
    want, _ = rle_numpy.resize_threshold(p, 120, 90, 0.5)
    assert ok and seg["size"] == [120, 90] and seg["counts"] == rle_numpy.rle_string(rle_numpy.rle_counts(want)) and raw["counts"] == seg["counts"]
    _seg, ok2, _ = eval_post.resize_mask(a, np.zeros((64, 64), np.float32), 120, 90)
    assert not ok2                                                  # eval.py:113-114: fewer than min_size * h * w pixels


@pytest.mark.gpu
def test_largest_component_matches_scipy():
    """rsis_largest_component against scipy.ndimage.label with the full 3x3 structure (== skimage.measure.label's default
    connectivity for 2-D, eval_cityscapes.py:139) + the most frequent label; ties resolved towards the first label in raster
    order on both sides"""
    from scipy import ndimage
    from rsis_amd import eval_post
    rng = np.random.default_rng(21)
    cases = []
    for (h, w, p, blob) in [(1, 1, 1.0, 1), (7, 9, 0.5, 1), (64, 48, 0.45, 2), (100, 132, 0.55, 3), (256, 512, 0.6, 4), (33, 65, 0.0, 1),
                            (40, 40, 1.0, 1), (128, 128, 0.5, 1)]:
        m = np.kron((rng.random((-(-h // blob), -(-w // blob))) < p).astype(np.uint8), np.ones((blob, blob), np.uint8))[:h, :w]
        cases.append(np.ascontiguousarray(m))
    spiral = np.zeros((41, 41), np.uint8)                 # a long thin component: deep union-find chains
    for k in range(0, 20, 2):
        spiral[k, k:41 - k] = 1; spiral[k:41 - k, 40 - k] = 1; spiral[40 - k, k:41 - k] = 1; spiral[k + 2:41 - k, k] = 1
    cases.append(spiral)
    diag = np.eye(30, dtype=np.uint8)                     # only diagonal contacts: 8-connectivity matters
    cases.append(diag)
    for m in cases:
        lab, nlab = ndimage.label(m, structure=np.ones((3, 3)))
        want = np.zeros_like(m)
        if nlab:
            cnt = np.bincount(lab.reshape(-1))[1:]
            want = (lab == (int(np.argmax(cnt)) + 1)).astype(np.uint8)        # argmax: first maximum = first in raster order
        got = eval_post.largest_component(torch.from_numpy(m[None]).cuda())[0].cpu().numpy()
        assert np.array_equal(got, want), m.shape
    # a batch in one call
    batch = np.stack([cases[2], cases[2][::-1].copy(), np.zeros_like(cases[2])])
    got = eval_post.largest_component(torch.from_numpy(batch).cuda()).cpu().numpy()
    for k in range(3):
        lab, nlab = ndimage.label(batch[k], structure=np.ones((3, 3)))
        want = (lab == (int(np.argmax(np.bincount(lab.reshape(-1))[1:])) + 1)).astype(np.uint8) if nlab else np.zeros_like(batch[k])
        assert np.array_equal(got[k], want)


@pytest.mark.gpu
def test_eval_driver_writes_coco_records(tmp_path):
    """python -m rsis_amd.eval --synthetic: inference + device post-processing + COCO-style records (eval.py:254-345)"""
    import json
    from rsis_amd.args import get_parser
    from rsis_amd.eval import Evaluate
    a = get_parser().parse_args(["--synthetic", "-model_name", "evtest", "-batch_size", "2", "-maxseqlen", "3", "-hidden_size", "32",
                                 "-synthetic_batches", "4", "-stop_th", "0.0", "-class_th", "0.0", "-min_size", "0.0"])
    a.models_root, a.imsize, a.num_classes = str(tmp_path), 64, 5
    torch.manual_seed(0)
    preds = Evaluate(a).run_eval()
    assert len(preds) == 2 * 3 * 4                               # images x timesteps x (classes - eos), nothing filtered at th = 0
    rec = preds[0]
    assert set(rec) == {"image_id", "category_id", "category_name", "segmentation", "score"} and rec["segmentation"]["size"] == [64, 64]
    with open(os.path.join(str(tmp_path), "evtest", "evtest_test_predictions.json")) as f:
        assert len(json.load(f)) == len(preds)
    # the record's RLE decodes (oracle side) to as many pixels as the thresholded mask has
    counts = rle_numpy_decode_area(rec["segmentation"]["counts"].encode("ascii"))
    assert 0 <= counts <= 64 * 64


def rle_numpy_decode_area(s):
    """area of a COCO compressed RLE string (inverse of rle_string, oracle side)"""
    cnts, p, m = [], 0, 0
    while p < len(s):
        x, k, more = 0, 0, True
        while more:
            c = s[p] - 48
            x |= (c & 0x1F) << (5 * k)
            more = bool(c & 0x20)
            p += 1
            k += 1
            if not more and (c & 0x10):
                x |= -1 << (5 * k)
        if m > 2:
            x += cnts[m - 2]
        cnts.append(x)
        m += 1
    return int(sum(cnts[1::2]))


def test_leaves_label_image_host_semantics():
    """eval_leaves.py:105-120 restated (rsis_amd.eval_post.leaves_label_image; no device work): per-mask bytescale before the resize (so
    the threshold is relative to the mask's own min / max), only timesteps whose stop probability exceeds -class_th paint, later
    timesteps overwrite, timestep 0 paints label 0."""
    import argparse
    from rsis_amd import eval_post
    a = argparse.Namespace(mask_th=0.5, class_th=0.5)
    T, Hm, Wm, h, w = 4, 8, 8, 16, 24
    m = np.zeros((T, Hm, Wm), np.float32)
    m[0, :4, :4] = 0.9          # timestep 0: label 0 -> invisible
    m[1, 2:6, 2:6] = 0.8
    m[2, 4:, 4:] = 0.3          # max 0.3 < mask_th, but bytescale stretches it to 255: it DOES paint (the reference's quirk)
    m[3, :, :] = 0.95           # would cover everything, but its stop probability is below class_th
    stop = np.array([[0.9], [0.9], [0.9], [0.2]], np.float32)
    lab = eval_post.leaves_label_image(a, torch.from_numpy(m), torch.from_numpy(stop), h, w)
    assert lab.shape == (h, w) and lab.dtype == np.uint8
    assert set(np.unique(lab)) == {0, 1, 2}
    assert lab[h // 2 - 1, w // 2 - 2] == 1 and lab[h - 1, w - 1] == 2 and lab[1, 1] == 0
    assert lab[h // 2 + 2, w // 2 + 3] == 2           # overlap of masks 1 and 2: the later timestep wins
    # imresize = bytescale + PIL bilinear: a constant array maps to 0, a 0 / 1 array to 0 / 255
    assert int(eval_post.imresize(np.full((4, 4), 0.7, np.float32), [8, 8]).max()) == 0
    r = eval_post.imresize(np.array([[0, 1], [0, 1]], np.uint8) * np.uint8(255), [2, 8])
    assert r.shape == (2, 8) and r[0, 0] == 0 and r[0, -1] == 255 and (np.diff(r[0].astype(int)) >= 0).all()


@pytest.mark.gpu
def test_cityscapes_and_leaves_writers(tmp_path):
    """the on-disk outputs of eval_cityscapes.py:118-167 (one PNG of the largest connected component per (timestep, class), resized to
    the image size, + `<sample>.txt` lines `<png> <cityscapes id> <class prob * objectness>`) and of eval_leaves.py:121-125 (label PNG
    renamed rgb -> label) from test()-shaped outputs"""
    import argparse
    from PIL import Image
    from rsis_amd import eval_post
    a = argparse.Namespace(mask_th=0.5, class_th=0.5)
    T, Hm, Wm, h, w, C = 3, 16, 32, 32, 64, 9
    g = np.random.default_rng(5)
    probs = np.zeros((T, Hm, Wm), np.float32)
    probs[0, 2:10, 3:20] = 0.9
    probs[0, 13:15, 28:31] = 0.8         # a second, smaller component: must not survive
    probs[1, 5:7, 5:7] = 0.7
    cls = g.random((T, C)).astype(np.float32)
    stop = np.array([[0.9], [0.6], [0.1]], np.float32)
    res = str(tmp_path / "city_results")
    lines = eval_post.write_cityscapes_results(a, "frankfurt_000000_000294", torch.from_numpy(probs).cuda(), torch.from_numpy(cls),
                                               torch.from_numpy(stop), h, w, res, "city_masks")
    assert len(lines) == T * (C - 1)
    txt = open(os.path.join(res, "frankfurt_000000_000294.txt")).read().splitlines()
    assert len(txt) == len(lines)
    name, cid, score = txt[0].split(" ")
    assert name == "city_masks/frankfurt_000000_000294_0.png" and cid == "24" and abs(float(score) - cls[0, 1] * 0.9) < 1e-6
    name, cid, score = txt[(C - 1) + 7].split(" ")
    assert name.endswith("_%d.png" % (C - 1 + 7)) and cid == "33" and abs(float(score) - float(cls[1, 8]) * float(stop[1, 0])) < 1e-6
    im = np.asarray(Image.open(os.path.join(res, "city_masks", "frankfurt_000000_000294_0.png")))
    assert im.shape == (h, w) and im.dtype == np.uint8 and im[10, 20] == 255 and im[28, 59] == 0 and im[0, 0] == 0
    empty = np.asarray(Image.open(os.path.join(res, "city_masks", "frankfurt_000000_000294_%d.png" % (2 * (C - 1)))))
    assert empty.max() == 0                                              # timestep 2 has no pixel above the threshold
    p = eval_post.write_leaves_result(a, "plant007_rgb", torch.from_numpy(probs).cuda(), torch.from_numpy(stop), h, w, str(tmp_path / "A1"))
    assert p.endswith("plant007_label.png")
    lab = np.asarray(Image.open(p))
    assert lab.shape == (h, w) and set(np.unique(lab)) <= {0, 1} and lab[11, 11] == 1


@pytest.mark.gpu
def test_eval_leaves_and_eval_cityscapes_drivers(tmp_path):
    """The two dataset evaluation scripts end to end (reference src/eval_leaves.py:91-125, src/eval_cityscapes.py:96-174; ADVICE r5:
    the writers had no caller): a checkpoint written by save_checkpoint is loaded, test() runs over the split, result files appear in
    the reference's layout.  eval_leaves over the 5-image val split of a synthesised CVPPP directory with batch 2 (so the last batch
    is ONE image: the reference would index past it); one label image is re-derived from test() by hand and must be identical."""
    import numpy as np
    from PIL import Image
    from rsis_amd.args import get_parser
    from rsis_amd.dataloader.leaves import synthesize_leaves_dir
    from rsis_amd.eval_post import leaves_label_image
    from rsis_amd.modules import FeatureExtractor, RSIS
    from rsis_amd.test import test as hip_test
    from rsis_amd.utils.utils import save_checkpoint
    from rsis_amd import eval_cityscapes, eval_leaves
    d = synthesize_leaves_dir(str(tmp_path / "A1"), n=101, size=(80, 96), seed=5)
    models = str(tmp_path / "models")
    a = get_parser().parse_args(["-model_name", "lv", "-dataset", "leaves", "-leaves_dir", d, "-leaves_test_dir", d, "-eval_split", "val",
                                 "-batch_size", "2", "-maxseqlen", "4", "-gt_maxseqlen", "6", "-num_classes", "2", "-imsize", "64",
                                 "--resize", "-hidden_size", "32", "-num_workers", "2", "-class_th", "0.0", "-models_root", models])
    torch.manual_seed(3)
    enc, dec = FeatureExtractor(a).cuda(), RSIS(a).cuda()
    a.epoch_resume, a.best_val_loss = 0, 0.0
    save_checkpoint(a, enc, dec, torch.optim.Adam(enc.parameters()), torch.optim.Adam(dec.parameters()), root=models)
    ev = eval_leaves.Evaluate(a)
    assert len(ev.sample_list) == 5 and len(ev.loader) == 3
    written = ev.create_figures()
    assert len(written) == 5 and all(p.endswith("_label.png") and os.path.exists(p) for p in written)
    assert os.path.dirname(written[0]) == os.path.join(models, "lv", "lv_results", "A1")
    img = np.asarray(Image.open(written[0]))
    assert img.shape == (80, 96) and img.dtype == np.uint8 and img.max() <= 3
    # by hand: the first val batch through test(), the first image's label map
    x = next(iter(ev.loader))[0]
    masks, _c, stops = hip_test(a, ev.encoder, ev.decoder, x)
    want = leaves_label_image(a, masks[0].view(4, 64, 64), stops[0], 80, 96)
    assert (img == want).all()

    c = get_parser().parse_args(["--synthetic", "-model_name", "cs", "-batch_size", "2", "-maxseqlen", "3", "-hidden_size", "32",
                                 "-synthetic_batches", "4", "-num_classes", "9", "-imsize", "64", "-models_root", models])
    n = eval_cityscapes.Evaluate(c).create_figures()
    assert n == 2 * 3 * 8                                      # images x timesteps x foreground classes
    res = os.path.join(models, "cs", "cs_results")
    lines = open(os.path.join(res, "synthetic_000000.txt")).read().splitlines()
    assert len(lines) == 24
    png, cid, score = lines[0].split(" ")
    assert int(cid) == 24 and 0.0 <= float(score) <= 1.0
    assert np.asarray(Image.open(os.path.join(res, png))).shape == (128, 128)     # the "original" size: twice the input
