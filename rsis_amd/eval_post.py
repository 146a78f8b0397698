"""Inference post-processing of the predicted instance masks on the GPU -- counterpart of reference src/eval.py:96-127
`resize_mask` (scipy.ndimage.zoom(order=1) to the original image size, `> args.mask_th`, ignore pixels cleared, minimum-size
test, pycocotools `mask.encode`), SURVEY.md section 8(f) row N2.  The resample + threshold + area and the run-length
encoding run in librsis_hip.so (rsis_mask_resize_threshold / rsis_rle_encode); only the few hundred run counts of each mask
come back to the host, where rsis_rle_to_string writes pycocotools' compressed text form."""
import ctypes
import os

import numpy as np
import torch

from ._lib import check, lib, ptr, stream


def encode_masks(prob, height, width, th, ignore=None, want_raw=True):
    """prob: (n, Hm, Wm) CUDA fp32 mask probabilities of one image -> (segs, areas, raws):
    segs / raws: lists of n COCO RLE dicts {'size': [height, width], 'counts': bytes} (raws: before the ignore mask; None when
    want_raw is False), areas: (n,) int64 numpy array of the set-pixel counts of segs."""
    L = lib()
    prob = prob.detach()
    if not prob.is_cuda or prob.dtype != torch.float32:
        raise ValueError("encode_masks: prob must be a CUDA float32 tensor")
    prob = prob.contiguous()
    n, Hm, Wm = prob.shape
    hw = height * width
    dev = prob.device
    seg = torch.empty((n, hw), dtype=torch.uint8, device=dev)
    raw = torch.empty((n, hw), dtype=torch.uint8, device=dev) if (want_raw and ignore is not None) else None
    area = torch.empty((n,), dtype=torch.int32, device=dev)
    ign = None
    if ignore is not None:
        ign = torch.as_tensor(np.ascontiguousarray(np.asarray(ignore).reshape(height, width)).astype(np.uint8)).to(dev)
    check(L.rsis_mask_resize_threshold(ptr(prob), n, Hm, Wm, ptr(ign), float(th), ptr(seg), ptr(raw), ptr(area), height, width, stream()),
          "rsis_mask_resize_threshold")
    segs = _rle_dicts(L, seg, n, hw, height, width)
    raws = None
    if want_raw:
        raws = _rle_dicts(L, raw, n, hw, height, width) if raw is not None else [dict(d) for d in segs]
    return segs, area.cpu().numpy().astype(np.int64), raws


def _rle_dicts(L, masks, n, hw, height, width):
    cap = min(hw + 1, 1 << 16)
    while True:
        counts = torch.empty((n, cap), dtype=torch.int32, device=masks.device)
        nruns = torch.empty((n,), dtype=torch.int32, device=masks.device)
        check(L.rsis_rle_encode(ptr(masks), n, hw, ptr(counts), cap, ptr(nruns), stream()), "rsis_rle_encode")
        nr = nruns.cpu().numpy()
        if (nr > 0).all():
            break
        cap = int(-nr.min())                                      # a mask with more runs than expected: retry with room for all
    m = int(nr.max())
    host = counts[:, :m].cpu().numpy().astype(np.uint32)          # ONE copy of the used prefix of every row
    out = []
    buf = ctypes.create_string_buffer(6 * m + 8)
    for k in range(n):
        row = np.ascontiguousarray(host[k, :nr[k]])
        ln = L.rsis_rle_to_string(row.ctypes.data_as(ctypes.c_void_p), int(nr[k]), buf, len(buf))
        if ln < 0:
            raise RuntimeError("rsis_rle_to_string: buffer too small")
        out.append({"size": [int(height), int(width)], "counts": buf.raw[:ln]})
    return out


def largest_component(masks):
    """masks: (n, h, w) CUDA uint8 / bool binary masks -> (n, h, w) uint8 mask of each one's largest 8-connected component
    (reference src/eval_cityscapes.py:131-150: skimage.measure.label + the most frequent label)."""
    m = masks.detach().to(torch.uint8).contiguous()
    if not m.is_cuda:
        raise ValueError("largest_component: masks must be a CUDA tensor")
    n, h, w = m.shape
    out = torch.empty_like(m)
    labels = torch.empty((n, h * w), dtype=torch.int32, device=m.device)
    counts = torch.empty_like(labels)
    best = torch.empty((n,), dtype=torch.int32, device=m.device)
    check(lib().rsis_largest_component(ptr(m), ptr(out), ptr(labels), ptr(counts), ptr(best), n, h, w, stream()),
          "rsis_largest_component")
    return out


def resize_mask(args, pred_mask, height, width, ignore_pixels=None):
    """reference src/eval.py:96-127, same arguments and return value: (segmentation, is_valid, segmentation_raw) with the two
    segmentations as COCO RLE dicts.  pred_mask: (Hm, Wm) numpy array or tensor of mask probabilities."""
    p = torch.as_tensor(np.asarray(pred_mask.detach().cpu() if torch.is_tensor(pred_mask) else pred_mask), dtype=torch.float32)
    segs, areas, raws = encode_masks(p.reshape(1, p.shape[-2], p.shape[-1]).cuda(), height, width, args.mask_th, ignore_pixels)
    is_valid = not (areas[0] < args.min_size * height * width)
    return segs[0], is_valid, raws[0]


# ------------------------------------------------------------------------------------------------------------------------------------
# Result writers of the two dataset-specific evaluation scripts: what they put on disk from the outputs of test().
# The reference's resampling goes through scipy.misc.imresize (removed from scipy; it was PIL underneath): an array becomes an 8-bit image
# by `bytescale` (min -> 0, max -> 255), is resized with PIL (default 'bilinear') and comes back as uint8.  Restated here with PIL.
# ------------------------------------------------------------------------------------------------------------------------------------
CITYSCAPES_CLASS_IDS = [24, 25, 26, 27, 28, 31, 32, 33]          # reference src/eval_cityscapes.py:113


def _bytescale(a):
    """scipy.misc.bytescale with its defaults: uint8 passes through, anything else is mapped linearly min..max -> 0..255"""
    a = np.asarray(a)
    if a.dtype == np.uint8:
        return a
    lo, hi = float(a.min()), float(a.max())
    scale = 255.0 / (hi - lo) if hi > lo else 1.0
    return ((a - lo) * scale + 0.5).clip(0, 255).astype(np.uint8)


def imresize(a, size):
    """scipy.misc.imresize(a, [h, w]) (interp='bilinear'): uint8 array of shape (h, w)"""
    from PIL import Image
    im = Image.fromarray(_bytescale(a), mode="L")
    return np.asarray(im.resize((int(size[1]), int(size[0])), resample=Image.BILINEAR))


def write_cityscapes_results(args, sample_idx, out_masks, class_scores, stop_probs, height, width, results_dir, masks_dir):
    """reference src/eval_cityscapes.py:118-167 for one image: per timestep the thresholded mask is reduced to its largest connected
    component (on the device: rsis_largest_component), scaled to 0 / 255, resized to the original image size and saved once per
    foreground class as `<masks_dir>/<sample>_<instance>.png`; `<results_dir>/<sample>.txt` gets one line `<png> <cityscapes class id>
    <class probability * objectness>` per (timestep, class) -- the format of the Cityscapes instance-level evaluation script.
    out_masks: (T, Hm, Wm) CUDA probabilities, class_scores: (T, C), stop_probs: (T, 1).  Returns the lines written."""
    from PIL import Image
    abs_masks = os.path.join(results_dir, masks_dir)
    os.makedirs(abs_masks, exist_ok=True)
    T = out_masks.shape[0]
    binm = (out_masks.detach() > args.mask_th)
    comp = largest_component(binm).cpu().numpy()                       # (T, Hm, Wm) 0 / 1; an empty mask stays empty
    cls = np.asarray(class_scores.detach().cpu(), dtype=np.float64)
    stop = np.asarray(stop_probs.detach().cpu(), dtype=np.float64).reshape(T, -1)
    lines, instance_id = [], 0
    for t in range(T):
        mask = imresize(comp[t] * np.uint8(255), [height, width])
        for i in range(cls.shape[1] - 1):                               # class 0 = <eos> (eval_cityscapes.py:156-162)
            name = "%s_%d.png" % (sample_idx, instance_id)
            score = cls[t][i + 1] * stop[t][0]
            Image.fromarray(mask, mode="L").save(os.path.join(abs_masks, name))
            cid = CITYSCAPES_CLASS_IDS[i] if i < len(CITYSCAPES_CLASS_IDS) else i + 1
            lines.append("%s/%s %s %s\n" % (masks_dir, name, cid, score))
            instance_id += 1
    with open(os.path.join(results_dir, sample_idx + ".txt"), "w") as f:
        f.writelines(lines)
    return lines


def leaves_label_image(args, out_masks, stop_probs, height, width):
    """reference src/eval_leaves.py:105-120 for one image: every timestep whose stop probability exceeds -class_th paints its mask --
    the probability map stretched to 0..255 (bytescale inside imresize: min -> 0, max -> 255 PER MASK), resized, thresholded at
    mask_th * 255 -- with the timestep index as label; later timesteps overwrite earlier ones and timestep 0 paints label 0 (the
    reference's behaviour, kept).  Returns the (height, width) uint8 label image (CVPPP A1 submission format)."""
    T = out_masks.shape[0]
    probs = np.asarray(out_masks.detach().cpu(), dtype=np.float32)
    stop = np.asarray(stop_probs.detach().cpu(), dtype=np.float64).reshape(T, -1)
    label = np.zeros([height, width])
    for t in range(T):
        mask = imresize(probs[t], [height, width])
        if stop[t][0] > args.class_th:
            label[mask > args.mask_th * 255] = t
    return label.astype(np.uint8)


def write_leaves_result(args, sample_idx, out_masks, stop_probs, height, width, results_dir):
    """eval_leaves.py:121-125: the label image as an 8-bit PNG named after the sample (`..._rgb` -> `..._label`)"""
    from PIL import Image
    os.makedirs(results_dir, exist_ok=True)
    path = os.path.join(results_dir, sample_idx + ".png").replace("rgb.png", "label.png")
    Image.fromarray(leaves_label_image(args, out_masks, stop_probs, height, width), mode="L").save(path)
    return path
