"""Inference post-processing of the predicted instance masks on the GPU -- counterpart of reference src/eval.py:96-127
`resize_mask` (scipy.ndimage.zoom(order=1) to the original image size, `> args.mask_th`, ignore pixels cleared, minimum-size
test, pycocotools `mask.encode`), SURVEY.md section 8(f) row N2.  The resample + threshold + area and the run-length
encoding run in librsis_hip.so (rsis_mask_resize_threshold / rsis_rle_encode); only the few hundred run counts of each mask
come back to the host, where rsis_rle_to_string writes pycocotools' compressed text form."""
import ctypes

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
