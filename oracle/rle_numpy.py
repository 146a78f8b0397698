"""TEST INFRASTRUCTURE ONLY.  numpy restatement of the COCO run-length mask encoding the reference's evaluation path relies
on (reference src/eval.py:97-127 `mask.encode(np.asfortranarray(seg))`; the arithmetic lives in
src/coco/common/maskApi.c:32-41 rleEncode and :196-209 rleToString).  Pinned against the reference library itself
(oracle/_ref/libmaskapi_ref.so, built by oracle/Makefile from the reference's sources) in tests/test_oracle_golden.py and
against tests/golden/rle.npz.

  counts: lengths of the alternating runs of 0s and 1s of the mask read in COLUMN-major order, starting with a (possibly
          empty) run of 0s.
  string: each count (for i > 2 the difference to counts[i-2]) as a little-endian base-32 varint with a sign-extension
          rule, 6 bits per character (5 payload bits + continuation bit), offset by 48 into printable ASCII."""
import numpy as np


def rle_counts(mask_hw):
    v = np.asarray(mask_hw, dtype=np.uint8).T.reshape(-1)            # column-major traversal of the (h, w) mask
    if v.size == 0:
        return np.zeros(1, dtype=np.uint32)
    change = np.flatnonzero(v[1:] != v[:-1]) + 1
    edges = np.concatenate(([0], change, [v.size]))
    counts = np.diff(edges)
    if v[0] != 0:                                                    # the first run counts zeros
        counts = np.concatenate(([0], counts))
    return counts.astype(np.uint32)


def rle_string(counts):
    out = bytearray()
    c = [int(x) for x in counts]
    for i, x in enumerate(c):
        if i > 2:
            x -= c[i - 2]
        while True:
            ch = x & 0x1F
            x >>= 5                                                  # arithmetic shift (python ints are signed)
            more = (x != -1) if (ch & 0x10) else (x != 0)
            if more:
                ch |= 0x20
            out.append(ch + 48)
            if not more:
                break
    return bytes(out)


def rle_area(counts):
    return int(np.asarray(counts, dtype=np.int64)[1::2].sum())


def resize_threshold(pred, height, width, th, ignore=None):
    """reference eval.py:96-117 resize_mask up to the encoding: scipy.ndimage.zoom(order=1) of the (Hm, Wm) probability map
    to (height, width) -- which samples the input at o * (in - 1) / (out - 1), i.e. align-corners bilinear --, `> th`,
    ignore pixels cleared.  Returns (segmentation uint8 (h, w), raw segmentation uint8 (h, w))."""
    from scipy.ndimage import zoom
    p = np.asarray(pred, dtype=np.float64)
    z = zoom(p.reshape(p.shape[0], p.shape[1], 1), [float(height) / p.shape[0], float(width) / p.shape[1], 1], order=1)
    raw = (z > th).astype(np.uint8).reshape(height, width)
    seg = raw.copy()
    if ignore is not None:
        seg[np.asarray(ignore).reshape(height, width) == 1] = 0
    return seg, raw
