"""TEST INFRASTRUCTURE ONLY.  Generates tests/golden/targets.npz with the reference's own
MyDataset.sequence_from_masks (reference src/dataloader/dataset.py:86-146), imported from /root/reference in the build
container (h5py / cv2 / the augmentation helpers are not needed by that method and are stubbed for the import)."""
import os
import sys
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("RSIS_REFERENCE", "/root/reference")


def main():
    for name in ("h5py", "cv2"):
        sys.modules.setdefault(name, types.ModuleType(name))
    du = types.ModuleType("dataset_utils")          # (scale / flip_crop: PIL augmentation helpers, not used by the method)
    du.scale = du.flip_crop = None
    sys.modules["dataset_utils"] = du
    sys.path[:0] = [os.path.join(ROOT, "oracle", "ref_shims"), os.path.join(REF, "src", "dataloader"), ROOT]
    import dataset as refds
    rng = np.random.default_rng(77)
    out = {}
    cases = [(16, 16, 3, 5), (24, 20, 7, 5), (32, 32, 12, 10), (20, 28, 0, 4), (16, 16, 4, 4), (12, 12, 6, 6)]
    for i, (h, w, ninst, T) in enumerate(cases):
        ins = np.zeros((h, w), np.int64)
        seg = np.zeros((h, w), np.int64)
        ids = rng.choice(np.arange(1, 200), size=ninst, replace=False) if ninst else []
        for k in ids:                                              # random rectangles, later ones overwrite earlier ones
            y0, x0 = rng.integers(0, h - 2), rng.integers(0, w - 2)
            y1, x1 = rng.integers(y0 + 1, h + 1), rng.integers(x0 + 1, w + 1)
            ins[y0:y1, x0:x1] = k
            seg[y0:y1, x0:x1] = rng.integers(1, 21)
        # instances of EQUAL area are ordered by np.argsort's default (unstable, implementation-defined) sort in the reference:
        # that order is not a property of the reference -> only tie-free cases are pinned
        areas = [int((ins == k).sum()) for k in np.unique(ins)[1:]]
        while len(set(areas)) != len(areas):
            k = int(rng.choice(np.unique(ins)[1:]))
            ys, xs = np.nonzero(ins == k)
            ins[ys[0], xs[0]] = 0
            seg[ys[0], xs[0]] = 0
            areas = [int((ins == k).sum()) for k in np.unique(ins)[1:]]
        self = types.SimpleNamespace(max_seq_len=T, classes=list(range(21)))
        tgt = refds.MyDataset.sequence_from_masks(self, ins.astype(np.float64), seg.astype(np.float64))
        out["ins%d" % i], out["seg%d" % i], out["T%d" % i], out["target%d" % i] = ins, seg, np.array(T), tgt
    out["n"] = np.array(len(cases))
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "targets.npz"), **out)
    print("wrote tests/golden/targets.npz (%d cases)" % len(cases))


if __name__ == "__main__":
    main()
