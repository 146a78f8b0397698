"""TEST INFRASTRUCTURE ONLY.  Generates tests/golden/rle.npz with the reference's own mask API (oracle/_ref/libmaskapi_ref.so,
compiled from /root/reference/src/coco/common/maskApi.c by oracle/Makefile): a set of masks with the run counts and the
compressed strings the reference produces for them.  Run in the build container: `make -C oracle && python oracle/make_golden_rle.py`."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import maskapi_ref as M   # noqa: E402
from oracle import rle_numpy as R     # noqa: E402


def main():
    rng = np.random.default_rng(2024)
    masks = []
    for (h, w, p, blob) in [(1, 1, 1.0, 1), (1, 9, 0.5, 1), (7, 1, 0.5, 1), (5, 7, 0.0, 1), (5, 7, 1.0, 1), (16, 16, 0.5, 1),
                            (33, 20, 0.3, 4), (64, 48, 0.5, 8), (100, 132, 0.4, 16), (200, 264, 0.35, 25), (256, 256, 0.5, 32)]:
        bh, bw = -(-h // blob), -(-w // blob)
        m = np.kron((rng.random((bh, bw)) < p).astype(np.uint8), np.ones((blob, blob), np.uint8))[:h, :w]
        masks.append(np.ascontiguousarray(m))
    out = {}
    for i, m in enumerate(masks):
        c, s = M.encode(m)
        assert np.array_equal(c, R.rle_counts(m)) and s == R.rle_string(c), "restatement != reference"
        out["mask%d" % i] = m
        out["counts%d" % i] = c
        out["string%d" % i] = np.frombuffer(s, dtype=np.uint8)
    out["n"] = np.array(len(masks))
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "rle.npz"), **out)
    print("wrote tests/golden/rle.npz with %d masks (reference library = oracle restatement on all of them)" % len(masks))


if __name__ == "__main__":
    main()
