#!/usr/bin/env python
"""`python -m rsis_amd.train --graph` against the eager launch mode on a loader whose batches stop at DIFFERENT decoder steps
(reference src/train.py:85-94: the sequence ends after the first step whose slot is empty in every image): a
synthesised CVPPP A1 directory (3-9 leaves per image), BASELINE configs[0]'s flag set (256 x 256, T = 16, batch 2) plus a batch-8
variant, `-max_epoch` epochs each way; prints the images / s of every epoch from the wall clock of train.py's own epoch lines and
the number of distinct t_run keys the loader produced.

    python tools/graph_cache_bench.py [--epochs 4] [--batch 2]
"""
import argparse
import os
import re
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--epochs", type=int, default=8)
    ap.add_argument("--batch", type=int, default=2)
    ap.add_argument("--imsize", type=int, default=256)
    o = ap.parse_args()
    from rsis_amd.dataloader.leaves import synthesize_leaves_dir
    with tempfile.TemporaryDirectory() as tmp:
        d = synthesize_leaves_dir(os.path.join(tmp, "A1"), n=104, size=(o.imsize + 16, o.imsize + 32), seed=3)
        for mode in ("eager", "graph"):
            cmd = [sys.executable, "-u", "-m", "rsis_amd.train", "-dataset", "leaves", "-leaves_dir", d, "-leaves_test_dir", d, "-imsize", str(o.imsize),
                   "--resize", "-batch_size", str(o.batch), "-maxseqlen", "16", "-gt_maxseqlen", "16", "-num_classes", "2", "--log_term",
                   "-max_epoch", str(o.epochs), "-print_every", "1000", "-model_name", "gcb_" + mode, "-models_root", os.path.join(tmp, "models"),
                   "-num_workers", "4", "-class_loss_after", "-1"] + (["--graph"] if mode == "graph" else [])
            t0 = time.time()
            p = subprocess.Popen(cmd, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
            marks = []
            for line in p.stdout:
                if re.match(r"Epoch \d+:.*\(train\)", line):
                    marks.append(time.time())
            p.wait()
            assert p.returncode == 0, "train.py failed in %s mode" % mode
            n_img = 96 // o.batch * o.batch
            per = [marks[0] - t0] + [b - a for a, b in zip(marks, marks[1:])]      # (each span also holds the 8-image validation pass before it)
            rest = sorted(per[1:])
            print("%-5s batch %d: epoch wall times %s s -> images/s %s (epoch 0 includes start-up%s); MEDIAN of epochs >= 1: %.1f images/s" % (
                mode, o.batch, " ".join("%.1f" % v for v in per), " ".join("%.1f" % (n_img / v) for v in per),
                " and every capture" if mode == "graph" else "", n_img / rest[len(rest) // 2]), flush=True)


if __name__ == "__main__":
    main()
