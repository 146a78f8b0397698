#!/usr/bin/env python
"""Inference throughput of test() (reference src/test.py:16-50: eval-mode encoder once, T decoder steps, upsample to the input
size, sigmoid) on synthetic 256x256 batches -- the R8 caller of SURVEY 8(a).   python tools/bench_inference.py [--batch 32] [--T 10]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
from rsis_amd.modules import FeatureExtractor, RSIS  # noqa: E402
from rsis_amd.test import GraphedTest, test  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--imsize", type=int, default=256)
    ap.add_argument("--T", type=int, default=10)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--dtype", default="fp32", choices=["fp32", "bf16"])
    ap.add_argument("--graph", action="store_true", help="replay test() as a captured hipGraph (rsis_amd.test.GraphedTest)")
    o = ap.parse_args()
    a = bench.bench_args(o.batch, o.imsize, o.T, o.dtype)
    torch.manual_seed(0)
    enc, dec = FeatureExtractor(a).cuda().eval(), RSIS(a).cuda().eval()
    x = torch.randn(o.batch, 3, o.imsize, o.imsize, device="cuda")
    run = GraphedTest(a, enc, dec) if o.graph else (lambda xin: test(a, enc, dec, xin))
    for _ in range(4):
        run(x)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(o.iters):
        run(x)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / o.iters
    print("test()%s: %.2f ms per batch of %d (T=%d, %dx%d, %s) = %.0f images/s" % (" as a replayed hipGraph" if o.graph else "", ms, o.batch, o.T,
                                                                                  o.imsize, o.imsize, o.dtype, o.batch / ms * 1e3))


if __name__ == "__main__":
    main()
