#!/usr/bin/env python
"""fp32 1x1 GEMM (conv_igemm_kernel V4) per trunk shape x forced tile: time and fraction of the exact-f32 MFMA peak.
  python tools/exp/gemm1x1_sweep.py [--batch 32] [--iters 20] [--tiles 13,14,15,16]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

from rsis_amd import ops  # noqa: E402
from rsis_amd._lib import check, int_array, lib, ptr, ptr_array, stream  # noqa: E402

# (Cin, Cout, HxW at 256^2, layers) forward; the data gradient of a layer is the mirrored shape
SHAPES = [(256, 1024, 16, 23), (1024, 256, 16, 22), (64, 256, 64, 4), (256, 64, 64, 2), (128, 512, 32, 4), (512, 128, 32, 3),
          (512, 2048, 8, 3), (2048, 512, 8, 2), (256, 128, 64, 1), (512, 256, 32, 1), (1024, 512, 16, 1)]


def timeit(fn, iters):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--tiles", default="0,13,14,15,16,5")
    o = ap.parse_args()
    L = lib()
    tiles = [int(t) for t in o.tiles.split(",")]
    tot = {t: 0.0 for t in tiles}
    tot_fl = 0.0
    print("%-22s %10s  " % ("shape (Cin->Cout @HW)", "GFLOP") + "  ".join("tile%-3d us (frac)" % t for t in tiles))
    for cin, cout, hw, count in SHAPES:
        for (ci, co) in ((cin, cout), (cout, cin)):          # forward, then its data gradient (a 1x1 conv of the mirrored shape)
            x = torch.randn(o.batch, ci, hw, hw, device="cuda")
            w = torch.randn(co, ci, 1, 1, device="cuda") / ci ** 0.5
            pack = ops.PackedConv(1, [ci], stride=1, pad=0, dtype=ops.DTYPE_F32)
            wp = pack.fwd(w)
            y = torch.empty(o.batch, co, hw, hw, device="cuda")
            pa, ia = ptr_array([x]), int_array([ci])
            fl = 2.0 * o.batch * hw * hw * ci * co
            row = []
            for t in tiles:
                ms = timeit(lambda: check(L.rsis_conv2d_fwd(pa, ia, 1, o.batch, hw, hw, ptr(wp), co, 1, 1, 0, None, None, ptr(y), hw, hw, t,
                                                            ops.DTYPE_F32, stream()), "fwd"), o.iters)
                row.append("%7.1f (%.3f)" % (1e3 * ms, fl / ms / 1e9 / 157.3))
                tot[t] += ms * count
            tot_fl += fl * count
            print("%-22s %10.2f  " % ("%d->%d @%d x%d" % (ci, co, hw, count), fl / 1e9) + "  ".join(row))
    print("per step (layer counts applied): " + "  ".join("tile%d %.3f ms (%.3f of peak)" % (t, tot[t], tot_fl / tot[t] / 1e9 / 157.3) for t in tiles))


if __name__ == "__main__":
    main()
