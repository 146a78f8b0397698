#!/usr/bin/env python
"""Winograd F(2x2, 3x3) (rsis_amd/csrc/conv_wino.hip, RSIS_DTYPE_F32_WINO) against the direct exact-f32 kernel, per shape: the 3x3 /
stride 1 convs of the ResNet-101 bottlenecks (torchvision Bottleneck.conv2 through reference src/modules/vision.py:16-19) at the
256 x 256 and 224 x 224 geometries, B = 32, forward and data gradient, HIP events over back-to-back launches.  Also the command
profiled for the MFMA-busy counters of the Winograd kernel (tools/make_profiles.sh).   python tools/wino_bench.py [--iters 30] [--only-wino]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from rsis_amd import ops  # noqa: E402
from rsis_amd._lib import check, int_array, lib, ptr, ptr_array, stream  # noqa: E402

SHAPES = [(256, 16, 22, "layer3 @256"), (256, 14, 22, "layer3 @224"), (128, 32, 3, "layer2 @256"), (128, 28, 3, "layer2 @224"),
          (64, 64, 3, "layer1 @256"), (64, 56, 3, "layer1 @224"), (512, 8, 2, "layer4 @256")]


def timeit(fn, iters):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=30)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--only-wino", action="store_true")
    o = ap.parse_args()
    L, B = lib(), o.batch
    print("%-14s %5s %4s | %21s | %21s | %s" % ("shape", "C", "HxW", "forward us (dir / wino)", "dgrad us (dir / wino)", "saved per step (fwd + dgrad, us)"))
    for C, hw, count, name in SHAPES:
        x = torch.randn(B, C, hw, hw, device="cuda")
        w = torch.randn(C, C, 3, 3, device="cuda") / (3.0 * C ** 0.5)
        y, dx = torch.empty_like(x), torch.empty_like(x)
        res = {}
        for tag, dt in (("direct", ops.DTYPE_F32), ("wino", ops.DTYPE_F32_WINO)):
            if o.only_wino and tag == "direct":
                continue
            pack = ops.PackedConv(3, [C], stride=1, pad=1, dtype=dt)
            wp, wd = pack.fwd(w), pack.dgrad(w)
            pa, ia, pd = ptr_array([x]), int_array([C]), ptr_array([dx])
            f = timeit(lambda: check(L.rsis_conv2d_fwd(pa, ia, 1, B, hw, hw, ptr(wp), C, 3, 1, 1, None, None, ptr(y), hw, hw, 100, dt, stream()), "fwd"), o.iters)
            d = timeit(lambda: check(L.rsis_conv2d_dgrad(ptr(y), B, C, hw, hw, ptr(wd), C, 3, 1, 1, pd, ia, 1, hw, hw, None, 0, dt, stream()), "dgrad"), o.iters)
            res[tag] = (f, d)
        if o.only_wino:
            print("%-14s %5d %4d | %21.1f | %21.1f |" % (name, C, hw, res["wino"][0], res["wino"][1]))
            continue
        gf = 2.0 * B * hw * hw * C * 9 * C / 1e9
        print("%-14s %5d %4d | %9.1f / %9.1f | %9.1f / %9.1f | %8.0f   (%d layers; direct %.0f TF/s, winograd %.0f TF/s direct-equivalent)" % (
            name, C, hw, res["direct"][0], res["wino"][0], res["direct"][1], res["wino"][1],
            count * (res["direct"][0] - res["wino"][0] + res["direct"][1] - res["wino"][1]), count, gf / res["direct"][0] * 1e3, gf / res["wino"][0] * 1e3))


if __name__ == "__main__":
    main()
