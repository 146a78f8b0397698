"""Micro-benchmark: the channel-blocked bf16 conv (rsis_blk_conv2d) against the fp32-storage bf16 conv (rsis_conv2d_fwd, dtype bf16)
on the stride-1 conv shapes of the ResNet-101 trunk at batch 32 (224x224 and 256x256 inputs).  HIP-event time per launch over
back-to-back launches; `--variants` sweeps the blk tile variants.  Usage: python tools/blk_bench.py [--imsize 224] [--variants]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rsis_amd import ops                                     # noqa: E402
from rsis_amd._lib import check, int_array, lib, ptr, ptr_array, stream   # noqa: E402


def timeit(fn, n=40, warm=5):
    for _ in range(warm):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3       # us


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--imsize", type=int, default=224)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--variants", action="store_true")
    a = ap.parse_args()
    B, S = a.batch, a.imsize
    shapes = []
    for planes, div in ((64, 4), (128, 8), (256, 16), (512, 32)):
        h = S // div
        shapes += [(planes, planes, h, 3), (planes, 4 * planes, h, 1), (4 * planes, planes, h, 1)]
    L = lib()
    print("%-26s %9s %9s %7s   %s" % ("Cin->Cout ks @HxW", "fp32st us", "blk us", "ratio", "blk: GB/s TF/s" + ("  variants us" if a.variants else "")))
    for cin, cout, h, ks in shapes:
        x = torch.randn(B, cin, h, h, device="cuda")
        w = torch.randn(cout, cin, ks, ks, device="cuda") / (cin * ks * ks) ** 0.5
        pk = ops.PackedConv(ks, [cin], stride=1, pad=ks // 2, dtype=ops.DTYPE_BF16)
        wp = pk.fwd(w)
        out = torch.empty(B, cout, h, h, device="cuda")
        pa, ia = ptr_array([x]), int_array([cin])

        def old():
            check(L.rsis_conv2d_fwd(pa, ia, 1, B, h, h, ptr(wp), cout, ks, 1, ks // 2, None, None, ptr(out), h, h, 0, ops.DTYPE_BF16, stream()), "old")
        xb = ops.blk_from_nchw(x)
        yb = torch.empty((B, cout // 8, h, h, 8), dtype=torch.bfloat16, device="cuda")

        def new(v=0):
            check(L.rsis_blk_conv2d(ptr(xb), B, cin, h, h, ptr(wp), cout, ks, None, ptr(yb), v, stream()), "blk")
        t_old, t_new = timeit(old), timeit(new)
        byts = (cin + cout) * B * h * h * 2 + cin * cout * ks * ks * 2
        fl = 2.0 * cin * cout * ks * ks * B * h * h
        extra = ""
        if a.variants:
            extra = "  " + " ".join("%d:%.1f" % (v, timeit(lambda: new(v))) for v in range(1, 7 if ks == 3 else 6))
        print("%-26s %9.1f %9.1f %7.2f   %6.0f %6.1f%s" % ("%d->%d %dx%d @%dx%d" % (cin, cout, ks, ks, h, h), t_old, t_new, t_old / t_new,
                                                         byts / t_new * 1e-3, fl / t_new * 1e-6, extra))


if __name__ == "__main__":
    main()
