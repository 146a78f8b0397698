#!/usr/bin/env python
"""Per-shape timing of the bf16 conv kernels on the trunk shapes (B=32): forward / data gradient under every forced variant, the
weight gradient as dispatched, next to two floors -- HBM (activation bytes once at 6 TB/s) and bf16 MFMA (at 1.5 PFLOP/s).
Prints the count-weighted gap per shape: where the step's bf16 time is."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

from bench import TRUNK_SHAPES, _time_launch  # noqa: E402
from rsis_amd import ops  # noqa: E402
from rsis_amd._lib import int_array, lib, ptr, ptr_array, stream  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--imsize", type=int, default=224)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--variants", default="1,2,3,4,5,6,7")
    ap.add_argument("--only", default="")
    o = ap.parse_args()
    L = lib()
    dt = ops.DTYPES[o.dtype]
    B = o.batch
    variants = [int(v) for v in o.variants.split(",") if v]
    tot = {"fwd": 0.0, "dgrad": 0.0, "wgrad": 0.0, "floor": 0.0}
    print("%-22s %4s | %8s %8s %8s | %7s %7s | forced fwd variants (us) / forced dgrad variants (us)" %
          ("cin->cout ks @hw", "n", "fwd", "dgrad", "wgrad", "hbm_us", "mfma_us"))
    for cin, cout, ks, hw, count in TRUNK_SHAPES:
        hw = hw * o.imsize // 256
        tag = "%d->%d k%d @%d" % (cin, cout, ks, hw)
        if o.only and o.only not in tag:
            continue
        pad = ks // 2
        x = torch.randn(B, cin, hw, hw, device="cuda")
        w = torch.randn(cout, cin, ks, ks, device="cuda") / (ks * cin ** 0.5)
        pack = ops.PackedConv(ks, [cin], stride=1, pad=pad, dtype=dt)
        wp, wd = pack.fwd(w), pack.dgrad(w)
        y = torch.zeros(B, cout, hw, hw, device="cuda")
        dx, dW = torch.zeros_like(x), torch.zeros_like(w)
        pa, ia, pd = ptr_array([x]), int_array([cin]), ptr_array([dx])

        def fwd(v):
            return L.rsis_conv2d_fwd(pa, ia, 1, B, hw, hw, ptr(wp), cout, ks, 1, pad, None, None, ptr(y), hw, hw, v, dt, stream())

        def dgr(v):
            return L.rsis_conv2d_dgrad(ptr(y), B, cout, hw, hw, ptr(wd), cin, ks, 1, pad, pd, ia, 1, hw, hw, None, v, dt, stream())

        def wgr():
            return L.rsis_conv2d_wgrad(ptr(y), ptr(x), ptr(dW), B, cin, hw, hw, cout, hw, hw, ks, 1, pad, cin, 0, 0, dt, stream())

        us_f = 1e3 * _time_launch(lambda: fwd(100), o.iters)      # 100 + 0: training call (split-K allowed), library heuristic
        us_d = 1e3 * _time_launch(lambda: dgr(0), o.iters)
        us_w = 1e3 * _time_launch(wgr, o.iters)
        fl = 2.0 * B * hw * hw * cin * ks * ks * cout
        hbm = 4.0 * B * hw * hw * (cin + cout) / 6e12 * 1e6
        mf = fl / 1.5e15 * 1e6
        ff, dd = [], []
        for v in variants:
            if fwd(100 + v) == 0:
                ff.append("%d:%.1f" % (v, 1e3 * _time_launch(lambda: fwd(100 + v), o.iters)))
            if dgr(v) == 0:
                dd.append("%d:%.1f" % (v, 1e3 * _time_launch(lambda: dgr(v), o.iters)))
        torch.cuda.synchronize()
        print("%-22s %4d | %8.1f %8.1f %8.1f | %7.1f %7.1f | %s / %s" % (tag, count, us_f, us_d, us_w, hbm, mf, " ".join(ff), " ".join(dd)))
        tot["fwd"] += us_f * count
        tot["dgrad"] += us_d * count
        tot["wgrad"] += us_w * count
        tot["floor"] += max(hbm, mf) * count
    print("per step (ms): fwd %.2f  dgrad %.2f  wgrad %.2f   floor per pass %.2f" %
          (tot["fwd"] / 1e3, tot["dgrad"] / 1e3, tot["wgrad"] / 1e3, tot["floor"] / 1e3))


if __name__ == "__main__":
    main()
