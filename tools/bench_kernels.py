#!/usr/bin/env python
"""Kernel micro-benchmark (GPU box): times individual C-ABI launches on the shapes of BASELINE config 2 with HIP
events and prints algorithmic TFLOP/s.  Used for kernel tuning and as the command profiled by rocprofv3.

  python tools/bench_kernels.py [--what gates|trunk|depth|c1|all] [--iters N] [--tile T] [--batch B]
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from rsis_amd import ops  # noqa: E402
from rsis_amd._lib import check, int_array, lib, ptr, ptr_array, stream  # noqa: E402

GATES = [([128], 128, 8), ([128, 128], 64, 16), ([64, 64], 32, 32), ([32, 32], 16, 64), ([16, 16], 8, 128)]
# ResNet-101 trunk groups at 256^2 (Cin, Cout, ks, stride, out HxW, count) -- SURVEY.md Appendix A
TRUNK = [(256, 256, 3, 1, 16, 22), (256, 1024, 1, 1, 16, 23), (1024, 256, 1, 1, 16, 22), (64, 64, 3, 1, 64, 3),
         (128, 128, 3, 1, 32, 3), (512, 512, 3, 1, 8, 2), (64, 256, 1, 1, 64, 4), (128, 512, 1, 1, 32, 4),
         (512, 128, 1, 1, 32, 3), (512, 2048, 1, 1, 8, 3), (3, 64, 7, 2, 128, 1), (2048, 512, 1, 1, 8, 2),
         (2048, 128, 3, 1, 8, 1), (1024, 128, 3, 1, 16, 1), (64, 16, 3, 1, 128, 1)]


def timeit(fn, iters):
    for _ in range(2):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


DT = 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--what", default="gates")
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--tile", type=int, default=0)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--dtype", default="fp32", choices=["fp32", "bf16"])
    o = ap.parse_args()
    global DT
    DT = ops.DTYPES[o.dtype]
    L = lib()
    B = o.batch
    tot_f = tot_ms = 0.0
    if o.what in ("gates", "all"):
        for segs, hid, hw in GATES:
            H = W = hw
            cin = sum(segs) + hid
            w = torch.randn(4 * hid, cin, 3, 3, device="cuda") / (3.0 * cin ** 0.5)
            bias = torch.randn(4 * hid, device="cuda") * 0.1
            pack = ops.PackedConv(3, segs + [hid], lstm_hid=hid, dtype=DT)
            wp = pack.fwd(w, bias)
            srcs = [torch.randn(B, c, H, W, device="cuda") for c in segs] + [torch.tanh(torch.randn(B, hid, H, W, device="cuda"))]
            c_prev = torch.randn(B, hid, H, W, device="cuda")
            h, c = torch.empty_like(c_prev), torch.empty_like(c_prev)
            act = torch.empty(B, 4 * hid, H, W, device="cuda")
            pa, ia = ptr_array(srcs), int_array(segs + [hid])
            ms = timeit(lambda: check(L.rsis_convlstm_fwd(pa, ia, len(srcs), B, H, W, ptr(wp), ptr(pack.bias_p), None, ptr(c_prev),
                                                          ptr(h), ptr(c), ptr(act), hid, 3, 1, o.tile, DT, stream()), "lstm"), o.iters)
            fl = 2.0 * B * H * W * cin * 9 * 4 * hid
            print("gate fwd  %3dx%-3d M=%7d K=%5d N=%4d  %8.1f us  %6.1f TF/s" % (H, W, B * H * W, cin * 9, 4 * hid, ms * 1e3, fl / ms / 1e9))
            tot_f += fl
            tot_ms += ms
            # dgrad (all sources) and wgrad of the same conv
            da = torch.randn(B, 4 * hid, H, W, device="cuda")
            wd = pack.dgrad(w)
            dxs = [torch.empty_like(s) for s in srcs]
            pd, idd = ptr_array(dxs), int_array(segs + [hid])
            ms = timeit(lambda: check(L.rsis_conv2d_dgrad(ptr(da), B, 4 * hid, H, W, ptr(wd), cin, 3, 1, 1, pd, idd, len(dxs), H, W,
                                                          None, o.tile, DT, stream()), "dgrad"), o.iters)
            print("gate dgrad %3dx%-3d %38s %8.1f us  %6.1f TF/s" % (H, W, "", ms * 1e3, fl / ms / 1e9))
            dW = torch.zeros_like(w)

            def wg():
                off = 0
                for s_ in srcs:
                    check(L.rsis_conv2d_wgrad(ptr(da), ptr(s_), ptr(dW), B, s_.shape[1], H, W, 4 * hid, H, W, 3, 1, 1, cin, off, hid,
                                              DT, stream()), "wgrad")
                    off += s_.shape[1]
            ms = timeit(wg, o.iters)
            print("gate wgrad %3dx%-3d %38s %8.1f us  %6.1f TF/s" % (H, W, "", ms * 1e3, fl / ms / 1e9))
        print("gates fwd total: %.1f us/timestep, %.1f TF/s" % (tot_ms * 1e3, tot_f / tot_ms / 1e9))
    if o.what in ("trunk", "all", "depth", "c1"):
        tf = tm = 0.0
        # "depth": the same 3x3 / 1x1 layer at growing input depth -> the intercept is the kernel's fixed cost
        shapes = TRUNK if o.what != "depth" else [(c, 256, k, 1, 16, 1) for k in (3, 1) for c in (8, 32, 64, 128, 256, 512, 1024)]
        if o.what == "c1":     # conv_out of the decoder (HBM-bound vector-ALU kernels): %s of HBM = bytes / time
            shapes = [(8, 1, 3, 1, 256, 1), (8, 1, 3, 1, 128, 1), (16, 1, 3, 1, 256, 1)]   # config 2: 8 channels at 256x256
        for cin, cout, ks, stride, hw, count in shapes:
            pad = ks // 2
            Hi = hw * stride
            x = torch.randn(B, cin, Hi, Hi, device="cuda")
            w = torch.randn(cout, cin, ks, ks, device="cuda") / (ks * cin ** 0.5)
            pack = ops.PackedConv(ks, [cin], stride=stride, pad=pad, dtype=DT)
            wp, wd = pack.fwd(w), pack.dgrad(w)
            y = torch.empty(B, cout, hw, hw, device="cuda")
            pa, ia = ptr_array([x]), int_array([cin])
            fl = 2.0 * B * hw * hw * cin * ks * ks * cout
            ms_f = timeit(lambda: check(L.rsis_conv2d_fwd(pa, ia, 1, B, Hi, Hi, ptr(wp), cout, ks, stride, pad, None, None, ptr(y), hw, hw,
                                                          o.tile, DT, stream()), "fwd"), o.iters)
            dx = torch.empty_like(x)
            pd = ptr_array([dx])
            ms_d = timeit(lambda: check(L.rsis_conv2d_dgrad(ptr(y), B, cout, hw, hw, ptr(wd), cin, ks, stride, pad, pd, ia, 1, Hi, Hi,
                                                            None, o.tile, DT, stream()), "dgrad"), o.iters)
            dW = torch.zeros_like(w)
            ms_w = timeit(lambda: check(L.rsis_conv2d_wgrad(ptr(y), ptr(x), ptr(dW), B, cin, Hi, Hi, cout, hw, hw, ks, stride, pad, cin, 0, 0,
                                                            DT, stream()), "wgrad"), o.iters)
            print("conv %4d->%4d k%d s%d @%3d^2 x%2d  %6.2f GF | fwd %7.1f us %6.1f TF | dgrad %7.1f us %6.1f TF | wgrad %7.1f us %6.1f TF"
                  % (cin, cout, ks, stride, hw, count, fl / 1e9, ms_f * 1e3, fl / ms_f / 1e9, ms_d * 1e3, fl / ms_d / 1e9, ms_w * 1e3,
                     fl / ms_w / 1e9))
            tf += 3 * fl * count
            tm += (ms_f + ms_d + ms_w) * count
        print("trunk fwd+dgrad+wgrad (weighted by layer count): %.2f ms, %.1f TF/s" % (tm, tf / tm / 1e9))


if __name__ == "__main__":
    main()
