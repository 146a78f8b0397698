"""Per-tile latency of the bf16 weight-gradient kernels: one weight gradient with ONE split (deterministic mode: every block walks all
spatial tiles of its dW tile) at growing batch -> the slope is the time one block needs per 64-pixel tile when it has the CU to itself."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from rsis_amd import ops          # noqa: E402
from rsis_amd._lib import check, lib, ptr, stream     # noqa: E402


def t_us(fn, n=20):
    for _ in range(3):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


ops.set_deterministic(True)
L = lib()
for (cin, cout, hw, ks) in ((1024, 256, 14, 1), (256, 1024, 14, 1), (256, 256, 14, 3), (64, 256, 56, 1)):
    for blk in (True, False):
        row = []
        for B in (8, 16, 32, 64):
            x = torch.randn(B, cin, hw, hw, device="cuda")
            dy = torch.randn(B, cout, hw, hw, device="cuda")
            dW = torch.zeros(cout, cin, ks, ks, device="cuda")
            if blk:
                xx, dd, dt = ops.blk_from_nchw(x), ops.blk_from_nchw(dy), ops.DTYPE_BF16_BLK
            else:
                xx, dd, dt = x, dy, ops.DTYPE_BF16
            row.append(t_us(lambda: check(L.rsis_conv2d_wgrad(ptr(dd), ptr(xx), ptr(dW), B, cin, hw, hw, cout, hw, hw, ks, 1, ks // 2, cin, 0, 0, dt,
                                                              stream()), "wgrad")))
        tiles = ((hw * hw + 63) // 64) if ks == 1 else (((hw + 3) // 4) * 1 if hw <= 16 else ((hw + 1) // 2) * ((hw + 31) // 32))
        print("%4d->%4d k%d @%2d^2 %s: us at B=8,16,32,64: %s  -> %.2f us per image (%d tiles per image)" % (
            cin, cout, ks, hw, "blk " if blk else "fp32", " ".join("%7.1f" % v for v in row), (row[3] - row[2]) / 32.0, tiles))
