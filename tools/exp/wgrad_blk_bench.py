"""Single-launch timing of the bf16 weight-gradient kernels on blk operands (rsis_conv2d_wgrad, RSIS_DTYPE_BF16_BLK) at the shapes of
a bf16 224^2 / batch-32 step: the time-batched decoder gates (320 images) and the trunk's 3x3 / 1x1 convs, next to the bytes each
must read once (GB/s of unique operand bytes) and the MFMA rate."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from rsis_amd import ops          # noqa: E402


def t_us(fn, n=10):
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


SHAPES = [  # B, Cin, Cout, HW, ks
    (320, 24, 32, 112, 3), (320, 48, 64, 56, 3), (320, 96, 128, 28, 3), (320, 192, 256, 14, 3), (320, 128, 512, 7, 3),
    (32, 64, 64, 56, 3), (32, 128, 128, 28, 3), (32, 256, 256, 14, 3), (32, 512, 512, 7, 3),
    (32, 256, 16, 56, 3), (32, 512, 32, 28, 3), (32, 1024, 64, 14, 3), (32, 2048, 128, 7, 3),
    (32, 64, 256, 56, 1), (32, 256, 64, 56, 1), (32, 128, 512, 28, 1), (32, 512, 128, 28, 1), (32, 256, 1024, 14, 1), (32, 1024, 256, 14, 1),
    (32, 512, 2048, 7, 1), (32, 2048, 512, 7, 1),
]
only = sys.argv[1] if len(sys.argv) > 1 else ""
for (B, cin, cout, hw, ks) in SHAPES:
    tag = "%dx%d->%d k%d @%d" % (B, cin, cout, ks, hw)
    if only and only not in tag:
        continue
    x = torch.randn(B, cin // 8, hw, hw, 8, device="cuda").to(torch.bfloat16)
    dy = torch.randn(B, cout // 8, hw, hw, 8, device="cuda").to(torch.bfloat16)
    dW = torch.zeros(cout, cin, ks, ks, device="cuda")
    us = t_us(lambda: ops.blk_conv_wgrad(dy, x, dW, ks))
    byts = 2.0 * B * hw * hw * (cin + cout)
    fl = 2.0 * B * hw * hw * cin * cout * ks * ks
    print("%-26s %8.1f us | unique operand bytes %7.1f MB -> %6.2f TB/s | %7.1f TFLOP/s" % (tag, us, byts / 1e6, byts / us / 1e6, fl / us / 1e6))

if os.environ.get("SWEEP"):      # library built with -DRSIS_W3T_SWEEP: tile height x ring depth of the 3x3 DMA kernel, per shape
    from rsis_amd._lib import lib, ptr, stream
    L = lib()
    for (B, cin, cout, hw, ks) in SHAPES:
        if ks != 3:
            continue
        x = torch.randn(B, cin // 8, hw, hw, 8, device="cuda").to(torch.bfloat16)
        dy = torch.randn(B, cout // 8, hw, hw, 8, device="cuda").to(torch.bfloat16)
        dW = torch.zeros(cout, cin, ks, ks, device="cuda")
        row = []
        for th in (2, 4, 8, 16):
            for nr in (2, 3, 4, 5):
                os.environ["RSIS_W3T_TH"], os.environ["RSIS_W3T_NR"] = str(th), str(nr)
                call = lambda: L.rsis_conv2d_wgrad(ptr(dy), ptr(x), ptr(dW), B, cin, hw, hw, cout, hw, hw, ks, 1, 1, cin, 0, 0, ops.DTYPE_BF16_BLK, stream())
                if call() != 0:
                    continue
                row.append((t_us(call), th, nr))
        del os.environ["RSIS_W3T_TH"]
        row.sort()
        print("%dx%d->%d @%d: " % (B, cin, cout, hw) + "  ".join("th%d/nr%d %.1f" % (th, nr, us) for us, th, nr in row[:8]) + "  ... worst %.1f" % row[-1][0])
