#!/usr/bin/env python
"""Per-layer distance from float64 of the 3x3 / stride-1 trunk convs (torchvision Bottleneck.conv2 through reference src/modules/vision.py:16-19)
under the four fp32 kernels an inference call could run: direct (one accumulation chain), direct with segmented sums (the inference default),
Winograd F(2x2, 3x3) (one chain per Winograd position) and Winograd with segmented sums (conv_wino_f32_kernel<FLUSH>).
    python tools/exp/wino_flush_accuracy.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

from rsis_amd import ops  # noqa: E402
from rsis_amd._lib import check, int_array, lib, ptr, ptr_array, stream  # noqa: E402

L = lib()
print("%-18s %12s %12s %12s %12s" % ("shape", "direct", "direct seg", "wino", "wino seg"))
for C, hw in ((256, 16), (256, 14), (128, 32), (64, 64), (512, 8)):
    torch.manual_seed(C + hw)
    B = 8
    x = torch.randn(B, C, hw, hw, device="cuda").relu_()                 # post-ReLU activations, as in the trunk
    w = torch.randn(C, C, 3, 3, device="cuda") / (3.0 * C ** 0.5)
    ref = torch.nn.functional.conv2d(x.double(), w.double(), padding=1)
    errs = []
    for dt in (ops.DTYPE_F32, ops.DTYPE_F32_WINO):
        pack = ops.PackedConv(3, [C], stride=1, pad=1, dtype=dt)
        wp = pack.fwd(w)
        for tile in (100, 0):                                            # training call (one chain) / inference call (segmented)
            y = torch.empty_like(x)
            check(L.rsis_conv2d_fwd(ptr_array([x]), int_array([C]), 1, B, hw, hw, ptr(wp), C, 3, 1, 1, None, None, ptr(y), hw, hw, tile, dt, stream()), "fwd")
            errs.append(float((y.double() - ref).abs().max()))
    print("%4d ch @%3d^2      %12.3e %12.3e %12.3e %12.3e   (|y|max %.2f)" % (C, hw, errs[0], errs[1], errs[2], errs[3], float(ref.abs().max())))
