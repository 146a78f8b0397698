"""What does a DEPENDENT launch cost inside a replayed hipGraph?  Chains of 200 launches on a layer-3-sized blocked bf16 tensor
(32 x 1024 x 14 x 14 = 12.8 MB and 32 x 256 x 14 x 14 = 3.2 MB): a stock elementwise kernel (x.mul_), the blk BatchNorm statistics /
apply pair, the single blk BatchNorm call, a blk 1x1 conv -- time per launch from HIP events around 20 replays."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from rsis_amd import ops          # noqa: E402


def chain(fn, n=200, replays=20):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        fn()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for _ in range(n):
                fn()
        g.replay()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(replays):
            g.replay()
        b.record()
        torch.cuda.synchronize()
    return a.elapsed_time(b) / replays / n * 1e3


for C in (256, 1024):
    x = ops.blk_from_nchw(torch.randn(32, C, 14, 14, device="cuda"))
    y = torch.empty_like(x)
    gamma, beta = torch.ones(C, device="cuda"), torch.zeros(C, device="cuda")
    rm, rv = torch.zeros(C, device="cuda"), torch.ones(C, device="cuda")
    w = torch.randn(C, C, 1, 1, device="cuda") / C ** 0.5
    pk = ops.PackedConv(1, [C], stride=1, pad=0, dtype=ops.DTYPE_BF16)
    wp = pk.fwd(w)
    print("C = %4d (%.1f MB per tensor):" % (C, x.numel() * 2 / 1e6))
    print("  x.mul_(1.0) [stock elementwise, in place]        %6.2f us per launch" % chain(lambda: x.mul_(1.0)))
    print("  blk_bn_fwd train (statistics + apply: 2 launches) %6.2f us per call" % chain(lambda: ops.blk_bn_fwd(x, None, gamma, beta, rm, rv, 1e-5, 0.1, True, True)))
    print("  blk_bn_fwd eval  (apply only: 1 launch)           %6.2f us per call" % chain(lambda: ops.blk_bn_fwd(x, None, gamma, beta, rm, rv, 1e-5, 0.1, True, False)))
    print("  blk_conv2d 1x1 C -> C                             %6.2f us per call" % chain(lambda: ops.blk_conv2d(x, wp, C, 1)))
    if C == 256:
        w3 = torch.randn(C, C, 3, 3, device="cuda") / (9 * C) ** 0.5
        pk3 = ops.PackedConv(3, [C], stride=1, pad=1, dtype=ops.DTYPE_BF16)
        wp3 = pk3.fwd(w3)
        print("  blk_conv2d 3x3 C -> C                             %6.2f us per call" % chain(lambda: ops.blk_conv2d(x, wp3, C, 3)))
    yb, sm, sr = ops.blk_bn_fwd(x, None, gamma, beta, rm, rv, 1e-5, 0.1, True, True)
    dy = ops.blk_from_nchw(torch.randn(32, C, 14, 14, device="cuda"))
    print("  blk_bn_bwd (sums + apply: 2 launches), mask from x %6.2f us per call" % chain(lambda: ops.blk_bn_bwd(dy, x, None, gamma, beta, sm, sr, True, False)))
    print("  blk_bn_bwd with y and the residual gradient        %6.2f us per call" % chain(lambda: ops.blk_bn_bwd(dy, x, yb, gamma, beta, sm, sr, True, True)))
