#!/usr/bin/env python
"""Shape stress: BASELINE.json configs[4] geometry (512x1024, T=20, batch 8 per GPU, 9 classes) -- a few eager training steps
through the same kernels (large maps: 256x512 at the finest pyramid level), finite losses, step time.
usage: tools/stress_config5.py [fp32|bf16]   (bf16: configs[4] as named; RSIS_BF16_STORAGE=0 for fp32 activations)"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
from rsis_amd.modules import FeatureExtractor, RSIS  # noqa: E402
from rsis_amd.synthetic import synthetic_batch  # noqa: E402
from rsis_amd.train import build_optimizers, runIter  # noqa: E402
from rsis_amd.utils.objectives import MaskedBCELoss, MaskedNLLLoss, softIoULoss  # noqa: E402

B, H, W, T = 8, 512, 1024, 20
DTYPE = sys.argv[1] if len(sys.argv) > 1 else "fp32"
a = bench.bench_args(B, H, T, DTYPE)
a.num_classes, a.gt_maxseqlen, a.maxseqlen = 9, 20, T
enc, dec = FeatureExtractor(a).cuda(), RSIS(a).cuda()
opts = list(build_optimizers(a, enc, dec))
crits = [softIoULoss(), MaskedNLLLoss(None), MaskedBCELoss(0.5)]
batch = synthetic_batch(1, B, H, W, 20, 15, 9, "cuda")
for i in range(5):
    torch.cuda.synchronize()
    t0 = time.time()
    losses, _outs, _perms = runIter(a, enc, dec, *batch, crits, opts, mode="train", sync_losses=True)
    torch.cuda.synchronize()
    print("step %d: %.1f ms  losses %s" % (i, (time.time() - t0) * 1e3, ["%.4f" % v for v in losses]))
    assert all(v == v and abs(v) < 1e6 for v in losses)
print("peak memory %.1f GB; %.1f images/s at 512x1024, T=20, B=8 (%s)" % (torch.cuda.max_memory_allocated() / 2**30, B / (time.time() - t0), DTYPE))
