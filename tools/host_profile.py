#!/usr/bin/env python
"""Host-side (Python) cost of one training step: cProfile over a few steps on the GPU box."""
import cProfile
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
from rsis_amd.modules import FeatureExtractor, RSIS  # noqa: E402
from rsis_amd.synthetic import synthetic_batch  # noqa: E402
from rsis_amd.train import build_optimizers, runIter  # noqa: E402
from rsis_amd.utils.objectives import MaskedBCELoss, MaskedNLLLoss, softIoULoss  # noqa: E402

a = bench.bench_args(32, 256, 10)
enc, dec = FeatureExtractor(a).cuda(), RSIS(a).cuda()
opts = list(build_optimizers(a, enc, dec))
crits = [softIoULoss(), MaskedNLLLoss(None), MaskedBCELoss(0.5)]
batch = synthetic_batch(1, 32, 256, 256, 20, 12, 21, "cuda")


def step():
    return runIter(a, enc, dec, *batch, crits, opts, mode="train", sync_losses=False)


for _ in range(3):
    step()
torch.cuda.synchronize()
# host-only time: how long the CPU needs to ENQUEUE a step (GPU drained before, not waited for after)
ts = []
for _ in range(5):
    torch.cuda.synchronize()
    t0 = time.time()
    step()
    ts.append(time.time() - t0)
    torch.cuda.synchronize()
print("host enqueue time per step (ms):", ["%.1f" % (t * 1e3) for t in ts])
pr = cProfile.Profile()
pr.enable()
for _ in range(3):
    step()
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(28)
