#!/usr/bin/env python
"""Which aten ops launch the small torch-side kernels of a training step (adds, copies, fills): torch.profiler over one step,
grouped by (op name, input shapes)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from torch.profiler import ProfilerActivity, profile  # noqa: E402

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
    return runIter(a, enc, dec, *batch, crits, opts, mode="train", sync_losses=False, t_run=10, want_outs=False)


for _ in range(3):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    step()
    torch.cuda.synchronize()
rows = []
for e in prof.key_averages(group_by_input_shape=True):
    if e.key.startswith("aten::") and e.self_device_time_total > 0:
        rows.append((e.count, e.key, str(e.input_shapes)[:110], e.self_device_time_total))
rows.sort(reverse=True)
for r in rows[:70]:
    print("%5d  %-28s %9.1f us  %s" % (r[0], r[1], r[3], r[2]))

# every op / runtime call seen >= 8 times in the step (finds the sources of small copies and fills)
print("---- calls >= 8 per step (any device time)")
rows = []
for e in prof.key_averages():
    if e.count >= 8:
        rows.append((e.count, e.key[:70], e.self_device_time_total, e.self_cpu_time_total))
rows.sort(reverse=True)
for r in rows[:80]:
    print("%5d  %-70s dev %9.1f us  cpu %9.1f us" % r)
