import os, sys, collections
sys.path.insert(0, os.getcwd())
import torch
from torch.profiler import ProfilerActivity, profile
import bench
from rsis_amd.modules import FeatureExtractor, RSIS
from rsis_amd.synthetic import synthetic_batch
from rsis_amd.train import build_optimizers, runIter
from rsis_amd.utils.objectives import MaskedBCELoss, MaskedNLLLoss, softIoULoss
a = bench.bench_args(32, 256, 10)
enc, dec = FeatureExtractor(a).cuda(), RSIS(a).cuda()
opts = list(build_optimizers(a, enc, dec))
crits = [softIoULoss(), MaskedNLLLoss(None), MaskedBCELoss(0.5)]
batch = synthetic_batch(1, 32, 256, 256, 20, 12, 21, "cuda")
def step(): return runIter(a, enc, dec, *batch, crits, opts, mode="train", sync_losses=False, t_run=10)
for _ in range(3): step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step(); torch.cuda.synchronize()
cnt = collections.Counter()
for e in prof.events():
    if 'emcpy' in e.name:
        chain = []
        q = e.cpu_parent
        while q is not None and len(chain) < 4:
            chain.append(q.name)
            q = q.cpu_parent
        cnt[(e.name, tuple(chain))] += 1
for k, v in cnt.most_common(20):
    print(v, k)
