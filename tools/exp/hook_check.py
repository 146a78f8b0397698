import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
os.environ.update(RSIS_FORCE_DIST="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29577", RANK="0", LOCAL_RANK="0", WORLD_SIZE="1")
import bench
from rsis_amd import ops
from rsis_amd.train import build_optimizers, init_distributed, runIter, steps_to_run
from rsis_amd.modules import FeatureExtractor, RSIS
from rsis_amd.optim import BucketedAllReduce
from rsis_amd.synthetic import synthetic_batch
from rsis_amd.utils.objectives import MaskedBCELoss, MaskedNLLLoss, softIoULoss
init_distributed()
a = bench.bench_args(32, 256, 10)
enc, dec = FeatureExtractor(a).cuda(), RSIS(a).cuda()
eo, do = build_optimizers(a, enc, dec)
red = BucketedAllReduce([do.group, eo.group], force=True)
fin = red.finish
def spy():
    print("buckets: %d, launched from hooks before finish(): %d" % (len(red.buckets), sum(1 for p in red._pending if p == 0)), "pending:", red._pending)
    return fin()
red.finish = spy
import time, torch.distributed as dist
_ar = dist.all_reduce
T0 = [0.0]
def logged(t, *a, **k):
    print("  all_reduce of %.1f MB issued %.1f ms after backward started (host time)" % (t.numel() * 4e-6, (time.time() - T0[0]) * 1e3))
    return _ar(t, *a, **k)
dist.all_reduce = logged
_bw = torch.Tensor.backward
def bw(self, *a, **k):
    T0[0] = time.time()
    r = _bw(self, *a, **k)
    print("  backward returned after %.1f ms (host time)" % ((time.time() - T0[0]) * 1e3))
    return r
torch.Tensor.backward = bw
crits = [softIoULoss(), MaskedNLLLoss(None), MaskedBCELoss(a.stop_balance_weight)]
batch = synthetic_batch(1, 32, 256, 256, a.gt_maxseqlen, 12, a.num_classes, "cuda")
for _ in range(3):
    runIter(a, enc, dec, *batch, crits, [eo, do], mode="train", reducer=red, sync_losses=False, t_run=steps_to_run(a, batch[3]))
torch.cuda.synchronize()
print("DIRECT_GRAD", ops.DIRECT_GRAD)
