import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from rsis_amd.modules import FeatureExtractor, RSIS
from rsis_amd.synthetic import synthetic_batch
from rsis_amd.train import GraphedStep, build_optimizers, runIter, steps_to_run
from rsis_amd.utils.objectives import MaskedBCELoss, MaskedNLLLoss, softIoULoss
if os.environ.get("ROOF", "1") == "1":
    r0 = bench.gate_kernel_roofline(32, 5, 256)
    r1 = bench.trunk_kernel_rooflines(32, 3, 256)
a = bench.bench_args(32, 256, 10)
torch.manual_seed(a.seed)
enc, dec = FeatureExtractor(a).cuda(), RSIS(a).cuda()
opts = list(build_optimizers(a, enc, dec))
crits = [softIoULoss(), MaskedNLLLoss(None), MaskedBCELoss(a.stop_balance_weight)]
batch = synthetic_batch(a.seed, 32, 256, 256, a.gt_maxseqlen, 12, a.num_classes, "cuda")
t_run = steps_to_run(a, batch[3])
g = GraphedStep(a, enc, dec, crits, opts, None, warm=2) if os.environ.get("GRAPH", "1") == "1" else None
ASYNC = os.environ.get("ASYNC", "0") == "1"
hist = []
for i in range(int(os.environ.get("N", 70))):
    r = g(batch, t_run) if g else runIter(a, enc, dec, *batch, crits, opts, mode="train", sync_losses=False, t_run=t_run, want_outs=False)
    if os.environ.get("EVT", "0") == "1" and i >= 3:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        r = g(batch, t_run)
        e1.record()
        e1.synchronize()
        continue
    if ASYNC and i >= 5:
        hist.append(torch.stack([v.detach().clone() for v in r[0]]))
        continue
    l = [float(v) for v in r[0]]
    if i < 6 or i % 10 == 0 or l[0] != l[0]:
        print(i, l, flush=True)
    if not all(v == v and abs(v) < 1e6 for v in l):
        bad = [k for k, p in list(enc.named_parameters()) + list(dec.named_parameters()) if not torch.isfinite(p).all()]
        print("non-finite params:", bad[:10], len(bad))
        break

if hist:
    torch.cuda.synchronize()
    H = torch.stack(hist).cpu()
    bad = (~torch.isfinite(H)).any(1).nonzero().flatten().tolist()
    print("async: first non-finite step:", (bad[0] + 5) if bad else None, "last losses", H[-1].tolist())

torch.cuda.synchronize()
for o in opts:
    gr = o.group
    print(gr.name, "non-finite m/v/g/p:", int((~torch.isfinite(gr.exp_avg)).sum()), int((~torch.isfinite(gr.exp_avg_sq)).sum()),
          int((~torch.isfinite(gr.flat_g)).sum()), int((~torch.isfinite(gr.flat_p)).sum()), "max|g| %.3e max v %.3e" % (float(gr.flat_g.abs().max()), float(gr.exp_avg_sq[torch.isfinite(gr.exp_avg_sq)].max())))
