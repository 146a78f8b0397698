import copy, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import torch
from helpers import mk_args
from rsis_amd.modules import FeatureExtractor, RSIS
from rsis_amd.synthetic import synthetic_batch
from rsis_amd.train import GraphedStep, build_optimizers, runIter, steps_to_run
from rsis_amd.utils.objectives import MaskedBCELoss, MaskedNLLLoss, softIoULoss
a = mk_args(hidden_size=32, maxseqlen=3, lr=1e-3, lr_cnn=1e-5, weight_decay=1e-6, weight_decay_cnn=1e-6, optim="adam",
            optim_cnn="adam", imsize=64, batch_size=4, seed=3)
batch = synthetic_batch(5, 4, 64, 64, a.gt_maxseqlen, 3, a.num_classes, "cuda")
t_run = steps_to_run(a, batch[3])
crits = [softIoULoss(), MaskedNLLLoss(None), MaskedBCELoss(a.stop_balance_weight)]
torch.manual_seed(0)
enc0, dec0 = FeatureExtractor(a).cuda(), RSIS(a).cuda()
for mode in ("eager", "eager", "graph", "graph_warm0"):
    enc, dec = copy.deepcopy(enc0), copy.deepcopy(dec0)
    opts = list(build_optimizers(a, enc, dec))
    g = GraphedStep(a, enc, dec, crits, opts, None, warm=2 if mode == "graph" else 0) if mode.startswith("graph") else None
    out = []
    for _ in range(6):
        r = g(batch, t_run) if g else runIter(a, enc, dec, *batch, crits, opts, mode="train", sync_losses=False, t_run=t_run, want_outs=False)
        out.append(float(r[0][0]))
    p = torch.cat([o.group.flat_p for o in opts])
    print(mode, " ".join("%.6f" % v for v in out), "|p| %.6f" % float(p.double().norm()), flush=True)
    if g: g.release()
