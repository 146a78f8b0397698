"""bf16 vs fp32 kernels, encoder in EVAL mode (running BatchNorm statistics: well conditioned), decoder T steps, mask-only loss:
relative L2 distance of every gradient; then a short training run of both."""
import os, sys, copy, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from helpers import mk_args
from oracle import filler
from test_gpu_round2 import _models, _rel_l2

S = int(os.environ.get("S", 128)); B, T = 2, 3
x = filler.tensor(5, "evalbn.x", (B, 3, S, S)).cuda()
res = {}
for dt in ("fp32", "bf16"):
    a = mk_args(maxseqlen=T, dtype=dt)
    enc, dec, _, _ = _models(a, 44, 45)
    enc.eval(); dec.train()
    for p in list(enc.parameters()) + list(dec.parameters()):
        p.grad = None
    feats = enc(x)
    hidden, loss = None, 0.0
    for t in range(T):
        m, c, s, hidden = dec(feats, hidden)
        loss = loss + (m * filler.tensor(5, "evalbn.gm%d" % t, m.shape).cuda()).sum()
    loss.backward()
    res[dt] = ({("dec." + k): p.grad.clone() for k, p in dec.named_parameters() if p.grad is not None} |
               {("enc." + k): p.grad.clone() for k, p in enc.named_parameters() if p.grad is not None}, float(loss))
print("loss fp32 %.5f bf16 %.5f" % (res["fp32"][1], res["bf16"][1]))
rows = sorted(((_rel_l2(res["bf16"][0][k], res["fp32"][0][k]), k) for k in res["fp32"][0]), reverse=True)
for r in rows[:12]:
    print("  %.3e %s" % r)
for grp in ("dec.", "enc.sk", "enc.bn", "enc.base.layer4", "enc.base.layer3", "enc.base.layer2", "enc.base.layer1", "enc.base.conv1", "enc.base.bn1"):
    v = [r[0] for r in rows if r[1].startswith(grp)]
    if v:
        print("  %-18s n=%3d median %.3e max %.3e" % (grp, len(v), statistics.median(v), max(v)))

# short training run
from rsis_amd.synthetic import synthetic_batch
from rsis_amd.train import build_optimizers, runIter
from rsis_amd.utils.objectives import MaskedBCELoss, MaskedNLLLoss, softIoULoss
from rsis_amd.modules import FeatureExtractor, RSIS
batch = synthetic_batch(5, 8, 64, 64, 20, 3, 21, "cuda")
crits = [softIoULoss(), MaskedNLLLoss(None), MaskedBCELoss(0.5)]
torch.manual_seed(0)
a0 = mk_args(hidden_size=32, maxseqlen=3, optim="adam", optim_cnn="adam", lr=1e-3, lr_cnn=1e-6, weight_decay=1e-6, weight_decay_cnn=1e-6)
enc0, dec0 = FeatureExtractor(a0).cuda(), RSIS(a0).cuda()
for dt in ("fp32", "bf16"):
    a = copy.copy(a0); a.dtype = dt
    enc, dec = FeatureExtractor(a).cuda(), RSIS(a).cuda()
    enc.load_state_dict(enc0.state_dict()); dec.load_state_dict(dec0.state_dict())
    opts = list(build_optimizers(a, enc, dec))
    ls = [float(runIter(a, enc, dec, *batch, crits, opts, mode="train")[0][0]) for _ in range(40)]
    print(dt, " ".join("%.4f" % v for v in ls[::4]), "final %.4f" % ls[-1])
