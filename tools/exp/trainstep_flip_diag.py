"""bf16 vs fp32 kernels on the trainstep_160 inputs: (a) arg-max flips of the global max-pool side features, (b) gradient agreement
with and without the class / stop losses (the only consumers of the side features)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from helpers import mk_args
from oracle import filler
from test_gpu_round2 import _models, _rel_l2
from rsis_amd.train import build_optimizers, runIter
from rsis_amd.utils.objectives import MaskedBCELoss, MaskedNLLLoss, softIoULoss

B, H, W, T = 4, 160, 160, 4
x = filler.tensor(88, "trainstep_160.x", (B, 3, H, W)).cuda()
y_mask, y_class, sw_mask, sw_class = [t.cuda() for t in filler.synthetic_targets(88, B, H, W, gt_maxseqlen=20, n_inst=6)]
for heads in (True, False):
    res = {}
    for dt in ("fp32", "bf16"):
        a = mk_args(maxseqlen=T, optim="adam", optim_cnn="adam", lr=0.0, lr_cnn=0.0, weight_decay=0.0, weight_decay_cnn=0.0, dtype=dt,
                    use_class_loss=heads, use_stop_loss=heads)
        enc, dec, _, _ = _models(a, 88, 89)
        picks = []
        orig = dec.forward
        def fwd(feats, hidden, _o=orig, _p=picks):
            out = _o(feats, hidden)
            _p.append([h.detach().flatten(2).argmax(-1).cpu() for h, _c in out[3]])
            return out
        dec.forward = fwd
        opts = build_optimizers(a, enc, dec)
        crits = [softIoULoss(), MaskedNLLLoss(None), MaskedBCELoss(a.stop_balance_weight)]
        losses, _o, _p = runIter(a, enc, dec, x, y_mask, y_class.clone(), sw_mask.double(), sw_class.double(), crits, list(opts), mode="train",
                                 want_outs=False)
        grads = {("dec." + k): p.grad.detach().clone() for k, p in dec.named_parameters()}
        grads.update({("enc." + k): p.grad.detach().clone() for k, p in enc.named_parameters() if not k.startswith("base.fc")})
        res[dt] = (picks, grads, losses)
    print("== class/stop losses %s: losses fp32 %s bf16 %s" % (heads, [round(float(v), 5) for v in res["fp32"][2]], [round(float(v), 5) for v in res["bf16"][2]]))
    nflip = 0
    for t in range(T):
        for i in range(5):
            a_, b_ = res["fp32"][0][t][i], res["bf16"][0][t][i]
            n = int((a_ != b_).sum())
            nflip += n
            if n and heads:
                print("   t=%d level %d: %d of %d planes pick another pixel" % (t, i, n, a_.numel()))
    print("   total flips:", nflip)
    rows = sorted(((_rel_l2(res["bf16"][1][k], res["fp32"][1][k]), k) for k in res["fp32"][1] if not k.endswith(".bias") or "sk" not in k), reverse=True)
    for r in rows[:10]:
        print("   relL2 bf16 vs fp32-hip %.3e  %s" % r)
    import statistics
    print("   median rel L2 over %d tensors: %.3e" % (len(rows), statistics.median(r[0] for r in rows)))
