"""Diagnostic for the deterministic mode (rsis_set_deterministic): which tensors differ between two runs of the same training steps,
and at which step a graph-replayed run leaves an eager one.  `python tools/det_diag.py [--dtype bf16] [--B 16 --S 128 --T 5 --hid 128]
[--steps 8] [--graph]`."""
import argparse
import copy
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dtype", default="fp32")
    ap.add_argument("--B", type=int, default=16)
    ap.add_argument("--S", type=int, default=128)
    ap.add_argument("--T", type=int, default=5)
    ap.add_argument("--hid", type=int, default=128)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--graph", action="store_true", help="second run = GraphedStep (2 eager warm-up steps, then replays)")
    ap.add_argument("--nondet", action="store_true")
    o = ap.parse_args()
    from helpers import mk_args
    from rsis_amd import ops
    from rsis_amd.modules import FeatureExtractor, RSIS
    from rsis_amd.synthetic import synthetic_batch
    from rsis_amd.train import GraphedStep, build_optimizers, runIter, steps_to_run
    from rsis_amd.utils.objectives import MaskedBCELoss, MaskedNLLLoss, softIoULoss
    ops.set_deterministic(not o.nondet)
    a = mk_args(hidden_size=o.hid, maxseqlen=o.T, lr=1e-3, lr_cnn=1e-6, weight_decay=1e-6, weight_decay_cnn=1e-6, optim="adam",
                optim_cnn="adam", imsize=o.S, batch_size=o.B, seed=3, dtype=o.dtype)
    torch.manual_seed(0)
    enc0, dec0 = FeatureExtractor(a).cuda(), RSIS(a).cuda()
    batch = synthetic_batch(5, o.B, o.S, o.S, a.gt_maxseqlen, o.T + 1, a.num_classes, "cuda")
    t_run = steps_to_run(a, batch[3])
    crits = [softIoULoss(), MaskedNLLLoss(None), MaskedBCELoss(a.stop_balance_weight)]

    def run(graphed):
        enc, dec = copy.deepcopy(enc0), copy.deepcopy(dec0)
        opts = list(build_optimizers(a, enc, dec))
        g = GraphedStep(a, enc, dec, crits, opts, None, warm=2) if graphed else None
        names = [("dec." + k, p) for k, p in dec.named_parameters()] + [("enc." + k, p) for k, p in enc.named_parameters()]
        hist = []
        for _ in range(o.steps):
            out = g(batch, t_run) if graphed else runIter(a, enc, dec, *batch, crits, opts, mode="train", sync_losses=False, t_run=t_run,
                                                          want_outs=False)
            torch.cuda.synchronize()
            hist.append(([float(v) for v in out[0]], {k: p.grad.detach().clone() for k, p in names if p.grad is not None},
                         {k: p.detach().clone() for k, p in names}))
        if graphed:
            assert g.graph is not None, g.failed
            g.release()
        return hist

    h1, h2 = run(False), run(o.graph)
    for s, ((l1, g1, p1), (l2, g2, p2)) in enumerate(zip(h1, h2)):
        dg = {k: float((g1[k].double() - g2[k].double()).abs().max()) for k in g1 if not torch.equal(g1[k], g2[k])}
        dp = {k: float((p1[k].double() - p2[k].double()).abs().max()) for k in p1 if not torch.equal(p1[k], p2[k])}
        print("step %d: loss %.9f vs %.9f | %d grads differ, %d params differ" % (s, l1[0], l2[0], len(dg), len(dp)))
        for k, v in sorted(dg.items(), key=lambda kv: -kv[1])[:12]:
            print("      grad %-50s max|d| %.3e  (|g|max %.3e)" % (k, v, float(g1[k].abs().max())))
        if dg or dp:
            if s >= 3:
                break


if __name__ == "__main__":
    main()
