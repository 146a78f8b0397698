#!/usr/bin/env python
"""bf16 training-quality evidence (reference src/train.py:159-187): the SAME 200 training iterations at
BASELINE configs[2]'s geometry (224 x 224, T = 10, batch 32, ResNet-101; all three losses on, both Adam optimizers stepping) under
`-dtype fp32` and `-dtype bf16`, from identical initial weights (torch's default initialisation: well conditioned, BatchNorm gamma 1)
and an identical stream of synthetic batches (16 distinct resident batches, cycled).  Prints / returns the two loss curves and the
soft-IoU loss component (1 - matched soft IoU, train.py:167) and the bands they stay within.

    python tools/bf16_training_quality.py [--iters 200] [--out profiles/r06_bf16_training_curve.txt]
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402


def run(iters=200, B=32, S=224, T=10, n_batches=16, seed=0, log=None, control=False):
    """control=True adds a third curve: fp32 again from the same weights perturbed by ONE PART IN A MILLION (w * (1 + 1e-6 n), n ~ N(0, 1)) --
    how far two fp32 trajectories of this chaotic system drift apart over the same iterations, i.e. the resolution of the comparison."""
    import bench
    from rsis_amd.modules import FeatureExtractor, RSIS
    from rsis_amd.synthetic import synthetic_batch
    from rsis_amd.train import GraphedStep, build_optimizers, steps_to_run
    from rsis_amd.utils.objectives import MaskedBCELoss, MaskedNLLLoss, softIoULoss
    batches = [synthetic_batch(100 + i, B, S, S, 20, 12, 21, "cuda") for i in range(n_batches)]
    init = None
    curves = {}
    for dtype in ("fp32", "bf16") + (("fp32_perturbed",) if control else ()):
        a = bench.bench_args(B, S, T, dtype.split("_")[0])
        torch.manual_seed(seed)
        enc, dec = FeatureExtractor(a).cuda(), RSIS(a).cuda()
        if init is None:
            init = ({k: v.clone() for k, v in enc.state_dict().items()}, {k: v.clone() for k, v in dec.state_dict().items()})
        enc.load_state_dict(init[0])
        dec.load_state_dict(init[1])
        if dtype == "fp32_perturbed":
            gen = torch.Generator(device="cuda").manual_seed(7)
            with torch.no_grad():
                for q in list(enc.parameters()) + list(dec.parameters()):
                    q.mul_(1.0 + 1e-6 * torch.randn(q.shape, device="cuda", generator=gen))
        opts = list(build_optimizers(a, enc, dec))
        crits = [softIoULoss(), MaskedNLLLoss(None), MaskedBCELoss(0.5)]
        t_run = steps_to_run(a, batches[0][3])
        g = GraphedStep(a, enc, dec, crits, opts, None, warm=2)
        tot, iou = [], []
        for it in range(iters):
            losses = g(batches[it % n_batches], t_run)[0]
            tot.append(losses[0].detach().clone())
            iou.append(losses[1].detach().clone())
        torch.cuda.synchronize()
        curves[dtype] = ([float(v) for v in tot], [float(v) for v in iou])
        g.release()
        del g, enc, dec, opts
        torch.cuda.empty_cache()
        if log:
            log("%s: loss %.4f -> %.4f, soft-IoU loss %.4f -> %.4f over %d iterations" % (
                dtype, curves[dtype][0][0], curves[dtype][0][-1], curves[dtype][1][0], curves[dtype][1][-1], iters))
    return curves


def bands(curves, window=20, other="bf16"):
    """the statements the test asserts: windowed means (20 iterations) of the two curves, relative distance per window"""
    out = {}
    for name, idx in (("loss", 0), ("soft_iou_loss", 1)):
        f, b = curves["fp32"][idx], curves[other][idx]
        n = len(f) // window
        wf = [sum(f[i * window:(i + 1) * window]) / window for i in range(n)]
        wb = [sum(b[i * window:(i + 1) * window]) / window for i in range(n)]
        out[name] = {"fp32": wf, "bf16": wb, "rel": [abs(x - y) / abs(x) for x, y in zip(wf, wb)]}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=200)
    ap.add_argument("--out", default="")
    o = ap.parse_args()
    lines = []

    def log(m):
        print(m, flush=True)
        lines.append(m)
    log("# bf16 vs fp32 training, configs[2] geometry (224x224, T=10, B=32), %d iterations, identical initial weights and batches" % o.iters)
    curves = run(o.iters, log=log, control=True)
    bd = bands(curves)
    for name, d in bd.items():
        log("%s, means over windows of 20 iterations:" % name)
        log("  fp32 " + " ".join("%.4f" % v for v in d["fp32"]))
        log("  bf16 " + " ".join("%.4f" % v for v in d["bf16"]))
        log("  |bf16 - fp32| / fp32 " + " ".join("%.4f" % v for v in d["rel"]) + "   (max %.4f)" % max(d["rel"]))
    bc = bands(curves, other="fp32_perturbed")
    for name, d in bc.items():
        log("CONTROL %s: fp32 from weights perturbed by 1e-6 relative, |perturbed - fp32| / fp32 per window " % name
            + " ".join("%.4f" % v for v in d["rel"]) + "   (max %.4f)" % max(d["rel"]))
    log("per-iteration loss, every 10th: fp32 " + " ".join("%.3f" % v for v in curves["fp32"][0][::10]))
    log("per-iteration loss, every 10th: bf16 " + " ".join("%.3f" % v for v in curves["bf16"][0][::10]))
    if o.out:
        with open(o.out, "w") as f:
            f.write("\n".join(lines) + "\n")


if __name__ == "__main__":
    main()
