#!/usr/bin/env python
"""What the announced-once fallbacks of rsis_amd/train.py cost ("none is exercised by a perf number"): eager
training steps (no graph: the host-assignment fallback synchronises) at BASELINE configs[1] (256 x 256, T = 10, batch 32, fp32) with
  default            : the sequence decoder node, fused soft-IoU kernels, device assignment (gt_maxseqlen 20)
  per-step decoder   : RSIS_DECODER_SEQ=0 (what runs when decoder_seq.supported() is False)
  soft-IoU via bmm   : gt_maxseqlen = 40 (>= 32 GT slots: ops.softiou_supported is False), device assignment still applies
  host assignment    : gt_maxseqlen = 72 (> 64 GT slots: scipy on the host, one D2H sync per step) -- also soft-IoU via bmm
each in its own process.   python tools/fallback_cost.py [--steps 10]"""
import argparse
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child(gt, steps):
    import time
    import torch
    import bench
    from rsis_amd.modules import FeatureExtractor, RSIS
    from rsis_amd.synthetic import synthetic_batch
    from rsis_amd.train import build_optimizers, runIter, steps_to_run
    from rsis_amd.utils.objectives import MaskedBCELoss, MaskedNLLLoss, softIoULoss
    a = bench.bench_args(32, 256, 10, "fp32")
    a.gt_maxseqlen = gt
    torch.manual_seed(0)
    enc, dec = FeatureExtractor(a).cuda(), RSIS(a).cuda()
    opts = list(build_optimizers(a, enc, dec))
    crits = [softIoULoss(), MaskedNLLLoss(None), MaskedBCELoss(0.5)]
    batch = synthetic_batch(1, 32, 256, 256, gt, 12, a.num_classes, "cuda")
    t_run = steps_to_run(a, batch[3])
    for _ in range(3):
        runIter(a, enc, dec, *batch, crits, opts, mode="train", sync_losses=False, t_run=t_run, want_outs=False)
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(steps):
        runIter(a, enc, dec, *batch, crits, opts, mode="train", sync_losses=False, t_run=t_run, want_outs=False)
    torch.cuda.synchronize()
    print("RESULT %.2f" % (1e3 * (time.time() - t0) / steps))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--child", type=int, default=0)
    o = ap.parse_args()
    if o.child:
        child(o.child, o.steps)
        return
    rows = [("default (gt_maxseqlen 20)", 20, {}), ("per-step decoder (RSIS_DECODER_SEQ=0)", 20, {"RSIS_DECODER_SEQ": "0"}),
            ("soft-IoU via torch.bmm (gt_maxseqlen 40)", 40, {}), ("host assignment + bmm (gt_maxseqlen 72)", 72, {})]
    base = None
    for name, gt, env in rows:
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", str(gt), "--steps", str(o.steps)], capture_output=True, text=True,
                           env=dict(os.environ, **env), cwd=ROOT)
        ms = [float(l.split()[1]) for l in r.stdout.splitlines() if l.startswith("RESULT")]
        said = [l for l in r.stderr.splitlines() if "[rsis]" in l or "fallback" in l.lower() or "runs per step" in l or "bmm" in l or "host" in l]
        if not ms:
            print("%-45s FAILED: %s" % (name, r.stderr[-300:]))
            continue
        base = base or ms[0]
        print("%-45s %7.2f ms per EAGER step (%+.2f ms)   announced: %s" % (name, ms[0], ms[0] - base, "; ".join(s.strip()[:90] for s in said[:2]) or "-"))


if __name__ == "__main__":
    main()
