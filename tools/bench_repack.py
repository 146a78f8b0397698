#!/usr/bin/env python
"""Time ops.repack_all() (rsis_conv_pack_batch: every packed weight copy of the model in one launch) on the GPU.
    python tools/bench_repack.py [--batch 8] [--imsize 128]
Prints the tile count, the packed / reference bytes and the launch time (HIP events, median of 20)."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--imsize", type=int, default=128)
    o = ap.parse_args()
    import bench
    from rsis_amd import ops
    from rsis_amd.modules import FeatureExtractor, RSIS
    from rsis_amd.synthetic import synthetic_batch
    from rsis_amd.train import build_optimizers, runIter, steps_to_run
    from rsis_amd.utils.objectives import MaskedBCELoss, MaskedNLLLoss, softIoULoss

    a = bench.bench_args(o.batch, o.imsize, 3)
    torch.manual_seed(0)
    encoder, decoder = FeatureExtractor(a).cuda(), RSIS(a).cuda()
    enc_opt, dec_opt = build_optimizers(a, encoder, decoder)
    crits = [softIoULoss(), MaskedNLLLoss(None), MaskedBCELoss(a.stop_balance_weight)]
    batch = synthetic_batch(1, o.batch, o.imsize, o.imsize, a.gt_maxseqlen, 12, a.num_classes, "cuda")
    t_run = steps_to_run(a, batch[3])
    for _ in range(2):      # creates every packed copy (forward + data-gradient)
        runIter(a, encoder, decoder, *batch, crits, [enc_opt, dec_opt], mode="train", sync_losses=False, t_run=t_run)
    torch.cuda.synchronize()
    packed = sum((p.wp.numel() if p.wp is not None else 0) + (p.wd.numel() if p.wd is not None else 0) for p in ops._PACKS)
    ref = sum(p.numel() for p in list(encoder.parameters()) + list(decoder.parameters()) if p.dim() == 4)
    times = []
    for _ in range(20):
        ops.bump_weight_epoch()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        ops.repack_all()
        e1.record()
        torch.cuda.synchronize()
        times.append(e0.elapsed_time(e1) * 1e3)
    times.sort()
    us = times[len(times) // 2]
    print("repack_all: %d jobs, %d tiles, packed %.1f MB, conv weights %.1f MB, median %.1f us (min %.1f)  ->  %.2f TB/s of "
          "write + read" % (ops._BATCH["n"], ops._BATCH["blocks"], packed * 4e-6, ref * 4e-6, us, times[0],
                            (packed * 4 + packed * 4) / us * 1e-6))


if __name__ == "__main__":
    main()
