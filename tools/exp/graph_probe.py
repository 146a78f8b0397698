"""Probe: can one whole training iteration (runIter: encoder, T decoder steps, matching, losses, backward, Adam, repack)
be captured into ONE hipGraph through torch.cuda.graph and replayed?  Prints eager vs replay step time and checks that the
replayed step produces the eager step's loss sequence."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

import bench  # noqa: E402
from rsis_amd.modules import FeatureExtractor, RSIS  # noqa: E402
from rsis_amd.synthetic import synthetic_batch  # noqa: E402
from rsis_amd.train import build_optimizers, runIter, steps_to_run  # noqa: E402
from rsis_amd.utils.objectives import MaskedBCELoss, MaskedNLLLoss, softIoULoss  # noqa: E402


def main():
    B, S, T = int(os.environ.get("B", 32)), int(os.environ.get("S", 256)), 10
    a = bench.bench_args(B, S, T)
    torch.manual_seed(a.seed)
    enc, dec = FeatureExtractor(a).cuda(), RSIS(a).cuda()
    enc_opt, dec_opt = build_optimizers(a, enc, dec)
    crits = [softIoULoss(), MaskedNLLLoss(None), MaskedBCELoss(a.stop_balance_weight)]
    batch = synthetic_batch(a.seed, B, S, S, a.gt_maxseqlen, 12, a.num_classes, "cuda")
    t_run = steps_to_run(a, batch[3])

    def step():
        return runIter(a, enc, dec, *batch, crits, [enc_opt, dec_opt], mode="train", reducer=None, sync_losses=False, t_run=t_run)[0]

    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3):
            step()
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(10):
        losses = step()
    torch.cuda.synchronize()
    print("eager: %.2f ms/step, loss %.5f" % ((time.time() - t0) * 100, float(losses[0])), flush=True)

    g = torch.cuda.CUDAGraph()
    t0 = time.time()
    with torch.cuda.graph(g):
        glosses = step()
    torch.cuda.synchronize()
    print("capture: %.2f s" % (time.time() - t0), flush=True)
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    t0 = time.time()
    n = 20
    for _ in range(n):
        g.replay()
    th = time.time() - t0
    torch.cuda.synchronize()
    dt = time.time() - t0
    print("replay: %.2f ms/step (host enqueue %.2f ms/step), loss %.5f" % (dt / n * 1000, th / n * 1000, float(glosses[0])), flush=True)


if __name__ == "__main__":
    main()
