"""Cityscapes evaluation driver -- python-3 / MI355X counterpart of reference src/eval_cityscapes.py (Python 2): checkpoint ->
`test()` -> per image the result files of the Cityscapes instance-level evaluation script (eval_cityscapes.py:96-174):
`<models_root>/<model_name>/<model_name>_results/<sample>.txt` with one line `<mask png> <class id> <score>` per (timestep, class)
and the mask PNGs (largest connected component of the thresholded mask, resized to the original image) under `<model_name>_masks/`.

    python -m rsis_amd.eval_cityscapes --synthetic -model_name <name> -num_classes 9 -imsize 512 -maxseqlen 20 -batch_size 8 [-dtype bf16]

The Cityscapes READER of the reference (src/dataloader/cityscapes.py: host-side PNG / json I/O) is out of scope,
so only `--synthetic` inputs are wired: square images of `-imsize` whose "original" size is taken as twice that, which exercises
the same resize path.  Everything after `test()` is the reference's procedure, through
`rsis_amd.eval_post.write_cityscapes_results`.  Deviations from the reference script: an empty mask is written as an
all-zero PNG (the reference reuses a stale `max_label` from the previous mask); scores are formatted from float64 (the reference
prints numpy float32 `str`), i.e. more digits of the same number.
"""
import os
import sys

import torch

from .args import get_parser
from .eval import load_models
from .eval_post import write_cityscapes_results
from .synthetic import SyntheticLoader
from .test import test


class Evaluate(object):
    def __init__(self, args):
        self.args, self.split, self.T = args, args.eval_split, args.maxseqlen
        if not getattr(args, "synthetic", False):
            raise Exception("only --synthetic inputs are wired in this build (the Cityscapes reader of the reference's src/dataloader is "
                            "host-side I/O outside the hot path: SURVEY.md section 8)")
        self.encoder, self.decoder = load_models(args)
        self.loader = SyntheticLoader(args, max(1, args.synthetic_batches // 4), args.seed + 7)
        self.sample_list = ["synthetic_%06d" % i for i in range(len(self.loader) * args.batch_size)]

    def create_figures(self):
        args = self.args
        results_dir = os.path.join(args.models_root, args.model_name, args.model_name + "_results")     # eval_cityscapes.py:99-104
        masks_dir = args.model_name + "_masks"
        os.makedirs(os.path.join(results_dir, masks_dir), exist_ok=True)
        print("Creating annotations for cityscapes validation...")
        acc, n_lines = 0, 0
        for x, _y_mask, _y_class, _sw_mask, _sw_class in self.loader:
            out_masks, out_scores, stop_probs = test(args, self.encoder, self.decoder, x)               # :108
            Hm, Wm = x.size(-2), x.size(-1)
            for s in range(out_masks.shape[0]):
                lines = write_cityscapes_results(args, self.sample_list[s + acc], out_masks[s].view(self.T, Hm, Wm), out_scores[s],
                                                 stop_probs[s], 2 * Hm, 2 * Wm, results_dir, masks_dir)
                n_lines += len(lines)
            acc += out_masks.shape[0]
        print("%d result lines for %d images -> %s" % (n_lines, acc, results_dir))
        return n_lines


if __name__ == "__main__":
    a = get_parser().parse_args()
    torch.manual_seed(a.seed)
    if not a.use_gpu or not torch.cuda.is_available():
        raise SystemExit("rsis_amd.eval_cityscapes needs the GPU: the HIP library is the only compute path")
    Evaluate(a).create_figures()
    sys.exit(0)
