"""CVPPP A1 evaluation driver -- python-3 / MI355X counterpart of reference src/eval_leaves.py (a Python-2 file with mixed tabs):
checkpoint -> `test()` over the -eval_split of the leaves directory -> one 8-bit label PNG per image in the CVPPP submission layout,
`<models_root>/<model_name>/<model_name>_results/A1/<sample>_label.png` (eval_leaves.py:91-125).

    python -m rsis_amd.eval_leaves -model_name <name> -dataset leaves -leaves_dir <dir> -eval_split val -batch_size 2 \
           -maxseqlen 16 -gt_maxseqlen 16 -imsize 256 --resize [-dtype bf16]

The label image itself is `rsis_amd.eval_post.leaves_label_image` (the reference's per-mask bytescale + bilinear resize + threshold,
later timesteps overwriting earlier ones).  Deviations from the reference script, all deliberate:
  * the last, short batch is evaluated sample by sample that exist (the reference indexes `range(batch_size)` past it and raises);
  * `-eval_split test` reads `-leaves_test_dir` and needs no ground truth (as the reference);
  * no matplotlib display path.
"""
import os
import sys

import torch

from .args import get_parser
from .dataloader.leaves import DeviceLoader, LeavesDataset
from .eval import load_models
from .eval_post import write_leaves_result
from .test import test


class Evaluate(object):
    def __init__(self, args):
        self.args, self.split, self.T = args, args.eval_split, args.maxseqlen
        self.dataset = LeavesDataset(args, split=self.split, augment=False, resize=args.resize, imsize=args.imsize)   # :36
        self.loader = DeviceLoader(self.dataset, args.batch_size, shuffle=False, num_workers=args.num_workers, seed=args.seed,
                                   drop_last=False)                                                                  # :38-41
        self.sample_list = self.dataset.get_sample_list()
        self.encoder, self.decoder = load_models(args)

    def create_figures(self):
        from PIL import Image
        args = self.args
        results_dir = os.path.join(args.models_root, args.model_name, args.model_name + "_results", "A1")
        os.makedirs(results_dir, exist_ok=True)
        print("Creating annotations for leaves validation...")
        acc, written = 0, []
        for x, _y_mask, _y_class, _sw_mask, _sw_class in self.loader:
            if x.size(0) == 1:                 # the reference's test() needs B >= 2 (model.py:169 squeeze); a lone tail image is doubled
                out_masks, _scores, stop_probs = test(args, self.encoder, self.decoder, torch.cat([x, x]))
                out_masks, stop_probs = out_masks[:1], stop_probs[:1]
            else:
                out_masks, _scores, stop_probs = test(args, self.encoder, self.decoder, x)      # eval_leaves.py:96
            Hm, Wm = x.size(-2), x.size(-1)
            for s in range(out_masks.shape[0]):
                path = self.sample_list[s + acc]
                with Image.open(path) as im:                                                    # :100-103 original size
                    w, h = im.size
                sample_idx = os.path.basename(path).split(".")[0]
                written.append(write_leaves_result(args, sample_idx, out_masks[s].view(self.T, Hm, Wm), stop_probs[s], h, w, results_dir))
            acc += out_masks.shape[0]
        print("%d label images -> %s" % (len(written), results_dir))
        return written


if __name__ == "__main__":
    a = get_parser().parse_args()
    torch.manual_seed(a.seed)
    if not a.use_gpu or not torch.cuda.is_available():
        raise SystemExit("rsis_amd.eval_leaves needs the GPU: the HIP library is the only compute path")
    Evaluate(a).create_figures()
    sys.exit(0)
