"""Evaluation driver -- python-3 / MI355X counterpart of reference src/eval.py (a Python-2 file: `print` statements,
`dict.iteritems`), for the part of it that sits on the hot path's output: inference (`test()`), the per-instance
post-processing (`resize_mask`: resample to the original image size, threshold, ignore pixels, minimum size, run-length
encoding -- all on the GPU, rsis_amd/eval_post.py) and the COCO-style prediction records (`create_annotation`,
eval.py:129-142) with the reference's thresholds (-stop_th, -class_th, -mask_th, -min_size; eval.py:299-340).

    python -m rsis_amd.eval --synthetic -model_name <name> -batch_size 32 -maxseqlen 10 [-eval_split test]

writes <models_root>/<model_name>/<model_name>_<eval_split>_predictions.json (list of {image_id, category_id, category_name,
segmentation: COCO RLE, score}).  Not built (out of scope): the dataset readers (so only `--synthetic`
inputs are wired), pycocotools' COCOeval AP computation, the matplotlib display path.
"""
import json
import os
import sys

import numpy as np
import torch

from .args import get_parser
from .eval_post import encode_masks, resize_mask  # noqa: F401  (resize_mask: reference signature, eval.py:96-127)
from .modules.model import RSIS, FeatureExtractor
from .synthetic import SyntheticLoader
from .test import test
from .utils.utils import check_parallel, load_checkpoint


def create_annotation(args, imname, pred_mask, class_id, score, classes, is_valid=True):
    """reference eval.py:129-142: annotation record in the COCO ground-truth format, or None for an invalid mask"""
    if not is_valid:
        return None
    return {"image_id": imname, "category_id": class_id, "category_name": classes[class_id], "segmentation": pred_mask,
            "score": score}


def load_models(args):
    """eval.py:229-246 (also eval_cityscapes.py:50-94, eval_leaves.py:45-89): the checkpoint of -model_name under -models_root ->
    eval-mode encoder / decoder on the device, built from the LOADED args; args.num_classes / hidden_size follow the checkpoint.
    Without a checkpoint directory the modules are randomly initialised (synthetic smoke runs), said on stderr."""
    model_dir = os.path.join(args.models_root, args.model_name)
    if os.path.exists(os.path.join(model_dir, "encoder.pt")):
        encoder_dict, decoder_dict, _, _, load_args = load_checkpoint(args.model_name, args.use_gpu, root=args.models_root)
        load_args.use_gpu = args.use_gpu
        if getattr(args, "dtype", None):
            load_args.dtype = args.dtype
        encoder, decoder = FeatureExtractor(load_args), RSIS(load_args)
        encoder_dict, decoder_dict = check_parallel(encoder_dict, decoder_dict)
        encoder.load_state_dict(encoder_dict)
        decoder.load_state_dict(decoder_dict)
        args.num_classes, args.hidden_size = load_args.num_classes, load_args.hidden_size
    else:
        print("no checkpoint at %s: evaluating randomly initialised weights" % model_dir, file=sys.stderr)
        encoder, decoder = FeatureExtractor(args), RSIS(args)
    return encoder.cuda().eval(), decoder.cuda().eval()


class Evaluate(object):
    def __init__(self, args):
        self.args = args
        self.split = args.eval_split
        if not getattr(args, "synthetic", False):
            raise Exception("only --synthetic inputs are wired in this build (the dataset readers of the reference's "
                            "src/dataloader are host-side I/O outside the hot path: SURVEY.md section 8)")
        self.encoder, self.decoder = load_models(self.args)
        self.class_names = ["<eos>"] + ["class%d" % i for i in range(1, self.args.num_classes)]
        self.loader = SyntheticLoader(args, max(1, args.synthetic_batches // 4), args.seed + 7)
        self.sample_list = ["synthetic_%06d" % i for i in range(len(self.loader) * args.batch_size)]

    def _create_json(self):
        """eval.py:254-345: one record per (instance, class) with score = class probability * objectness"""
        args = self.args
        predictions, shown, acc = [], [], 0
        for inputs, _y_mask, _y_class, _sw_mask, _sw_class in self.loader:
            x = inputs
            out_masks, out_scores, stop_probs = test(args, self.encoder, self.decoder, x)       # eval.py:262
            scores = out_scores.cpu().numpy()
            stops = stop_probs.cpu().numpy()
            classes = np.argmax(scores, axis=-1)
            h, w = x.size(-2), x.size(-1)                          # (synthetic images: the "original" size is the input size)
            for s in range(out_masks.shape[0]):
                sample_idx = self.sample_list[s + acc]
                # all T masks of the image in one launch each: resample + threshold + area, then run-length encoding
                segs, areas, raws = encode_masks(out_masks[s], h, w, args.mask_th, None)
                for i in range(out_masks.shape[1]):
                    objectness = float(stops[s][i][0])
                    if objectness < args.stop_th:                  # eval.py:303-304
                        continue
                    max_class = 1 if args.class_th == 0.0 else int(classes[s][i])
                    is_valid = not (areas[i] < args.min_size * h * w)                            # eval.py:113-114
                    for cls_id in range(1, len(self.class_names)):                               # eval.py:316-319 (0 = eos)
                        score = float(scores[s][i][cls_id]) * objectness
                        ann = create_annotation(args, sample_idx, _jsonable(segs[i]), cls_id, score, self.class_names, is_valid)
                        if ann is None:
                            continue
                        if cls_id == max_class and score >= args.class_th:                       # eval.py:333
                            shown.append(create_annotation(args, sample_idx, _jsonable(raws[i]), cls_id, score, self.class_names,
                                                           is_valid))
                        predictions.append(ann)
            acc += out_masks.shape[0]
        return predictions, shown

    def run_eval(self):
        predictions, shown = self._create_json()
        out_dir = os.path.join(self.args.models_root, self.args.model_name)
        os.makedirs(out_dir, exist_ok=True)
        path = os.path.join(out_dir, "%s_%s_predictions.json" % (self.args.model_name, self.split))
        with open(path, "w") as f:
            json.dump(predictions, f)
        print("%d prediction records (%d above -class_th for display) from %d images -> %s" %
              (len(predictions), len(shown), len(self.sample_list), path))
        return predictions


def _jsonable(rle):
    return {"size": rle["size"], "counts": rle["counts"].decode("ascii")}


if __name__ == "__main__":
    parser = get_parser()
    a = parser.parse_args()
    torch.manual_seed(a.seed)
    if not a.use_gpu or not torch.cuda.is_available():
        raise SystemExit("rsis_amd.eval needs the GPU: the HIP library is the only compute path")
    Evaluate(a).run_eval()
