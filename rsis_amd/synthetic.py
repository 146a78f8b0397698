"""Synthetic batches of SURVEY.md section 8(d) / BASELINE.md section 4, generated once on the host and kept resident
in HBM: x ~ N(0,1) (B,3,H,W); `n_inst` axis-aligned rectangles per image (2-20 % area, binary masks);
classes U{1..C-1}; sw_mask[:, :n]=1, sw_class[:, :n+1]=1.  Target layout = the reference's batch_to_var split of the
[B, gt_T, H*W+3] tensor (dataset.py:142-144, utils/utils.py:119-122) but fp32 instead of float64."""
import numpy as np
import torch


def synthetic_targets(seed, B, H, W, gt_maxseqlen=20, n_inst=12, num_classes=21):
    r = np.random.default_rng([int(seed), 777])
    n_inst = min(int(n_inst), int(gt_maxseqlen))      # (the reference keeps the gt_maxseqlen largest instances: dataset.py:126-131)
    y_mask = np.zeros((B, gt_maxseqlen, H * W), np.float32)
    y_class = np.zeros((B, gt_maxseqlen), np.int64)
    sw_mask = np.zeros((B, gt_maxseqlen), np.float32)
    sw_class = np.zeros((B, gt_maxseqlen), np.float32)
    for b in range(B):
        for g in range(n_inst):
            area = r.uniform(0.02, 0.20) * H * W
            ar = r.uniform(0.5, 2.0)
            h = int(np.clip(round(np.sqrt(area * ar)), 1, H))
            w = int(np.clip(round(area / max(h, 1)), 1, W))
            y0 = int(r.integers(0, H - h + 1))
            x0 = int(r.integers(0, W - w + 1))
            m = np.zeros((H, W), np.float32)
            m[y0:y0 + h, x0:x0 + w] = 1
            y_mask[b, g] = m.reshape(-1)
            y_class[b, g] = int(r.integers(1, num_classes))
        sw_mask[b, :n_inst] = 1
        sw_class[b, :min(n_inst + 1, gt_maxseqlen)] = 1
    return torch.from_numpy(y_mask), torch.from_numpy(y_class), torch.from_numpy(sw_mask), torch.from_numpy(sw_class)


def synthetic_batch(seed, B, H, W, gt_maxseqlen=20, n_inst=12, num_classes=21, device="cuda"):
    g = np.random.default_rng([int(seed), 123])
    x = torch.from_numpy(g.standard_normal((B, 3, H, W), dtype=np.float32))
    y_mask, y_class, sw_mask, sw_class = synthetic_targets(seed, B, H, W, gt_maxseqlen, n_inst, num_classes)
    return tuple(t.to(device) for t in (x, y_mask, y_class, sw_mask, sw_class))


class SyntheticLoader(object):
    """Iterates `n_batches` resident synthetic batches (a few distinct ones, cycled)."""

    def __init__(self, args, n_batches, seed, device="cuda", distinct=2, rank=0, batch_size=None):
        H = W = args.imsize
        self.batches = [synthetic_batch(seed + 1000 * rank + i, batch_size or args.batch_size, H, W, args.gt_maxseqlen,
                                        getattr(args, "synthetic_instances", 12), args.num_classes, device)
                        for i in range(distinct)]
        self.n = n_batches
        self._t_run = {}

    def steps_to_run(self, args, sw_mask):
        """early-stop rule of train.py:80-92 for a resident batch, evaluated once (no per-iteration host sync)"""
        from .train import steps_to_run
        key = (sw_mask.data_ptr(), getattr(args, "limit_seqlen_to", None), args.maxseqlen)
        if key not in self._t_run:
            self._t_run[key] = steps_to_run(args, sw_mask)
        return self._t_run[key]

    def __len__(self):
        return self.n

    def __iter__(self):
        for i in range(self.n):
            yield self.batches[i % len(self.batches)]
