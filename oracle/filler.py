"""Deterministic weight / input filler shared by the oracle, the golden generator and the
parity tests -- TEST INFRASTRUCTURE (see rsis_oracle.py header).

Uses numpy's default_rng (not torch RNG) so that the build container, the GPU box and
any torch version produce bit-identical tensors from a seed.  Scales are chosen so that
activations stay O(1) through 100+ layers and mask logits reach O(1-5) (default
init gives |logit|~0.1 where a 1e-4 check is too easy).
"""
import zlib
import numpy as np
import torch


def _rng(seed, name):
    return np.random.default_rng([int(seed), zlib.crc32(name.encode())])


def fill_module(module, seed=0, conv_gain=1.0, gates_gain=2.0):
    """Overwrite every parameter and BN buffer of `module` in state_dict-key order-independent
    fashion (each tensor's stream depends only on (seed, key))."""
    sd = module.state_dict()
    out = {}
    for key, t in sd.items():
        r = _rng(seed, key)
        shape = tuple(t.shape)
        if key.endswith("num_batches_tracked"):
            out[key] = torch.zeros_like(t)
            continue
        if key.endswith("running_mean"):
            v = r.normal(0.0, 0.1, shape)
        elif key.endswith("running_var"):
            v = r.uniform(0.5, 1.5, shape)
        elif t.dim() == 1 and (".bn" in "." + key or "downsample.1" in key):
            # BN affine: weight ~ U(0.5,1.5), bias ~ N(0,0.1)
            v = r.uniform(0.5, 1.5, shape) if key.endswith("weight") else r.normal(0.0, 0.1, shape)
        elif t.dim() == 4:
            fan_in = shape[1] * shape[2] * shape[3]
            g = gates_gain if "Gates" in key or "conv_out" in key else conv_gain
            v = r.normal(0.0, g * np.sqrt(1.0 / fan_in), shape)
        elif t.dim() == 2:
            v = r.normal(0.0, 2.0 * np.sqrt(1.0 / shape[1]), shape)
        else:  # biases
            v = r.normal(0.0, 0.1, shape)
        out[key] = torch.from_numpy(np.ascontiguousarray(v)).to(t.dtype)
    module.load_state_dict(out)
    return module


def tensor(seed, name, shape, scale=1.0, dtype=torch.float32):
    v = _rng(seed, name).normal(0.0, scale, tuple(shape))
    return torch.from_numpy(np.ascontiguousarray(v)).to(dtype)


def synthetic_targets(seed, B, H, W, gt_maxseqlen=20, n_inst=12, num_classes=21):
    """SURVEY 8(d) synthetic targets: n_inst axis-aligned rectangles per image (2-20 % area),
    classes U{1..C-1}, sw_mask[:, :n]=1, sw_class[:, :n+1]=1.  Returns fp32 masks (not the
    reference's float64) + long classes + float sample weights."""
    r = np.random.default_rng([int(seed), 777])
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
    return (torch.from_numpy(y_mask), torch.from_numpy(y_class), torch.from_numpy(sw_mask), torch.from_numpy(sw_class))
