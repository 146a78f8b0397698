"""Print, for the trainstep_160 fixture, the HIP path's distance to the float64 truth next to the reference's own fp32 distance
(fp32 and bf16 kernels)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from helpers import sub_idx
from test_gpu_round2 import _train_step, _rel_l2

for dt in ("fp32", "bf16"):
    g, losses, outs, perms, named, pre, (B, T, H, W) = _train_step(dt)
    print("==", dt)
    for k, v in zip(("loss", "loss_mask_iou", "loss_stop", "loss_class"), losses):
        print("  %-14s hip-f64 %.3e   ref32-f64 %.3e" % (k, abs(float(v) - float(g["f64." + k])), abs(float(g[k]) - float(g["f64." + k]))))
    for nm, got in (("out_masks_sub", outs[0].view(B, T, H, W)[:, :, ::4, ::4]), ("out_classes", outs[1])):
        f64, ref = torch.from_numpy(g["f64." + nm]), torch.from_numpy(g[nm]).double()
        got = got.detach().double().cpu().reshape(f64.shape)
        print("  %-14s max|hip-f64| %.3e  max|ref32-f64| %.3e  relL2 hip %.3e ref %.3e" % (nm, float((got - f64).abs().max()),
              float((ref - f64).abs().max()), _rel_l2(got, f64), _rel_l2(ref, f64)))
    rows = []
    for k, p in named:
        flat = p.grad.detach().reshape(-1)
        cap = 2048 if (k.startswith("dec.") or not k.startswith("enc.base.")) else 64
        got = flat[sub_idx(flat.numel(), cap)].double().cpu()
        f64, ref = torch.from_numpy(g["f64.grad." + k]), torch.from_numpy(g["grad." + k]).double()
        sc = float(f64.abs().max()) + 1e-30
        rows.append((float((got - f64).abs().max()) / sc, float((ref - f64).abs().max()) / sc, _rel_l2(got, f64), _rel_l2(ref, f64), k))
    rows.sort(key=lambda r: -(r[0] / (r[1] + 1e-30)) if r[1] > 1e-12 else 0)
    print("  worst ratio (hip err / ref floor), max-norm relative to max|f64|:")
    for r in rows[:8]:
        print("   hip %.3e ref %.3e | relL2 hip %.3e ref %.3e  %s" % r)
    grp = [r for r in rows if not r[4].startswith("enc.base.")]
    grp.sort(key=lambda r: -r[2])
    print("  dec_opt group, largest rel L2:")
    for r in grp[:12]:
        print("   hip %.3e ref %.3e | relL2 hip %.3e ref %.3e  %s" % r)
