import os, sys, torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from rsis_amd import blk_trunk, ops
from rsis_amd.modules.vision import Bottleneck, HipBatchNorm2d
def rel(a, b):
    a, b = a.double().flatten(), b.double().flatten()
    return float((a - b).norm() / (b.norm() + 1e-30))
def bf(x): return x.to(torch.bfloat16).float()
torch.manual_seed(4)
blk = Bottleneck(256, 64).cuda().train()
for m in blk.modules():
    if isinstance(m, HipBatchNorm2d):
        m.weight.data.uniform_(0.5, 1.5); m.bias.data.normal_(0, 0.2)
ops.set_dtype(blk, "bf16")
x0 = bf(torch.randn(8, 256, 28, 28, device="cuda").relu())
gy = bf(torch.randn(8, 256, 28, 28, device="cuda"))
res = {}
for name in ("fp32st", "blk"):
    blk.zero_grad(set_to_none=True)
    x = x0.clone().requires_grad_()
    y = blk_trunk.to_nchw(blk_trunk.layer_forward(torch.nn.Sequential(blk), blk_trunk.to_blk(x))) if name == "blk" else blk(x)
    y.backward(gy)
    res[name] = (y.detach().clone(), x.grad.clone(), {k: p.grad.clone() for k, p in blk.named_parameters()})
# float64 reference with bf16-rounded conv weights
xd = x0.double().requires_grad_()
P = {k: p.detach().double().requires_grad_() for k, p in blk.named_parameters()}
def bn(t, pre): return F.batch_norm(t, None, None, P[pre + ".weight"], P[pre + ".bias"], True, 0.0, 1e-5)
W = {k: bf(blk.state_dict()[k]).double().requires_grad_() for k in ("conv1.weight", "conv2.weight", "conv3.weight")}
o = F.relu(bn(F.conv2d(xd, W["conv1.weight"]), "bn1"))
o = F.relu(bn(F.conv2d(o, W["conv2.weight"], padding=1), "bn2"))
o = F.relu(bn(F.conv2d(o, W["conv3.weight"]), "bn3") + xd)
o.backward(gy.double())
for name in ("fp32st", "blk"):
    y, dx, g = res[name]
    print(name, "y %.4f dx %.4f" % (rel(y, o), rel(dx, xd.grad)), " ".join("%s %.4f" % (k.replace(".weight", ".w").replace(".bias", ".b"), rel(g[k], (W[k] if k in W else P[k]).grad)) for k in sorted(g)))
