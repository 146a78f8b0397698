import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from rsis_amd import blk_trunk, ops
from rsis_amd.modules.vision import ResNet101
def rel(a, b):
    a, b = a.double().flatten(), b.double().flatten()
    return float((a - b).norm() / (b.norm() + 1e-30))
torch.manual_seed(0)
net = ResNet101().cuda().train()
ops.set_dtype(net, "bf16")
x = torch.randn(4, 3, 128, 128, device="cuda")
sd0 = {k: v.clone() for k, v in net.state_dict().items()}
outs = []
for on in (False, True):
    blk_trunk.ENABLED[0] = on
    net.load_state_dict(sd0)
    with torch.no_grad():
        outs.append([o.clone() for o in net(x)])
for i, (a, b) in enumerate(zip(outs[1], outs[0])):
    print("x%d" % (5 - i), tuple(a.shape), "rel", rel(a, b))
# block by block through layer1 on the same input
blk_trunk.ENABLED[0] = False
net.load_state_dict(sd0)
with torch.no_grad():
    x1 = net.bn1(net.conv1(x), relu=True)
    xp = ops.maxpool3x3s2(x1)
    ref = xp
    cur = ops.blk_from_nchw(xp)
    for li, layer in enumerate([net.layer1, net.layer2, net.layer3, net.layer4]):
        for bi, blk in enumerate(layer):
            net.load_state_dict(sd0)
            ref_out = blk(ops.blk_to_nchw(cur))          # fp32-storage block on the SAME (bf16-valued) input
            new, _ = blk_trunk._block_forward(blk, cur, False)
            print("layer%d.%d rel %.4f  |ref| %.3g" % (li + 1, bi, rel(ops.blk_to_nchw(new), ref_out), float(ref_out.abs().max())))
            cur = new
            if bi > 3:
                break
