import sys, os, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from rsis_amd import ops
from rsis_amd.decoder_fused import _SideUpFn
from rsis_amd.modules.vision import HipConv2d
torch.manual_seed(3)
for (B, C, Hi, Wi) in [(2, 8, 16, 16), (2, 8, 128, 128), (1, 16, 12, 8)]:
    conv = HipConv2d(C, 1, 3, padding=1).cuda()
    h = torch.randn(B, C, Hi, Wi, device="cuda")
    size = (2 * Hi, 2 * Wi)
    with torch.no_grad():
        side, out = ops.side_upconv_out(h, conv.weight, conv.bias, size)
        s0, up = _SideUpFn.apply(None, 0, h, size)
        o0 = conv(up)
        ref = torch.nn.functional.conv2d(torch.nn.functional.interpolate(h.double(), size=size, mode="bilinear", align_corners=True), conv.weight.double(), conv.bias.double(), padding=1)
    d = (out - o0).abs()
    print((B, C, Hi, Wi), "fused vs unfused max %.3e at %s ; fused vs fp64 %.3e ; unfused vs fp64 %.3e" % (float(d.max()), tuple(int(v) for v in (d == d.max()).nonzero()[0]), float((out.double() - ref).abs().max()), float((o0.double() - ref).abs().max())))
B, C, Hi, Wi = 2, 8, 16, 16
conv = HipConv2d(C, 1, 3, padding=1).cuda()
h = torch.randn(B, C, Hi, Wi, device="cuda")
size = (2 * Hi, 2 * Wi)
with torch.no_grad():
    side, out = ops.side_upconv_out(h, conv.weight, conv.bias, size)
    s0, up = _SideUpFn.apply(None, 0, h, size)
    o0 = conv(up)
bad = ((out - o0).abs() > 1e-5).nonzero()
print("bad count", len(bad), "of", out.numel())
import collections
print("bad rows:", sorted(collections.Counter(int(v[2]) for v in bad).items()))
print("bad cols:", sorted(collections.Counter(int(v[3]) for v in bad).items()))
print("bad b:", sorted(collections.Counter(int(v[0]) for v in bad).items()))
