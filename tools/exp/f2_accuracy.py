import sys, os, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from rsis_amd import ops
torch.manual_seed(0)
for (B, Cin, Cout, H) in [(4, 256, 256, 32), (4, 512, 512, 16), (4, 128, 128, 64)]:
    x = torch.randn(B, Cin, H, H); w = torch.randn(Cout, Cin, 3, 3) / (3 * Cin ** 0.5)
    ref = torch.nn.functional.conv2d(x.double(), w.double(), None, 2, 1)
    pack = ops.PackedConv(3, [Cin], stride=2, pad=1)
    y = ops.conv2d([x.cuda()], w.cuda(), None, 2, 1, pack).cpu().double()
    e = y - ref
    print("conv %d->%d @%d s2: rms err %.3e  max err %.3e  (|ref| rms %.3f)" % (Cin, Cout, H, float(e.pow(2).mean().sqrt()), float(e.abs().max()), float(ref.pow(2).mean().sqrt())))
