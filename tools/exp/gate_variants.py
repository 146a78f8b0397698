"""time the product-form gate launches (and their data gradients) of the two shallow levels under forced tile variants"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from rsis_amd import ops
from rsis_amd._lib import check, int_array, lib, ptr, ptr_array, stream
L = lib(); B = 32
for (segs, hid, hw) in bench.GATE_LAYERS[2:]:
    H = W = hw
    c_up, c_skip = segs[0], segs[1]
    cin = sum(segs) + hid
    w = torch.randn(4 * hid, cin, 3, 3, device="cuda") * (1.0 / (3.0 * cin ** 0.5))
    dyn = ops.PackedConv(3, [c_up, hid], lstm_hid=hid, offs=[0, c_up + c_skip])
    wd = dyn.fwd(w); wdg = dyn.dgrad(w)
    G = torch.randn(B, 4 * hid, H, W, device="cuda")
    up, h_prev, c_prev = (torch.randn(B, c, H, W, device="cuda") for c in (c_up, hid, hid))
    h, c, act = torch.empty_like(c_prev), torch.empty_like(c_prev), torch.empty(B, 4 * hid, H, W, device="cuda")
    pd, idd = ptr_array([up, h_prev]), int_array([c_up, hid])
    dxs = [torch.empty_like(up), torch.empty_like(h_prev)]
    px = ptr_array(dxs)
    f = 2.0 * B * H * W * (c_up + hid) * 9 * 4 * hid
    for v in (0, 3, 5, 7, 8, 9, 4):
        try:
            ms = bench._time_launch(lambda: check(L.rsis_convlstm_fwd(pd, idd, 2, B, H, W, ptr(wd), None, ptr(G), ptr(c_prev), ptr(h), ptr(c), ptr(act),
                                                                       hid, 3, 1, v, 0, stream()), "f"), 20)
            md = bench._time_launch(lambda: check(L.rsis_conv2d_dgrad(ptr(act), B, 4 * hid, H, W, ptr(wdg), c_up + hid, 3, 1, 1, px, idd, 2, H, W, None, v, 0,
                                                                       stream()), "d"), 20)
            print("%dx%d hid %d variant %d: fwd %.1f us %.1f TF | dgrad %.1f us %.1f TF" % (H, W, hid, v, ms * 1e3, f / ms / 1e9, md * 1e3, f / md / 1e9))
        except Exception as e:
            print("variant", v, "failed", e)
# trunk / skip 3x3 layers
from tools.bench_kernels import TRUNK
for cin, cout, ks, stride, hw, count in TRUNK:
    if ks != 3 or stride != 1 or hw < 16:
        continue
    x = torch.randn(B, cin, hw, hw, device="cuda"); w = torch.randn(cout, cin, 3, 3, device="cuda") / (3 * cin ** 0.5)
    pack = ops.PackedConv(3, [cin]); wp = pack.fwd(w); y = torch.empty(B, cout, hw, hw, device="cuda")
    pa, ia = ptr_array([x]), int_array([cin]); fl = 2.0 * B * hw * hw * cin * 9 * cout
    for v in (0, 3, 2, 7, 8, 9):
        ms = bench._time_launch(lambda: check(L.rsis_conv2d_fwd(pa, ia, 1, B, hw, hw, ptr(wp), cout, 3, 1, 1, None, None, ptr(y), hw, hw, v, 0, stream()), "f"), 20)
        print("conv %d->%d @%d variant %d: %.1f us %.1f TF" % (cin, cout, hw, v, ms * 1e3, fl / ms / 1e9))
