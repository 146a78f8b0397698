"""GPU parity of the Winograd F(2x2, 3x3) path (rsis_amd/csrc/conv_wino.hip, RSIS_DTYPE_F32_WINO): the 3x3 / stride 1 / pad 1 convs of
the ResNet bottlenecks (torchvision Bottleneck.conv2 through reference src/modules/vision.py:16-19) forward and data gradient against a
FLOAT64 convolution on the host, next to the direct exact-f32 kernel on the same inputs.  Bars, stated before the kernel ran:

  * forward / data gradient within 2e-6 * sqrt(K) + 1e-6 of float64 on O(1) data (K = 9 Cin resp. 9 Cout) -- the bar of an fp32
    summation, one order below the 2e-5 * sqrt(K) the direct kernel's op test allows;
  * and no further from float64 than 2 x the DIRECT kernel's own error + 1e-6 on the same inputs;
  * the weight gradient is the direct path's (same kernel): checked at its usual bar;
  * batched repack (rsis_conv_pack_batch) == single-shot pack bit for bit; addend and bias in the epilogue; partial regions on maps
    that are not multiples of 16 (14 x 14 at 224 inputs, 13 x 17 at 200 x 264), several regions per image (32 x 64), odd widths."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from helpers import assert_close

pytestmark = pytest.mark.gpu


def _rng_t(seed, shape, scale=1.0):
    return torch.from_numpy(np.random.default_rng(seed).normal(0, scale, shape).astype(np.float32))


WINO_CASES = [
    # (B, Cin, Cout, H, W, bias)
    (2, 256, 256, 16, 16, False),     # layer 3 at 256 x 256
    (3, 256, 256, 14, 14, False),     # layer 3 at 224 x 224: one partial region
    (2, 64, 96, 13, 17, True),        # odd sizes, two regions, bias, Cin != Cout
    (1, 128, 128, 32, 64, False),     # 2 x 4 regions per image (layer 3 at 512 x 1024)
    (2, 32, 32, 5, 3, True),          # smaller than one region, smallest channel counts
    (9, 64, 64, 16, 16, False),       # region count not a multiple of the 8 XCD ranges
    (2, 512, 512, 8, 8, False),       # layer 4 geometry (a quarter of the tile slots used)
]


@pytest.mark.parametrize("case", WINO_CASES)
def test_winograd_conv_fwd_bwd_against_float64(case):
    from rsis_amd import ops
    from rsis_amd._lib import lib
    B, Cin, Cout, H, W, has_bias = case
    assert lib().rsis_conv_uses_wino(3, 1, 1, Cin, Cout, 1, 0) == 1
    x = _rng_t(1, (B, Cin, H, W))
    w = _rng_t(2, (Cout, Cin, 3, 3), 1.0 / np.sqrt(Cin * 9))
    b = _rng_t(3, (Cout,)) if has_bias else None
    x64, w64 = x.double().requires_grad_(), w.double().requires_grad_()
    b64 = b.double().requires_grad_() if has_bias else None
    ref = F.conv2d(x64, w64, b64, stride=1, padding=1)
    gy = _rng_t(4, tuple(ref.shape))
    ref.backward(gy.double())
    res = {}
    for name, dt in (("direct", ops.DTYPE_F32), ("wino", ops.DTYPE_F32_WINO)):
        xd, wd = x.cuda().requires_grad_(), w.cuda().requires_grad_()
        bd = b.cuda().requires_grad_() if has_bias else None
        pack = ops.PackedConv(3, [Cin], stride=1, pad=1, dtype=dt)
        out = ops.conv2d([xd], wd, bd, 1, 1, pack)          # (a training call: the weight requires grad -> the Winograd kernel under dt = WINO)
        out.backward(gy.cuda())
        torch.cuda.synchronize()
        res[name] = (out.detach().double().cpu(), xd.grad.double().cpu(), wd.grad.double().cpu(), bd.grad.double().cpu() if has_bias else None)
    e = {k: [float((v[0] - ref.detach()).abs().max()), float((v[1] - x64.grad).abs().max())] for k, v in res.items()}
    print("\n%r: |fwd - f64| direct %.3e winograd %.3e; |dx - f64| direct %.3e winograd %.3e" % (case, e["direct"][0], e["wino"][0], e["direct"][1], e["wino"][1]))
    assert_close("fwd", res["wino"][0], ref.detach(), 2e-6 * np.sqrt(9 * Cin) + 1e-6, 2e-6)
    assert_close("dx", res["wino"][1], x64.grad, 2e-6 * np.sqrt(9 * Cout) + 1e-6, 2e-6)
    assert e["wino"][0] <= 2 * e["direct"][0] + 1e-6 and e["wino"][1] <= 2 * e["direct"][1] + 1e-6
    assert_close("dW", res["wino"][2], w64.grad, 1e-4 * max(1.0, float(w64.grad.abs().max())), 1e-5)
    if has_bias:
        assert_close("db", res["wino"][3], b64.grad, 1e-4 * max(1.0, float(b64.grad.abs().max())), 1e-5)


def test_winograd_epilogue_addend_and_batched_repack():
    """(a) the data gradient with the residual branch's gradient summed in the epilogue (`addend`, the Bottleneck's identity path);
    (b) the packed copies written by the batched repack of a training step (rsis_conv_pack_batch, modes 7 / 8) are bit for bit
    the single-shot ones."""
    from rsis_amd import ops
    from rsis_amd._lib import check, int_array, lib, ptr, ptr_array, stream
    L = lib()
    B, C, H, W = 2, 64, 14, 18
    w = _rng_t(5, (C, C, 3, 3), 1.0 / 24).cuda()
    dy, add = _rng_t(6, (B, C, H, W)).cuda(), _rng_t(7, (B, C, H, W)).cuda()
    pack = ops.PackedConv(3, [C], stride=1, pad=1, dtype=ops.DTYPE_F32_WINO)
    wd = pack.dgrad(w)
    dx = torch.empty_like(dy)
    check(L.rsis_conv2d_dgrad(ptr(dy), B, C, H, W, ptr(wd), C, 3, 1, 1, ptr_array([dx]), int_array([C]), 1, H, W, ptr(add), 0, ops.DTYPE_F32_WINO,
                              stream()), "dgrad + addend")
    ref = F.conv_transpose2d(dy.double().cpu(), w.double().cpu(), stride=1, padding=1) + add.double().cpu()
    assert_close("dgrad + addend", dx, ref, 2e-6 * np.sqrt(9 * C) + 1e-6, 2e-6)
    # batched repack: change the weight, repack everything, compare with fresh single-shot packs
    wp1, wd1 = pack.fwd(w).clone(), pack.dgrad(w).clone()
    w2 = (w * 1.5 + 0.01).contiguous()
    fresh = ops.PackedConv(3, [C], stride=1, pad=1, dtype=ops.DTYPE_F32_WINO)
    f_wp, f_wd = fresh.fwd(w2).clone(), fresh.dgrad(w2).clone()
    w.copy_(w2)
    ops.bump_weight_epoch() if hasattr(ops, "bump_weight_epoch") else None
    ops.repack_all()
    torch.cuda.synchronize()
    assert not torch.equal(pack.wp, wp1) and torch.equal(pack.wp, f_wp) and torch.equal(pack.wd, f_wd)


def test_resnet_layer3_convs_take_the_winograd_copy():
    """the module rule (RSIS_WINOGRAD, default "64,128,256"): the stride-1 3x3 convs of layers 1-3 (3 + 3 + 22) get DTYPE_F32_WINO, nothing else
    does (not the stride-2 first blocks, not layer 4, not the skip convs); under -dtype bf16 none does"""
    from helpers import mk_args
    from rsis_amd import ops
    from rsis_amd.modules import FeatureExtractor
    enc = FeatureExtractor(mk_args()).cuda()
    wino = [k for k, m in enc.named_modules() if getattr(getattr(m, "_pack", None), "dtype", None) == ops.DTYPE_F32_WINO]
    if ops.WINOGRAD[0] is None:
        assert not wino
        return
    if ops.WINOGRAD[0] == {64, 128, 256}:
        assert len(wino) == 28 and all(k.startswith(("base.layer1.", "base.layer2.", "base.layer3.")) and k.endswith(".conv2") for k in wino), wino
        assert sum(k.startswith("base.layer3.") for k in wino) == 22
    ops.set_dtype(enc, "bf16")
    assert not [k for k, m in enc.named_modules() if getattr(getattr(m, "_pack", None), "dtype", None) == ops.DTYPE_F32_WINO]
    ops.set_dtype(enc, "fp32")
    assert len([k for k, m in enc.named_modules() if getattr(getattr(m, "_pack", None), "dtype", None) == ops.DTYPE_F32_WINO]) == len(wino)


def test_inference_calls_of_a_winograd_conv_run_the_direct_kernel():
    """no_grad calls (test(), eval.py) keep the direct kernel with segmented accumulation unless RSIS_WINOGRAD_INFER=1: bit-identical to a
    plain RSIS_DTYPE_F32 conv; with the switch on, the Winograd kernel answers (different bits, same value to fp32 rounding)."""
    from rsis_amd import ops
    x = _rng_t(11, (2, 64, 14, 14)).cuda()
    w = _rng_t(12, (64, 64, 3, 3), 1.0 / 24).cuda()
    pw, pd = ops.PackedConv(3, [64], stride=1, pad=1, dtype=ops.DTYPE_F32_WINO), ops.PackedConv(3, [64], stride=1, pad=1, dtype=ops.DTYPE_F32)
    prev = ops.WINOGRAD_INFER[0]
    try:
        with torch.no_grad():
            ops.WINOGRAD_INFER[0] = False
            a, b = ops.conv2d([x], w, None, 1, 1, pw), ops.conv2d([x], w, None, 1, 1, pd)
            ops.WINOGRAD_INFER[0] = True
            c = ops.conv2d([x], w, None, 1, 1, pw)
        assert torch.equal(a, b)
        assert not torch.equal(a, c) and float((a - c).abs().max()) < 2e-5
    finally:
        ops.WINOGRAD_INFER[0] = prev


@pytest.mark.parametrize("shape", [(2, 128, 64, 16, 16, True), (2, 64, 32, 32, 32, True), (3, 128, 64, 14, 18, False), (2, 64, 32, 20, 12, True)])
def test_winograd_gate_data_gradient_two_destinations(shape):
    """The decoder's gate data gradient (d(up) | dh_prev of reference clstm.py:43-47, explicit BPTT) through the Winograd kernel: the pack
    of the DYNAMIC channels of a ConvLSTM Gates weight (two channel segments with offsets, gate-interleaved rows: pack mode 8 with a
    segment map), two destinations splitting the output channels, grouped with a second job in one launch (rsis_conv2d_dgrad_batch) --
    against float64 and against the direct kernel's packed copy on the same inputs.  with_h = False: the t = 0 form (d(up) only)."""
    from rsis_amd import _lib, ops
    from rsis_amd._lib import check, lib, ptr, stream
    B, c_up, hid, H, W, with_h = shape
    c_skip = c_up
    L = lib()
    Ctot = c_up + c_skip + hid
    w = _rng_t(31, (4 * hid, Ctot, 3, 3), 1.0 / np.sqrt(9 * Ctot)).cuda()
    da = _rng_t(32, (B, 4 * hid, H, W)).cuda()            # gate-interleaved rows: row 4 j + gate
    segs, offs = [c_up, hid], [0, c_up + c_skip]
    res = {}
    for tag, dt in (("direct", ops.DTYPE_F32), ("wino", ops.DTYPE_F32_WINO)):
        pack = ops.PackedConv(3, segs, lstm_hid=hid, stride=1, pad=1, offs=offs, dtype=dt)
        wd = pack.dgrad(w)
        dup, dhp = torch.full((B, c_up, H, W), 7.0, device="cuda"), torch.full((B, hid, H, W), 7.0, device="cuda")
        dxs = [dup] + ([dhp] if with_h else [])
        jobs = (_lib.DgradJob * 2)()
        for j in jobs:       # the same job twice (second copy into scratch): exercises the grouped launch
            (j.dy, j.B, j.Cout, j.Hy, j.Wy, j.Wd, j.Cin_packed, j.ks, j.stride, j.pad, j.ndst, j.Hx, j.Wx, j.addend, j.tile, j.dtype) = (
                ptr(da), B, 4 * hid, H, W, ptr(wd), pack.cin, 3, 1, 1, len(dxs), H, W, None, 0, dt)
        scratch = [torch.empty_like(x) for x in dxs]
        for k, x in enumerate(dxs):
            jobs[0].dx[k], jobs[0].Cdx[k] = x.data_ptr(), x.shape[1]
            jobs[1].dx[k], jobs[1].Cdx[k] = scratch[k].data_ptr(), x.shape[1]
        check(L.rsis_conv2d_dgrad_batch(jobs, 2, stream()), "rsis_conv2d_dgrad_batch")
        torch.cuda.synchronize()
        assert all(torch.equal(a, b) for a, b in zip(dxs, scratch))
        res[tag] = [x.double().cpu() for x in dxs]
    # float64: the reference-layout weight has rows [i | f | o | g] x hid; interleaved row 4 j + gate <- reference row gate * hid + j
    perm = torch.arange(4 * hid).view(4, hid).t().reshape(-1)
    da_ref = torch.empty_like(da)
    da_ref[:, perm] = da                                   # da in reference row order
    full = F.conv_transpose2d(da_ref.double().cpu(), w.double().cpu(), stride=1, padding=1)      # gradient of all Ctot input channels
    want = [full[:, :c_up]] + ([full[:, c_up + c_skip:]] if with_h else [])
    for k, ref in enumerate(want):
        e_d, e_w = float((res["direct"][k] - ref).abs().max()), float((res["wino"][k] - ref).abs().max())
        print("dst %d: |direct - f64| %.3e, |winograd - f64| %.3e" % (k, e_d, e_w))
        assert_close("winograd dst %d" % k, res["wino"][k], ref, 2e-6 * np.sqrt(9 * 4 * hid) + 1e-6, 2e-6)
        assert e_w <= 2 * e_d + 1e-6
