"""Inference: the eval-mode BatchNorm (+ residual add) (+ ReLU) folded into the fp32 conv epilogues (rsis_conv2d_fwd_bn_eval; reference path:
the conv -> BatchNorm pairs of src/modules/model.py:59-63 and of every torchvision Bottleneck behind src/modules/vision.py:12-19, as
test() / eval.py run them: .eval(), no autograd graph).

The fold shares ONE definition of the BatchNorm arithmetic with the stand-alone launch (csrc/common.h: rsis_bn_affine / rsis_bn_apply),
so the bar is equality of bits: per op on every kernel family the trunk uses (1x1 GEMM, strided 1x1, direct 3x3 on every tile variant
the dispatcher picks, the 7x7 stem, skip convs with bias), for the whole encoder, and for test() end to end -- whose outputs therefore
keep every golden-vector bar of tests/test_gpu_modules.py / test_gpu_hot.py unchanged (those tests now run the folded path)."""
import pytest
import torch

from helpers import gold, mk_args

pytestmark = pytest.mark.gpu


def _bn(C, seed):
    from rsis_amd.modules.vision import HipBatchNorm2d
    g = torch.Generator().manual_seed(seed)
    bn = HipBatchNorm2d(C)
    with torch.no_grad():
        bn.weight.copy_(torch.rand(C, generator=g) + 0.5)
        bn.bias.copy_(torch.randn(C, generator=g) * 0.3)
        bn.running_mean.copy_(torch.randn(C, generator=g) * 0.5)
        bn.running_var.copy_(torch.rand(C, generator=g) * 2 + 0.05)
    return bn.cuda().eval()


CASES = [
    # cin, cout, ks, stride, H, W, bias, relu, res          which kernel
    (64, 256, 1, 1, 64, 64, False, False, False),          # 1x1 GEMM (LDS-DMA path), downsample-style: no ReLU
    (256, 64, 1, 1, 64, 64, False, True, False),           # conv1 of a bottleneck
    (256, 1024, 1, 1, 16, 16, False, True, True),          # conv3 + residual + ReLU
    (512, 2048, 1, 1, 8, 8, False, True, True),
    (1024, 256, 1, 1, 14, 14, False, True, False),         # 224^2 geometry: H * W % 4 == 0
    (256, 512, 1, 2, 64, 64, False, False, False),         # strided 1x1 (sub-sampled copy)
    (64, 100, 1, 1, 15, 15, False, True, False),           # generic 1x1 (H * W % 4 != 0: not the LDS-DMA form), Cout tail inside a tile
    (64, 64, 3, 1, 64, 64, False, True, False),            # direct 3x3, 512-thread variant
    (128, 128, 3, 1, 32, 32, False, True, False),          # 32-row 16 x 8 tile
    (256, 256, 3, 1, 16, 16, False, True, False),
    (512, 512, 3, 1, 8, 8, False, True, False),            # 8 x 8 maps: K-split variant
    (256, 256, 3, 1, 14, 14, False, True, False),          # ragged map
    (2048, 128, 3, 1, 8, 8, True, False, False),           # sk5: bias, BatchNorm, no ReLU; 256 chunks deep
    (256, 32, 3, 1, 64, 64, True, False, False),           # sk2
    (64, 16, 3, 1, 128, 128, True, False, False),          # sk1: 16 output channels
    (3, 64, 7, 2, 128, 128, False, True, False),           # the stem
    (128, 128, 3, 2, 64, 64, False, True, False),          # 3x3 / stride 2 (conv2 of the first block of layers 2-4): 16 x 8 output tile
    (512, 512, 3, 2, 16, 16, False, True, False),          # ... 8 x 8 output tile, 64 rows
    (64, 64, 3, 2, 31, 29, False, True, False),            # ... odd map
    (24, 40, 3, 1, 20, 12, True, True, True),              # odd channel counts (3 chunks, a row tail), ragged map, everything on
]


@pytest.mark.parametrize("case", CASES, ids=lambda c: "%d-%d_k%ds%d_%dx%d%s%s%s" % (c[0], c[1], c[2], c[3], c[4], c[5], "_b" if c[6] else "",
                                                                                 "_relu" if c[7] else "", "_res" if c[8] else ""))
def test_fold_equals_the_two_launches(case):
    from rsis_amd import ops
    from rsis_amd.modules.vision import HipConv2d, conv_bn
    cin, cout, ks, stride, H, W, bias, relu, res = case
    torch.manual_seed(cin * 131 + cout)
    conv = HipConv2d(cin, cout, ks, stride=stride, padding=ks // 2 if ks > 1 else 0, bias=bias).cuda().eval()
    bn = _bn(cout, cin + 7 * cout)
    x = torch.randn(4, cin, H, W, device="cuda")
    with torch.no_grad():
        two = bn(conv(x), relu=False)                       # shape of the output
        r = torch.randn_like(two) if res else None
        two = bn(conv(x), res=r, relu=relu)
        one = ops.conv2d_bn_eval(x, conv.weight, conv.bias, conv.stride, conv.padding, conv._pack, bn.weight, bn.bias, bn.running_mean,
                                 bn.running_var, bn.eps, relu=relu, res=r)
        assert one is not None, "no folded epilogue for this conv: the case list names only covered ones"
        assert torch.equal(one, two), "max |diff| %.3e" % float((one - two).abs().max())
        assert torch.equal(conv_bn(conv, bn, x, relu=relu, res=r), two)
    if relu:
        assert float(one.min()) >= 0.0


def test_uncovered_convs_fall_back():
    """a conv without a folded epilogue (output channels not a multiple of 4: conv_out) is refused before anything is launched and conv_bn runs the
    two modules; so does any call that records an autograd graph or meets a BatchNorm in training mode"""
    from rsis_amd import ops
    from rsis_amd.modules.vision import HipConv2d, conv_bn
    torch.manual_seed(0)
    conv = HipConv2d(64, 6, 3, padding=1, bias=True).cuda().eval()
    bn = _bn(6, 3)
    x = torch.randn(2, 64, 32, 32, device="cuda")
    with torch.no_grad():
        assert ops.conv2d_bn_eval(x, conv.weight, conv.bias, 1, 1, conv._pack, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.eps) is None
        assert torch.equal(conv_bn(conv, bn, x, relu=True), bn(conv(x), relu=True))
    bn = _bn(64, 4)
    conv1 = HipConv2d(64, 64, 1, bias=False).cuda()
    assert ops.conv2d_bn_eval(x, conv1.weight, None, 1, 0, conv1._pack, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.eps) is None   # grad mode
    y = conv_bn(conv1, bn, x.requires_grad_(True), relu=True)
    y.sum().backward()
    assert x.grad is not None and conv1.weight.grad is not None


def test_library_refuses_before_launching():
    """rsis_conv2d_fwd_bn_eval on a conv without the epilogue (6 output channels) returns RSIS_ERR_UNSUPPORTED and leaves the output untouched"""
    from rsis_amd import ops
    from rsis_amd._lib import int_array, lib, ptr, ptr_array, stream
    x = torch.randn(2, 64, 32, 32, device="cuda")
    w = torch.randn(6, 64, 3, 3, device="cuda")
    pack = ops.PackedConv(3, [64], stride=1, pad=1)
    out = torch.full((2, 6, 32, 32), 7.0, device="cuda")
    v = torch.ones(6, device="cuda")
    rc = lib().rsis_conv2d_fwd_bn_eval(ptr_array([x]), int_array([64]), 1, 2, 32, 32, ptr(pack.fwd(w)), 6, 3, 1, 1, None, None, ptr(v), ptr(v),
                                       ptr(v), ptr(v), 1e-5, 1, ptr(out), 32, 32, 0, 0, stream())
    torch.cuda.synchronize()
    assert rc == 3 and bool((out == 7.0).all())
    rc = lib().rsis_conv2d_fwd_bn_eval(ptr_array([x]), int_array([64]), 1, 2, 32, 32, ptr(pack.fwd(w)), 6, 3, 1, 1, None, None, None, ptr(v),
                                       ptr(v), ptr(v), 1e-5, 1, ptr(out), 32, 32, 0, 0, stream())
    assert rc not in (0, 3)        # a missing BatchNorm array is an argument error


@pytest.mark.parametrize("hw", [(64, 64), (96, 80)])
def test_encoder_eval_fold_is_bit_identical(hw):
    """the whole FeatureExtractor in eval mode: folded (default) == RSIS_EVAL_FOLD=0, all five skip features"""
    from oracle import filler
    from oracle import rsis_oracle as O
    from rsis_amd import ops
    from rsis_amd.modules import FeatureExtractor
    a = mk_args(hidden_size=32)
    oenc = filler.fill_module(O.FeatureExtractor(a), seed=1).eval()
    enc = FeatureExtractor(a).cuda().eval()
    enc.load_state_dict(oenc.state_dict())
    x = filler.tensor(3, "fold.x", (2, 3) + hw).cuda()
    old = ops.EVAL_FOLD[0]
    try:
        with torch.no_grad():
            ops.EVAL_FOLD[0] = True
            f1 = enc(x)
            ops.EVAL_FOLD[0] = False
            f0 = enc(x)
    finally:
        ops.EVAL_FOLD[0] = old
    for k, (p, q) in enumerate(zip(f1, f0)):
        assert torch.equal(p, q), "skip feature %d: max |diff| %.3e" % (k, float((p - q).abs().max()))


def test_test_entry_point_fold_is_bit_identical_and_launches_less():
    """test() on the north-star fixture: same bits with and without the fold; the folded run launches no eval-mode bn_apply for the
    covered convs (counted through the library's launch counter when it has one, otherwise by timing-free inspection of the module path)"""
    from oracle import filler
    from oracle import rsis_oracle as O
    from rsis_amd import ops
    from rsis_amd.modules import RSIS, FeatureExtractor
    from rsis_amd.test import test as hip_test
    g = gold("e2e_256")
    a = mk_args(maxseqlen=int(g["T"]))
    oenc = filler.fill_module(O.FeatureExtractor(a), seed=44)
    odec = filler.fill_module(O.RSIS(a), seed=45)
    enc, dec = FeatureExtractor(a).cuda(), RSIS(a).cuda()
    enc.load_state_dict(oenc.state_dict())
    dec.load_state_dict(odec.state_dict())
    x = filler.tensor(44, "e2e_256.x", tuple(int(v) for v in g["shape"])).cuda()
    old = ops.EVAL_FOLD[0]
    calls = []
    real = ops.batchnorm

    def counting(*args, **kw):
        calls.append(1)
        return real(*args, **kw)

    try:
        ops.batchnorm = counting
        ops.EVAL_FOLD[0] = True
        o1 = hip_test(a, enc, dec, x, return_logits=True)
        n_fold = len(calls)
        del calls[:]
        ops.EVAL_FOLD[0] = False
        o0 = hip_test(a, enc, dec, x, return_logits=True)
        n_plain = len(calls)
    finally:
        ops.batchnorm = real
        ops.EVAL_FOLD[0] = old
    for p, q in zip(o1, o0):
        assert torch.equal(p, q)
    assert n_plain == 109 and n_fold == 0, (n_plain, n_fold)     # 104 trunk + 5 skip BatchNorms: every one of them in a conv epilogue


def test_bf16_model_stem_folds_too():
    """under -dtype bf16 the 7x7 stem (and any conv the library runs on the fp32 kernels anyway) takes the folded epilogue: the library, not
    the binding, decides; a conv that the bf16 kernels run is refused"""
    from rsis_amd import ops
    from rsis_amd.modules.vision import HipConv2d, conv_bn
    torch.manual_seed(5)
    conv = HipConv2d(3, 64, 7, stride=2, padding=3, bias=False).cuda().eval()
    conv._set_rsis_dtype(ops.DTYPE_BF16)
    bn = _bn(64, 11)
    x = torch.randn(2, 3, 96, 96, device="cuda")
    with torch.no_grad():
        one = ops.conv2d_bn_eval(x, conv.weight, None, 2, 3, conv._pack, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.eps, relu=True)
        assert one is not None and torch.equal(one, bn(conv(x), relu=True)) and torch.equal(conv_bn(conv, bn, x, relu=True), one)
    c3 = HipConv2d(64, 64, 3, padding=1, bias=False).cuda().eval()
    c3._set_rsis_dtype(ops.DTYPE_BF16)
    xx = torch.randn(2, 64, 16, 16, device="cuda")
    with torch.no_grad():
        assert ops.conv2d_bn_eval(xx, c3.weight, None, 1, 1, c3._pack, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.eps) is None
