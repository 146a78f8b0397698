"""GPU parity tests of the channel-blocked bf16 kernels (rsis_amd/csrc/conv_blk.hip, blk_norm.hip): the storage half of the bf16
path (BASELINE.json configs[2..4]).  A blk tensor is the logical [B][C][H][W] tensor stored as bf16 [B][C/8][H][W][8].

Semantics under test (stated before measuring): the kernels compute in fp32 on EXACT bf16 inputs and round ONCE to bf16 at the
store.  The reference is therefore the same op in float64 on the same bf16-valued inputs (and bf16-rounded weights for the convs);
a result may differ from it by half a bf16 ulp of the exact value (2^-9 relative) plus the fp32 accumulation noise
(1e-5 of the output scale), and nothing else."""
import numpy as np
import pytest
import torch

from helpers import assert_close

pytestmark = pytest.mark.gpu

HALF_ULP = 2.0 ** -8        # |x - bf16(x)| <= 2^-9 |x|; the bar leaves a factor 2 for values next to a binade boundary


def to_blk(x):
    """fp32 NCHW -> the blk layout, by torch (round-to-nearest-even)"""
    B, C, H, W = x.shape
    return x.view(B, C // 8, 8, H, W).permute(0, 1, 3, 4, 2).contiguous().to(torch.bfloat16)


def from_blk(y):
    B, Cb, H, W, _ = y.shape
    return y.float().permute(0, 1, 4, 2, 3).reshape(B, Cb * 8, H, W)


def _bf16(x):
    return x.to(torch.bfloat16).float()


def test_layout_converters_round_trip():
    from rsis_amd import ops
    torch.manual_seed(0)
    x = torch.randn(3, 24, 5, 7, device="cuda") * 3
    y = ops.blk_from_nchw(x)
    assert y.dtype == torch.bfloat16 and tuple(y.shape) == (3, 3, 5, 7, 8)
    assert torch.equal(y, to_blk(x))
    assert torch.equal(ops.blk_to_nchw(y), _bf16(x))


SHAPES = [  # (B, Cin, Cout, H, W, ks)
    (2, 64, 64, 56, 56, 3), (2, 256, 256, 14, 14, 3), (3, 512, 512, 7, 7, 3), (2, 128, 128, 28, 28, 3), (2, 32, 48, 10, 13, 3),
    (2, 64, 256, 56, 56, 1), (2, 1024, 256, 14, 14, 1), (3, 2048, 512, 7, 7, 1), (2, 256, 64, 56, 56, 1), (2, 40, 24, 9, 11, 1),
    (2, 16, 8, 5, 5, 3), (1, 8, 8, 3, 3, 1),
]


@pytest.mark.parametrize("shape", SHAPES, ids=lambda s: "x".join(str(v) for v in s))
@pytest.mark.parametrize("variant", [0, 1, 2, 3, 4, 5, 6])
def test_blk_conv_forward_and_data_gradient(shape, variant):
    """rsis_blk_conv2d as the forward conv and, on the data-gradient pack, as conv_transpose (reference: the trunk convs of
    vision.py:12-19 / torchvision bottlenecks, bias-free) -- every tile variant on trunk shapes, ragged maps and channel tails."""
    from rsis_amd import ops
    B, Cin, Cout, H, W, ks = shape
    if ks == 1 and variant == 6:
        pytest.skip("five 1x1 variants")
    torch.manual_seed(sum(shape))
    x = _bf16(torch.randn(B, Cin, H, W, device="cuda"))
    w = torch.randn(Cout, Cin, ks, ks, device="cuda") / (Cin * ks * ks) ** 0.5
    pk = ops.PackedConv(ks, [Cin], stride=1, pad=ks // 2, dtype=ops.DTYPE_BF16)
    y = from_blk(ops.blk_conv2d(to_blk(x), pk.fwd(w), Cout, ks, variant))
    ref = torch.nn.functional.conv2d(x.double(), _bf16(w).double(), padding=ks // 2)
    scale = float(ref.abs().max())
    assert_close("fwd", y, ref, 1e-5 * scale, HALF_ULP)
    dy = _bf16(torch.randn(B, Cout, H, W, device="cuda"))
    dx = from_blk(ops.blk_conv2d(to_blk(dy), pk.dgrad(w), Cin, ks, variant))
    dref = torch.nn.functional.conv_transpose2d(dy.double(), _bf16(w).double(), padding=ks // 2)
    assert_close("dgrad", dx, dref, 1e-5 * float(dref.abs().max()), HALF_ULP)
    if variant == 0:      # + addend (the identity branch's gradient joining a residual block's first data gradient)
        add = _bf16(torch.randn(B, Cin, H, W, device="cuda"))
        dx2 = from_blk(ops.blk_conv2d(to_blk(dy), pk.dgrad(w), Cin, ks, 0, addend=to_blk(add)))
        ref2 = dref + add.double()
        assert_close("dgrad+addend", dx2, ref2, 1e-5 * float(ref2.abs().max()), HALF_ULP)


@pytest.mark.parametrize("shape", [(4, 64, 14, 14), (2, 256, 28, 28), (32, 16, 7, 7), (3, 24, 5, 9), (2, 8, 1, 1), (8, 64, 56, 56), (32, 24, 14, 14)],
                         ids=lambda s: "x".join(str(v) for v in s))
@pytest.mark.parametrize("relu,res", [(True, False), (True, True), (False, False)])
def test_blk_batchnorm_forward_backward(shape, relu, res):
    """Train-mode nn.BatchNorm2d (+ residual) (+ ReLU) on blk tensors -- the bn1 / bn2 / bn3 + add + relu of a torchvision bottleneck
    (vision.py:12-19) -- against the same graph in float64 on the same bf16-valued inputs: output and gradients within half a bf16
    ulp + fp32 noise, batch statistics / running statistics / parameter gradients (fp32 outputs) to 1e-5 relative."""
    from rsis_amd import ops
    B, C, H, W = shape
    torch.manual_seed(C + H)
    x = _bf16(torch.randn(B, C, H, W, device="cuda") * 2 + 0.5)
    r = _bf16(torch.randn(B, C, H, W, device="cuda")) if res else None
    gamma = torch.rand(C, device="cuda") + 0.5
    beta = torch.randn(C, device="cuda") * 0.3
    rm, rv = torch.randn(C, device="cuda") * 0.1, torch.rand(C, device="cuda") + 0.5
    rm0, rv0 = rm.clone(), rv.clone()
    y, sm, sr = ops.blk_bn_fwd(to_blk(x), to_blk(r) if res else None, gamma, beta, rm, rv, 1e-5, 0.1, relu, True)
    xd = x.double().requires_grad_()
    rd = r.double().requires_grad_() if res else None
    gd, bd = gamma.double().requires_grad_(), beta.double().requires_grad_()
    rmd, rvd = rm0.double(), rv0.double()
    ref = torch.nn.functional.batch_norm(xd, rmd, rvd, gd, bd, True, 0.1, 1e-5)
    if res:
        ref = ref + rd
    if relu:
        ref = torch.relu(ref)
    scale = float(ref.detach().abs().max()) + 1e-6
    assert_close("y", from_blk(y), ref, 2e-5 * scale, HALF_ULP)
    N = B * H * W
    if N > 1:
        assert_close("save_mean", sm, x.double().mean((0, 2, 3)), 1e-5, 1e-5)
        assert_close("save_rstd", sr, 1.0 / torch.sqrt(x.double().var((0, 2, 3), unbiased=False) + 1e-5), 1e-5, 1e-4)
        assert_close("running_mean", rm, rmd, 1e-5, 1e-5)
        assert_close("running_var", rv, rvd, 1e-5, 1e-4)
    # backward: the mask of the reference is taken from the PRODUCT's rounded output (y == 0 exactly where the bf16 result is 0)
    dy = _bf16(torch.randn(B, C, H, W, device="cuda"))
    dx, dres, dg, db = ops.blk_bn_bwd(to_blk(dy), to_blk(x), y if (relu and res) else None, gamma, beta, sm, sr, relu, res)
    ref.backward(dy.double())
    gs = float(xd.grad.abs().max()) + 1e-6
    # elements whose pre-activation is within rounding of 0 may fall on either side of the ReLU: compare where |pre| is clear of 0
    pre = torch.nn.functional.batch_norm(x.double(), None, None, gd.detach(), bd.detach(), True, 0.0, 1e-5) + (r.double() if res else 0)
    clear = (pre.abs() > 2.0 ** -7 * (pre.abs() + 1)).float() if relu else torch.ones_like(pre).float()
    frac = float(clear.mean())
    assert frac > 0.9
    # sums over the batch see the few flipped elements: bars scale with their count
    nflip = float((1 - clear).sum())
    # (2 x 8 x 1 x 1: two samples per channel, dx is pure cancellation -- the absolute bar is fp32 noise of the terms that cancel)
    cancel = 1e-5 * float((gamma * sr).max()) * float(dy.abs().max())
    assert_close("dx", from_blk(dx) * clear, xd.grad * clear, 2e-4 * gs + 8.0 * nflip / max(N, 1) * gs + cancel, 2 * HALF_ULP)
    if res:
        assert_close("dres", from_blk(dres) * clear, rd.grad * clear, 1e-6, HALF_ULP)
    tol = 1e-4 * (float(gd.grad.abs().max()) + float(bd.grad.abs().max()) + 1) + 4.0 * nflip
    assert_close("dgamma", dg, gd.grad, tol, 1e-4)
    assert_close("dbeta", db, bd.grad, tol, 1e-4)


def test_blk_batchnorm_eval_mode_and_reproducibility():
    from rsis_amd import ops
    torch.manual_seed(3)
    B, C, H, W = 4, 32, 9, 9
    x = _bf16(torch.randn(B, C, H, W, device="cuda"))
    gamma, beta = torch.rand(C, device="cuda") + 0.5, torch.randn(C, device="cuda")
    rm, rv = torch.randn(C, device="cuda") * 0.1, torch.rand(C, device="cuda") + 0.5
    y, _sm, _sr = ops.blk_bn_fwd(to_blk(x), None, gamma, beta, rm.clone(), rv.clone(), 1e-5, 0.1, True, False)
    ref = torch.relu(torch.nn.functional.batch_norm(x.double(), rm.double(), rv.double(), gamma.double(), beta.double(), False, 0.1, 1e-5))
    assert_close("eval y", from_blk(y), ref, 2e-5 * float(ref.abs().max()), HALF_ULP)
    outs = [ops.blk_bn_fwd(to_blk(x), None, gamma, beta, rm.clone(), rv.clone(), 1e-5, 0.1, True, True) for _ in range(3)]
    assert all(torch.equal(o[0], outs[0][0]) and torch.equal(o[1], outs[0][1]) and torch.equal(o[2], outs[0][2]) for o in outs[1:])


@pytest.mark.parametrize("hw", [(8, 8), (7, 7), (9, 14)])
def test_blk_subsample_and_its_transpose(hw):
    from rsis_amd import ops
    H, W = hw
    torch.manual_seed(1)
    x = _bf16(torch.randn(2, 16, H, W, device="cuda"))
    y = ops.blk_subsample(to_blk(x), 2)
    assert torch.equal(from_blk(y), x[:, :, ::2, ::2])
    dy = _bf16(torch.randn_like(x[:, :, ::2, ::2]))
    dx = from_blk(ops.blk_upscatter(to_blk(dy.contiguous()), H, W, 2))
    want = torch.zeros_like(x)
    want[:, :, ::2, ::2] = dy
    assert torch.equal(dx, want)


# every (rows per block, tile width) instantiation of the 3x3 DMA / transposing-read kernel (conv_wgrad_bf16.hip, wgrad3_tr_body):
# 32 / 64 / 128 rows x 32- / 16- / 8-pixel tiles, ragged maps, channel counts that leave partial slabs, several images per split
W3T_SHAPES = [(2, 24, 32, 112, 112, 3), (3, 40, 24, 14, 14, 3), (5, 16, 8, 7, 7, 3), (2, 520, 72, 28, 28, 3), (2, 520, 72, 14, 14, 3),
              (3, 520, 72, 7, 7, 3), (8, 128, 128, 7, 7, 3), (2, 64, 48, 56, 56, 3), (3, 48, 16, 33, 20, 3), (2, 72, 136, 5, 11, 3)]


@pytest.mark.parametrize("shape", SHAPES + W3T_SHAPES + [(32, 256, 256, 14, 14, 3), (32, 64, 64, 56, 56, 1), (4, 32, 64, 28, 28, 3),
                                                         (4, 24, 40, 17, 9, 3)],
                         ids=lambda s: "x".join(str(v) for v in s))
def test_blk_conv_weight_gradient(shape):
    """dW of the stride-1 convs from blk dy / x (rsis_conv2d_wgrad, RSIS_DTYPE_BF16_BLK: 1x1 -- 8 x 8 register transposition at
    staging; 3x3 -- cells by DMA into LDS, operands by the transposing LDS read)
    against float64 on the same bf16-valued operands: fp32 accumulation only -- 1e-5 of the gradient scale + 1e-4 relative
    (split-K partial sums meet in fp32 atomics) -- and a second call accumulates."""
    from rsis_amd import ops
    B, Cin, Cout, H, W, ks = shape
    torch.manual_seed(sum(shape) + 1)
    x = _bf16(torch.randn(B, Cin, H, W, device="cuda"))
    dy = _bf16(torch.randn(B, Cout, H, W, device="cuda"))
    dW = torch.zeros(Cout, Cin, ks, ks, device="cuda")
    ops.blk_conv_wgrad(to_blk(dy), to_blk(x), dW, ks)
    xd = x.double()
    wd = torch.zeros(Cout, Cin, ks, ks, dtype=torch.float64, device="cuda", requires_grad=True)
    torch.nn.functional.conv2d(xd, wd, padding=ks // 2).backward(dy.double())
    scale = float(wd.grad.abs().max())
    assert_close("dW", dW, wd.grad, 2e-5 * scale, 1e-4)
    ops.blk_conv_wgrad(to_blk(dy), to_blk(x), dW, ks)
    assert_close("dW x2", dW, 2 * wd.grad, 4e-5 * scale, 1e-4)


def _rel_l2(a, b):
    a, b = a.double().flatten(), b.double().flatten()
    return float((a - b).norm() / (b.norm() + 1e-30))


@pytest.mark.parametrize("kind", ["identity", "downsample_s1", "downsample_s2"])
def test_blk_bottleneck_forward_backward(kind):
    """One torchvision bottleneck (vision.py:12-19: conv1x1-BN-ReLU, conv3x3(/s)-BN-ReLU, conv1x1-BN, + identity or 1x1(/s)-BN
    downsample, ReLU), train mode, forward + backward: the blk node of rsis_amd/blk_trunk.py against the same block in float64 (bf16
    conv weights, bf16-valued input).  Any two roundings of the forward flip the ReLU mask of the elements next to 0, and a relative
    L2 of gradients is the square root of the flipped fraction -- ~5 % here for the fp32-storage bf16 kernels too.  So the bar is
    RELATIVE to that path (same kernels on fp32 NCHW activations, RSIS_BF16_STORAGE=0), evaluated on the same fixture: every
    tensor's distance to float64 at most 1.5 x the fp32-storage path's + 1 %; the output itself within 1 %."""
    import torch.nn.functional as F
    from rsis_amd import blk_trunk, ops
    from rsis_amd.modules.vision import Bottleneck, HipBatchNorm2d, HipConv2d
    torch.manual_seed(4)
    if kind == "identity":
        cin, planes, stride = 256, 64, 1
        blk = Bottleneck(cin, planes)
    else:
        stride = 1 if kind == "downsample_s1" else 2
        cin, planes = (64, 64) if stride == 1 else (256, 128)
        blk = Bottleneck(cin, planes, stride, torch.nn.Sequential(HipConv2d(cin, planes * 4, 1, stride=stride, bias=False), HipBatchNorm2d(planes * 4)))
    blk = blk.cuda().train()
    for m in blk.modules():
        if isinstance(m, HipBatchNorm2d):
            m.weight.data.uniform_(0.5, 1.5)
            m.bias.data.normal_(0, 0.2)
    ops.set_dtype(blk, "bf16")
    x0 = _bf16(torch.randn(8, cin, 28, 28, device="cuda").relu())
    gy = None
    res = {}
    for name in ("fp32_storage", "blk"):
        blk.zero_grad(set_to_none=True)
        x = x0.clone().requires_grad_()
        if name == "blk":
            y = blk_trunk.to_nchw(blk_trunk.layer_forward(torch.nn.Sequential(blk), blk_trunk.to_blk(x)))
        else:
            y = blk(x)
        if gy is None:
            gy = _bf16(torch.randn_like(y))
        y.backward(gy)
        res[name] = (y.detach().clone(), x.grad.clone(), {k: p.grad.clone() for k, p in blk.named_parameters()})
    # float64
    xd = x0.double().requires_grad_()
    P = {k: (_bf16(p.detach()) if p.dim() == 4 else p.detach()).double().requires_grad_() for k, p in blk.named_parameters()}

    def bn(t, pre):
        return F.batch_norm(t, None, None, P[pre + ".weight"], P[pre + ".bias"], True, 0.0, 1e-5)
    o = F.relu(bn(F.conv2d(xd, P["conv1.weight"]), "bn1"))
    o = F.relu(bn(F.conv2d(o, P["conv2.weight"], stride=stride, padding=1), "bn2"))
    o = bn(F.conv2d(o, P["conv3.weight"]), "bn3")
    idn = xd if kind == "identity" else bn(F.conv2d(xd, P["downsample.0.weight"], stride=stride), "downsample.1")
    o = F.relu(o + idn)
    o.backward(gy.double())
    assert _rel_l2(res["blk"][0], o.detach()) < 1e-2
    worst = []
    for what, ref in [("dx", xd.grad)] + [("grad." + k, P[k].grad) for k in sorted(P)]:
        e_blk = _rel_l2(res["blk"][1] if what == "dx" else res["blk"][2][what[5:]], ref)
        e_f32 = _rel_l2(res["fp32_storage"][1] if what == "dx" else res["fp32_storage"][2][what[5:]], ref)
        worst.append((e_blk / (1.5 * e_f32 + 1e-2), what, e_blk, e_f32))
    worst.sort()
    assert len(worst) >= 10 and worst[-1][0] <= 1.0, "further from float64 than 1.5 x the fp32-storage path + 1 %%: %s" % worst[-4:]


def test_blk_trunk_eval_forward_against_fp32_storage():
    """The whole trunk (vision.py:11-21) in eval mode (running statistics: well conditioned at any depth), forward only: layers 1-4 on
    blk activations against the same bf16 kernels on fp32 activations -- the five feature maps within 3 % relative L2 (2^-9 per
    element and layer, ~100 layers)."""
    from rsis_amd import blk_trunk, ops
    from rsis_amd.modules.vision import HipBatchNorm2d, ResNet101
    torch.manual_seed(0)
    net = ResNet101().cuda()
    for m in net.modules():      # running statistics of a net whose activations stay O(1): var = fan-in gain, damped residual branches
        if isinstance(m, HipBatchNorm2d):
            m.running_var.fill_(0.4)
            m.weight.data.fill_(0.5)
    net.eval()
    ops.set_dtype(net, "bf16")
    x = torch.randn(2, 3, 160, 96, device="cuda")
    outs = []
    was = blk_trunk.ENABLED[0]
    try:
        for on in (False, True):
            blk_trunk.ENABLED[0] = on
            with torch.no_grad():
                outs.append([o.clone() for o in net(x)])
    finally:
        blk_trunk.ENABLED[0] = was
    for i, (a, b) in enumerate(zip(outs[1], outs[0])):
        assert float(b.abs().max()) > 1e-3 and torch.isfinite(b).all()
        assert _rel_l2(a, b) < 3e-2, "x%d: rel L2 %.3g" % (5 - i, _rel_l2(a, b))


@pytest.mark.parametrize("shape", [(2, 64, 256, 14, 14, 1), (2, 64, 64, 17, 9, 3), (3, 128, 40, 7, 7, 3), (2, 256, 128, 28, 28, 1)],
                         ids=lambda s: "x".join(str(v) for v in s))
@pytest.mark.parametrize("res,relu", [(False, True), (True, True), (False, False)])
def test_blk_conv_with_the_eval_batchnorm_in_its_epilogue(shape, res, relu):
    """rsis_blk_conv2d_bn_eval.  single_rounding = 0: BIT FOR BIT what rsis_blk_conv2d followed by the eval-mode blk BatchNorm launch
    writes (the fold is then a scheduling change only).  single_rounding = 1: relu?(conv * sc + sh (+ addend)) in fp32 with one rounding --
    against float64 on the same bf16-valued operands, half a bf16 ulp + 1e-5 of the output's scale."""
    from rsis_amd import ops
    B, Cin, Cout, H, W, ks = shape
    torch.manual_seed(sum(shape) + int(res) + 2 * int(relu))
    x = _bf16(torch.randn(B, Cin, H, W, device="cuda"))
    w = torch.randn(Cout, Cin, ks, ks, device="cuda") / (ks * Cin ** 0.5)
    pack = ops.PackedConv(ks, [Cin], stride=1, pad=ks // 2, dtype=ops.DTYPE_BF16)
    wp = pack.fwd(w)
    gamma, beta = torch.rand(Cout, device="cuda") + 0.5, torch.randn(Cout, device="cuda")
    mean, var, eps = torch.randn(Cout, device="cuda") * 0.3, torch.rand(Cout, device="cuda") + 0.2, 1e-5
    r = _bf16(torch.randn(B, Cout, H, W, device="cuda")) if res else None
    rb = to_blk(r) if res else None
    xb = to_blk(x)
    # the separate launches
    a = ops.blk_conv2d(xb, wp, Cout, ks)
    want, _sm, _sr = ops.blk_bn_fwd(a, rb, gamma, beta, mean.clone(), var.clone(), eps, 0.1, relu, False)
    got = ops.blk_conv2d(xb, wp, Cout, ks, addend=rb, bn=(gamma, beta, mean, var, eps), relu=relu)
    assert torch.equal(got.view(torch.int16), want.view(torch.int16)), "folded BatchNorm differs from conv + BatchNorm launch"
    # one rounding
    y = ops.blk_conv2d(xb, wp, Cout, ks, addend=rb, bn=(gamma, beta, mean, var, eps), relu=relu, single_rounding=True)
    sc = gamma.double() / torch.sqrt(var.double() + eps)
    ref = torch.nn.functional.conv2d(x.double(), _bf16(w).double(), padding=ks // 2) * sc.view(1, -1, 1, 1) + (beta.double() - mean.double() * sc).view(1, -1, 1, 1)
    if res:
        ref = ref + r.double()
    if relu:
        ref = ref.clamp_min(0)
    assert_close("y", from_blk(y), ref, 2e-5 * float(ref.abs().max()), HALF_ULP)


def test_blk_trunk_eval_with_folded_batchnorm():
    """inference: the blk trunk with every eval-mode BatchNorm (+ residual) (+ ReLU) in its conv's epilogue (blk_trunk._block_forward_eval).
    Mode 1 (default) reproduces the conv -> BatchNorm launches bit for bit; mode 2 (one rounding per layer) stays within 1 % relative L2 of
    them and is no farther from the fp32-storage kernels."""
    from rsis_amd import blk_trunk, ops
    from rsis_amd.modules.vision import HipBatchNorm2d, ResNet101
    torch.manual_seed(0)
    net = ResNet101().cuda()
    for m in net.modules():
        if isinstance(m, HipBatchNorm2d):
            m.running_var.fill_(0.4)
            m.running_mean.normal_(0, 0.05)
            m.weight.data.fill_(0.5)
            m.bias.data.normal_(0, 0.05)
    net.eval()
    ops.set_dtype(net, "bf16")
    x = torch.randn(2, 3, 160, 96, device="cuda")
    outs = {}
    was, wasf = blk_trunk.ENABLED[0], blk_trunk.EVAL_FOLD[0]
    try:
        for name, on, fold in (("fp32 storage", False, 0), ("unfolded", True, 0), ("folded", True, 1), ("single rounding", True, 2)):
            blk_trunk.ENABLED[0], blk_trunk.EVAL_FOLD[0] = on, fold
            with torch.no_grad():
                outs[name] = [o.clone() for o in net(x)]
    finally:
        blk_trunk.ENABLED[0], blk_trunk.EVAL_FOLD[0] = was, wasf
    for i in range(5):
        a, b, c, d = outs["folded"][i], outs["unfolded"][i], outs["fp32 storage"][i], outs["single rounding"][i]
        assert torch.isfinite(a).all() and float(c.abs().max()) > 1e-3
        assert torch.equal(a, b), "x%d: the folded trunk differs from the unfolded one" % (5 - i)
        assert _rel_l2(d, b) < 1e-2, "x%d: single rounding vs unfolded rel L2 %.3g" % (5 - i, _rel_l2(d, b))
        assert _rel_l2(d, c) < _rel_l2(b, c) + 1e-2, "x%d: %.3g vs %.3g" % (5 - i, _rel_l2(d, c), _rel_l2(b, c))
