"""CPU test of the HOST logic of the blocked bf16 trunk (rsis_amd/blk_trunk.py): the hand-written backward of a layer of
torchvision bottlenecks (reference src/modules/vision.py:12-19) -- which gradient joins which, the residual / downsample routing,
the strided layers as "stride-1 conv + sub-sampling" with zero-inserted gradients, where parameter gradients are accumulated.

The device ops are replaced by float64 torch emulations of their CONTRACTS (include/rsis_hip.h: rsis_blk_conv2d, rsis_blk_bn_fwd /
_bwd, rsis_blk_subsample2d / _upscatter2d, the blk weight gradient), on plain NCHW tensors standing in for blk tensors; the kernels
themselves are held to those contracts by tests/test_gpu_blk.py.  With exact arithmetic on both sides the node must reproduce
autograd of the same bottlenecks to float64 rounding."""
import pytest
import torch
import torch.nn.functional as F

from rsis_amd import blk_trunk, ops
from rsis_amd.modules.vision import Bottleneck, HipBatchNorm2d, HipConv2d


def _bn_fwd(x, res, gamma, beta, run_mean, run_var, eps, momentum, relu, train, stats=None):
    assert train
    dims = (0, 2, 3)
    mean, var = x.mean(dims), x.var(dims, unbiased=False)
    rstd = 1.0 / torch.sqrt(var + eps)
    y = (x - mean[None, :, None, None]) * (rstd * gamma.double())[None, :, None, None] + beta.double()[None, :, None, None]
    if res is not None:
        y = y + res
    if relu:
        y = y.clamp_min(0)
    return y, mean, rstd


def _bn_bwd(dy, x, y, gamma, beta, save_mean, save_rstd, relu, want_dres, dgamma=None, dbeta=None, accumulate=False):
    xh = (x - save_mean[None, :, None, None]) * save_rstd[None, :, None, None]
    g = dy
    if relu:
        on = (y > 0) if y is not None else (xh * gamma.double()[None, :, None, None] + beta.double()[None, :, None, None] > 0)
        g = dy * on
    dims = (0, 2, 3)
    db, dg = g.sum(dims), (g * xh).sum(dims)
    n = x.numel() / x.shape[1]
    dx = (gamma.double() * save_rstd)[None, :, None, None] * (g - (db / n)[None, :, None, None] - xh * (dg / n)[None, :, None, None])
    return dx, (g if want_dres else None), dg.to(gamma.dtype), db.to(gamma.dtype)


@pytest.fixture
def emulated(monkeypatch):
    def conv(c, x, stats=False):
        return F.conv2d(x, c.weight.detach().double(), padding=c.kernel_size // 2)

    def dgrad(c, dy, addend=None):
        dx = F.conv_transpose2d(dy, c.weight.detach().double(), padding=c.kernel_size // 2)
        return dx if addend is None else dx + addend

    def wgrad(c, dy, x):
        with torch.enable_grad():         # (the node's backward runs with grad mode off)
            w = c.weight.detach().double().requires_grad_()
            F.conv2d(x.detach(), w, padding=c.kernel_size // 2).backward(dy.detach())
        blk_trunk._acc(c.weight, w.grad.to(c.weight.dtype))

    monkeypatch.setattr(blk_trunk, "_conv", conv)
    monkeypatch.setattr(blk_trunk, "_dgrad", dgrad)
    monkeypatch.setattr(blk_trunk, "_wgrad", wgrad)
    monkeypatch.setattr(ops, "blk_bn_fwd", _bn_fwd)
    monkeypatch.setattr(ops, "blk_bn_bwd", _bn_bwd)
    monkeypatch.setattr(ops, "blk_subsample", lambda x, s: x[:, :, ::s, ::s].contiguous())

    def upscatter(dy, H, W, s):
        dx = dy.new_zeros(dy.shape[0], dy.shape[1], H, W)
        dx[:, :, ::s, ::s] = dy
        return dx
    monkeypatch.setattr(ops, "blk_upscatter", upscatter)


def _reference(layer, x):
    """the same bottlenecks by plain torch ops under autograd (float64)"""
    for blk in layer:
        def bn(m, t):
            return F.batch_norm(t, None, None, m.weight.double(), m.bias.double(), True, 0.0, m.eps)
        o = F.relu(bn(blk.bn1, F.conv2d(x, blk.conv1.weight.double())))
        o = F.relu(bn(blk.bn2, F.conv2d(o, blk.conv2.weight.double(), stride=blk.stride, padding=1)))
        o = bn(blk.bn3, F.conv2d(o, blk.conv3.weight.double()))
        idn = x if blk.downsample is None else bn(blk.downsample[1], F.conv2d(x, blk.downsample[0].weight.double(), stride=blk.stride))
        x = F.relu(o + idn)
    return x


@pytest.mark.parametrize("stride", [1, 2])
def test_layer_node_backward_equals_autograd(emulated, stride):
    torch.manual_seed(stride)
    inpl, planes = 16, 8
    down = torch.nn.Sequential(HipConv2d(inpl, planes * 4, 1, stride=stride, bias=False), HipBatchNorm2d(planes * 4))
    layer = torch.nn.Sequential(Bottleneck(inpl, planes, stride, down), Bottleneck(planes * 4, planes), Bottleneck(planes * 4, planes)).train()
    for m in layer.modules():
        if isinstance(m, HipBatchNorm2d):
            m.weight.data.uniform_(0.5, 1.5)
            m.bias.data.normal_(0, 0.3)
    x0 = torch.randn(3, inpl, 9, 7, dtype=torch.float64)
    gy = torch.randn(3, planes * 4, (9 - 1) // stride + 1, (7 - 1) // stride + 1, dtype=torch.float64)
    # reference
    xr = x0.clone().requires_grad_()
    _reference(layer, xr).backward(gy)
    want = {k: p.grad.clone() for k, p in layer.named_parameters()}
    want_dx = xr.grad.clone()
    nbt = [m._nbt_pending for m in layer.modules() if isinstance(m, HipBatchNorm2d)]
    layer.zero_grad(set_to_none=True)
    # the node
    xn = x0.clone().requires_grad_()
    out = blk_trunk.layer_forward(layer, xn)
    assert torch.allclose(out, _reference(layer, x0), rtol=1e-10, atol=1e-10)
    out.backward(gy)
    assert torch.allclose(xn.grad, want_dx, rtol=1e-8, atol=1e-10)
    for k, p in layer.named_parameters():
        assert p.grad is not None, k
        assert torch.allclose(p.grad.double(), want[k].double(), rtol=1e-4, atol=1e-6), (k, float((p.grad.double() - want[k].double()).abs().max()))
    # train-mode bookkeeping: every BatchNorm counted one more batch (num_batches_tracked, flushed by state_dict())
    assert [m._nbt_pending for m in layer.modules() if isinstance(m, HipBatchNorm2d)] == [n + 1 for n in nbt]


def test_layer_node_without_input_gradient_still_trains_its_parameters(emulated):
    """a frozen stem in front of a trainable layer: x carries no gradient, the node stays in the graph through its anchor parameter"""
    torch.manual_seed(5)
    layer = torch.nn.Sequential(Bottleneck(32, 8)).train()
    x = torch.randn(2, 32, 5, 5, dtype=torch.float64)
    out = blk_trunk.layer_forward(layer, x)
    assert out.requires_grad
    out.sum().backward()
    assert all(p.grad is not None for p in layer.parameters())
    with torch.no_grad():
        assert not blk_trunk.layer_forward(layer, x).requires_grad
