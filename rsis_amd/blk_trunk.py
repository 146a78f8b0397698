"""Layers 1-4 of the ResNet-101 trunk on channel-blocked bf16 activations -- the storage half of `-dtype bf16` (BASELINE.json
configs[2..4]; the layers are torchvision's bottlenecks behind reference src/modules/vision.py:12-19).

Under `-dtype bf16` the convs already multiply bf16 operands, but every activation, BatchNorm and gradient tensor of the trunk was
fp32 NCHW: the convs spent their time converting while staging and BatchNorm moved 4-byte elements.  Here a logical [B][C][H][W]
tensor lives as bf16 [B][C/8][H][W][8] ("blk", csrc/conv_blk.hip): conv forward / data gradient are LDS-DMA rings over 16-byte
cells, BatchNorm (+ residual + ReLU) works on cells, the weight gradients transpose 8 x 8 blocks in registers while staging.
Parameters, their gradients, the BatchNorm statistics and the optimizer stay fp32; so do the stem (3 input channels), the max-pool
and everything outside the trunk -- the five feature maps leave as fp32 NCHW.

A layer (nn.Sequential of Bottlenecks) is ONE autograd node with a hand-written backward: no per-op autograd bookkeeping, the
gradient of a block's input is the first conv's data gradient with the identity / downsample branch's gradient as its addend, and
parameter gradients are accumulated by the kernels -- straight into the flat gradient buffers when ops.DIRECT_GRAD is on (weight
gradients parked for the grouped launch of ops.flush_wgrads, as the fp32-storage path does).

Strided layers (the first block of layers 2-4): a stride-s conv is the stride-1 conv followed by a sub-sampling -- with the bf16
MFMA the 4x extra work of three convs costs less than a second kernel family -- and its gradients use the zero-inserted dy.
"""
import os

import torch

from . import ops
from ._lib import lib

# RSIS_BF16_STORAGE=0: keep fp32 NCHW activations under -dtype bf16 (the round-2 path)
ENABLED = [os.environ.get("RSIS_BF16_STORAGE", "1") != "0"]


class _ToBlkFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        return ops.blk_from_nchw(x)

    @staticmethod
    def backward(ctx, dy):
        return ops.blk_to_nchw(dy.contiguous())


class _ToNchwFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        return ops.blk_to_nchw(x)

    @staticmethod
    def backward(ctx, dy):
        return ops.blk_from_nchw(dy)


def to_blk(x):
    return _ToBlkFn.apply(x)


def to_nchw(x):
    return _ToNchwFn.apply(x)


def _pack(conv):
    """the stride-1 bf16 pack of a conv's weight (forward and data-gradient copies), whatever the conv's own stride"""
    pk = getattr(conv, "_blk_pack", None)
    if pk is None:
        ks = conv.kernel_size
        pk = conv._blk_pack = ops.PackedConv(ks, [conv.in_channels], stride=1, pad=ks // 2, dtype=ops.DTYPE_BF16)
    return pk


def _conv(conv, x):
    return ops.blk_conv2d(x, _pack(conv).fwd(conv.weight), conv.out_channels, conv.kernel_size)


def _dgrad(conv, dy, addend=None):
    return ops.blk_conv2d(dy, _pack(conv).dgrad(conv.weight), conv.in_channels, conv.kernel_size, addend=addend)


def _bn(bn, x, res, relu):
    if bn.training:
        bn._nbt_pending += 1
    y, sm, sr = ops.blk_bn_fwd(x, res, bn.weight.detach(), bn.bias.detach(), bn.running_mean, bn.running_var, bn.eps, bn.momentum, relu,
                               bn.training)
    return y, (sm, sr)


def _acc(p, g):
    """parameter gradient outside DIRECT_GRAD: what autograd's accumulation would do"""
    if p.grad is None:
        p.grad = g
    else:
        p.grad += g


def _bn_bwd(bn, dy, x, y, stats, relu, want_dres):
    tg, tb = ops._direct_target(bn.weight), ops._direct_target(bn.bias)
    direct = tg is not None and tb is not None
    need = bn.weight.requires_grad
    dx, dres, dg, db = ops.blk_bn_bwd(dy, x, y, bn.weight.detach(), bn.bias.detach(), stats[0], stats[1], relu, want_dres,
                                      dgamma=tg if direct else None, dbeta=tb if direct else None, accumulate=direct)
    if need and not direct:
        _acc(bn.weight, dg)
        _acc(bn.bias, db)
    return dx, dres


def _wgrad(conv, dy, x):
    """dW of a stride-1 'same' conv from blk dy / x: into the flat gradient buffer (parked for the grouped launch) or a new tensor"""
    w = conv.weight
    if not w.requires_grad:
        return
    B, _cb, H, W, _ = dy.shape
    tgt = ops._direct_target(w)
    dW = tgt if tgt is not None else torch.zeros_like(w)
    ks = conv.kernel_size
    ops.wgrad_launch(lib(), dy, x, dW, B, conv.in_channels, H, W, conv.out_channels, H, W, ks, 1, ks // 2, conv.in_channels, 0, 0,
                     ops.DTYPE_BF16_BLK, "rsis_conv2d_wgrad(blk)", tgt is not None)
    if tgt is None:
        _acc(w, dW)


# inference: the eval-mode BatchNorm (+ residual) (+ ReLU) behind every trunk conv in the conv's epilogue.  "1" (default): in the
# arithmetic of the separate launches (the product rounded to bf16, then the BatchNorm: the same bits, minus 103 launches per forward);
# "2": the product kept in fp32 (one rounding per layer: closer to the fp32 features, but not the bits the parity tests were pinned on);
# "0": separate BatchNorm launches
EVAL_FOLD = [int(os.environ.get("RSIS_BLK_EVAL_FOLD", "1"))]


def _conv_bn(conv, bn, x, res, relu):
    return ops.blk_conv2d(x, _pack(conv).fwd(conv.weight), conv.out_channels, conv.kernel_size, addend=res,
                          bn=(bn.weight.detach(), bn.bias.detach(), bn.running_mean, bn.running_var, bn.eps), relu=relu,
                          single_rounding=EVAL_FOLD[0] == 2)


def _block_forward_eval(blk, x):
    """the bottleneck for inference: three (four) convs, every eval-mode BatchNorm, the residual add and the ReLUs in their epilogues --
    no BatchNorm launch.  (A stride-s conv is the stride-1 conv followed by the sub-sampling; the pointwise epilogue commutes with it.)"""
    s = blk.stride
    y1 = _conv_bn(blk.conv1, blk.bn1, x, None, True)
    y2 = _conv_bn(blk.conv2, blk.bn2, y1, None, True)
    if s != 1:
        y2 = ops.blk_subsample(y2, s)
    if blk.downsample is None:
        res = x
    else:
        xs = x if s == 1 else ops.blk_subsample(x, s)
        res = _conv_bn(blk.downsample[0], blk.downsample[1], xs, None, False)
    return _conv_bn(blk.conv3, blk.bn3, y2, res, True)


def _block_forward(blk, x, keep):
    if not keep and EVAL_FOLD[0] and not blk.bn1.training:
        return _block_forward_eval(blk, x), None
    s = blk.stride
    a1 = _conv(blk.conv1, x)
    y1, m1 = _bn(blk.bn1, a1, None, True)
    a2 = _conv(blk.conv2, y1)
    if s != 1:
        a2 = ops.blk_subsample(a2, s)
    y2, m2 = _bn(blk.bn2, a2, None, True)
    a3 = _conv(blk.conv3, y2)
    xs = ad = md = None
    if blk.downsample is None:
        res = x
    else:
        xs = x if s == 1 else ops.blk_subsample(x, s)
        ad = _conv(blk.downsample[0], xs)
        res, md = _bn(blk.downsample[1], ad, None, False)
    out, m3 = _bn(blk.bn3, a3, res, True)
    if keep:
        return out, (x, a1, y1, m1, a2, y2, m2, a3, m3, out, xs, ad, md)
    return out, None


def _block_backward(blk, dout, saved):
    x, a1, y1, m1, a2, y2, m2, a3, m3, out, xs, ad, md = saved
    s = blk.stride
    da3, dres = _bn_bwd(blk.bn3, dout, a3, out, m3, True, True)
    _wgrad(blk.conv3, da3, y2)
    dy2 = _dgrad(blk.conv3, da3)
    da2, _ = _bn_bwd(blk.bn2, dy2, a2, None, m2, True, False)
    if s != 1:
        da2 = ops.blk_upscatter(da2, y1.shape[2], y1.shape[3], s)
    _wgrad(blk.conv2, da2, y1)
    dy1 = _dgrad(blk.conv2, da2)
    da1, _ = _bn_bwd(blk.bn1, dy1, a1, None, m1, True, False)
    _wgrad(blk.conv1, da1, x)
    if blk.downsample is None:
        return _dgrad(blk.conv1, da1, addend=dres)
    dad, _ = _bn_bwd(blk.downsample[1], dres, ad, None, md, False, False)
    _wgrad(blk.downsample[0], dad, xs)
    dxs = _dgrad(blk.downsample[0], dad)
    if s != 1:
        dxs = ops.blk_upscatter(dxs, x.shape[2], x.shape[3], s)
    return _dgrad(blk.conv1, da1, addend=dxs)


class _LayerFn(torch.autograd.Function):
    """one `layerN` (nn.Sequential of Bottlenecks): blk in, blk out; the parameters are read from the modules"""

    @staticmethod
    def forward(ctx, layer, x, _anchor):
        keep = any(ctx.needs_input_grad)          # (grad mode is off inside forward: this is "somebody will call backward")
        tape = []
        for blk in layer:
            x, saved = _block_forward(blk, x, keep)
            tape.append(saved)
        ctx.layer, ctx.tape = layer, tape
        return x

    @staticmethod
    def backward(ctx, dout):
        layer, tape = ctx.layer, ctx.tape
        ctx.tape = None
        if not layer[0].bn1.training:
            raise RuntimeError("blk trunk: the backward exists for train-mode BatchNorm only (eval-mode trunks keep fp32 activations)")
        dx = dout.contiguous()
        for blk, saved in zip(reversed(list(layer)), reversed(tape)):
            dx = _block_backward(blk, dx, saved)
        return None, dx, None


class _SkipsBlkFn(torch.autograd.Function):
    """The skip branches of reference model.py:59-63 -- x_k_skip = bn_k(sk_k(x_k)), a 3x3 conv with bias and a BatchNorm without ReLU --
    for the trunk features that are already blk tensors, blk in / blk out: the convs of all given levels in ONE grouped launch
    (rsis_blk_conv3x3_batch), blk BatchNorm, and in the backward one grouped data-gradient launch, the weight gradients parked for the
    grouped flush.  No layout converters between the trunk, the skip branches and the decoder."""

    @staticmethod
    def forward(ctx, mods, _anchor, *xs):
        # mods: [(HipConv2d, HipBatchNorm2d)] per level, xs: the blk inputs; _anchor: see skips_forward
        keep = any(ctx.needs_input_grad)
        zs, jobs = [], []
        for (conv, _unused), x in zip(mods, xs):
            B, _cb, H, W, _ = x.shape
            z = torch.empty((B, conv.out_channels // 8, H, W, 8), dtype=torch.bfloat16, device=x.device)
            jobs.append(ops.blk_conv_job([x], conv._pack.fwd(conv.weight), conv.out_channels, bias=conv.bias.detach(), dsts=[z]))
            zs.append(z)
        ops.blk_conv3x3_batch(jobs)
        ys, stats = [], []
        for (_unused, bn), z in zip(mods, zs):
            y, st = _bn(bn, z, None, False)
            ys.append(y)
            stats.append(st)
        if keep:
            ctx.mods, ctx.xs, ctx.zs, ctx.stats = mods, xs, zs, stats
        return tuple(ys)

    @staticmethod
    def backward(ctx, *dys):
        mods, xs, zs, stats = ctx.mods, ctx.xs, ctx.zs, ctx.stats
        ctx.xs = ctx.zs = ctx.stats = None
        L = lib()
        dzs, jobs, dxs = [], [], []
        for (conv, bn), x, z, st, dy in zip(mods, xs, zs, stats, dys):
            if not bn.training:
                raise RuntimeError("blk skip branches: the backward exists for train-mode BatchNorm only")
            dz, _ = _bn_bwd(bn, dy.contiguous(), z, None, st, False, False)
            dzs.append(dz)
            _wgrad(conv, dz, x)
            if conv.bias is not None and conv.bias.requires_grad:
                tb = ops._direct_target(conv.bias)
                db = tb if tb is not None else torch.zeros_like(conv.bias)
                B, _cb, H, W, _ = dz.shape
                from ._lib import check, ptr, stream
                check(L.rsis_blk_bias_grad(ptr(dz), ptr(db), B, conv.out_channels, H * W, 0, stream()), "rsis_blk_bias_grad")
                if tb is None:
                    _acc(conv.bias, db)
        for i, ((conv, _unused), x, dz) in enumerate(zip(mods, xs, dzs)):
            if ctx.needs_input_grad[2 + i]:
                dx = torch.empty_like(x)
                jobs.append(ops.blk_conv_job([dz], conv._pack.dgrad(conv.weight), conv.in_channels, cpack=conv._pack.cin, dsts=[dx]))
                dxs.append(dx)
            else:
                dxs.append(None)
        if jobs:
            ops.blk_conv3x3_batch(jobs)
        return (None, None) + tuple(dxs)


def skips_forward(mods, xs):
    """[(sk_k, bn_k)] applied to the blk features xs -> the blk skip features (one autograd node).  As in layer_forward, an anchor
    (any skip parameter that requires grad) keeps the node in the graph when none of the features carries a gradient -- a frozen
    trunk in front of trainable skip branches (the reference's update_encoder=False phase still trains nothing of the encoder, but
    a caller freezing only `base` must not silently lose sk / bn gradients)."""
    anchor = None
    if torch.is_grad_enabled() and not any(x.requires_grad for x in xs):
        for conv, bn in mods:
            for p in list(conv.parameters()) + list(bn.parameters()):
                if p.requires_grad:
                    anchor = p
                    break
            if anchor is not None:
                break
    return list(_SkipsBlkFn.apply(mods, anchor, *xs))


def layer_forward(layer, x):
    """x (blk) through one layer.  The anchor keeps the node in the graph when x itself carries no gradient (a frozen stem in front
    of a trainable layer): any parameter of the layer that requires grad."""
    anchor = None
    if torch.is_grad_enabled() and not x.requires_grad:
        for p in layer.parameters():
            if p.requires_grad:
                anchor = p
                break
    return _LayerFn.apply(layer, x, anchor)


def usable(trunk, x):
    """blk storage applies: enabled, bf16 kernels selected, and no backward through eval-mode BatchNorm (no blk kernel for that)"""
    return bool(ENABLED[0] and getattr(trunk, "_blk", False) and x.is_cuda and (trunk.training or not torch.is_grad_enabled()))
