"""ResNet-101 feature pyramid -- drop-in for reference src/modules/vision.py:6-21 on MI355X.

The reference subclasses torchvision's ResNet(Bottleneck, [3,4,23,3]) (un-vendored third party); this module owns the
same module tree / state_dict keys (conv1, bn1, layer1..4.{j}.{conv1,bn1,conv2,bn2,conv3,bn3,downsample.{0,1}}, fc)
and runs every conv / BN / ReLU / residual add / max-pool in librsis_hip.so.  Stride sits on the 3x3 conv
(torchvision), max-pool is 3x3/2 pad 1.  `avgpool`/`fc` are never called by the reference (vision.py:11-21); `fc` is
kept only for checkpoint-key compatibility.
"""
import math

import torch
from torch import nn

from .. import blk_trunk, ops


class HipConv2d(nn.Module):
    """nn.Conv2d parameters (reference layout) + the gfx950 implicit-GEMM forward/backward."""

    def __init__(self, cin, cout, kernel_size, stride=1, padding=0, bias=True):
        super().__init__()
        self.in_channels, self.out_channels = int(cin), int(cout)
        self.kernel_size, self.stride, self.padding = int(kernel_size), int(stride), int(padding)
        self.weight = nn.Parameter(torch.empty(self.out_channels, self.in_channels, self.kernel_size, self.kernel_size))
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        if bias:
            self.bias = nn.Parameter(torch.empty(self.out_channels))
            bound = 1.0 / math.sqrt(self.in_channels * self.kernel_size ** 2)
            nn.init.uniform_(self.bias, -bound, bound)
        else:
            self.register_parameter("bias", None)
        self._pack = None
        self._set_rsis_dtype(ops.DTYPE_F32)

    def _set_rsis_dtype(self, d):
        """f32 or bf16 MFMA kernels for this conv (ops.set_dtype); the parameters stay fp32.  Under fp32 the 3x3 convs that
        ops.conv_dtype selects (RSIS_WINOGRAD) get the Winograd copy of their weight (same arithmetic type, fewer matrix flops)."""
        import torch as _t
        if _t.cuda.is_available():       # (the rule asks the library which geometries its kernel covers; CPU-side module construction -- tests of
            d = ops.conv_dtype(d, self.kernel_size, self.stride, self.padding, self.in_channels, self.out_channels)   # the parameter layout -- has none)
        if self._pack is None or self._pack.dtype != d:
            self._pack = ops.PackedConv(self.kernel_size, [self.in_channels], stride=self.stride, pad=self.padding, dtype=d)

    def forward(self, x, grad_slot=None, park_slot=None):
        return ops.conv2d([x], self.weight, self.bias, self.stride, self.padding, self._pack, grad_slot=grad_slot,
                          park_slot=park_slot)


class HipBatchNorm2d(nn.Module):
    """nn.BatchNorm2d state (weight, bias, running_mean, running_var, num_batches_tracked) + fused HIP kernels.
    forward(x, res=None, relu=False) computes act(bn(x) + res)."""

    def __init__(self, num_features, eps=1e-5, momentum=0.1):
        super().__init__()
        self.num_features, self.eps, self.momentum = int(num_features), eps, momentum
        self.weight = nn.Parameter(torch.ones(self.num_features))
        self.bias = nn.Parameter(torch.zeros(self.num_features))
        self.register_buffer("running_mean", torch.zeros(self.num_features))
        self.register_buffer("running_var", torch.ones(self.num_features))
        self.register_buffer("num_batches_tracked", torch.tensor(0, dtype=torch.long))
        self._nbt_pending = 0     # increments of num_batches_tracked not yet written to the buffer (flushed by state_dict())
        self._arena = None        # (fwd, bwd) slices of a zeroed float64 stats arena, set per iteration by the owner
        self._register_state_dict_hook(HipBatchNorm2d._flush_nbt_hook)

    @staticmethod
    def _flush_nbt_hook(module, state_dict, prefix, local_metadata):
        if module._nbt_pending:
            module.num_batches_tracked += module._nbt_pending
            module._nbt_pending = 0
            state_dict[prefix + "num_batches_tracked"] = module.num_batches_tracked

    def _load_from_state_dict(self, state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys, error_msgs):
        # torch-0.2 era checkpoints (the reference's) have no num_batches_tracked
        key = prefix + "num_batches_tracked"
        if key not in state_dict:
            state_dict[key] = torch.tensor(0, dtype=torch.long)
        self._nbt_pending = 0
        super()._load_from_state_dict(state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys, error_msgs)

    def forward(self, x, res=None, relu=False, res_slot=None):
        if self.training:
            self._nbt_pending += 1      # host-side counter: no device launch per layer per step
        arena, self._arena = self._arena, None
        return ops.batchnorm(x, self.weight, self.bias, self.running_mean, self.running_var, self.training, relu=relu, res=res,
                             eps=self.eps, momentum=self.momentum, arena=arena if self.training else None, res_slot=res_slot)


def conv_bn(conv, bn, x, relu=False, res=None):
    """act(bn(conv(x)) + res).  A call that records no autograd graph with the BatchNorm in eval mode (test() / eval.py) runs as ONE
    launch -- the BatchNorm, the residual add and the ReLU in the conv's epilogue (ops.conv2d_bn_eval) -- where the library has that
    epilogue; every other call is the two modules."""
    if not bn.training and not torch.is_grad_enabled() and x.is_cuda:
        y = ops.conv2d_bn_eval(x, conv.weight, conv.bias, conv.stride, conv.padding, conv._pack, bn.weight, bn.bias, bn.running_mean,
                               bn.running_var, bn.eps, relu=relu, res=res)
        if y is not None:
            return y
    return bn(conv(x), res=res, relu=relu)


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = HipConv2d(inplanes, planes, 1, bias=False)
        self.bn1 = HipBatchNorm2d(planes)
        self.conv2 = HipConv2d(planes, planes, 3, stride=stride, padding=1, bias=False)
        self.bn2 = HipBatchNorm2d(planes)
        self.conv3 = HipConv2d(planes, planes * 4, 1, bias=False)
        self.bn3 = HipBatchNorm2d(planes * 4)
        self.downsample = downsample
        self.stride = stride
        self._slot = ops.GradSlot()       # gradient of x handed to conv1's data-gradient kernel
        self._slot_in = ops.GradSlot()    # (downsample blocks) gradient of x from outside the trunk, see ResNet101.forward

    def forward(self, x):
        # In training x has two consumers inside the block (conv1 and the identity branch / downsample conv); instead of
        # letting autograd add their gradients, the second one is handed to conv1's data-gradient kernel (ops.GradSlot).
        if not self.training and not torch.is_grad_enabled():          # inference: each conv + BatchNorm (+ residual) (+ ReLU) one launch
            out = conv_bn(self.conv1, self.bn1, x, relu=True)
            out = conv_bn(self.conv2, self.bn2, out, relu=True)
            residual = x if self.downsample is None else conv_bn(self.downsample[0], self.downsample[1], x)
            return conv_bn(self.conv3, self.bn3, out, relu=True, res=residual)
        hand = self.training and torch.is_grad_enabled() and x.requires_grad
        slot = self._slot if hand else None
        out = self.bn1(self.conv1(x, grad_slot=slot), relu=True)
        out = self.bn2(self.conv2(out), relu=True)
        if self.downsample is None:
            return self.bn3(self.conv3(out), res=x, relu=True, res_slot=slot)
        residual = self.downsample[1](self.downsample[0](x, grad_slot=self._slot_in if hand else None, park_slot=slot))
        return self.bn3(self.conv3(out), res=residual, relu=True)


class ResNet101(nn.Module):
    """Returns intermediate features (x5,x4,x3,x2,x1) -- reference vision.py:11-21."""

    def __init__(self):
        super().__init__()
        self.inplanes = 64
        self.conv1 = HipConv2d(3, 64, 7, stride=2, padding=3, bias=False)
        self.bn1 = HipBatchNorm2d(64)
        self.layer1 = self._make_layer(64, 3)
        self.layer2 = self._make_layer(128, 4, stride=2)
        self.layer3 = self._make_layer(256, 23, stride=2)
        self.layer4 = self._make_layer(512, 3, stride=2)
        self.fc = nn.Linear(2048, 1000)  # present in reference checkpoints, never used in forward (vision.py:11-21)
        self._slot_x1 = ops.GradSlot()
        # cut_layer3: cut the autograd graph between layer2 and layer3 (FeatureExtractor.split_backward == 2): the backward of
        # layers 3-4 (41 M of the trunk's 44.5 M parameters) then ends at a leaf copy of x3 and FeatureExtractor.backward_trunk
        # continues from it after a callback -- their gradients can travel while layers 2, 1 and the stem back-propagate
        self.cut_layer3 = False
        self._cut3 = None
        self._blk = False                 # -dtype bf16: layers 1-4 on channel-blocked bf16 activations (rsis_amd/blk_trunk.py)

    def _set_rsis_dtype(self, d):
        self._blk = int(d) == ops.DTYPE_BF16

    def _make_layer(self, planes, blocks, stride=1):
        downsample = None
        if stride != 1 or self.inplanes != planes * 4:
            downsample = nn.Sequential(HipConv2d(self.inplanes, planes * 4, 1, stride=stride, bias=False),
                                       HipBatchNorm2d(planes * 4))
        layers = [Bottleneck(self.inplanes, planes, stride, downsample)]
        self.inplanes = planes * 4
        for _ in range(1, blocks):
            layers.append(Bottleneck(self.inplanes, planes))
        return nn.Sequential(*layers)

    def forward(self, x, blk_out=False):
        """blk_out (internal fast path of FeatureExtractor.forward(blk_skips=True)): under -dtype bf16 return x5..x2 as the blk tensors
        the trunk computes in, without the converters to fp32 NCHW"""
        x1 = conv_bn(self.conv1, self.bn1, x, relu=True)   # vision.py:12-14 (x1 is the post-ReLU stem)
        hand = self.training and torch.is_grad_enabled() and x1.requires_grad
        x = ops.maxpool3x3s2(x1, grad_slot=self._slot_x1 if hand else None)   # :15
        if blk_trunk.usable(self, x):
            return self._forward_blk(x, x1, hand, blk_out)
        x2 = self.layer1(x)
        x3 = self.layer2(x2)
        self._cut3 = None
        if self.cut_layer3 and hand and x3.requires_grad:
            # (the skip connection of this level then taps the LEAF copy: a tap on x3 itself would make layers 2-1 reachable
            #  from the first half's roots, and autograd would run them there with undefined gradients)
            x3_in = x3.detach().requires_grad_(True)
            self._cut3 = (x3, x3_in)
            x4 = self.layer3(x3_in)
            x3 = x3_in
        else:
            x4 = self.layer3(x3)
        x5 = self.layer4(x4)
        if self.training:
            # x2..x4 also leave the trunk (skip connections): their outside gradient is parked for the next stage's strided
            # downsample conv, which accumulates into it in place (no memset, no autograd add)
            x1 = ops.grad_tap(x1, self._slot_x1)          # (x1 feeds the stem max-pool and the first skip conv)
            x2 = ops.grad_tap(x2, self.layer2[0]._slot_in)
            x3 = ops.grad_tap(x3, self.layer3[0]._slot_in)
            x4 = ops.grad_tap(x4, self.layer4[0]._slot_in)
        return x5, x4, x3, x2, x1

    def _forward_blk(self, x, x1, hand, blk_out=False):
        """layers 1-4 (vision.py:16-19) on channel-blocked bf16 activations: one autograd node per layer, the four feature maps
        converted back to fp32 NCHW where they leave the trunk (autograd adds a tap's gradient to the next layer's)"""
        x2b = blk_trunk.layer_forward(self.layer1, blk_trunk.to_blk(x))
        x3b = blk_trunk.layer_forward(self.layer2, x2b)
        self._cut3 = None
        if self.cut_layer3 and hand and x3b.requires_grad:
            x3_in = x3b.detach().requires_grad_(True)       # (the level-3 skip connection taps the LEAF copy, as in forward)
            self._cut3 = (x3b, x3_in)
            x3b = x3_in
        x4b = blk_trunk.layer_forward(self.layer3, x3b)
        x5b = blk_trunk.layer_forward(self.layer4, x4b)
        if self.training:
            x1 = ops.grad_tap(x1, self._slot_x1)
        if blk_out:
            return x5b, x4b, x3b, x2b, x1
        return blk_trunk.to_nchw(x5b), blk_trunk.to_nchw(x4b), blk_trunk.to_nchw(x3b), blk_trunk.to_nchw(x2b), x1
