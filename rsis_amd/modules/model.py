"""FeatureExtractor + RSIS decoder -- drop-in for reference src/modules/model.py on MI355X.

Same class names, constructor argument (`args` namespace), forward signatures / return values, child-module names
and state_dict keys; every conv / BN / ConvLSTM / upsample / max-pool runs in librsis_hip.so.
Differences that are deliberate: no network download in the constructor (weights come from
load_state_dict or random init), python-3 integer division for the channel pyramid, no modules constructed inside
forward.
"""
import torch
from torch import nn

import os

from .. import decoder_fused, decoder_seq, ops
from ..utils.utils import get_skip_dims
from .clstm import ConvLSTMCell
from .vision import HipBatchNorm2d, HipConv2d, ResNet101, conv_bn


# RSIS_FROZEN_TRUNK_BACKWARD=1: compute (and discard) the trunk's backward while the encoder is not being updated, as the reference does
FROZEN_TRUNK = [os.environ.get("RSIS_FROZEN_TRUNK_BACKWARD", "0") != "1"]


class FeatureExtractor(nn.Module):
    """Returns base network to extract visual features from image (reference model.py:15-70)."""

    def __init__(self, args):
        super().__init__()
        skip_dims_in = get_skip_dims(args.base_model)
        if args.base_model == "resnet101":
            self.base = ResNet101()   # model.py:29-31; pretrained weights are loaded by the caller (no download here)
        else:
            raise Exception("The base model you chose is not supported !")   # model.py:37 (resnet34/50/vgg16 out of scope)
        self.hidden_size = int(args.hidden_size)
        self.kernel_size = int(args.kernel_size)
        self.padding = 0 if self.kernel_size == 1 else 1
        hs, k, p = self.hidden_size, self.kernel_size, self.padding
        self.sk5 = HipConv2d(skip_dims_in[0], hs, k, padding=p)        # model.py:43-47
        self.sk4 = HipConv2d(skip_dims_in[1], hs, k, padding=p)
        self.sk3 = HipConv2d(skip_dims_in[2], hs // 2, k, padding=p)
        self.sk2 = HipConv2d(skip_dims_in[3], hs // 4, k, padding=p)
        self.sk1 = HipConv2d(skip_dims_in[4], hs // 8, k, padding=p)
        self.bn5 = HipBatchNorm2d(hs)                                   # model.py:50-54
        self.bn4 = HipBatchNorm2d(hs)
        self.bn3 = HipBatchNorm2d(hs // 2)
        self.bn2 = HipBatchNorm2d(hs // 4)
        self.bn1 = HipBatchNorm2d(hs // 8)
        self._bns = None
        # split_backward: cut the autograd graph between the trunk and the skip convs (train.GraphedStep with a gradient exchange):
        # the skip convs / decoder then back-propagate into leaf copies of x5..x1, and backward_trunk() continues from there --
        # so that the decoder-group gradients are final (and can travel) before the trunk's backward starts
        self.split_backward = 0          # 0 off, 1 cut at the skip convs, 2 also in front of layer3 (ResNet101.cut_layer3)
        self._cut = None
        # trunk_grad = False (train.runIter sets it from args.update_encoder): a training forward whose trunk gradients nobody will use.
        # The reference back-propagates through the ResNet on every iteration and only skips enc_opt.step() (train.py:184-187; the skip
        # convs / BatchNorms belong to the decoder's optimizer, utils.py:get_skip_params) -- until `-finetune_after` epochs have passed
        # (20 in scripts/train_cityscapes.sh) the trunk's backward is computed and thrown away.  With trunk_grad False the trunk runs
        # under no_grad (train-mode BatchNorm, running statistics updated as before) and the backward ends at the skip convs.
        self.trunk_grad = True
        ops.set_dtype(self, getattr(args, "dtype", "fp32"))      # `-dtype bf16`: bf16-operand MFMA kernels where they exist

    def _arm_bn_arena(self, device):
        """one zeroed float64 arena per iteration for the batch statistics of all 109 BN layers (forward + backward
        partial sums) instead of two memsets per layer"""
        if self._bns is None:
            self._bns = [m for m in self.modules() if isinstance(m, HipBatchNorm2d)]
            self._bn_total = sum(2 * m.num_features for m in self._bns)
        arena = torch.zeros(2 * self._bn_total, dtype=torch.float64, device=device)
        off = 0
        for m in self._bns:
            n = 2 * m.num_features
            m._arena = (arena[off:off + n], arena[self._bn_total + off:self._bn_total + off + n])
            off += n

    def forward(self, x, semseg=False, raw=False, blk_skips=False):
        """blk_skips (internal, used by train.runIter under -dtype bf16): the skip features of the levels whose trunk feature is a
        channel-blocked bf16 tensor are returned as blk tensors ([B][C/8][H][W][8]; rsis_amd.decoder_seq consumes them as they are) --
        trunk, skip branches and decoder then exchange no fp32 NCHW copies.  The default returns fp32 NCHW, as the reference does."""
        if self.training and x.is_cuda:
            self._arm_bn_arena(x.device)
        # (the second cut is armed only on the path that records the first one below: a semseg / raw caller, or one outside a training
        #  iteration, gets the uncut graph and a plain loss.backward() reaches every layer)
        frozen = self.training and torch.is_grad_enabled() and not self.trunk_grad and not (semseg or raw) and FROZEN_TRUNK[0]
        self.base.cut_layer3 = int(self.split_backward) >= 2 and not (semseg or raw) and self.training and torch.is_grad_enabled() and not frozen
        self.base._cut3 = None
        self._cut = None
        blk_skips = bool(blk_skips) and not (semseg or raw) and (self.training or not torch.is_grad_enabled()) and self.kernel_size == 3
        prev_tf, ops.TRAINING_FORWARD[0] = ops.TRAINING_FORWARD[0], frozen or ops.TRAINING_FORWARD[0]
        try:
            with torch.set_grad_enabled(torch.is_grad_enabled() and not frozen):
                x5, x4, x3, x2, x1 = self.base(x, blk_out=True) if blk_skips else self.base(x)            # model.py:57
        finally:
            ops.TRAINING_FORWARD[0] = prev_tf
        if semseg:
            return x5
        if raw:
            return x5, x4, x3, x2, x1
        if self.split_backward and self.training and torch.is_grad_enabled() and x5.requires_grad:
            roots = (x5, x4, x3, x2, x1)
            leaves = tuple(t.detach().requires_grad_(True) for t in roots)
            self._cut = (roots, leaves)
            x5, x4, x3, x2, x1 = leaves
        if blk_skips and x5.dtype == torch.bfloat16:
            # the four trunk features arrive as blk tensors (blk trunk): their skip branches as one blk autograd node; x1 (the fp32 stem
            # output) keeps the fp32 branch
            from .. import blk_trunk
            s5, s4, s3, s2 = blk_trunk.skips_forward([(self.sk5, self.bn5), (self.sk4, self.bn4), (self.sk3, self.bn3), (self.sk2, self.bn2)],
                                                     [x5, x4, x3, x2])
            return s5, s4, s3, s2, conv_bn(self.sk1, self.bn1, x1)
        x5_skip = conv_bn(self.sk5, self.bn5, x5)    # model.py:59-63 (BN, no ReLU)
        x4_skip = conv_bn(self.sk4, self.bn4, x4)
        x3_skip = conv_bn(self.sk3, self.bn3, x3)
        x2_skip = conv_bn(self.sk2, self.bn2, x2)
        x1_skip = conv_bn(self.sk1, self.bn1, x1)
        return x5_skip, x4_skip, x3_skip, x2_skip, x1_skip

    def backward_trunk(self, between=None):
        """second half of a split backward (see split_backward): back-propagate the gradients the skip convs left on the leaf
        copies of x5..x1 through the trunk.  The taps of x1..x4 (ops.grad_tap, ResNet101.forward) are the youngest nodes of the
        trunk's graph, so autograd runs them first: they park their gradient for the in-place hand-over exactly as in the
        unsplit backward.  With the second cut (split_backward == 2) this call first runs layers 4-3 down to the leaf copy of x3,
        flushes their parked weight gradients, calls between("trunk_hi") -- the gradients of layers 3-4 are final -- and then
        continues through layers 2-1 and the stem."""
        if self._cut is None:
            return False
        roots, leaves = self._cut
        self._cut = None
        trip = [(k, r, l.grad) for k, (r, l) in enumerate(zip(roots, leaves)) if l.grad is not None]      # k: x5, x4, x3, x2, x1
        pairs = [(r, g) for _k, r, g in trip]
        cut3, self.base._cut3 = self.base._cut3, None
        if cut3 is None:
            torch.autograd.backward([r for r, _g in pairs], [g for _r, g in pairs])
            return True
        # first half: x5 and the taps of x4 / x3 (the x3 tap sits on the leaf copy) -> layers 4-3; the taps of x2 / x1 wait for the
        # second half, whose graph (layers 2-1, stem) consumes what they park
        hi = [(r, g) for k, r, g in trip if k < 3]
        lo = [(r, g) for k, r, g in trip if k >= 3]
        torch.autograd.backward([r for r, _g in hi], [g for _r, g in hi])
        ops.flush_wgrads()
        if between is not None:
            between("trunk_hi")
        lo = [(cut3[0], cut3[1].grad)] + lo
        torch.autograd.backward([r for r, _g in lo], [g for _r, g in lo])
        return True


class RSIS(nn.Module):
    """The recurrent decoder (reference model.py:72-184)."""

    def __init__(self, args):
        super().__init__()
        self.hidden_size = int(args.hidden_size)
        self.num_classes = args.num_classes
        self.kernel_size = int(args.kernel_size)
        padding = 0 if self.kernel_size == 1 else 1
        self.padding = padding
        self.dropout = args.dropout
        self.dropout_stop = args.dropout_stop
        self.dropout_cls = args.dropout_cls
        self.skip_mode = args.skip_mode
        hs = self.hidden_size
        skip_dims_out = [hs, hs // 2, hs // 4, hs // 8, hs // 16]       # model.py:91-93
        self.clstm_list = nn.ModuleList()
        for i in range(len(skip_dims_out)):                              # model.py:98-106
            if i == 0:
                clstm_in_dim = hs
            else:
                clstm_in_dim = skip_dims_out[i - 1]
                if self.skip_mode == "concat":
                    clstm_in_dim *= 2
            self.clstm_list.append(ConvLSTMCell(args, clstm_in_dim, skip_dims_out[i], self.kernel_size, padding=padding))
        self.conv_out = HipConv2d(skip_dims_out[-1], 1, self.kernel_size, padding=padding)   # model.py:109
        fc_dim = sum(skip_dims_out)                                      # model.py:115-117
        self.fc_class = nn.Linear(fc_dim, self.num_classes)              # model.py:119
        self.fc_stop = nn.Linear(fc_dim, 1)                              # model.py:120
        # private per-iteration cache of the fused path (time-invariant hoisting + time-batched weight gradients)
        self._tcap = int(getattr(args, "maxseqlen", 10))
        self._tape = None
        ops.set_dtype(self, getattr(args, "dtype", "fp32"))
        self.fused = os.environ.get("RSIS_DECODER_FUSED", "1") != "0"

    def _heads(self, clstm_in, side_feats, hidden_list, keys=None):
        # keys: (per-level max-pool keys, per-level arg-max buffers) when the gate kernels pooled the hidden states themselves
        # (decoder_fused.decoder_sequence); the heads launch then decodes them into side_feats' own storage first
        out_mask = self.conv_out(clstm_in)                               # model.py:167
        if self.dropout_cls == 0 and self.dropout_stop == 0 and ops.heads_supported(side_feats, self.fc_class, self.fc_stop):
            class_probs, stop_probs = ops.heads(side_feats, self.fc_class, self.fc_stop, keys)   # model.py:169-182 in one launch
            return out_mask, class_probs, stop_probs, hidden_list
        assert keys is None, "pooled keys need the fused heads kernel"
        side_feats = torch.cat(side_feats, 1).squeeze()                  # model.py:169 (drops the batch dim at B == 1)
        if self.dropout_cls > 0:
            class_feats = nn.functional.dropout(side_feats, self.dropout_cls, training=True)
        else:
            class_feats = side_feats
        class_feats = self.fc_class(class_feats)                         # model.py:174
        if self.dropout_stop > 0:
            stop_feats = nn.functional.dropout(side_feats, self.dropout_stop, training=True)
        else:
            stop_feats = side_feats
        stop_probs = self.fc_stop(stop_feats)                            # model.py:179 (a logit)
        # model.py:182: implicit-dim nn.Softmax() -> dim 1 for 2-D input (dim 0 for the 1-D B == 1 quirk)
        class_probs = torch.softmax(class_feats, dim=1 if class_feats.dim() == 2 else 0)
        return out_mask, class_probs, stop_probs, hidden_list            # model.py:184

    def forward_sequence(self, skip_feats, T):
        """T timesteps from the zero state: what `hidden = None; for t in range(T): out_mask, cls, stop, hidden = decoder(feats, hidden)`
        computes (reference train.py:85-94, test.py:37-38), returned as ([(out_mask, class_probs, stop_probs)] * T, hidden_list).
        Runs the (level, timestep) wavefront schedule of rsis_amd.decoder_fused.decoder_sequence where it applies, the plain loop
        over forward() otherwise."""
        # (an instance whose forward() has been wrapped -- tests that record per-step intermediates -- sees every step through it)
        if decoder_seq.supported(self, skip_feats, T):
            # the whole sequence as one autograd node (explicit BPTT, rsis_amd/decoder_seq.py); the per-step tuples are views of its
            # stacked outputs (training loops use forward_sequence_stacked and never slice)
            masks, probs, stops, hidden, (Hm, Wm) = decoder_seq.decoder_sequence_stacked(self, skip_feats, T)
            B = masks.shape[0]
            return [(masks[:, t].view(B, 1, Hm, Wm), probs[:, t], stops[:, t]) for t in range(T)], hidden
        if "forward" not in self.__dict__ and decoder_fused.sequence_supported(self, skip_feats, T):
            return decoder_fused.decoder_sequence(self, skip_feats, T)
        hidden, outs = None, []
        for _t in range(T):
            out_mask, out_class, out_stop, hidden = self.forward(skip_feats, hidden)
            outs.append((out_mask, out_class, out_stop))
        return outs, hidden

    def forward_sequence_stacked(self, skip_feats, T, want_hidden=True):
        """forward_sequence with the per-step outputs stacked the way reference train.py:117-120 stacks them: (out_masks (B, T, H*W) logits,
        class_probs (B, T, C), stop logits (B, T, 1), hidden_list, (H, W) of the masks) -- or None where only the per-step path applies"""
        if not decoder_seq.supported(self, skip_feats, T):
            return None
        return decoder_seq.decoder_sequence_stacked(self, skip_feats, T, want_hidden)

    def forward(self, skip_feats, prev_hidden_list):
        if self.fused and self.skip_mode == "concat" and self.dropout == 0 and len(skip_feats) == len(self.clstm_list):
            res = decoder_fused.decoder_levels(self, skip_feats, prev_hidden_list)
            if res is not None:
                hidden_list, side_feats, up = res
                return self._heads(up, side_feats, hidden_list)
        clstm_in = [skip_feats[0]]                                       # model.py:124
        skip_feats = skip_feats[1:]
        side_feats = []
        hidden_list = []
        for i in range(len(skip_feats) + 1):                             # model.py:129
            state = self.clstm_list[i].forward_multi(clstm_in, None if prev_hidden_list is None else prev_hidden_list[i])
            hidden_list.append(state)                                    # model.py:137 (pre-dropout state recurs)
            hidden = state[0]
            if self.dropout > 0:
                hidden = nn.functional.dropout2d(hidden, self.dropout, training=True)   # model.py:141
            side_feats.append(ops.global_maxpool(hidden))                # model.py:143
            if i < len(skip_feats):
                skip_vec = skip_feats[i]
                hidden = ops.upsample_bilinear_ac(hidden, skip_vec.shape[-2:])          # model.py:149-150
                if self.skip_mode == "concat":
                    clstm_in = [hidden, skip_vec]                        # model.py:153 (concat by pointer)
                elif self.skip_mode == "sum":
                    clstm_in = [hidden + skip_vec]
                elif self.skip_mode == "mul":
                    clstm_in = [hidden * skip_vec]
                elif self.skip_mode == "none":
                    clstm_in = [hidden]
                else:
                    raise Exception("Skip connection mode not supported !")
            else:
                hidden = ops.upsample_bilinear_ac(hidden, (hidden.shape[-2] * 2, hidden.shape[-1] * 2))   # model.py:163-164
                clstm_in = [hidden]
        return self._heads(clstm_in[0], side_feats, hidden_list)
