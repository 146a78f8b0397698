"""CPU ORACLE for the RSIS hot path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

A plain-PyTorch (CPU, fp32/fp64) restatement of the reference's algorithm for the
encoder -> recurrent ConvLSTM decoder path.  Only `tests/`, `__graft_entry__.smoke()`
and the `cpu_baseline` leg of `bench.py` may import this module, and only as the
checker / reported baseline -- never as the thing shipped.  The product path
(`rsis_amd/`) never imports it and fails loudly when its HIP library is missing.

Parity pinning: the reference (imatge-upc/rsis, Python) has NO tests or golden
vectors of its own.  This restatement is pinned against the
reference *itself*, imported unmodified in the build container through the shims in
`oracle/ref_shims/` by `oracle/make_golden.py`, which (i) asserts oracle == reference
on every fixture case and (ii) writes the reference's outputs to `tests/golden/*.npz`.
Two third-party pieces of arithmetic are absent from /root/reference and therefore
"parity unpinned" by the reference: torchvision's ResNet-101 trunk (un-pinned
`pip install torchvision`, README.md:17; restated below from its published
definition) and munkres==1.0.12 (requirements.txt:12; restated through
scipy.optimize.linear_sum_assignment -- the optimal cost is unique, tie-breaking of
equal-cost permutations is not pinned).

Every function cites the reference file:line it follows (paths relative to
/root/reference/).
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F


# ----------------------------------------------------------------------------------
# src/utils/utils.py:129-137  get_skip_dims
# ----------------------------------------------------------------------------------
def get_skip_dims(model_name):
    if model_name in ("resnet50", "resnet101"):
        return [2048, 1024, 512, 256, 64]
    if model_name == "resnet34":
        return [512, 256, 128, 64, 64]
    if model_name == "vgg16":
        return [512, 512, 256, 128, 64]
    raise Exception("The base model you chose is not supported !")


# ----------------------------------------------------------------------------------
# src/modules/clstm.py:7-62  ConvLSTMCell
# ----------------------------------------------------------------------------------
class ConvLSTMCell(nn.Module):
    """clstm.py:12-17: one Conv2d(in+hid -> 4*hid, k, padding) named `Gates`."""

    def __init__(self, args, input_size, hidden_size, kernel_size, padding):
        super().__init__()
        self.use_gpu = getattr(args, "use_gpu", False)
        self.input_size = int(input_size)
        self.hidden_size = int(hidden_size)
        self.Gates = nn.Conv2d(self.input_size + self.hidden_size, 4 * self.hidden_size,
                               kernel_size, padding=padding)

    def forward(self, input_, prev_state):
        # clstm.py:22-37: zero state when prev_state is None
        b = input_.size(0)
        spatial = list(input_.shape[2:])
        if prev_state is None:
            z = torch.zeros([b, self.hidden_size] + spatial, dtype=input_.dtype, device=input_.device)
            prev_state = (z, z.clone())
        prev_hidden, prev_cell = prev_state
        # clstm.py:43-44: channel order [x | h_prev]
        gates = self.Gates(torch.cat((input_, prev_hidden), 1))
        # clstm.py:47: output-channel order [i | f | o | g]
        in_gate, remember_gate, out_gate, cell_gate = gates.chunk(4, 1)
        in_gate = torch.sigmoid(in_gate)          # clstm.py:50
        remember_gate = torch.sigmoid(remember_gate)  # :51
        out_gate = torch.sigmoid(out_gate)        # :52
        cell_gate = torch.tanh(cell_gate)         # :55
        cell = remember_gate * prev_cell + in_gate * cell_gate  # :57
        hidden = out_gate * torch.tanh(cell)      # :58
        return [hidden, cell]                     # :60-62 (a python list)


# ----------------------------------------------------------------------------------
# torchvision.models.resnet (third-party, un-vendored; call sites vision.py:1,9,12-19,
# model.py:29-31).  Restated from the published definition: Bottleneck with the stride
# on the 3x3 conv, BN after every conv, downsample = 1x1 stride-s conv + BN;
# stem 7x7/2 p3 -> BN -> ReLU -> maxpool 3/2 p1.
# ----------------------------------------------------------------------------------
class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, kernel_size=1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, kernel_size=3, stride=stride, padding=1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, planes * 4, kernel_size=1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * 4)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample
        self.stride = stride

    def forward(self, x):
        residual = x
        out = self.relu(self.bn1(self.conv1(x)))
        out = self.relu(self.bn2(self.conv2(out)))
        out = self.bn3(self.conv3(out))
        if self.downsample is not None:
            residual = self.downsample(x)
        out = out + residual
        return self.relu(out)


class BasicBlock(nn.Module):  # only so that `from torchvision.models.resnet import BasicBlock` resolves
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 3, stride, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = nn.Conv2d(planes, planes, 3, 1, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.downsample = downsample
        self.stride = stride

    def forward(self, x):
        residual = x
        out = self.relu(self.bn1(self.conv1(x)))
        out = self.bn2(self.conv2(out))
        if self.downsample is not None:
            residual = self.downsample(x)
        return self.relu(out + residual)


class ResNet(nn.Module):
    """torchvision.models.resnet.ResNet (conv1,bn1,relu,maxpool,layer1..4,avgpool,fc)."""

    def __init__(self, block, layers, num_classes=1000):
        self.inplanes = 64
        super().__init__()
        self.conv1 = nn.Conv2d(3, 64, kernel_size=7, stride=2, padding=3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(kernel_size=3, stride=2, padding=1)
        self.layer1 = self._make_layer(block, 64, layers[0])
        self.layer2 = self._make_layer(block, 128, layers[1], stride=2)
        self.layer3 = self._make_layer(block, 256, layers[2], stride=2)
        self.layer4 = self._make_layer(block, 512, layers[3], stride=2)
        self.avgpool = nn.AvgPool2d(7)
        self.fc = nn.Linear(512 * block.expansion, num_classes)

    def _make_layer(self, block, planes, blocks, stride=1):
        downsample = None
        if stride != 1 or self.inplanes != planes * block.expansion:
            downsample = nn.Sequential(
                nn.Conv2d(self.inplanes, planes * block.expansion, kernel_size=1, stride=stride, bias=False),
                nn.BatchNorm2d(planes * block.expansion),
            )
        layers = [block(self.inplanes, planes, stride, downsample)]
        self.inplanes = planes * block.expansion
        for _ in range(1, blocks):
            layers.append(block(self.inplanes, planes))
        return nn.Sequential(*layers)

    def forward(self, x):
        x = self.maxpool(self.relu(self.bn1(self.conv1(x))))
        x = self.layer4(self.layer3(self.layer2(self.layer1(x))))
        x = self.avgpool(x)
        return self.fc(x.view(x.size(0), -1))


class ResNet101(ResNet):
    """src/modules/vision.py:6-21: returns (x5,x4,x3,x2,x1); x1 is the post-ReLU stem."""

    def __init__(self):
        super().__init__(Bottleneck, [3, 4, 23, 3], 1000)

    def forward(self, x):
        x = self.conv1(x)
        x = self.bn1(x)
        x1 = self.relu(x)
        x = self.maxpool(x1)
        x2 = self.layer1(x)
        x3 = self.layer2(x2)
        x4 = self.layer3(x3)
        x5 = self.layer4(x4)
        return x5, x4, x3, x2, x1


# ----------------------------------------------------------------------------------
# src/modules/model.py:15-70  FeatureExtractor
# ----------------------------------------------------------------------------------
class FeatureExtractor(nn.Module):
    def __init__(self, args):
        super().__init__()
        skip_dims_in = get_skip_dims(args.base_model)
        if args.base_model != "resnet101":
            raise Exception("The base model you chose is not supported !")  # model.py:37 (others out of scope)
        self.base = ResNet101()  # model.py:29-31 (pretrained download replaced by caller-provided weights)
        hs = int(args.hidden_size)
        self.hidden_size = hs
        self.kernel_size = args.kernel_size
        self.padding = 0 if self.kernel_size == 1 else 1
        k, p = self.kernel_size, self.padding
        # model.py:43-47 (python-2 integer division)
        self.sk5 = nn.Conv2d(skip_dims_in[0], hs, k, padding=p)
        self.sk4 = nn.Conv2d(skip_dims_in[1], hs, k, padding=p)
        self.sk3 = nn.Conv2d(skip_dims_in[2], hs // 2, k, padding=p)
        self.sk2 = nn.Conv2d(skip_dims_in[3], hs // 4, k, padding=p)
        self.sk1 = nn.Conv2d(skip_dims_in[4], hs // 8, k, padding=p)
        # model.py:50-54
        self.bn5 = nn.BatchNorm2d(hs)
        self.bn4 = nn.BatchNorm2d(hs)
        self.bn3 = nn.BatchNorm2d(hs // 2)
        self.bn2 = nn.BatchNorm2d(hs // 4)
        self.bn1 = nn.BatchNorm2d(hs // 8)

    def forward(self, x, semseg=False, raw=False):
        x5, x4, x3, x2, x1 = self.base(x)          # model.py:57
        x5_skip = self.bn5(self.sk5(x5))           # :59-63 (no ReLU after BN)
        x4_skip = self.bn4(self.sk4(x4))
        x3_skip = self.bn3(self.sk3(x3))
        x2_skip = self.bn2(self.sk2(x2))
        x1_skip = self.bn1(self.sk1(x1))
        if semseg:
            return x5
        if raw:
            return x5, x4, x3, x2, x1
        return x5_skip, x4_skip, x3_skip, x2_skip, x1_skip


# ----------------------------------------------------------------------------------
# src/modules/model.py:72-184  RSIS (the recurrent decoder)
# ----------------------------------------------------------------------------------
def _upsample_ac(x, size):
    """nn.UpsamplingBilinear2d(size) == bilinear, align_corners=True."""
    return F.interpolate(x, size=tuple(int(s) for s in size), mode="bilinear", align_corners=True)


class RSIS(nn.Module):
    def __init__(self, args):
        super().__init__()
        hs = int(args.hidden_size)
        self.hidden_size = hs
        self.num_classes = args.num_classes
        self.kernel_size = args.kernel_size
        padding = 0 if self.kernel_size == 1 else 1
        self.dropout = args.dropout
        self.dropout_stop = args.dropout_stop
        self.dropout_cls = args.dropout_cls
        self.skip_mode = args.skip_mode
        skip_dims_out = [hs, hs // 2, hs // 4, hs // 8, hs // 16]  # model.py:91-93
        self.clstm_list = nn.ModuleList()
        for i in range(len(skip_dims_out)):                         # model.py:98-106
            if i == 0:
                clstm_in_dim = hs
            else:
                clstm_in_dim = skip_dims_out[i - 1]
                if self.skip_mode == "concat":
                    clstm_in_dim *= 2
            self.clstm_list.append(ConvLSTMCell(args, clstm_in_dim, skip_dims_out[i], self.kernel_size, padding))
        self.conv_out = nn.Conv2d(skip_dims_out[-1], 1, self.kernel_size, padding=padding)  # :109
        fc_dim = sum(skip_dims_out)                                 # :115-117
        self.fc_class = nn.Linear(fc_dim, self.num_classes)         # :119
        self.fc_stop = nn.Linear(fc_dim, 1)                         # :120

    def forward(self, skip_feats, prev_hidden_list):
        clstm_in = skip_feats[0]                                    # model.py:124
        skip_feats = skip_feats[1:]
        side_feats, hidden_list = [], []
        for i in range(len(skip_feats) + 1):                        # :129
            state = self.clstm_list[i](clstm_in, None if prev_hidden_list is None else prev_hidden_list[i])
            hidden_list.append(state)                               # :137 (pre-dropout state recurs)
            hidden = state[0]
            if self.dropout > 0:
                hidden = F.dropout2d(hidden, self.dropout, training=True)  # :141 fresh module => training
            # :143 global max over the whole map
            side_feats.append(F.max_pool2d(hidden, kernel_size=tuple(clstm_in.shape[2:])))
            if i < len(skip_feats):
                skip_vec = skip_feats[i]
                hidden = _upsample_ac(hidden, skip_vec.shape[-2:])  # :149-150
                if self.skip_mode == "concat":
                    clstm_in = torch.cat([hidden, skip_vec], 1)    # :153
                elif self.skip_mode == "sum":
                    clstm_in = hidden + skip_vec
                elif self.skip_mode == "mul":
                    clstm_in = hidden * skip_vec
                elif self.skip_mode == "none":
                    clstm_in = hidden
                else:
                    raise Exception("Skip connection mode not supported !")
            else:
                hidden = _upsample_ac(hidden, (hidden.shape[-2] * 2, hidden.shape[-1] * 2))  # :163-164
                clstm_in = hidden
        out_mask = self.conv_out(clstm_in)                          # :167
        side_feats = torch.cat(side_feats, 1).squeeze()             # :169 (drops batch dim at B==1)
        class_feats = F.dropout(side_feats, self.dropout_cls, training=True) if self.dropout_cls > 0 else side_feats
        class_feats = self.fc_class(class_feats)                    # :174
        stop_feats = F.dropout(side_feats, self.dropout_stop, training=True) if self.dropout_stop > 0 else side_feats
        stop_probs = self.fc_stop(stop_feats)                       # :179 (a logit)
        # :182 implicit-dim nn.Softmax(): dim=1 for 2-D input, dim=0 for the 1-D (B==1) quirk
        class_probs = F.softmax(class_feats, dim=1 if class_feats.dim() == 2 else 0)
        return out_mask, class_probs, stop_probs, hidden_list       # :184


# ----------------------------------------------------------------------------------
# src/test.py:16-50  test()  (inference caller)
# ----------------------------------------------------------------------------------
@torch.no_grad()
def test(args, encoder, decoder, x, return_logits=False):
    T = args.maxseqlen
    hidden = None
    out_masks, out_classes, out_stops = [], [], []
    encoder.eval()
    decoder.eval()
    feats = encoder(x)                                              # test.py:35
    for _t in range(T):
        out_mask, out_class, out_stop, hidden = decoder(feats, hidden)   # :38
        out_mask = _upsample_ac(out_mask, x.shape[-2:])            # :39-40
        out_masks.append(out_mask)
        out_classes.append(out_class)
        out_stops.append(out_stop)
    out_masks = torch.cat(out_masks, 1)                             # :46
    out_classes = torch.cat(out_classes, 1).view(out_class.size(0), len(out_classes), -1)  # :47
    out_stops = torch.cat(out_stops, 1).view(out_stop.size(0), len(out_stops), -1)         # :48
    if return_logits:
        return out_masks, out_classes, out_stops
    return torch.sigmoid(out_masks), out_classes, torch.sigmoid(out_stops)  # :50


# ----------------------------------------------------------------------------------
# src/utils/hungarian.py
# ----------------------------------------------------------------------------------
def MaskedNLL(target, probs, balance_weights=None):
    """hungarian.py:10-32 (no epsilon: a zero probability gives inf, as in the reference)."""
    log_probs = torch.log(probs)
    if balance_weights is not None:
        log_probs = log_probs * balance_weights
    return (-torch.gather(log_probs, dim=1, index=target)).squeeze()


def StableBalancedMaskedBCE(target, out, balance_weight=None):
    """hungarian.py:34-59."""
    if balance_weight is None:
        num_positive = target.sum()
        num_negative = (1 - target).sum()
        balance_weight = num_positive / (num_positive + num_negative)
    max_val = (-out).clamp(min=0)
    loss_values = out - out * target + max_val + ((-max_val).exp() + (-out - max_val).exp()).log()
    loss_positive = loss_values * target
    loss_negative = loss_values * (1 - target)
    return ((1 - balance_weight) * loss_positive + balance_weight * loss_negative).squeeze()


def softIoU(target, out, e=1e-6):
    """hungarian.py:62-89: cost = 1 - sum(p*y) / (sum(p + y - p*y) + e)."""
    out = torch.sigmoid(out)
    num = (out * target).sum(1, True)
    den = (out + target - out * target).sum(1, True) + e
    return (1 - num / den).squeeze()


def munkres_compute(cost):
    """munkres.Munkres().compute(cost) (third-party munkres==1.0.12, requirements.txt:12):
    list of (row, col) pairs of a minimum-cost assignment of a (possibly rectangular) matrix."""
    from scipy.optimize import linear_sum_assignment
    r, c = linear_sum_assignment(np.asarray(cost, dtype=np.float64))
    return list(zip(r.tolist(), c.tolist()))


def match(masks, classes, overlaps):
    """hungarian.py:91-125: rows = GT slots, cols = predictions; perm[b, col] = row."""
    overlaps = overlaps.detach().cpu().numpy().tolist()
    t_mask, p_mask = masks
    t_class, _p_class = classes
    t_mask_cpu = t_mask.detach().cpu().numpy().copy()
    t_class_cpu = t_class.detach().cpu().numpy().copy()
    permute_indices = np.zeros((t_mask.size(0), t_mask.size(1)), dtype=int)
    for sample in range(p_mask.size(0)):
        for row, column in munkres_compute(overlaps[sample]):
            permute_indices[sample, column] = row
        t_mask_cpu[sample] = t_mask_cpu[sample, permute_indices[sample], :]
        t_class_cpu[sample] = t_class_cpu[sample, permute_indices[sample]]
    return t_mask_cpu, t_class_cpu, permute_indices


# src/utils/objectives.py:6-33 (masked_select by the sample weights)
def MaskedNLLLoss(y_true, y_pred, sw, balance_weight=None):
    costs = MaskedNLL(y_true, y_pred, balance_weight).view(-1, 1)
    return torch.masked_select(costs, sw.bool())


def MaskedBCELoss(y_true, y_pred, sw, balance_weight=None):
    costs = StableBalancedMaskedBCE(y_true, y_pred, balance_weight).view(-1, 1)
    return torch.masked_select(costs, sw.bool())


def softIoULoss(y_true, y_pred, sw):
    costs = softIoU(y_true, y_pred).view(-1, 1)
    return torch.mean(torch.masked_select(costs, sw.bool()))


# ----------------------------------------------------------------------------------
# src/train.py:54-197  runIter  (train.py is Python-2 only and cannot be imported; this
# restates its arithmetic on top of the pieces above, each of which IS pinned against the
# imported reference).  Returns the loss tensors and leaves .backward()/optimizer to the caller.
# ----------------------------------------------------------------------------------
def run_iter_forward(args, encoder, decoder, x, y_mask, y_class, sw_mask, sw_class, mode="train", assignment=None):
    """assignment: None = the Hungarian matching of train.py:137; or a (B, gt_maxseqlen) integer array perm[b, prediction] = GT slot
    to evaluate the iteration under a GIVEN assignment (tests: an assignment that ties with the optimum inside fp32 noise)."""
    T = args.maxseqlen
    hidden = None
    out_masks, out_classes, out_stops = [], [], []
    encoder.train(mode == "train")                                  # train.py:71-76
    decoder.train(mode == "train")
    feats = encoder(x)                                              # :77
    scores = torch.ones(y_mask.size(0), args.gt_maxseqlen, args.maxseqlen)  # :78
    if getattr(args, "curriculum_learning", False):
        T = min(args.maxseqlen, args.limit_seqlen_to)               # :80-81
    stop_next = False
    for t in range(T):                                              # :85
        if stop_next:
            break
        if float(sw_mask[:, t].sum()) == 0:                         # :91
            stop_next = True
        out_mask, out_class, out_stop, hidden = decoder(feats, hidden)   # :94
        out_mask = _upsample_ac(out_mask, x.shape[-2:])             # :96-97
        out_mask = out_mask.view(out_mask.size(0), -1)              # :98
        # :102-109 prediction repeated against every GT slot
        y_pred_i = out_mask.unsqueeze(1).repeat(1, y_mask.size(1), 1).view(y_mask.size(0) * y_mask.size(1), y_mask.size(2))
        y_true_p = y_mask.view(y_mask.size(0) * y_mask.size(1), y_mask.size(2))
        c = args.iou_weight * softIoU(y_true_p, y_pred_i)
        scores[:, :, t] = c.view(sw_mask.size(0), -1).detach()     # :110
        out_masks.append(out_mask)
        out_classes.append(out_class)
        out_stops.append(out_stop)
    t = len(out_masks)                                              # :117
    out_masks = torch.cat(out_masks, 1).view(out_mask.size(0), t, -1)
    out_classes = torch.cat(out_classes, 1).view(out_class.size(0), t, -1)
    out_stops = torch.cat(out_stops, 1).view(out_stop.size(0), t, -1)
    # :127-131 validity of (gt slot g, prediction t) pairs, invalid -> 10
    sw_g = sw_mask.unsqueeze(-1).repeat(1, 1, args.maxseqlen).bool()
    sw_t = sw_mask[:, 0:args.maxseqlen].unsqueeze(-1).repeat(1, 1, args.gt_maxseqlen).permute(0, 2, 1).bool()
    valid = (sw_g & sw_t).float()
    scores = scores * valid + (1 - valid) * 10
    if assignment is None:
        y_mask_perm, y_class_perm, perm_idx = match([y_mask, out_masks], [y_class, out_classes], scores)  # :137
    else:
        perm_idx = np.asarray(assignment).astype(int)
        y_mask_perm = np.stack([y_mask[b].detach().numpy()[perm_idx[b]] for b in range(y_mask.size(0))])
        y_class_perm = np.stack([y_class[b].detach().numpy()[perm_idx[b]] for b in range(y_class.size(0))])
    y_mask_perm = torch.from_numpy(y_mask_perm[:, 0:t])             # :140-141
    y_class_perm = torch.from_numpy(y_class_perm[:, 0:t])
    sw_mask_t = sw_mask[:, 0:t].contiguous().float()                # :147-148
    sw_class_t = sw_class[:, 0:t].contiguous().float()
    loss_class = torch.mean(MaskedNLLLoss(y_class_perm.reshape(-1, 1), out_classes.reshape(-1, out_classes.size(-1)),
                                          sw_mask_t.view(-1, 1)))   # :159-161
    loss_mask_iou = torch.mean(softIoULoss(y_mask_perm.reshape(-1, y_mask_perm.size(-1)),
                                           out_masks.reshape(-1, out_masks.size(-1)), sw_mask_t.view(-1, 1)))  # :162-163
    loss_stop = torch.mean(MaskedBCELoss(sw_mask_t, out_stops.squeeze(), sw_class_t.view(-1, 1),
                                         balance_weight=args.stop_balance_weight))  # :167-168, train.py:267
    loss = args.iou_weight * loss_mask_iou                          # :171
    if args.use_class_loss:
        loss = loss + args.class_weight * loss_class                # :173-174
    if args.use_stop_loss:
        loss = loss + args.stop_weight * loss_stop                  # :175-176
    return dict(loss=loss, loss_mask_iou=loss_mask_iou, loss_stop=loss_stop, loss_class=loss_class,
                out_masks=out_masks, out_classes=out_classes, out_stops=out_stops,
                y_mask_perm=y_mask_perm, y_class_perm=y_class_perm, scores=scores, assignment=perm_idx)
