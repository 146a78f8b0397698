"""Training driver -- python-3 / MI355X counterpart of reference src/train.py (which is Python-2 only).

`runIter` keeps the reference's signature and return value (train.py:54-197) and reproduces its arithmetic
(SURVEY.md section 8(a) row R9 / Appendix E), restructured for one-process-per-GPU execution:
  * ONE host sync up front for the early-stop rule instead of one per timestep (train.py:91),
  * the all-pairs soft-IoU score matrix is one batched contraction after the T decoder steps instead of a
    gt_T-fold `repeat` per step + a D2H copy per step (train.py:102-110),
  * the Hungarian assignment runs on ONE D2H copy of the (B, gt_T, T) scores; GT masks are permuted on the device
    (the reference moves every GT mask to the host and back: hungarian.py:110-111, train.py:140-145),
  * masked means are computed as sum(c*sw)/sum(sw) (no data-dependent-size masked_select => no sync),
  * gradients are all-reduced by bucketed RCCL collectives overlapped with the encoder backward (rsis_amd/optim.py)
    instead of nn.DataParallel replication (train.py:269-274).
Launch: `python -m rsis_amd.train --synthetic ...` or `torchrun --nproc_per_node=8 -m rsis_amd.train ...`.
"""
import os
import pickle
import random
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

from . import ops
from .args import get_parser
from .modules.model import RSIS, FeatureExtractor
from .optim import BucketedAllReduce, FlatAdam
from .synthetic import SyntheticLoader
from .utils.hungarian import match_indices, softIoU_matrix
from .utils.objectives import MaskedBCELoss, MaskedNLLLoss, softIoULoss
from .utils.utils import (base_param_multiplicity, check_parallel, get_base_params, get_skip_params, load_checkpoint, make_dir,
                          save_checkpoint)


BLK_SKIPS = [os.environ.get("RSIS_BLK_SKIPS", "1") != "0"]       # A/B switch: blk skip features between encoder and decoder


def _blk_skips_ok(encoder, decoder, x, T=1):
    """the trunk computes in blk tensors and the decoder's sequence node will take blk skip features for this input -- asked of
    decoder_seq.supported itself (through shape probes), not re-derived here"""
    from . import blk_trunk, decoder_seq
    if not (blk_trunk.ENABLED[0] and getattr(encoder.base, "_blk", False) and x.dim() == 4 and getattr(encoder, "kernel_size", 3) == 3):
        return False
    skips = [encoder.sk5, encoder.sk4, encoder.sk3, encoder.sk2, encoder.sk1]
    if any(s.in_channels % 8 for s in skips[:4]):
        return False
    return decoder_seq.supported_for_input(decoder, [s.out_channels for s in skips], x.shape[0], x.shape[-2], x.shape[-1], max(1, int(T)))


_LOGGED = set()


def log_once(key, msg):
    """one line on stderr (rank 0) the first time a slower fallback is taken -- the step silently costing more is how perf cliffs hide"""
    if key in _LOGGED:
        return
    _LOGGED.add(key)
    if int(os.environ.get("RANK", "0")) == 0:
        print("[rsis_amd] fallback: %s" % msg, file=sys.stderr, flush=True)


def _masked_mean(costs, sw):
    """mean(masked_select(costs, sw)) for sw in {0,1} without a data-dependent shape (objectives.py:13,23,32)."""
    sw = sw.reshape(costs.shape).to(costs.dtype)
    return torch.where(sw > 0, costs, torch.zeros_like(costs)).sum() / sw.sum()


def steps_to_run(args, sw_mask):
    """train.py:80-92: T (curriculum-capped); step t still runs when sw_mask[:, t] is all zero, the next does not."""
    T = args.maxseqlen
    if getattr(args, "curriculum_learning", False):
        T = min(args.maxseqlen, args.limit_seqlen_to)
    empty = (sw_mask[:, :T].sum(0) == 0).cpu().numpy()      # the ONE early-stop sync of the iteration
    idx = np.nonzero(empty)[0]
    return T if len(idx) == 0 else int(idx[0]) + 1


def apply_update(args, optims, gscale=1.0):
    """the two optimizer steps of train.py:185-187 (dec_opt always, enc_opt when the encoder is being updated) and the rebuild of
    every packed weight copy; gscale = 1 / world_size after a SUM all-reduce of the gradients"""
    enc_opt, dec_opt = optims
    dec_opt.gscale = gscale
    enc_opt.gscale = gscale
    dec_opt.step()                                                    # :185
    if args.update_encoder:
        enc_opt.step()                                                # :186-187
    ops.repack_all()                                                  # every packed weight copy in one launch


def runIter(args, encoder, decoder, x, y_mask, y_class, sw_mask, sw_class, crits, optims, mode="train", reducer=None,
            sync_losses=True, t_run=None, want_outs=True, do_update=True, between=None):
    """Runs forward, computes loss and (if train mode) updates parameters for the provided batch (train.py:54-197).
    Returns (losses [total, iou, stop, class], outs [sigmoid(masks), class probs], perms [y_mask_perm, y_class_perm, assignment]).
    want_outs=False (training loops that only log the losses, as the reference's trainIters does: train.py:344-356): the
    sigmoid of all T full-resolution masks (train.py:191) is not computed and outs[0] holds the raw logits; the permuted GT masks
    (train.py:140, 84 MB of gather per step at batch 32) are not materialised either: perms[0] is None, perms[1] is y_class_perm.
    do_update=False: stop after the backward (gradients left in the flat buffers; the caller all-reduces them and calls
    apply_update) -- the split GraphedStep uses when a gradient exchange sits between the backward and the optimizer.
    between: callback of a split backward (encoder.split_backward): between("dec") when the decoder + skip-conv gradients are
    final and the trunk's backward is about to start, between("trunk_hi") when those of trunk layers 3-4 are (second cut)."""
    from .utils.hungarian import MaskedNLL, StableBalancedMaskedBCE, softIoU
    mask_siou, class_crit, stop_xentropy = crits
    enc_opt, dec_opt = optims
    train = mode == "train"
    # gradient exchange of an eager step (one process per GPU): STAGED by default -- the backward is cut where a range of the flat
    # gradient buffers becomes final (exchange_plan: decoder + skip group | trunk layers 3-4 | the rest) and each range is
    # all-reduced asynchronously while the next part of the backward runs: three weight-gradient flushes and three collectives per
    # step, the same schedule (and the same arithmetic) as the cut hipGraphs of GraphedStep.  reducer.staged = False
    # (RSIS_EXCHANGE=hooks) keeps the hook-driven bucketed exchange of optim.BucketedAllReduce (one flush per completed bucket).
    staged = bool(train and do_update and between is None and reducer is not None and getattr(reducer, "active", False)
                  and getattr(reducer, "staged", False) and hasattr(encoder, "split_backward"))
    restore = None
    if staged:
        restore = (encoder.split_backward, reducer.hooks_enabled)      # (not sticky: a later forward + backward outside runIter gets the uncut graph)
        encoder.split_backward, reducer.hooks_enabled = EXCHANGE_CUTS, False
        direct = getattr(reducer, "direct", None)          # RCCL bound directly (rsis_amd/comm.py) when it could be built
        between = exchange = StagedExchange(exchange_plan(encoder, optims, EXCHANGE_CUTS, bool(args.update_encoder)), direct if direct is not None else
                                            (lambda b, a: dist.all_reduce(b, op=dist.ReduceOp.SUM, group=reducer.pg, async_op=a)))
    encoder.train(train)                                             # train.py:71-76
    decoder.train(train)
    if hasattr(encoder, "trunk_grad"):
        # the trunk's gradients are used by enc_opt.step() only (train.py:186-187): while the encoder is not being updated its backward is
        # not computed at all (FeatureExtractor.trunk_grad; RSIS_FROZEN_TRUNK_BACKWARD=1 restores the reference's compute-and-discard)
        encoder.trunk_grad = bool(args.update_encoder)
    y_mask = y_mask.float()
    sw_mask = sw_mask.float()
    sw_class = sw_class.float()
    if t_run is None:       # callers that know the batch (data loader / resident synthetic batch) pass it: no sync here
        t_run = steps_to_run(args, sw_mask)

    hidden = None
    out_masks, out_classes, out_stops = [], [], []
    with torch.set_grad_enabled(train):
        # train.py:77 (once per iteration).  Under -dtype bf16 with the blk trunk and the blk decoder the skip features stay
        # channel-blocked bf16 between them (FeatureExtractor.forward(blk_skips=True)); everywhere else fp32 NCHW as in the reference
        blk_ok = (train and hasattr(decoder, "forward_sequence_stacked") and isinstance(encoder, FeatureExtractor) and
                  BLK_SKIPS[0] and _blk_skips_ok(encoder, decoder, x, t_run))
        feats = encoder(x, blk_skips=True) if blk_ok else encoder(x)
        # train.py:85-94: t_run decoder steps from the zero state -- RSIS.forward_sequence runs them in wavefront order with the
        # gate kernels of a (level, step) diagonal in one launch; same nodes, same results as t_run calls of decoder(feats, hidden)
        # (train.py never reads the final recurrent state: it is not materialised)
        stacked = decoder.forward_sequence_stacked(feats, t_run, want_hidden=False) if hasattr(decoder, "forward_sequence_stacked") else None
        if stacked is None and any(f.dim() == 5 for f in feats):
            # (cannot happen while _blk_skips_ok asks decoder_seq.supported itself; kept as a conversion instead of an error)
            from . import blk_trunk
            log_once("blk-feats-to-nchw", "blk skip features converted to fp32 NCHW: the decoder's sequence node does not apply")
            feats = [blk_trunk.to_nchw(f) if f.dim() == 5 else f for f in feats]
        if stacked is None:
            log_once("per-step-decoder", "decoder runs per step (decoder_seq.supported is False for this model / input)")
        if stacked is not None:
            # the whole sequence as ONE autograd node (rsis_amd/decoder_seq.py): outputs already in the (B, t, .) layout of :118-120
            out_masks, out_classes, out_stops, hidden, (Hm, Wm) = stacked
            t = out_masks.size(1)
            if (Hm, Wm) != (x.size(-2), x.size(-1)):                 # :96-97 (identity when the pyramid doubles up to the input size)
                up = ops.upsample_bilinear_ac(out_masks.view(-1, 1, Hm, Wm), (x.size(-2), x.size(-1)))
                out_masks = up.view(out_masks.size(0), t, -1)
        else:
            steps, hidden = decoder.forward_sequence(feats, t_run)
            for out_mask, out_class, out_stop in steps:              # train.py:85
                out_mask = ops.upsample_bilinear_ac(out_mask, (x.size(-2), x.size(-1)))         # :96-97
                out_masks.append(out_mask.reshape(out_mask.size(0), -1))                        # :98
                out_classes.append(out_class)
                out_stops.append(out_stop)
            t = len(out_masks)                                       # :117
            out_masks = torch.stack(out_masks, 1)                    # (B, t, N)   :118
            out_classes = torch.stack(out_classes, 1)                # (B, t, C)   :119
            out_stops = torch.stack(out_stops, 1)                    # (B, t, 1)   :120

    # ---- scores + matching (no grad) : train.py:78,102-110,127-137 ----
    fused_iou = ops.softiou_supported(out_masks, y_mask)    # one pass over logits + GT masks gives every soft-IoU sum
    with torch.no_grad():
        scores = torch.ones(y_mask.size(0), args.gt_maxseqlen, args.maxseqlen, device=x.device)
        if fused_iou:
            iou_sums = ops.softiou_sums(out_masks, y_mask)
            scores[:, :, :t] = args.iou_weight * ops.softiou_cost_matrix(iou_sums)
        else:
            log_once("softiou-bmm", "soft-IoU scores through torch.bmm (ops.softiou_supported is False for these shapes / dtypes)")
            scores[:, :, :t] = args.iou_weight * softIoU_matrix(y_mask, out_masks)
        valid = (sw_mask.unsqueeze(-1) * sw_mask[:, 0:args.maxseqlen].unsqueeze(1) > 0).float()   # :127-130
        scores = scores * valid + (1 - valid) * 10                                                 # :131
        if scores.is_cuda and scores.size(1) <= 64 and scores.size(1) >= scores.size(2):
            perm = ops.assign_min_cost(scores)                                                     # :137 on the device: no sync
        else:
            log_once("host-assignment", "assignment on the host (scipy, one D2H sync per step): %d GT slots x %d steps" % (scores.size(1), scores.size(2)))
            perm = torch.from_numpy(match_indices(scores)).to(x.device)                           # more than 64 GT slots: host assignment (scipy), as the reference does
        idx = perm[:, 0:t]
        # :140 -- the permuted GT masks are a RETURN value here (the matched loss below reads y_mask through `perm`): 84 MB of
        # gather per step that a training loop which only logs the losses (want_outs=False) does not need; the fallback loss does
        need_perm_masks = want_outs or not fused_iou
        y_mask_perm = torch.gather(y_mask, 1, idx.unsqueeze(-1).expand(-1, -1, y_mask.size(2))) if need_perm_masks else None
        y_class_perm = torch.gather(y_class, 1, idx)                                               # :141
    sw_mask_t = sw_mask[:, 0:t].contiguous()                         # :147
    sw_class_t = sw_class[:, 0:t].contiguous()                       # :148

    # ---- losses : train.py:159-176 (masked means) ----
    with torch.set_grad_enabled(train):
        N, C = out_masks.size(-1), out_classes.size(-1)
        if fused_iou:
            siou = ops.softiou_matched(out_masks, y_mask, perm, iou_sums)                          # same sums, no second pass
        else:
            siou = softIoU(y_mask_perm.reshape(-1, N), out_masks.reshape(-1, N))
        cls_w, bw = getattr(class_crit, "balance_weight", None), getattr(stop_xentropy, "balance_weight", None)
        if out_classes.is_cuda and out_classes.dtype == torch.float32:
            # the three masked means and their weighted sum (:159-176) in one launch each way
            cw = cls_w.to(out_classes.device, torch.float32) if cls_w is not None else None
            loss, parts = ops.loss_tail(out_classes, y_class_perm, out_stops.squeeze(-1), siou.reshape(y_class_perm.shape), sw_mask_t,
                                        sw_class_t, cw, bw, args.iou_weight, args.class_weight if args.use_class_loss else 0.0,
                                        args.stop_weight if args.use_stop_loss else 0.0)
            loss_mask_iou, loss_stop, loss_class = parts[0], parts[1], parts[2]
        else:
            nll = MaskedNLL(y_class_perm.reshape(-1, 1), out_classes.reshape(-1, C), cls_w)
            loss_class = _masked_mean(nll.reshape(-1, 1), sw_mask_t.reshape(-1, 1))               # :159-161
            loss_mask_iou = _masked_mean(siou.reshape(-1, 1), sw_mask_t.reshape(-1, 1))           # :162-163
            bce = StableBalancedMaskedBCE(sw_mask_t, out_stops.squeeze(-1), bw)
            loss_stop = _masked_mean(bce.reshape(-1, 1), sw_class_t.reshape(-1, 1))               # :167-168
            loss = args.iou_weight * loss_mask_iou                                                  # :171
            if args.use_class_loss:
                loss = loss + args.class_weight * loss_class                                        # :173-174
            if args.use_stop_loss:
                loss = loss + args.stop_weight * loss_stop                                          # :175-176

    enc_opt.zero_grad()                                               # :178-181
    dec_opt.zero_grad()
    if train:
        if reducer is not None:
            reducer.reset()
        # FlatAdam keeps every .grad as a zeroed view of one flat buffer: let the wgrad kernels accumulate into it directly
        direct = isinstance(enc_opt, FlatAdam) and isinstance(dec_opt, FlatAdam)
        if isinstance(dec_opt, FlatAdam):
            # torch.optim.Adam skips parameters whose grad is None: the heads get their first gradient when their loss is
            # switched on (train.py:173-176); until then they are not touched (no weight decay, no moments, no step count)
            if args.use_class_loss:
                dec_opt.mark_has_grad(decoder.fc_class.parameters())
            if args.use_stop_loss:
                dec_opt.mark_has_grad(decoder.fc_stop.parameters())
        prev, ops.DIRECT_GRAD[0] = ops.DIRECT_GRAD[0], direct
        prev_defer, ops.WGRAD_DEFER[0] = ops.WGRAD_DEFER[0], direct and WGRAD_DEFER_DEFAULT[0]
        try:
            loss.backward()                                           # :184
            ops.flush_wgrads()         # the weight gradients parked during backward (ops.wgrad_launch), as grouped launches
            if getattr(encoder, "_cut", None) is not None:            # split backward: the trunk's half
                if between is not None:
                    between("dec")
                encoder.backward_trunk(between)
                ops.flush_wgrads()
        finally:
            ops.DIRECT_GRAD[0] = prev
            ops.WGRAD_DEFER[0] = prev_defer
            del ops._WGRAD_QUEUE[:]
            if staged:                # (restored on the error path too: a later hook-driven step must not find its hooks off)
                encoder.split_backward, reducer.hooks_enabled = restore
        if staged:
            exchange.finish()         # waits; reduces "rest" and the range of any stage whose cut did not materialise
            apply_update(args, optims, 1.0 / reducer.world)
        elif do_update:
            apply_update(args, optims, reducer.finish() if reducer is not None else 1.0)

    losses = [loss.detach(), loss_mask_iou.detach(), loss_stop.detach(), loss_class.detach()]
    if sync_losses:
        losses = [float(v) for v in torch.stack(losses).cpu()]        # :189 (one D2H for all four)
    outs = [torch.sigmoid(out_masks.detach()) if want_outs else out_masks.detach(), out_classes.detach()]  # :191-192
    perms = [y_mask_perm, y_class_perm, perm]     # (+ the assignment itself: perm[b, prediction] = matched GT slot; hungarian.py:110)
    return losses, outs, perms


# --------------------------------------------------------------------------------------------------
# park the weight gradients during backward and flush them as grouped launches (rsis_amd.ops.wgrad_launch); RSIS_WGRAD_DEFER=0: launch
# each one where autograd reaches it
WGRAD_DEFER_DEFAULT = [os.environ.get("RSIS_WGRAD_DEFER", "1") != "0"]


def init_distributed():
    """one process per GPU; torchrun provides RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*."""
    # (before the first HIP call of the process: the host driver of this stack supports dmabuf IPC only, and without this RCCL fails
    #  with `hipIpcGetMemHandle: invalid argument`)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if torch.cuda.is_available():
        if os.environ.get("RSIS_SHARE_GPU", "") == "1":     # test hook: several ranks on one GPU (gloo backend only)
            local_rank = local_rank % torch.cuda.device_count()
        torch.cuda.set_device(local_rank)
    if world > 1 and os.environ.get("RSIS_PIN_THREADS", "1") != "0":
        # one process per GPU on one host: give every rank its own slice of the cores (the step is launched from one Python
        # thread per rank; 8 ranks all spinning up cores/1 intra-op threads fight each other)
        try:
            cores = sorted(os.sched_getaffinity(0))
            per = max(1, len(cores) // int(os.environ.get("LOCAL_WORLD_SIZE", world)))
            mine = cores[local_rank * per:(local_rank + 1) * per]
            if mine:
                os.sched_setaffinity(0, mine)
                torch.set_num_threads(max(1, min(per, 8)))
        except (AttributeError, OSError):
            pass
    force = os.environ.get("RSIS_FORCE_DIST", "") == "1"      # test hook: run the collective path even at world size 1
    if (world > 1 or force) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        backend = os.environ.get("RSIS_DIST_BACKEND", "nccl" if torch.cuda.is_available() else "gloo")
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local_rank, world


def build_optimizers(args, encoder, decoder):
    """train.py:236-240: dec_opt = decoder + skip convs/BNs (lr), enc_opt = trunk (lr_cnn).  Fused flat Adam.
    The reference's get_base_params yields trunk tensors 1-4 times, i.e. an effective 1x/3x/4x
    lr_cnn; `--enc_lr_quirk` reproduces it as a per-tensor learning-rate multiplier, the default steps each tensor once."""
    if args.optim != "adam" or args.optim_cnn != "adam":
        raise Exception("only -optim adam / -optim_cnn adam run on the fused HIP optimizer")
    decoder_params = list(decoder.parameters()) + list(get_skip_params(encoder))
    # fc_class / fc_stop receive no gradient until their loss is enabled: torch.optim.Adam leaves such parameters alone
    lazy = [] if args.use_class_loss else list(decoder.fc_class.parameters())
    lazy += [] if args.use_stop_loss else list(decoder.fc_stop.parameters())
    dec_opt = FlatAdam(decoder_params, lr=args.lr, weight_decay=args.weight_decay, name="dec", lazy=lazy)
    mult = base_param_multiplicity(encoder) if getattr(args, "enc_lr_quirk", False) else None
    enc_opt = FlatAdam(list(get_base_params(args, encoder)), lr=args.lr_cnn, weight_decay=args.weight_decay_cnn, name="enc",
                       lr_mult=mult)
    return enc_opt, dec_opt


EXCHANGE_CUTS = int(os.environ.get("RSIS_EXCHANGE_CUTS", "2"))     # 0 / 1 / 2, see GraphedStep


class StagedExchange(object):
    """The `between` callback of a split backward (runIter) for an eager staged gradient exchange: between(stage) launches the
    asynchronous SUM all-reduce of the flat-gradient ranges that are final after that stage (exchange_plan).  finish() waits for them
    and reduces, synchronously, the "rest" range AND every range whose stage never fired -- a cut of the backward only materialises
    when the tensor it sits on requires grad (frozen stem + layers 1-2, a wrapped encoder, ...); without this the ranges of the
    missing stages would reach the optimizer un-reduced and the replicas would drift apart silently."""

    def __init__(self, plan, reduce):
        self.plan, self.reduce, self.fired, self.pending = plan, reduce, [], []

    def __call__(self, stage):
        self.fired.append(stage)
        self.pending.extend(self.reduce(b, True) for b in self.plan[stage])

    def finish(self):
        for h in self.pending:
            h.wait()
        self.pending = []
        for stage in ("dec", "trunk_hi"):
            if stage not in self.fired:
                for b in self.plan[stage]:
                    self.reduce(b, False)
        for b in self.plan["rest"]:
            self.reduce(b, False)


def exchange_plan(encoder, optims, cuts, update_encoder=True):
    """flat-gradient ranges that are final after each stage of a split backward (encoder.split_backward = cuts):
    {"dec": [...], "trunk_hi": [...], "rest": [...]} -- "dec" when BPTT and the skip convs are done, "trunk_hi" when trunk layers
    4-3 are, "rest" at the end.  optims = [enc_opt, dec_opt] (FlatAdam).  update_encoder False (every rank has the same args): the
    trunk's gradients are neither computed (FeatureExtractor.trunk_grad) nor applied (train.py:186-187), so its 178 MB do not travel."""
    groups = [o.group for o in optims if isinstance(o, FlatAdam)]
    enc_g = optims[0].group
    dec = [g.flat_g for g in groups if g is not enc_g]
    from .modules import model as _model
    if not update_encoder and _model.FROZEN_TRUNK[0] and hasattr(encoder, "trunk_grad"):
        return {"dec": dec, "trunk_hi": [], "rest": []}
    if cuts <= 0:
        return {"dec": [], "trunk_hi": [], "rest": dec + [enc_g.flat_g]}
    if cuts == 1:
        return {"dec": dec, "trunk_hi": [], "rest": [enc_g.flat_g]}
    first = next(encoder.base.layer3.parameters())
    off3 = enc_g.offsets[enc_g._index[id(first)]][0]      # parameters lie in forward order: [stem, layer1, layer2 | layer3, layer4]
    return {"dec": dec, "trunk_hi": [enc_g.flat_g[off3:]], "rest": [enc_g.flat_g[:off3]]}


class GraphedStep(object):
    """One training iteration (runIter: encoder, T decoder steps, matching, losses, backward, gradient exchange, both Adam
    steps, weight repack) captured ONCE as hipGraphs and replayed: the ~1000 kernel launches of a step cost the host a
    hipGraphLaunch instead of ~35 us of Python each (the reference's train.py:85-115 loop is host-bound in the same way).
    The first `warm` calls run eagerly on a side stream (they are real training steps), the next one captures.  A capture is
    valid for one (input shapes, t_run, loss switches, update_encoder, active parameter set) key; the caller keeps one per key.
    Inputs are copied into static buffers; the returned losses / outs / perms are static device tensors overwritten by the
    next replay.  State that flows from one iteration to the next (parameters, packed weights and gate biases, Adam moments and
    step counts, BatchNorm statistics) must live in buffers allocated BEFORE the capture and be updated in place: a tensor
    re-allocated inside the capture is read at its OLD address by the part of the graph captured before it (ops.PackedConv._pack_bias;
    found by tests/test_gpu_determinism.py).  The library zero-fills with a kernel, never hipMemsetAsync (common.h rsis_zero_async).

    With a gradient exchange (`reducer` active: one process per GPU) the iteration is SEVERAL graphs, because RCCL refuses stream
    capture on this stack (ProcessGroupNCCL raises hipErrorStreamCaptureUnsupported from its watchdog thread, which terminates the
    process): the collectives are launched eagerly BETWEEN graphs, cut where a range of the flat gradient buffers becomes final
    (`cuts`, default RSIS_EXCHANGE_CUTS = 2):
        graph A  : forward, matching, losses, BPTT through the decoder and the skip convs, their weight-gradient flush
        RCCL     : all-reduce of the decoder + skip group (24 MB at hidden 128), asynchronous          ||  graph B1
        graph B1 : backward of trunk layers 4-3 + their weight-gradient flush          (cuts >= 1; with cuts == 1: the whole trunk)
        RCCL     : all-reduce of the gradients of layers 3-4 (165 of the trunk's 178 MB), asynchronous ||  graph B2
        graph B2 : backward of layers 2-1 and the stem + flush                         (cuts == 2)
        RCCL     : wait for both, all-reduce of what is left (5.6 MB: the exposed part of the exchange)
        graph C  : both Adam steps (gradient scale 1 / world) + the batched weight repack
    i.e. the bucketed, overlapped schedule of the eager path (optim.BucketedAllReduce) at the granularity a capture allows.  Every
    cut costs ~0.3 ms per step (a graph-launch bubble and a less well filled weight-gradient group launch); cuts = 0 is the
    two-graph form (backward | all-reduce of everything, exposed | update).  `timing = True` brackets the segments with HIP events
    (bench.py prints their averages)."""

    def __init__(self, args, encoder, decoder, crits, optims, reducer=None, warm=2, pool=None, cuts=None):
        self.args, self.encoder, self.decoder, self.crits, self.optims, self.reducer = args, encoder, decoder, crits, optims, reducer
        self.split = reducer is not None and getattr(reducer, "active", False)
        # direct: the collectives go through rsis_amd.comm.DirectReducer and live INSIDE one captured graph (forked stream); else the
        # iteration is cut into graphs with torch.distributed collectives between them
        self.direct = getattr(reducer, "direct", None) if self.split else None
        self.cuts = max(0, min(2, EXCHANGE_CUTS if cuts is None else int(cuts))) if self.split else 0
        if self.split:
            # this schedule all-reduces the flat gradient buffers itself: the reducer's per-bucket hooks must not fire inside
            # these backward passes (they would reduce the first warm-up step's gradients twice)
            reducer.hooks_enabled = False
            encoder.split_backward = self.cuts
        self.warm, self.pool = warm, pool
        self.graphs, self.graph_update, self.static, self.result, self.t_run = None, None, None, None, None
        self.stream = torch.cuda.Stream()
        self.n_eager = 0
        self._bns = None
        self.failed = None
        self._in_src = None
        self.timing, self._events = False, []

    # (`graph` is what callers test for "captured": the first graph of the iteration)
    @property
    def graph(self):
        return self.graphs[0] if self.graphs else None

    @property
    def graph_b(self):
        return self.graphs[1] if self.graphs and len(self.graphs) > 1 else None

    def _run(self, batch, t_run, do_update=True, between=None):
        return runIter(self.args, self.encoder, self.decoder, *batch, self.crits, self.optims, mode="train",
                       reducer=None if self.split else self.reducer, sync_losses=False, t_run=t_run, want_outs=False, do_update=do_update,
                       between=between)

    def _groups(self):
        return [o.group for o in self.optims if isinstance(o, FlatAdam)]

    def _reduce(self, buf, async_op=False):
        """SUM all-reduce of (a range of) a flat gradient buffer, issued from the current stream (RCCL runs it on its own stream
        after the work already enqueued here; `wait()` -- or the synchronous form -- makes the current stream wait for it)"""
        if self.direct is not None:
            return self.direct(buf, async_op)
        return dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.reducer.pg, async_op=async_op)

    def _plan(self):
        return exchange_plan(self.encoder, self.optims, self.cuts, bool(self.args.update_encoder))

    def __call__(self, batch, t_run):
        if self.graphs is None and self.failed is None and self.n_eager >= self.warm:
            self._capture(batch, t_run)
        if self.graphs is None:                    # warm-up steps (or capture not possible): plain eager iterations
            self.n_eager += 1
            self.stream.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(self.stream):
                if self.split:
                    exchange = StagedExchange(self._plan(), self._reduce)
                    res = self._run(batch, t_run, do_update=False, between=exchange)
                    exchange.finish()     # (also covers a backward whose cuts did not materialise: every range is reduced exactly once)
                    apply_update(self.args, self.optims, 1.0 / self.reducer.world)
                else:
                    res = self._run(batch, t_run)
            torch.cuda.current_stream().wait_stream(self.stream)
            return res
        if t_run != self.t_run:
            raise RuntimeError("GraphedStep: captured for t_run=%d, called with %d" % (self.t_run, t_run))
        # inputs -> static buffers.  Skipped only for a resident batch that is already there: the SAME tensor objects (held here by
        # strong references, so their addresses cannot have been recycled by the caching allocator for another batch), unmodified
        # since the last copy.  A loader that yields fresh tensors per step always pays the copy.
        same = (self._in_src is not None and len(self._in_src) == len(batch) and
                all(a is b and a._version == v for a, (b, v) in zip(batch, self._in_src)))
        if not same:
            for dst, src in zip(self.static, batch):
                if dst.data_ptr() != src.data_ptr():
                    dst.copy_(src, non_blocking=True)
            self._in_src = [(t, t._version) for t in batch]
        if not self.split:
            self.graphs[0].replay()
        elif self.direct is not None:              # ONE graph: backward, the all-reduces on their forked branch, both Adam steps, repack
            if self.timing:
                ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
                ev[0].record()
                self.graphs[0].replay()
                ev[1].record()
                self._events.append(ev)
            else:
                self.graphs[0].replay()
        else:
            n = len(self.graphs)
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 3)] if self.timing else None
            mark = (lambda k: ev[k].record()) if ev is not None else (lambda k: None)
            plan, pending = self._plan(), []
            stages = ["dec", "trunk_hi"]
            mark(0)
            for k, g in enumerate(self.graphs):
                g.replay()                                         # A, B1, B2 ...
                mark(k + 1)
                if k + 1 < n:                                      # this range is final: it travels while the next graph runs
                    pending.extend(self._reduce(b, True) for b in plan[stages[k]])
            for h in pending:
                h.wait()
            for b in plan["rest"]:                                 # the exposed part
                self._reduce(b)
            mark(n + 1)
            self.graph_update.replay()                             # C
            mark(n + 2)
            if ev is not None:
                self._events.append(ev)

        for g in self._groups():
            if self.args.update_encoder or g is not self.optims[0].group:
                g.note_replay()
        for m in self._bns:                        # HipBatchNorm2d counts its training calls on the host
            m._nbt_pending += 1
        return self.result

    def segment_ms(self):
        """average milliseconds per segment over the timed replays so far (`timing`); synchronises"""
        if not self._events:
            return None
        torch.cuda.synchronize()
        n, ng = len(self._events), len(self.graphs)
        if self.direct is not None:
            ms = sum(e[0].elapsed_time(e[1]) for e in self._events) / n
            self._events = []
            return {"one_graph_fwd_bwd_allreduce_adam_repack": ms, "exposed_allreduce": 0.0, "replays": n, "cuts": 0,
                    "mode": "RCCL bound directly: the all-reduces are nodes of the iteration's graph (forked branch)"}
        seg = [sum(e[k].elapsed_time(e[k + 1]) for e in self._events) / n for k in range(ng + 2)]
        self._events = []
        names = (["graph_A_fwd_bptt"] + ["graph_B%d_trunk_bwd" % (k + 1) for k in range(ng - 1)])
        out = {nm: seg[k] for k, nm in enumerate(names)}
        out["exposed_allreduce"] = seg[ng]
        out["graph_C_adam_repack"] = seg[ng + 1]
        out["replays"], out["cuts"] = n, self.cuts
        return out

    def _capture(self, batch, t_run):
        from .modules.vision import HipBatchNorm2d
        self.static = [t.clone() for t in batch]
        self.t_run = t_run
        self._bns = [m for mod in (self.encoder, self.decoder) for m in mod.modules() if isinstance(m, HipBatchNorm2d)]
        pend = [m._nbt_pending for m in self._bns]
        groups = self._groups()
        for g in groups:
            g.begin_graph()
        graphs, graph_u = [torch.cuda.CUDAGraph()], torch.cuda.CUDAGraph()
        try:
            torch.cuda.synchronize()
            if dist.is_initialized():
                # Let the RCCL watchdog thread retire the (now complete) collectives of the warm-up steps before the capture
                # starts: it polls its work list every 100 ms with event queries, and an event query that lands while this
                # stream captures kills the process from inside the watchdog (hipErrorStreamCaptureUnsupported under the global
                # capture mode, hipErrorCapturedEvent under the thread-local one: 1 run in 4-6 of tests/test_gpu_ddp.py's
                # world-size-1 worker).  No collective is issued during the capture, so an empty list stays empty.
                time.sleep(0.6)
            # thread-local capture mode: under the default (global) mode an event query from ANY thread while this one captures
            # is an error, and the RCCL watchdog thread polls the events of the eager all-reduces that ran just before -- one
            # run in six died with hipErrorStreamCaptureUnsupported raised inside the watchdog (the loader's staging threads
            # are the other candidate)
            if not self.split:
                with torch.cuda.graph(graphs[0], pool=self.pool, stream=self.stream, capture_error_mode="thread_local"):
                    self.result = self._run(self.static, t_run)
            elif self.direct is not None:
                # the split backward still tells WHEN a range of the flat gradient buffers is final (between("dec") / ("trunk_hi"));
                # its all-reduce is captured on the forked stream right there and joined before the optimizer steps
                with torch.cuda.graph(graphs[0], pool=self.pool, stream=self.stream, capture_error_mode="thread_local"):
                    exchange = StagedExchange(self._plan(), self._reduce)
                    self.result = self._run(self.static, t_run, do_update=False, between=exchange)
                    exchange.finish()
                    apply_update(self.args, self.optims, 1.0 / self.reducer.world)
            else:
                # one pass of runIter, cut into graphs where its split backward calls back (same thread: the callback runs in
                # runIter between the backward calls), then graph C; all share one memory pool (they always replay in this
                # order, so blocks freed by one may be reused by the next exactly as inside a single capture)
                def cut(_stage):
                    graphs[-1].capture_end()
                    graphs.append(torch.cuda.CUDAGraph())
                    graphs[-1].capture_begin(pool=graphs[0].pool(), capture_error_mode="thread_local")
                import gc
                gc.collect()
                torch.cuda.empty_cache()
                self.stream.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(self.stream):
                    graphs[0].capture_begin(pool=self.pool, capture_error_mode="thread_local")
                    try:
                        self.result = self._run(self.static, t_run, do_update=False, between=cut)
                    finally:
                        graphs[-1].capture_end()
                    graph_u.capture_begin(pool=graphs[0].pool(), capture_error_mode="thread_local")
                    try:
                        apply_update(self.args, self.optims, 1.0 / self.reducer.world)
                    finally:
                        graph_u.capture_end()
                torch.cuda.current_stream().wait_stream(self.stream)
                if len(graphs) != self.cuts + 1:
                    raise RuntimeError("GraphedStep: expected %d cuts of the backward, got %d" % (self.cuts, len(graphs) - 1))
                self.graph_update = graph_u
            self.graphs = graphs
        except Exception as e:  # noqa: BLE001  (capture refused: stay eager)
            self.failed = repr(e)
            for g in groups:
                g.end_graph()
            self.static = None
        for m, v in zip(self._bns, pend):          # the capture pass itself executed nothing
            m._nbt_pending = v

    def release(self):
        for g in self._groups():
            g.end_graph()
        self.graphs = self.graph_update = self.result = self.static = self._in_src = None


def init_dataloaders(args, rank=0, world=1):
    """`-batch_size` is the GLOBAL batch, as in the reference (`-batch_size B -ngpus N` under nn.DataParallel splits B over the
    GPUs, train.py:269-274): every rank of a torchrun job takes B / world samples, so reference hyper-parameters carry over.
    (The loss is the mean of the per-rank masked means -- equal to the global masked mean for equal shard sizes.)"""
    if not getattr(args, "synthetic", False):
        if args.dataset != "leaves":
            raise Exception("data: --synthetic, or -dataset leaves (CVPPP A1, BASELINE configs[0]); the Pascal VOC / Cityscapes readers "
                            "of the reference's src/dataloader are host-side I/O outside the hot path (SURVEY.md section 8(f) row N3)")
        from .dataloader.leaves import DeviceLoader, LeavesDataset
        if args.batch_size % world != 0:
            raise Exception("-batch_size %d (the global batch) is not divisible by the %d ranks" % (args.batch_size, world))
        loaders = {}
        for split in ("train", "val"):                                       # train.py:31-49
            ds = LeavesDataset(args, split=split, augment=args.augment and split == "train", resize=args.resize, imsize=args.imsize)
            loaders[split] = DeviceLoader(ds, args.batch_size // world, shuffle=True, num_workers=args.num_workers,
                                          seed=args.seed + (0 if split == "train" else 1), rank=rank, world=world)
        return loaders, ds.get_classes()
    if args.batch_size % world != 0:
        raise Exception("-batch_size %d (the global batch) is not divisible by the %d ranks" % (args.batch_size, world))
    per_rank = args.batch_size // world
    loaders = {"train": SyntheticLoader(args, args.synthetic_batches, args.seed, rank=rank, batch_size=per_rank),
               "val": SyntheticLoader(args, max(1, args.synthetic_batches // 4), args.seed + 7, rank=rank, batch_size=per_rank)}
    return loaders, ["<eos>"] + ["class%d" % i for i in range(1, args.num_classes)]


def trainIters(args):
    rank, _local_rank, world = init_distributed()
    epoch_resume = 0
    model_dir = os.path.join(args.models_root, args.model_name)
    enc_opt_dict = dec_opt_dict = None
    resuming = bool(args.resume)          # (args is replaced by the checkpoint's namespace below, whose `resume` is False)
    if args.resume:                                                       # train.py:204-215
        encoder_dict, decoder_dict, enc_opt_dict, dec_opt_dict, load_args = load_checkpoint(args.model_name, args.use_gpu,
                                                                                            root=args.models_root)
        epoch_resume = load_args.epoch_resume
        # command-line settings of THIS run that override the checkpoint's namespace -- copied BEFORE the modules are built: the
        # modules read `dtype` (which MFMA kernels they run) from the namespace they are constructed with, and the namespace that
        # is re-saved next to the checkpoint must say what actually ran
        for k in ("synthetic", "synthetic_batches", "synthetic_instances", "models_root", "max_epoch", "log_term", "graph", "dtype",
                  "enc_lr_quirk"):
            setattr(load_args, k, getattr(args, k, None))
        encoder, decoder = FeatureExtractor(load_args), RSIS(load_args)
        encoder_dict, decoder_dict = check_parallel(encoder_dict, decoder_dict)
        encoder.load_state_dict(encoder_dict)
        decoder.load_state_dict(decoder_dict)
        args = load_args
    elif args.transfer:                                                   # train.py:217-224
        encoder_dict, decoder_dict, enc_opt_dict, dec_opt_dict, load_args = load_checkpoint(args.transfer_from, args.use_gpu,
                                                                                            root=args.models_root)
        load_args.dtype = getattr(args, "dtype", "fp32")                   # (the architecture comes from the checkpoint, the arithmetic from this run)
        encoder, decoder = FeatureExtractor(load_args), RSIS(load_args)
        encoder_dict, decoder_dict = check_parallel(encoder_dict, decoder_dict)
        encoder.load_state_dict(encoder_dict)
        decoder.load_state_dict(decoder_dict)
        if load_args.num_classes != args.num_classes:                     # train.py:249-251: new classification head
            decoder.fc_class = torch.nn.Linear(decoder.fc_class.weight.size(1), args.num_classes)
            decoder.num_classes = args.num_classes
    else:
        encoder, decoder = FeatureExtractor(args), RSIS(args)             # train.py:227-228
    if not args.use_gpu:
        raise Exception("--cpu: this build has no CPU path (the HIP library is the product)")
    encoder.cuda()
    decoder.cuda()
    if world > 1:                                                         # identical replicas
        for p in list(encoder.parameters()) + list(decoder.parameters()) + list(encoder.buffers()):
            dist.broadcast(p.data, 0)
    if rank == 0:
        make_dir(args.models_root)
        make_dir(model_dir)
        pickle.dump(args, open(os.path.join(model_dir, "args.pkl"), "wb"))   # train.py:234
    enc_opt, dec_opt = build_optimizers(args, encoder, decoder)
    if (resuming or args.transfer) and enc_opt_dict is not None:
        # own format or the reference's torch.optim.Adam state_dict ('state' / 'param_groups'); load_state_dict dispatches
        ok_e = enc_opt.load_state_dict(enc_opt_dict)
        new_head = args.transfer and "exp_avg" in dec_opt_dict and dec_opt_dict["exp_avg"].numel() != dec_opt.group.exp_avg.numel()
        ok_d = False if new_head else dec_opt.load_state_dict(dec_opt_dict)
        if rank == 0 and not (ok_e and ok_d):
            print("optimizer state: %s restored, %s restart from zero moments"
                  % (", ".join(n for n, ok in (("encoder", ok_e), ("decoder", ok_d)) if ok) or "nothing",
                     ", ".join(n for n, ok in (("encoder", ok_e), ("decoder", ok_d)) if not ok)))
    reducer = BucketedAllReduce([dec_opt.group, enc_opt.group]) if world > 1 else None
    if reducer is not None and reducer.active:
        from .comm import make_direct_reducer
        reducer.direct = make_direct_reducer(lambda m: print("[train] %s" % m, file=sys.stderr, flush=True) if rank == 0 else None)
    if not args.log_term and rank == 0:                                   # train.py:253-256
        print("Training logs will be saved to:", os.path.join(model_dir, "train.log"))
        sys.stdout = open(os.path.join(model_dir, "train.log"), "w")
        sys.stderr = open(os.path.join(model_dir, "train.err"), "w")
    if rank == 0:
        print(args)
    crits = [softIoULoss(), MaskedNLLLoss(balance_weight=None), MaskedBCELoss(balance_weight=args.stop_balance_weight)]
    optims = [enc_opt, dec_opt]
    torch.cuda.synchronize()
    start = time.time()
    best_val_loss = args.best_val_loss
    acc_patience = 0
    mt_val = -1
    if args.curriculum_learning and epoch_resume == 0:
        args.limit_seqlen_to = 2                                          # train.py:299-300
    loaders, _class_names = init_dataloaders(args, rank, world)
    graphs = {}

    def reload_best():                                                    # train.py:456-460 etc.
        e_d, d_d, eo, do, _ = load_checkpoint(args.model_name, args.use_gpu, root=args.models_root)
        encoder.load_state_dict(e_d)
        decoder.load_state_dict(d_d)
        for gs in graphs.get("cache", {}).values():      # the optimizer state is about to change behind the captured graphs
            gs.release()
        graphs.clear()
        enc_opt.load_state_dict(eo)
        dec_opt.load_state_dict(do)
        ops.bump_weight_epoch()

    for e in range(args.max_epoch):
        if rank == 0:
            print("Epoch", e + epoch_resume)
        epoch_losses = {s: {"total": [], "iou": [], "stop": [], "class": []} for s in ("train", "val")}
        ep = e + epoch_resume
        if ep >= args.finetune_after and not args.update_encoder and not args.finetune_after == -1:     # :313-318
            print("Starting to update encoder")
            args.update_encoder = True
            acc_patience, mt_val = 0, -1
        if ep >= args.class_loss_after and not args.use_class_loss and not args.class_loss_after == -1:  # :319-324
            print("Starting to learn class loss")
            args.use_class_loss = True
            best_val_loss, acc_patience, mt_val = 1000, 0, -1
        if ep >= args.stop_loss_after and not args.use_stop_loss and not args.stop_loss_after == -1:     # :325-339
            if (not args.curriculum_learning) or args.limit_seqlen_to > args.min_steps:
                print("Starting to learn stop loss")
                args.use_stop_loss = True
                best_val_loss, acc_patience, mt_val = 1000, 0, -1
        for split in ("train", "val"):                                    # :341
            n_img, t_split = 0, time.time()
            for batch_idx, (x, y_mask, y_class, sw_mask, sw_class) in enumerate(loaders[split]):
                t_run = loaders[split].steps_to_run(args, sw_mask) if hasattr(loaders[split], "steps_to_run") else None
                if split == "train" and getattr(args, "graph", False) and t_run is not None:
                    # one captured hipGraph per (shapes, T, loss switches, encoder update, active parameter set)
                    key = (tuple(x.shape), tuple(y_mask.shape), t_run, args.use_class_loss, args.use_stop_loss, args.update_encoder)
                    if graphs.get("mode", key[3:]) != key[3:]:       # loss switches / encoder update changed: other launch set
                        for gs in graphs.pop("cache", {}).values():
                            gs.release()
                        graphs.pop("key", None)
                    graphs["mode"] = key[3:]
                    if graphs.get("key") != key:
                        # (a data set whose batches stop at different steps alternates between a few keys: keep the last few
                        #  captures instead of re-capturing on every change; a resident synthetic batch only ever has one)
                        cache = graphs.setdefault("cache", {})
                        if key not in cache:
                            # Policy (round 6, measured with tools/graph_cache_bench.py): a loader whose batches stop at different steps
                            # produces at most `maxseqlen` values of t_run per mode, so keeping maxseqlen + 1 captures means every key is
                            # captured ONCE and replayed from then on -- with the four kept until round 5 a CVPPP-like set (3-9 leaves per
                            # image, 7+ keys) re-captured continually (2 eager steps + a capture each time).  The captures share one graph
                            # memory pool; what a key costs is its static input copies (the GT masks: B x gt_maxseqlen x H x W floats).
                            keep = max(4, int(args.maxseqlen) + 1)
                            while len(cache) >= keep:              # evict the least recently USED capture (dicts keep insertion order)
                                cache.pop(next(iter(cache))).release()
                            # the captures never replay concurrently and their results are cloned at once: ONE graph memory pool for all
                            # of them (a private pool each held a full step's activations per key)
                            if "pool" not in graphs:
                                graphs["pool"] = torch.cuda.graph_pool_handle()
                            cache[key] = GraphedStep(args, encoder, decoder, crits, optims, reducer, pool=graphs["pool"])
                        else:
                            cache[key] = cache.pop(key)            # most recently used goes last
                        graphs["key"], graphs["step"] = key, cache[key]
                    losses, _outs, _perm = graphs["step"]((x, y_mask, y_class, sw_mask, sw_class), t_run)
                    if graphs["step"].failed is not None and not graphs.get("warned") and rank == 0:
                        print("[train] hipGraph capture failed (%s): this configuration runs eagerly" % graphs["step"].failed)
                        graphs["warned"] = True
                    losses = [v.clone() for v in losses]      # (static tensors of the graph: keep this step's values)
                else:
                    losses, _outs, _perm = runIter(args, encoder, decoder, x, y_mask, y_class, sw_mask, sw_class, crits, optims,
                                                   mode=split, reducer=reducer, sync_losses=False, t_run=t_run, want_outs=False)
                for k, v in zip(("total", "iou", "stop", "class"), losses):
                    epoch_losses[split][k].append(v)
                n_img += x.size(0) * world
                if (batch_idx + 1) % args.print_every == 0:               # :360-401
                    m = {k: float(torch.stack(v).mean()) for k, v in epoch_losses[split].items()}
                    torch.cuda.synchronize()
                    te = time.time() - start
                    if rank == 0:
                        print("iter %d:\ttotal:%.4f\tclass:%.4f\tiou:%.4f\tstop:%.4f\ttime:%.4f\timg/s:%.1f"
                              % (batch_idx, m["total"], m["class"], m["iou"], m["stop"], te,
                                 args.print_every * x.size(0) * world / max(te, 1e-9)))
                    start = time.time()
            # (a split without a single batch -- a validation set smaller than one batch -- gives nan means, as np.mean([]) does in the
            #  reference, train.py:403-405: no checkpoint is saved on it, the epoch goes on)
            m = {k: (float(torch.stack(v).mean()) if v else float("nan")) for k, v in epoch_losses[split].items()}
            if world > 1:                                                 # epoch means over all ranks
                tv = torch.tensor([m["total"], m["iou"], m["stop"], m["class"]], device="cuda")
                dist.all_reduce(tv)
                m = dict(zip(("total", "iou", "stop", "class"), (tv / world).tolist()))
            mt = m["total"]
            if split == "val" and args.smooth_curves:                     # :406-411
                mt = mt if mt_val == -1 else 0.9 * mt_val + 0.1 * mt
                mt_val = mt
            args.epoch_resume = ep
            if rank == 0:
                print("Epoch %d:\ttotal:%.4f\tclass:%.4f\tiou:%.4f\tstop:%.4f\t(%s)" % (e, mt, m["class"], m["iou"], m["stop"], split))
        if mt < (best_val_loss - args.min_delta):                         # :440-446
            best_val_loss = mt
            args.best_val_loss = best_val_loss
            if rank == 0:
                print("Saving checkpoint.")
                save_checkpoint(args, encoder, decoder, enc_opt, dec_opt, root=args.models_root)
            if world > 1:
                dist.barrier()
            acc_patience = 0
        else:
            acc_patience += 1
        if acc_patience > args.patience and not args.use_class_loss and not args.class_loss_after == -1:   # :450-460
            print("Starting to learn class loss")
            acc_patience, args.use_class_loss, best_val_loss, mt_val = 0, True, 1000, -1
            reload_best()
        if acc_patience > args.patience and args.curriculum_learning and args.limit_seqlen_to < args.maxseqlen:  # :461-467
            acc_patience = 0
            args.limit_seqlen_to += args.steps_cl
            print("Adding one step more:", args.limit_seqlen_to)
            best_val_loss, mt_val = 1000, -1
        if acc_patience > args.patience and not args.update_encoder and not args.finetune_after == -1:     # :469-479
            print("Starting to update encoder")
            acc_patience, args.update_encoder, best_val_loss, mt_val = 0, True, 1000, -1
            reload_best()
        if acc_patience > args.patience and not args.use_stop_loss and not args.stop_loss_after == -1:     # :480-499
            if (not args.curriculum_learning) or args.limit_seqlen_to > args.min_steps:
                print("Starting to learn stop loss")
                acc_patience, args.use_stop_loss, best_val_loss, mt_val = 0, True, 1000, -1
            reload_best()
        if acc_patience > args.patience_stop:                             # :501-502
            break
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    parser = get_parser()
    args = parser.parse_args()
    torch.manual_seed(args.seed)                                          # train.py:509-513
    random.seed(args.seed)
    if args.use_gpu and torch.cuda.is_available():
        torch.cuda.manual_seed(args.seed)
    trainIters(args)
