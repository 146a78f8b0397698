"""Flat-buffer optimizer + bucketed gradient all-reduce for one-process-per-GPU data parallelism.

Replaces the reference's two torch.optim.Adam instances (utils/utils.py:83-84, train.py:239-240,185-187) and its
single-process nn.DataParallel replication (train.py:269-274).  Parameters of a group are re-homed as views of ONE
flat fp32 buffer (so are their grads): the Adam step is a single fused HIP kernel over the flat range
(rsis_adam_step), and the data-parallel gradient exchange is a handful of large RCCL all-reduces over xGMI, one per
bucket, launched from autograd hooks as soon as a bucket's gradients are final (decoder + skip bucket first -- it
overlaps the whole encoder backward).
"""
import os

import torch
import torch.distributed as dist

from . import ops


class FlatGroup(object):
    """A parameter group living in one flat buffer. `params` keep their identity (module attributes still work).

    torch.optim.Adam semantics per PARAMETER (the reference's optimizer, utils/utils.py:83-84): a parameter that has never
    received a gradient (`p.grad is None` there) is skipped entirely -- no weight decay, no moment update, no step count --
    and its bias correction starts at step 1 when its first gradient arrives.  Here every `.grad` is a view of the flat
    gradient buffer, so "has a gradient" is tracked explicitly: parameters listed in `lazy` start inactive and are switched on
    by `mark_has_grad` (train.runIter does it for fc_class / fc_stop when their loss is enabled, train.py:173-176); once active
    a parameter stays active, as a zero-filled `.grad` does in the reference.  Consecutive parameters with the same state are
    stepped by one kernel launch.

    lr_mult: optional {param: multiplier} -- the reference hands duplicated trunk tensors to Adam (utils/utils.py:34-52,
    SURVEY.md Appendix C), i.e. it applies 1/3/4 identical updates per step; pass utils.base_param_multiplicity(...) to reproduce
    that as a per-range learning-rate multiplier (first-order identical; off by default)."""

    def __init__(self, params, lr, weight_decay=0.0, betas=(0.9, 0.999), eps=1e-8, name="group", lazy=(), lr_mult=None):
        self.params = [p for p in params if p.requires_grad]
        self.lr, self.weight_decay, self.betas, self.eps, self.name = lr, weight_decay, betas, eps, name
        n = sum(p.numel() for p in self.params)
        dev = self.params[0].device if self.params else torch.device("cpu")
        self.flat_p = torch.empty(n, dtype=torch.float32, device=dev)
        self.flat_g = torch.zeros(n, dtype=torch.float32, device=dev)
        self.exp_avg = torch.zeros(n, dtype=torch.float32, device=dev)
        self.exp_avg_sq = torch.zeros(n, dtype=torch.float32, device=dev)
        self.offsets = []
        off = 0
        with torch.no_grad():
            for p in self.params:
                k = p.numel()
                self.flat_p[off:off + k].copy_(p.detach().reshape(-1))
                p.data = self.flat_p[off:off + k].view_as(p)
                p.grad = self.flat_g[off:off + k].view_as(p)
                self.offsets.append((off, k))
                off += k
        lazy_ids = set(id(p) for p in lazy)
        self.active = [id(p) not in lazy_ids for p in self.params]
        self.steps = [0] * len(self.params)                 # torch.optim.Adam's per-parameter state['step']
        self.mult = [float(lr_mult.get(p, 1.0)) if lr_mult else 1.0 for p in self.params]
        self._index = {id(p): i for i, p in enumerate(self.params)}
        self._ranges = None
        self._dev = None          # (ranges, int32 device counters) while captured graphs own the step counts
        self._dev_users = 0       # live captures sharing those counters (train.GraphedStep instances)
        ops.bump_weight_epoch()

    # ---- torch.optim.Adam's "skip parameters without a gradient" ----
    def mark_has_grad(self, params):
        for p in params:
            i = self._index.get(id(p))
            if i is not None and not self.active[i]:
                self.active[i] = True
                self._ranges = None

    @property
    def step_count(self):
        return max(self.steps) if self.steps else 0

    @step_count.setter
    def step_count(self, v):
        self.steps = [int(v)] * len(self.steps)
        self._ranges = None

    def state_key(self):
        """changes whenever the launch ranges change (a captured graph is valid for one key only)"""
        return tuple(self.active)

    def ranges(self):
        """[(offset, numel, step, lr multiplier, [param indices])] of the ACTIVE parameters, consecutive equal states merged"""
        if self._ranges is None:
            out = []
            for i, (off, k) in enumerate(self.offsets):
                if not self.active[i]:
                    continue
                if out and out[-1][0] + out[-1][1] == off and out[-1][2] == self.steps[i] and out[-1][3] == self.mult[i]:
                    last = out[-1]
                    out[-1] = (last[0], last[1] + k, last[2], last[3], last[4] + [i])
                else:
                    out.append((off, k, self.steps[i], self.mult[i], [i]))
            self._ranges = out
        return self._ranges

    def zero_grad(self):
        self.flat_g.zero_()
        for p, (off, k) in zip(self.params, self.offsets):  # re-attach (a caller may have set .grad = None)
            if p.grad is None or p.grad.data_ptr() != self.flat_g.data_ptr() + 4 * off:
                p.grad = self.flat_g[off:off + k].view_as(p)

    def step(self, gscale=1.0):
        if not self.flat_p.is_cuda:
            raise RuntimeError("FlatGroup.step: the fused Adam step runs on the GPU only")
        if self._dev is not None:
            # graph mode: the update counts live on the device and are advanced on the stream (captured + replayed)
            ranges, counters = self._dev
            counters.add_(1)
            for k, (off, n, _st, mult, _idx) in enumerate(ranges):
                ops.adam_step_flat(self.flat_p[off:off + n], self.flat_g[off:off + n], self.exp_avg[off:off + n],
                                   self.exp_avg_sq[off:off + n], self.lr * mult, self.betas[0], self.betas[1], self.eps,
                                   self.weight_decay, 1, gscale, step_dev=counters[k:k + 1], bump=False)
            if not torch.cuda.is_current_stream_capturing():
                self.note_replay()          # an EAGER step while a capture owns the counts (warm-up of another capture): mirror it
            ops.bump_weight_epoch()
            return
        ranges = self.ranges()
        for off, n, st, mult, idx in ranges:
            ops.adam_step_flat(self.flat_p[off:off + n], self.flat_g[off:off + n], self.exp_avg[off:off + n],
                               self.exp_avg_sq[off:off + n], self.lr * mult, self.betas[0], self.betas[1], self.eps,
                               self.weight_decay, st + 1, gscale, bump=False)
            for i in idx:
                self.steps[i] = st + 1
        self._ranges = None if len(ranges) > 1 else [(r[0], r[1], r[2] + 1, r[3], r[4]) for r in ranges]
        ops.bump_weight_epoch()

    # ---- hipGraph support (train.GraphedStep) ----
    def begin_graph(self):
        """freeze the launch ranges and move their update counts to the device; call before capturing step().  Several captures
        may be alive at once (one per input shape / step count): they share ONE set of device counters, which every replay and
        every eager step advances, so each of them always reads the live count."""
        ranges = [tuple(r) for r in self.ranges()]
        same = self._dev is not None and [(r[0], r[1], r[3], r[4]) for r in self._dev[0]] == [(r[0], r[1], r[3], r[4]) for r in ranges]
        if self._dev is not None and not same:
            raise RuntimeError("FlatGroup.begin_graph: the active parameter set changed while a capture is alive (release it first)")
        if self._dev is None:
            self._dev = (ranges, torch.tensor([r[2] for r in ranges], dtype=torch.int32, device=self.flat_p.device))
        self._dev_users += 1

    def note_replay(self):
        """a captured step() was replayed: mirror its count increments on the host"""
        for _off, _n, _st, _m, idx in self._dev[0]:
            for i in idx:
                self.steps[i] += 1
        self._ranges = None

    def end_graph(self):
        self._dev_users = max(0, self._dev_users - 1)
        if self._dev_users == 0:
            self._dev = None

    def state_dict(self):
        return {"step": self.step_count, "steps": list(self.steps), "active": list(self.active), "exp_avg": self.exp_avg,
                "exp_avg_sq": self.exp_avg_sq, "lr": self.lr, "weight_decay": self.weight_decay, "betas": self.betas, "eps": self.eps}

    def load_state_dict(self, sd):
        if "param_groups" in sd:
            return self._load_torch_adam(sd)
        if self._dev is not None:
            raise RuntimeError("FlatGroup.load_state_dict: leave graph mode first (end_graph)")
        if "steps" in sd and len(sd["steps"]) == len(self.steps):
            self.steps = [int(v) for v in sd["steps"]]
            self.active = [bool(v) for v in sd["active"]]
        else:                                   # round-1 checkpoints: one count for the whole group
            self.steps = [int(sd["step"])] * len(self.steps)
            self.active = [True] * len(self.steps)
        self._ranges = None
        self.exp_avg.copy_(sd["exp_avg"])
        self.exp_avg_sq.copy_(sd["exp_avg_sq"])
        return True

    def _load_torch_adam(self, sd):
        """torch.optim.Adam.state_dict() as the reference saved it (utils/utils.py:93-94): per-parameter moments in
        param_groups order.  They are adopted when they line up with this group's parameters one to one (a parameter without
        an entry had no gradient yet: it stays inactive); otherwise (e.g. the reference's trunk group lists tensors several
        times) the moments restart at zero."""
        ids = [i for g in sd.get("param_groups", []) for i in g["params"]]
        st = sd.get("state", {})
        ok = len(ids) == len(self.params) and all(i not in st or tuple(st[i]["exp_avg"].shape) == tuple(p.shape)
                                                  for i, p in zip(ids, self.params))
        if not ok or not st:
            if st:
                print("FlatGroup(%s): optimizer state does not match the parameter list; moments restart at zero" % self.name)
            return False
        for k, (i, (off, n)) in enumerate(zip(ids, self.offsets)):
            if i in st:
                self.exp_avg[off:off + n].copy_(st[i]["exp_avg"].reshape(-1))
                self.exp_avg_sq[off:off + n].copy_(st[i]["exp_avg_sq"].reshape(-1))
                self.steps[k] = int(st[i]["step"])
                self.active[k] = True
            else:
                self.steps[k], self.active[k] = 0, False
        self._ranges = None
        return True


class FlatAdam(object):
    """torch.optim.Adam semantics (incl. L2 weight decay, parameters without a gradient skipped) over a FlatGroup;
    `.step()` / `.zero_grad()` / state_dict."""

    def __init__(self, params, lr=1e-3, weight_decay=0.0, betas=(0.9, 0.999), eps=1e-8, name="adam", lazy=(), lr_mult=None):
        self.group = FlatGroup(list(params), lr, weight_decay, betas, eps, name, lazy=lazy, lr_mult=lr_mult)
        self.gscale = 1.0

    def zero_grad(self):
        self.group.zero_grad()

    def step(self):
        self.group.step(self.gscale)

    def mark_has_grad(self, params):
        self.group.mark_has_grad(params)

    def state_dict(self):
        return self.group.state_dict()

    def load_state_dict(self, sd):
        return self.group.load_state_dict(sd)


class BucketedAllReduce(object):
    """Sum-all-reduce flat gradient buffers in buckets, launched asynchronously from post-accumulate-grad hooks.

    groups: FlatGroups in the order their gradients become final during backward (decoder group first).
    The 1/world scaling is folded into the Adam kernel (gscale), so the collective is a plain SUM.
    Works with any torch.distributed backend (`nccl` = RCCL over xGMI on the GPU box, `gloo` in the CPU tests).
    """

    def __init__(self, groups, bucket_bytes=64 << 20, process_group=None, force=False):
        self.pg = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.active = dist.is_initialized() and (self.world > 1 or force)   # force: exercise the collective path at world 1
        self.hooks_enabled = True
        # staged: train.runIter exchanges the gradients itself at the cuts of a split backward (three ranges, three collectives)
        # instead of through the per-bucket hooks below; RSIS_EXCHANGE=hooks (or staged = False) selects the hooks
        self.staged = os.environ.get("RSIS_EXCHANGE", "staged") != "hooks"
        self.buckets = []      # (flat_g view, n_params)
        self._pending = []
        self._handles = []
        self._hooks = []
        self._param_bucket = {}
        for g in groups:
            # Parameters are laid out in forward order, so a group's FIRST bucket is the last one whose gradients become final
            # (the stem / layer1 of the trunk at the very end of backward): nothing is left to hide its all-reduce behind.
            # Keep it small (1/8 of the bucket size, then 1/2, then full buckets) so that the exposed tail is a short
            # latency-bound collective instead of a 64 MB one.
            start, count, first, nb = None, 0, 0, 0
            for i, (p, (off, k)) in enumerate(zip(g.params, g.offsets)):
                if start is None:
                    start, first = off, i
                count += k
                last = i == len(g.params) - 1
                cap = bucket_bytes // 8 if nb == 0 else (bucket_bytes // 2 if nb == 1 else bucket_bytes)
                if count * 4 >= cap or last:
                    nb += 1
                    bi = len(self.buckets)
                    self.buckets.append((g.flat_g[start:start + count], i - first + 1))
                    for q in g.params[first:i + 1]:
                        self._param_bucket[q] = bi
                    start, count = None, 0
        if self.active:
            for p, bi in self._param_bucket.items():
                self._hooks.append(p.register_post_accumulate_grad_hook(self._make_hook(bi)))
        self.reset()

    def _make_hook(self, bi):
        def hook(_p):
            if not self.hooks_enabled:         # split-graph replay (train.GraphedStep) exchanges the flat buffers itself
                return
            self._pending[bi] -= 1
            if self._pending[bi] == 0:
                ops.flush_wgrads()     # weight gradients parked by ops.wgrad_launch must be in the flat buffer before it travels
                self._handles.append(dist.all_reduce(self.buckets[bi][0], op=dist.ReduceOp.SUM, group=self.pg, async_op=True))
        return hook

    def reset(self):
        """call before every backward"""
        self._pending = [n for (_v, n) in self.buckets]
        self._handles = []

    def finish(self):
        """call after backward: launches the buckets whose hooks did not all fire (params without grad this step,
        e.g. fc_class / fc_stop while their losses are off) and waits for everything."""
        if self.active:
            for bi, left in enumerate(self._pending):
                if left > 0:
                    self._handles.append(dist.all_reduce(self.buckets[bi][0], op=dist.ReduceOp.SUM, group=self.pg, async_op=True))
                    self._pending[bi] = 0
            for h in self._handles:
                h.wait()
        self._handles = []
        return 1.0 / self.world
