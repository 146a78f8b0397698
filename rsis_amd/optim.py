"""Flat-buffer optimizer + bucketed gradient all-reduce for one-process-per-GPU data parallelism.

Replaces the reference's two torch.optim.Adam instances (utils/utils.py:83-84, train.py:239-240,185-187) and its
single-process nn.DataParallel replication (train.py:269-274).  Parameters of a group are re-homed as views of ONE
flat fp32 buffer (so are their grads): the Adam step is a single fused HIP kernel over the flat range
(rsis_adam_step), and the data-parallel gradient exchange is a handful of large RCCL all-reduces over xGMI, one per
bucket, launched from autograd hooks as soon as a bucket's gradients are final (decoder + skip bucket first -- it
overlaps the whole encoder backward).
"""
import torch
import torch.distributed as dist

from . import ops


class FlatGroup(object):
    """A parameter group living in one flat buffer. `params` keep their identity (module attributes still work)."""

    def __init__(self, params, lr, weight_decay=0.0, betas=(0.9, 0.999), eps=1e-8, name="group"):
        self.params = [p for p in params if p.requires_grad]
        self.lr, self.weight_decay, self.betas, self.eps, self.name = lr, weight_decay, betas, eps, name
        self.step_count = 0
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
        ops.bump_weight_epoch()

    def zero_grad(self):
        self.flat_g.zero_()
        for p, (off, k) in zip(self.params, self.offsets):  # re-attach (a caller may have set .grad = None)
            if p.grad is None or p.grad.data_ptr() != self.flat_g.data_ptr() + 4 * off:
                p.grad = self.flat_g[off:off + k].view_as(p)

    def step(self, gscale=1.0):
        self.step_count += 1
        if self.flat_p.is_cuda:
            ops.adam_step_flat(self.flat_p, self.flat_g, self.exp_avg, self.exp_avg_sq, self.lr, self.betas[0], self.betas[1],
                               self.eps, self.weight_decay, self.step_count, gscale)
        else:
            raise RuntimeError("FlatGroup.step: the fused Adam step runs on the GPU only")

    def state_dict(self):
        return {"step": self.step_count, "exp_avg": self.exp_avg, "exp_avg_sq": self.exp_avg_sq, "lr": self.lr,
                "weight_decay": self.weight_decay, "betas": self.betas, "eps": self.eps}

    def load_state_dict(self, sd):
        if "param_groups" in sd:
            return self._load_torch_adam(sd)
        self.step_count = int(sd["step"])
        self.exp_avg.copy_(sd["exp_avg"])
        self.exp_avg_sq.copy_(sd["exp_avg_sq"])

    def _load_torch_adam(self, sd):
        """torch.optim.Adam.state_dict() as the reference saved it (utils/utils.py:93-94): per-parameter moments in
        param_groups order.  They are adopted when they line up with this group's parameters one to one; otherwise
        (e.g. the reference's trunk group lists tensors several times) the moments restart at zero."""
        ids = [i for g in sd.get("param_groups", []) for i in g["params"]]
        st = sd.get("state", {})
        ok = len(ids) == len(self.params) and all(i in st and tuple(st[i]["exp_avg"].shape) == tuple(p.shape)
                                                  for i, p in zip(ids, self.params))
        if not ok:
            if st:
                print("FlatGroup(%s): optimizer state does not match the parameter list; moments restart at zero" % self.name)
            return False
        steps = []
        for i, (off, k) in zip(ids, self.offsets):
            self.exp_avg[off:off + k].copy_(st[i]["exp_avg"].reshape(-1))
            self.exp_avg_sq[off:off + k].copy_(st[i]["exp_avg_sq"].reshape(-1))
            steps.append(int(st[i]["step"]))
        self.step_count = max(steps) if steps else 0
        return True


class FlatAdam(object):
    """torch.optim.Adam semantics (incl. L2 weight decay) over FlatGroups; `.step()` / `.zero_grad()` / state_dict."""

    def __init__(self, params, lr=1e-3, weight_decay=0.0, betas=(0.9, 0.999), eps=1e-8, name="adam"):
        self.group = FlatGroup(list(params), lr, weight_decay, betas, eps, name)
        self.gscale = 1.0

    def zero_grad(self):
        self.group.zero_grad()

    def step(self):
        self.group.step(self.gscale)

    def state_dict(self):
        return self.group.state_dict()

    def load_state_dict(self, sd):
        self.group.load_state_dict(sd)


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
            self._pending[bi] -= 1
            if self._pending[bi] == 0:
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
