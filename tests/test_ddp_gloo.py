"""CPU tests of the multi-GPU path (one process per GPU, gradient all-reduce in buckets): world_size 2 over gloo.
The data path shards the batch with no collective; the only exchange is the bucketed SUM all-reduce of the flat
gradient buffers (rsis_amd/optim.py), whose 1/world scaling is folded into the Adam kernel."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from rsis_amd.optim import BucketedAllReduce, FlatGroup
    torch.manual_seed(0)                       # identical replicas
    dec = torch.nn.Sequential(torch.nn.Linear(7, 5), torch.nn.Linear(5, 3))
    enc = torch.nn.Sequential(torch.nn.Linear(4, 7), torch.nn.Linear(7, 7), torch.nn.Linear(7, 7))
    unused = torch.nn.Linear(3, 3)             # e.g. fc_class while its loss is off: never receives a gradient
    g_dec = FlatGroup(list(dec.parameters()) + list(unused.parameters()), lr=1e-3, name="dec")
    g_enc = FlatGroup(list(enc.parameters()), lr=1e-6, name="enc")
    red = BucketedAllReduce([g_dec, g_enc], bucket_bytes=200)     # tiny buckets -> several per group
    assert len(red.buckets) >= 4
    # per-rank shard of a global batch of 6 samples
    torch.manual_seed(123)
    X = torch.randn(6, 4)
    shard = X[rank * 3:(rank + 1) * 3]
    for it in range(2):
        g_dec.zero_grad()
        g_enc.zero_grad()
        red.reset()
        loss = dec(enc(shard)).square().mean()
        loss.backward()
        gscale = red.finish()
        assert gscale == 1.0 / world
    flat = torch.cat([g_dec.flat_g, g_enc.flat_g]) * gscale
    # reference: single process, mean of the per-rank means == full-batch mean here (equal shard sizes)
    torch.manual_seed(0)
    dec2 = torch.nn.Sequential(torch.nn.Linear(7, 5), torch.nn.Linear(5, 3))
    enc2 = torch.nn.Sequential(torch.nn.Linear(4, 7), torch.nn.Linear(7, 7), torch.nn.Linear(7, 7))
    dec2(enc2(X)).square().mean().backward()
    ref = torch.cat([p.grad.reshape(-1) for p in dec2.parameters()] + [torch.zeros(12)] +
                    [p.grad.reshape(-1) for p in enc2.parameters()])
    q.put((rank, float((flat - ref).abs().max()), flat.tolist()))      # (plain data: a tensor's shared-memory handle can outlive its sender)
    dist.barrier()
    dist.destroy_process_group()


def test_bucketed_allreduce_world2_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert res[0][1] < 1e-6 and res[1][1] < 1e-6, res          # averaged grads == single-process full-batch grads
    assert res[0][2] == res[1][2]                                # and identical on both ranks


def _worker_real_layout(rank, world, port, q):
    """BucketedAllReduce over the product's own parameter lists (ResNet-101 trunk group + decoder / skip group; modules are built
    on the CPU, no forward): bucket boundaries, the small first buckets, base.fc left out, heads without a gradient."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from helpers import mk_args
    from rsis_amd.modules import RSIS, FeatureExtractor
    from rsis_amd.optim import BucketedAllReduce, FlatGroup
    from rsis_amd.utils.utils import get_base_params, get_skip_params
    a = mk_args(hidden_size=32)
    torch.manual_seed(0)
    enc, dec = FeatureExtractor(a), RSIS(a)
    dec_params = list(dec.parameters()) + list(get_skip_params(enc))
    lazy = list(dec.fc_stop.parameters())                       # stop loss off: fc_stop never receives a gradient
    g_dec = FlatGroup(dec_params, lr=1e-3, name="dec", lazy=lazy)
    g_enc = FlatGroup(list(get_base_params(a, enc)), lr=1e-6, name="enc")
    red = BucketedAllReduce([g_dec, g_enc], bucket_bytes=16 << 20)
    n_enc = sum(p.numel() for p in g_enc.params)
    assert all(not k.startswith("base.fc") for k, p in enc.named_parameters() if any(p is q for q in g_enc.params))
    assert n_enc == sum(p.numel() for k, p in enc.named_parameters() if k.startswith("base.") and not k.startswith("base.fc"))
    sizes = [v.numel() * 4 for v, _n in red.buckets]
    covered = sum(v.numel() for v, _n in red.buckets)
    assert covered == g_dec.flat_g.numel() + g_enc.flat_g.numel()          # every gradient element is in exactly one bucket
    # a "backward": every parameter except fc_stop receives rank-dependent gradients through autograd (hooks fire in autograd's order)
    coef = {id(p): torch.full_like(p, float(rank + 1)) * (1 + (i % 7)) for i, p in enumerate(g_dec.params + g_enc.params)}
    skip_ids = {id(p) for p in lazy}
    fired = []
    for bi, (_v, _n) in enumerate(red.buckets):
        pass
    orig = dist.all_reduce

    def spy(t, *args, **kw):
        fired.append(t.numel())
        return orig(t, *args, **kw)
    dist.all_reduce = spy
    g_dec.zero_grad()
    g_enc.zero_grad()
    red.reset()
    loss = sum((p * coef[id(p)]).sum() for p in g_dec.params + g_enc.params if id(p) not in skip_ids)
    loss.backward()
    n_from_hooks = len(fired)
    gscale = red.finish()
    dist.all_reduce = orig
    flat = torch.cat([g_dec.flat_g, g_enc.flat_g])
    rank_sum = world * (world + 1) / 2.0                      # sum over the ranks of (rank + 1)
    want = torch.cat([(coef[id(p)] * 0 if id(p) in skip_ids else coef[id(p)] / (rank + 1) * rank_sum).reshape(-1) for p in g_dec.params + g_enc.params])
    res = (rank, float((flat - want).abs().max()), gscale, len(red.buckets), n_from_hooks, len(fired), sizes[:3], g_dec.state_key().count(False))
    if world > 2:
        # ... and the STAGED exchange of the split backward (train.exchange_plan + StagedExchange: what GraphedStep's cut schedule and the
        # eager staged path run) over the same flat buffers, with the collectives of torch.distributed: each range reduced exactly once,
        # in stage order, asynchronous ones waited for; the result equals the rank sum element for element
        from rsis_amd.train import StagedExchange, exchange_plan

        class _Opt(object):                                    # exchange_plan only reads .group of FlatAdam instances
            pass
        from rsis_amd.optim import FlatAdam
        eo, do = FlatAdam.__new__(FlatAdam), FlatAdam.__new__(FlatAdam)
        eo.group, do.group = g_enc, g_dec
        pattern = torch.cat([torch.arange(g.flat_g.numel(), dtype=torch.float32) % 251 for g in (g_dec, g_enc)])
        g_dec.flat_g.copy_(pattern[:g_dec.flat_g.numel()] * (rank + 1))
        g_enc.flat_g.copy_(pattern[g_dec.flat_g.numel():] * (rank + 1))
        plan = exchange_plan(enc, [eo, do], 2)
        calls = []

        def reduce(buf, async_op):
            calls.append((buf.numel(), bool(async_op)))
            return dist.all_reduce(buf, async_op=True) if async_op else dist.all_reduce(buf)
        ex = StagedExchange(plan, reduce)
        ex("dec")
        ex("trunk_hi")
        ex.finish()
        got = torch.cat([g_dec.flat_g, g_enc.flat_g])
        res = res + (float((got - pattern * rank_sum).abs().max()), calls, [sum(t.numel() for t in plan[k]) for k in ("dec", "trunk_hi", "rest")])
    q.put(res)
    dist.barrier()
    dist.destroy_process_group()


def test_bucketed_allreduce_real_parameter_layout_world2_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_real_layout, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=300) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    for rank, err, gscale, nb, n_hooks, n_all, sizes, inactive in res:
        assert err == 0.0 and gscale == 0.5                      # sum over ranks 1 and 2 of (rank * c) == 3 c, exactly
        assert nb >= 12 and n_all == nb                          # every bucket reduced exactly once
        assert n_hooks == nb - 1                                 # all but the bucket holding fc_stop were launched from hooks
        assert sizes[0] <= (16 << 20) // 8 + (8 << 20)           # graduated first buckets (1/8, 1/2, full) of a group
        assert inactive == 2                                     # fc_stop.weight / .bias: skipped by the Adam step


def test_bucketed_allreduce_and_staged_exchange_real_parameter_layout_world8_gloo():
    """The 8-rank shape of BASELINE configs[3] / [4]: eight gloo ranks, each with the product's real parameter
    layout (ResNet-101 trunk group, decoder + skip group), (1) the bucketed all-reduce driven by autograd hooks and (2) the three-range
    staged exchange of the split backward.  Exact sums (small integers), every bucket / range reduced exactly once on every rank."""
    world = 8
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_real_layout, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=900) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(300)
        assert p.exitcode == 0
    for rank, err, gscale, nb, n_hooks, n_all, sizes, inactive, err2, calls, range_elems in res:
        assert err == 0.0 and gscale == 1.0 / world
        assert nb >= 12 and n_all == nb and n_hooks == nb - 1 and inactive == 2
        assert err2 == 0.0
        assert [a for _n, a in calls] == [True, True, False]          # dec and layers 3-4 asynchronous at their cuts, the rest at the end
        assert [n for n, _a in calls] == range_elems and range_elems[1] > 0.9 * (range_elems[1] + range_elems[2])
    assert len({tuple(r[10]) for r in res}) == 1                       # the same three ranges on every rank


def test_flatgroup_views_and_zero_grad():
    from rsis_amd.optim import FlatGroup
    lin = torch.nn.Linear(3, 2)
    w0 = lin.weight.detach().clone()
    g = FlatGroup(list(lin.parameters()), lr=1e-3)
    assert torch.equal(lin.weight.detach(), w0)                  # values preserved, now views of one flat buffer
    assert lin.weight.data_ptr() == g.flat_p.data_ptr()
    lin(torch.ones(1, 3)).sum().backward()
    assert float(g.flat_g.abs().sum()) > 0                       # autograd accumulated INTO the flat buffer
    g.zero_grad()
    assert float(g.flat_g.abs().sum()) == 0
    with pytest.raises(RuntimeError):
        g.step()                                                  # the fused Adam step is GPU-only: no CPU path


def test_steps_to_run_early_stop_rule():
    import argparse
    from rsis_amd.train import steps_to_run
    a = argparse.Namespace(maxseqlen=10, curriculum_learning=False)
    sw = torch.zeros(2, 20)
    sw[:, :4] = 1
    assert steps_to_run(a, sw) == 5       # train.py:87-92: the first all-zero step still runs, the next does not
    sw[:, :] = 1
    assert steps_to_run(a, sw) == 10
    a.curriculum_learning, a.limit_seqlen_to = True, 3
    assert steps_to_run(a, sw) == 3       # train.py:80-81


def test_host_loss_math_matches_golden():
    """rsis_amd.utils.hungarian / objectives (the product's host-side mirrors) against the reference golden values."""
    import numpy as np
    from oracle import filler
    from helpers import assert_close, gold
    from rsis_amd.utils import hungarian as Hn
    from rsis_amd.utils.objectives import MaskedBCELoss, MaskedNLLLoss, softIoULoss
    g = gold("losses")
    B, G, T, N, C = 3, 20, 10, 96, 21
    P = filler.tensor(55, "f5.P", (B * G, N), 2.0)
    Y = (filler.tensor(55, "f5.Y", (B * G, N)) > 0.3).float()
    probs = torch.softmax(filler.tensor(55, "f5.logits", (B * T, C)), 1)
    tgt = torch.from_numpy(np.random.default_rng(55).integers(0, C, (B * T, 1)))
    stop_logit = filler.tensor(55, "f5.stop", (B, T), 3.0)
    stop_tgt = (filler.tensor(55, "f5.stopt", (B, T)) > 0).float()
    sw = (filler.tensor(55, "f5.sw", (B * T, 1)) > -0.5).float()
    assert_close("softIoU", Hn.softIoU(Y, P), g["softIoU"], 1e-6)
    assert_close("MaskedNLL", Hn.MaskedNLL(tgt, probs), g["MaskedNLL"], 1e-6)
    assert_close("BCE", Hn.StableBalancedMaskedBCE(stop_tgt, stop_logit, 0.5), g["BCE_bw05"], 1e-6)
    assert_close("softIoULoss", softIoULoss()(Y[:B * T], P[:B * T], sw), g["softIoULoss"], 1e-6)
    assert_close("MaskedNLLLoss", MaskedNLLLoss()(tgt, probs, sw), g["MaskedNLLLoss"], 1e-6)
    assert_close("MaskedBCELoss", MaskedBCELoss(0.5)(stop_tgt, stop_logit, sw), g["MaskedBCELoss"], 1e-6)
    # all-pairs score matrix == the reference's repeat + softIoU
    Yb, Pb = Y.view(B, G, N), P.view(B, G, N)[:, :T]
    M = Hn.softIoU_matrix(Yb, Pb)
    ref = torch.stack([Hn.softIoU(Yb[b].repeat_interleave(T, 0), Pb[b].repeat(G, 1)).view(G, T) for b in range(B)])
    assert_close("softIoU_matrix", M, ref, 1e-6)
    scores = torch.from_numpy(np.random.default_rng(56).uniform(0, 1, (B, G, T))).float()
    ym = (filler.tensor(56, "f5.ym", (B, G, N)) > 0).float()
    yc = torch.from_numpy(np.random.default_rng(57).integers(0, C, (B, G)))
    m, c, perm = Hn.match([ym, None], [yc, None], scores)
    assert (perm == g["match_perm"]).all() and (c.numpy() == g["match_class"]).all()
    assert np.allclose(m.sum(-1).numpy(), g["match_mask_sum"])


def test_exchange_plan_partitions_the_flat_gradient_buffers():
    """train.exchange_plan (the gradient ranges that become final at the cuts of a split backward -- what GraphedStep and the staged
    eager exchange all-reduce, and when): for every cut count the ranges are disjoint views that together cover both flat gradient
    buffers exactly once; with two cuts the trunk splits in front of layer3 (parameters lie in forward order), layers 3-4 holding
    the bulk of the trunk's parameters."""
    from helpers import mk_args
    from rsis_amd.modules import FeatureExtractor, RSIS
    from rsis_amd.optim import FlatAdam
    from rsis_amd.train import exchange_plan
    from rsis_amd.utils.utils import get_base_params, get_skip_params
    a = mk_args()
    enc, dec = FeatureExtractor(a), RSIS(a)
    dec_opt = FlatAdam(list(dec.parameters()) + list(get_skip_params(enc)), lr=1e-3, name="dec")
    enc_opt = FlatAdam(list(get_base_params(a, enc)), lr=1e-6, name="enc")
    n_enc, n_dec = enc_opt.group.flat_g.numel(), dec_opt.group.flat_g.numel()
    for cuts in (0, 1, 2):
        plan = exchange_plan(enc, [enc_opt, dec_opt], cuts)
        spans = []
        for stage in ("dec", "trunk_hi", "rest"):
            for t in plan[stage]:
                base = enc_opt.group.flat_g if t.untyped_storage().data_ptr() == enc_opt.group.flat_g.untyped_storage().data_ptr() else dec_opt.group.flat_g
                spans.append((id(base), t.storage_offset(), t.storage_offset() + t.numel()))
        by = {}
        for b, lo, hi in spans:
            by.setdefault(b, []).append((lo, hi))
        assert sorted(by[id(enc_opt.group.flat_g)]) and sum(hi - lo for lo, hi in by[id(enc_opt.group.flat_g)]) == n_enc
        assert sum(hi - lo for lo, hi in by[id(dec_opt.group.flat_g)]) == n_dec
        for v in by.values():
            v.sort()
            assert v[0][0] == 0 and all(x[1] == y[0] for x, y in zip(v, v[1:]))
        assert (len(plan["dec"]) > 0) == (cuts >= 1) and (len(plan["trunk_hi"]) > 0) == (cuts == 2)
    hi = exchange_plan(enc, [enc_opt, dec_opt], 2)["trunk_hi"][0]
    first3 = next(enc.base.layer3.parameters())
    assert hi.data_ptr() == first3.grad.data_ptr()                       # the range starts at layer3's first parameter
    assert hi.numel() > 0.9 * n_enc                                      # layers 3-4: 41 M of the trunk's 42.5 M optimised parameters


def test_staged_exchange_reduces_the_ranges_of_stages_that_never_fired():
    """train.StagedExchange (the eager staged gradient exchange of runIter / GraphedStep's warm-up steps): a cut of the split backward
    only materialises when the tensor it sits on requires grad; the ranges of a stage that never called back must still be reduced
    (exactly once) before the optimizer step -- otherwise decoder / layer 3-4 gradients reach Adam un-summed and the replicas diverge."""
    from rsis_amd.train import StagedExchange

    class H(object):
        def __init__(self, log, b):
            self.log, self.b = log, b

        def wait(self):
            self.log.append(("wait", self.b))

    for fired in ([], ["dec"], ["dec", "trunk_hi"], ["trunk_hi"]):
        log = []
        plan = {"dec": ["d0", "d1"], "trunk_hi": ["hi"], "rest": ["lo"]}
        ex = StagedExchange(plan, lambda b, a: (log.append(("async" if a else "sync", b)), H(log, b))[1])
        for st in fired:
            ex(st)
        ex.finish()
        reduced = [b for kind, b in log if kind in ("async", "sync")]
        assert sorted(reduced) == ["d0", "d1", "hi", "lo"], (fired, log)          # every range exactly once
        waited = [b for kind, b in log if kind == "wait"]
        assert sorted(waited) == sorted(b for st in fired for b in plan[st])     # every asynchronous launch was waited for
        assert [b for kind, b in log if kind == "async"] == [b for st in fired for b in plan[st]]
