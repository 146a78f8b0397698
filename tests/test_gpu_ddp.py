"""GPU test of the data-parallel training step: two ranks (gloo, both on the one GPU of the test box) run runIter on different
shards with the bucketed gradient all-reduce hooked into the backward; afterwards the summed gradients and the updated
parameters must be identical on both ranks, and the all-reduced gradient must equal the sum of the two ranks' local
gradients.  (nccl = RCCL needs one GPU per rank; the collective path itself is backend-agnostic.)"""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    # RSIS_CONV_SPLITK=0: bit-reproducible forward, so that the two lr = 0 steps below see the same scores / the same assignment
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0",
                      RSIS_CONV_SPLITK="0")
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from helpers import mk_args
    from rsis_amd.modules import RSIS, FeatureExtractor
    from rsis_amd.optim import BucketedAllReduce
    from rsis_amd.synthetic import synthetic_batch
    from rsis_amd.train import build_optimizers, runIter
    from rsis_amd.utils.objectives import MaskedBCELoss, MaskedNLLLoss, softIoULoss
    torch.cuda.set_device(0)
    a = mk_args(hidden_size=32, maxseqlen=3, optim="adam", optim_cnn="adam", lr=1e-3, lr_cnn=1e-4, weight_decay=0.0, weight_decay_cnn=0.0)
    a.gt_maxseqlen, a.num_classes = 5, 7
    a.use_class_loss = a.use_stop_loss = a.update_encoder = True
    torch.manual_seed(0)                                   # identical replicas
    enc, dec = FeatureExtractor(a).cuda(), RSIS(a).cuda()
    enc_opt, dec_opt = build_optimizers(a, enc, dec)
    crits = [softIoULoss(), MaskedNLLLoss(None), MaskedBCELoss(0.5)]
    batch = synthetic_batch(10 + rank, 2, 64, 64, 5, 3, 7, "cuda")          # a different shard per rank
    # local gradient of this rank (no reducer, no optimizer effect on the comparison: snapshot params first)
    p0 = torch.cat([dec_opt.group.flat_p, enc_opt.group.flat_p]).clone()
    # two steps with lr = 0: the local gradient first (the reducer's hooks are live from its construction on, so it is built
    # afterwards), then the all-reduced one, to be compared with the sum of the local gradients of both ranks
    for g in (enc_opt.group, dec_opt.group):
        g.lr_saved, g.lr = g.lr, 0.0
    runIter(a, enc, dec, *batch, crits, [enc_opt, dec_opt], mode="train", reducer=None, sync_losses=True)
    local = torch.cat([dec_opt.group.flat_g, enc_opt.group.flat_g]).clone()
    red = BucketedAllReduce([dec_opt.group, enc_opt.group], bucket_bytes=8 << 20)
    assert red.active and len(red.buckets) >= 3
    runIter(a, enc, dec, *batch, crits, [enc_opt, dec_opt], mode="train", reducer=red, sync_losses=True)
    summed = torch.cat([dec_opt.group.flat_g, enc_opt.group.flat_g]).clone()
    both = [torch.empty_like(local).cpu() for _ in range(world)]
    dist.all_gather(both, local.cpu())
    want = both[0] + both[1]
    err = float((summed.cpu() - want).abs().max() / want.abs().max().clamp_min(1e-12))
    assert float((torch.cat([dec_opt.group.flat_p, enc_opt.group.flat_p]) - p0).abs().max()) == 0.0      # lr 0: parameters untouched
    # now a real step: parameters must stay identical across ranks
    for g in (enc_opt.group, dec_opt.group):      # (the lr = 0 step without a reducer fed rank-local gradients to the Adam moments)
        g.lr = g.lr_saved
        g.exp_avg.zero_()
        g.exp_avg_sq.zero_()
        g.step_count = 0
    runIter(a, enc, dec, *batch, crits, [enc_opt, dec_opt], mode="train", reducer=red, sync_losses=True)
    p1 = torch.cat([dec_opt.group.flat_p, enc_opt.group.flat_p]).cpu()
    # (small summaries only: tensors that travel through an mp.Queue live in shared memory of a process that is about to exit)
    def digest(t):
        t = t.double().cpu()
        return (float(t.sum()), float(t.abs().sum()), t[::997].numpy().copy())
    q.put((rank, err, digest(summed), digest(p1), float((p1 - p0.cpu()).abs().max())))
    dist.barrier()
    dist.destroy_process_group()


def test_runiter_world2_gradients_and_parameters_agree():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    import queue
    import time
    res, t0 = [], time.time()
    while len(res) < 2:                                    # fail fast when a rank dies instead of waiting for the timeout
        try:
            res.append(q.get(timeout=5))
        except queue.Empty:
            assert all(p.is_alive() or p.exitcode == 0 for p in procs), "a rank crashed: %s" % [p.exitcode for p in procs]
            assert time.time() - t0 < 300, "timed out"
    res.sort(key=lambda t: t[0])
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    # the two launches of the same shard differ only by the order of the fp32 atomics in the backward
    assert res[0][1] < 1e-3 and res[1][1] < 1e-3, (res[0][1], res[1][1])
    import numpy as np
    for i, what in ((2, "all-reduced gradients differ between ranks"), (3, "parameters diverged between ranks")):
        a, b = res[0][i], res[1][i]
        assert a[0] == b[0] and a[1] == b[1] and np.array_equal(a[2], b[2]), what
    assert res[0][4] > 0, "the optimizer step did not change the parameters"


def _worker_split_graph(rank, world, port, q):
    """the cut-graph schedules of train.GraphedStep (graph A | all-reduce(decoder group) || graph B1 | all-reduce(layers 3-4) ||
    graph B2 | all-reduce(rest) | graph C, with 2 / 1 / 0 cuts) and the staged exchange of an eager step against the hook-driven
    bucketed exchange (optim.BucketedAllReduce), both ranks on the one GPU over gloo, library in its deterministic mode: all of
    them are the same arithmetic, so after three steps from identical replicas on different shards every parameter must be
    BIT-identical between them -- and between the ranks"""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0",
                      RSIS_DETERMINISTIC="1")
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import copy
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from helpers import mk_args
    from rsis_amd import ops
    from rsis_amd.modules import RSIS, FeatureExtractor
    from rsis_amd.optim import BucketedAllReduce
    from rsis_amd.synthetic import synthetic_batch
    from rsis_amd.train import GraphedStep, build_optimizers, runIter, steps_to_run
    from rsis_amd.utils.objectives import MaskedBCELoss, MaskedNLLLoss, softIoULoss
    torch.cuda.set_device(0)
    assert ops.is_deterministic()
    a = mk_args(hidden_size=32, maxseqlen=3, optim="adam", optim_cnn="adam", lr=1e-3, lr_cnn=1e-4, weight_decay=0.0, weight_decay_cnn=0.0)
    a.gt_maxseqlen, a.num_classes = 5, 7
    torch.manual_seed(0)                                   # identical replicas
    enc0, dec0 = FeatureExtractor(a).cuda(), RSIS(a).cuda()
    crits = [softIoULoss(), MaskedNLLLoss(None), MaskedBCELoss(0.5)]
    batch = synthetic_batch(10 + rank, 2, 64, 64, 5, 3, 7, "cuda")          # a different shard per rank
    t_run = steps_to_run(a, batch[3])
    finals, losses = [], []
    # legs: the hook-driven bucketed exchange (eager) | the staged exchange of an eager step (runIter's default) | the cut hipGraphs
    for graphed, cuts, staged in ((False, 0, False), (False, 2, True), (True, 2, True), (True, 1, True), (True, 0, True)):
        enc, dec = copy.deepcopy(enc0), copy.deepcopy(dec0)
        enc_opt, dec_opt = build_optimizers(a, enc, dec)
        red = BucketedAllReduce([dec_opt.group, enc_opt.group], bucket_bytes=8 << 20)
        red.staged = staged
        g = GraphedStep(a, enc, dec, crits, [enc_opt, dec_opt], red, warm=1, cuts=cuts) if graphed else None
        for _ in range(3):
            if graphed:
                out = g(batch, t_run)
            else:
                out = runIter(a, enc, dec, *batch, crits, [enc_opt, dec_opt], mode="train", reducer=red, sync_losses=False, t_run=t_run,
                              want_outs=False)
        torch.cuda.synchronize()
        losses.append(float(out[0][0]))
        if graphed:
            assert g.graph is not None and len(g.graphs) == cuts + 1 and g.graph_update is not None, "capture failed: %s" % g.failed
            assert dec_opt.group.steps[0] == 3 and enc_opt.group.steps[0] == 3
        finals.append((torch.cat([dec_opt.group.flat_p, enc_opt.group.flat_p]).clone(),
                       torch.cat([dec_opt.group.flat_g, enc_opt.group.flat_g]).clone()))
        if graphed:
            g.release()
        for h in red._hooks:
            h.remove()
    (pe, ge), (pg, gg) = finals[0], finals[1]
    dp = max(float((pe - f[0]).abs().max()) for f in finals[1:])          # every cut schedule against the eager bucketed one
    dg = max(float((ge - f[1]).abs().max()) for f in finals[1:])
    moved = float((pe[:1000] - torch.cat([p.detach().reshape(-1) for p in dec0.parameters()])[:1000]).abs().max())
    assert moved > 0, "the optimizer steps did not change the parameters"
    t = pg.double().cpu()
    q.put((rank, dp, dg, float(t.sum()), float(t.abs().sum()), t[::997].numpy().copy(), losses, moved))
    dist.barrier()
    dist.destroy_process_group()


def test_split_graph_schedule_equals_eager_bucketed_schedule_world2():
    import numpy as np
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_split_graph, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    import queue
    import time
    res, t0 = [], time.time()
    while len(res) < 2:
        try:
            res.append(q.get(timeout=5))
        except queue.Empty:
            assert all(p.is_alive() or p.exitcode == 0 for p in procs), "a rank crashed: %s" % [p.exitcode for p in procs]
            assert time.time() - t0 < 300, "timed out"
    res.sort(key=lambda t: t[0])
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    for r in res:
        assert r[1] == 0.0 and r[2] == 0.0, "rank %d: split-graph vs eager schedule: max |d param| %.3e, max |d grad| %.3e" % (r[0], r[1], r[2])
        assert all(v == r[6][0] for v in r[6]) and r[6][0] == r[6][0]
    a, b = res
    assert a[3] == b[3] and a[4] == b[4] and np.array_equal(a[5], b[5]), "parameters diverged between ranks"


def _worker_nccl(rank, world, port, q):
    """one process per GPU over RCCL: identical replicas, different shards, two real steps (eager, then hipGraph replay)"""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import torch.distributed as dist
    from helpers import mk_args
    from rsis_amd.modules import RSIS, FeatureExtractor
    from rsis_amd.optim import BucketedAllReduce
    from rsis_amd.synthetic import synthetic_batch
    from rsis_amd.train import GraphedStep, build_optimizers, init_distributed, runIter, steps_to_run
    from rsis_amd.utils.objectives import MaskedBCELoss, MaskedNLLLoss, softIoULoss
    init_distributed()                                      # backend nccl (= RCCL), device = LOCAL_RANK
    a = mk_args(hidden_size=32, maxseqlen=3, optim="adam", optim_cnn="adam", lr=1e-3, lr_cnn=1e-4, weight_decay=0.0, weight_decay_cnn=0.0)
    torch.manual_seed(0)
    enc, dec = FeatureExtractor(a).cuda(), RSIS(a).cuda()
    for p in list(enc.parameters()) + list(dec.parameters()) + list(enc.buffers()):
        dist.broadcast(p.data, 0)
    enc_opt, dec_opt = build_optimizers(a, enc, dec)
    red = BucketedAllReduce([dec_opt.group, enc_opt.group], bucket_bytes=8 << 20)
    crits = [softIoULoss(), MaskedNLLLoss(None), MaskedBCELoss(0.5)]
    batch = synthetic_batch(10 + rank, 2, 64, 64, 20, 3, 21, "cuda")
    t_run = steps_to_run(a, batch[3])
    runIter(a, enc, dec, *batch, crits, [enc_opt, dec_opt], mode="train", reducer=red, sync_losses=False, t_run=t_run)
    g = GraphedStep(a, enc, dec, crits, [enc_opt, dec_opt], red, warm=1)
    for _ in range(3):
        g(batch, t_run)
    torch.cuda.synchronize()
    p1 = torch.cat([dec_opt.group.flat_p, enc_opt.group.flat_p]).double().cpu()
    q.put((rank, float(p1.sum()), float(p1.abs().sum()), p1[::997].numpy().copy(), g.graph is not None, g.failed))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs 2 GPUs (RCCL: one GPU per rank)")
def test_runiter_world2_rccl_parameters_agree():
    """torch.distributed backend `nccl` (RCCL over xGMI), 2 ranks on 2 GPUs: after one eager and three further steps (captured as
    two hipGraphs around an eager all-reduce of the flat gradient buffers) the replicas hold identical parameters."""
    import numpy as np
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_nccl, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=600) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    a, b = res
    assert a[1] == b[1] and a[2] == b[2] and np.array_equal(a[3], b[3]), "parameters diverged between ranks"


def _worker_force_dist_graph(q):
    """world size 1 with the collective path forced on (RCCL all-reduce at one rank): the iteration replays as two hipGraphs with
    the eager all-reduce of the flat gradient buffers between them (RCCL refuses stream capture on this stack)"""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", RSIS_FORCE_DIST="1")
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from helpers import mk_args
    from rsis_amd.modules import RSIS, FeatureExtractor
    from rsis_amd.optim import BucketedAllReduce
    from rsis_amd.synthetic import synthetic_batch
    from rsis_amd.train import GraphedStep, build_optimizers, init_distributed, steps_to_run
    from rsis_amd.utils.objectives import MaskedBCELoss, MaskedNLLLoss, softIoULoss
    init_distributed()
    a = mk_args(hidden_size=32, maxseqlen=3, optim="adam", optim_cnn="adam", lr=1e-3, lr_cnn=1e-4, weight_decay=0.0, weight_decay_cnn=0.0)
    torch.manual_seed(0)
    enc, dec = FeatureExtractor(a).cuda(), RSIS(a).cuda()
    opts = list(build_optimizers(a, enc, dec))
    red = BucketedAllReduce([opts[1].group, opts[0].group], bucket_bytes=8 << 20, force=True)
    crits = [softIoULoss(), MaskedNLLLoss(None), MaskedBCELoss(0.5)]
    batch = synthetic_batch(10, 2, 64, 64, 20, 3, 21, "cuda")
    t_run = steps_to_run(a, batch[3])
    g = GraphedStep(a, enc, dec, crits, opts, red, warm=1)
    ls = [float(g(batch, t_run)[0][0]) for _ in range(6)]
    torch.cuda.synchronize()
    q.put((g.graph is not None, g.failed, ls))
    import torch.distributed as dist
    dist.destroy_process_group()


def test_graph_capture_with_rccl_collectives_world1():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_worker_force_dist_graph, args=(q,))
    p.start()
    captured, failed, ls = q.get(timeout=300)
    p.join(120)
    assert p.exitcode == 0
    assert captured, "capture refused: %s" % failed
    assert all(v == v for v in ls) and ls[-1] < ls[0], ls


def _worker_exchange_mode(q, mode):
    """world size 1 with the collective path forced on; mode "direct": RCCL bound directly, the all-reduces inside ONE captured graph
    (rsis_amd/comm.py); mode "cuts": torch.distributed collectives between cut graphs.  Deterministic library mode: the two schedules
    must produce the same bits."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", RSIS_FORCE_DIST="1",
                      RSIS_EXCHANGE=mode, RSIS_DETERMINISTIC="1")
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from helpers import mk_args
    from rsis_amd.comm import make_direct_reducer
    from rsis_amd.modules import RSIS, FeatureExtractor
    from rsis_amd.optim import BucketedAllReduce
    from rsis_amd.synthetic import synthetic_batch
    from rsis_amd.train import GraphedStep, build_optimizers, init_distributed, steps_to_run
    from rsis_amd.utils.objectives import MaskedBCELoss, MaskedNLLLoss, softIoULoss
    init_distributed()
    a = mk_args(hidden_size=32, maxseqlen=3, optim="adam", optim_cnn="adam", lr=1e-3, lr_cnn=1e-4, weight_decay=0.0, weight_decay_cnn=0.0)
    torch.manual_seed(0)
    enc, dec = FeatureExtractor(a).cuda(), RSIS(a).cuda()
    opts = list(build_optimizers(a, enc, dec))
    red = BucketedAllReduce([opts[1].group, opts[0].group], bucket_bytes=8 << 20, force=True)
    notes = []
    red.direct = make_direct_reducer(notes.append)
    crits = [softIoULoss(), MaskedNLLLoss(None), MaskedBCELoss(0.5)]
    batch = synthetic_batch(10, 2, 64, 64, 20, 3, 21, "cuda")
    t_run = steps_to_run(a, batch[3])
    g = GraphedStep(a, enc, dec, crits, opts, red, warm=1)
    ls = [float(g(batch, t_run)[0][0]) for _ in range(6)]
    torch.cuda.synchronize()
    flat = torch.cat([o.group.flat_p.detach().double().cpu() for o in opts])
    q.put((g.graph is not None, g.failed, ls, red.direct is not None, len(g.graphs or []), g.graph_update is not None, notes,
           flat.numpy().tobytes()))
    import torch.distributed as dist
    dist.destroy_process_group()


def test_direct_rccl_exchange_lives_inside_one_graph_and_equals_the_cut_schedule_world1():
    """train.GraphedStep with the gradient exchange bound directly to RCCL (rsis_comm_*): the iteration is ONE hipGraph whose
    all-reduces sit on a forked branch -- no cuts, no torch.distributed collective per step -- and, in the deterministic mode, six steps
    end at bit-identical parameters to the cut-graph schedule over torch.distributed.  (World size 1: the box has one GPU; the
    communicator, the capture of ncclAllReduce and the fork / join are the real thing.)"""
    ctx = mp.get_context("spawn")
    out = {}
    for mode in ("direct", "cuts"):
        q = ctx.Queue()
        p = ctx.Process(target=_worker_exchange_mode, args=(q, mode))
        p.start()
        out[mode] = q.get(timeout=300)
        p.join(120)
        assert p.exitcode == 0
    captured, failed, ls, direct, ngraphs, has_update_graph, notes, params = out["direct"]
    assert direct, "the direct communicator was not built: %s" % (notes,)
    assert captured and ngraphs == 1 and not has_update_graph, "capture: %s (failed=%s), %d graphs" % (captured, failed, ngraphs)
    assert all(v == v for v in ls) and ls[-1] < ls[0], ls
    c2 = out["cuts"]
    assert c2[0] and not c2[3] and c2[4] == 3 and c2[5], "the cut schedule did not run as three graphs + the update graph: %s" % (c2[:6],)
    assert ls == c2[2], "losses differ between the two schedules: %s vs %s" % (ls, c2[2])
    assert params == c2[7], "parameters after six steps differ between the direct and the cut schedule"


def _worker_probe(q, timeout):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", RSIS_FORCE_DIST="1")
    from rsis_amd import comm
    from rsis_amd.train import init_distributed
    init_distributed()
    notes = []
    q.put((comm.probe_direct(notes.append, timeout=timeout), notes))
    import torch.distributed as dist
    dist.destroy_process_group()


@pytest.mark.parametrize("timeout,want", [(240.0, True), (0.05, False)])
def test_out_of_process_probe_of_the_direct_exchange(timeout, want):
    """comm.probe_direct: the communicator + eager + captured all-reduce are proven in a child process per rank under a
    deadline before the training process commits to the direct schedule at world > 1; a child that does not finish in time is
    killed and the answer is False on every rank (here: a deadline no child can meet)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_worker_probe, args=(q, timeout))
    p.start()
    ok, notes = q.get(timeout=400)
    p.join(120)
    assert p.exitcode == 0
    assert ok is want, notes
    if not want:
        assert notes and "timeout" in notes[0], notes
