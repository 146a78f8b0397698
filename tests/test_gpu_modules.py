"""GPU parity tests, module level: the product modules (rsis_amd.modules, HIP path) against the committed golden
vectors (outputs of the unmodified reference) and against the CPU oracle run live on the same seeded inputs.
Bar (BASELINE.json north_star): per-timestep mask / class / stop outputs within 1e-4 (fp32)."""
import numpy as np
import pytest
import torch

from helpers import assert_close, gold, mk_args

pytestmark = pytest.mark.gpu


def _loaded_native():
    """fail loudly if the HIP library is not the thing that ran"""
    maps = open("/proc/self/maps").read()
    assert "librsis_hip.so" in maps, "librsis_hip.so is not loaded: the HIP path did not run"


@pytest.mark.parametrize("name", ["cell_small", "cell_l4like", "cell_wide"])
def test_cell_golden(name):
    from oracle import filler
    from oracle import rsis_oracle as O
    from rsis_amd.modules import ConvLSTMCell
    g = gold(name)
    B, Cin, hid, H, W = [int(v) for v in g["shape"]]
    ocell = filler.fill_module(O.ConvLSTMCell(mk_args(), Cin, hid, 3, 1), seed=11)
    cell = ConvLSTMCell(mk_args(), Cin, hid, 3, 1).cuda()
    cell.load_state_dict(ocell.state_dict())
    x0 = filler.tensor(11, name + ".x0", (B, Cin, H, W)).cuda().requires_grad_()
    x1 = filler.tensor(11, name + ".x1", (B, Cin, H, W)).cuda().requires_grad_()
    gh = filler.tensor(11, name + ".gh", (B, hid, H, W)).cuda()
    gc = filler.tensor(11, name + ".gc", (B, hid, H, W)).cuda()
    h0, c0 = cell(x0, None)
    h1, c1 = cell(x1, (h0, c0))
    ((h1 * gh).sum() + (c1 * gc).sum()).backward()
    got = dict(h0=h0, c0=c0, h1=h1, c1=c1, dx0=x0.grad, dx1=x1.grad, dW=cell.Gates.weight.grad, db=cell.Gates.bias.grad)
    for k, v in got.items():
        ref = g[k]
        assert_close(name + "." + k, v, ref, 1e-4 * max(1.0, float(np.abs(ref).max())), 1e-4)
    _loaded_native()


@pytest.mark.parametrize("name", ["dec_pow2", "dec_odd"])
def test_decoder_golden(name):
    from oracle import filler
    from oracle import rsis_oracle as O
    from rsis_amd.modules import RSIS
    g = gold(name)
    hs, B, T = int(g["hidden_size"]), int(g["B"]), int(g["T"])
    sizes = [tuple(int(v) for v in s) for s in g["sizes"]]
    odec = filler.fill_module(O.RSIS(mk_args(hidden_size=hs)), seed=22)
    dec = RSIS(mk_args(hidden_size=hs)).cuda()
    dec.load_state_dict(odec.state_dict())
    chans = [hs, hs, hs // 2, hs // 4, hs // 8]
    feats = [filler.tensor(22, "%s.f%d" % (name, i), (B, chans[i]) + sizes[i]).cuda().requires_grad_() for i in range(5)]
    hidden, loss = None, 0.0
    for t in range(T):
        m, c, s, hidden = dec(feats, hidden)
        assert_close("%s.mask%d" % (name, t), m, g["mask%d" % t], 1e-4)
        assert_close("%s.class%d" % (name, t), c, g["class%d" % t], 1e-5)
        assert_close("%s.stop%d" % (name, t), s, g["stop%d" % t], 1e-4)
        loss = loss + (m * filler.tensor(22, "%s.gm%d" % (name, t), m.shape).cuda()).sum() \
            + (c * filler.tensor(22, "%s.gc%d" % (name, t), c.shape).cuda()).sum() \
            + (s * filler.tensor(22, "%s.gs%d" % (name, t), s.shape).cuda()).sum()
    for i, (h, c) in enumerate(hidden):
        assert_close("%s.h%d" % (name, i), h, g["h%d" % i], 1e-4)
        assert_close("%s.c%d" % (name, i), c, g["c%d" % i], 1e-4)
    loss.backward()
    for i, f in enumerate(feats):
        ref = g["dfeat%d" % i]
        assert_close("%s.dfeat%d" % (name, i), f.grad, ref, 2e-4 * max(1.0, float(np.abs(ref).max())), 1e-4)
    for k, p in dec.named_parameters():
        ref = g["grad." + k]
        assert_close("%s.grad.%s" % (name, k), p.grad, ref, 2e-4 * max(1.0, float(np.abs(ref).max())), 1e-4)


def _noise_floor(o32, o64):
    """fp32 noise floor of the reference op graph itself: max |oracle fp32 - oracle fp64|.  Train-mode BatchNorm over
    the 8..32 samples/channel of a 64x64 fixture makes the deep pyramid levels ill-conditioned (the REFERENCE's own
    fp32 result moves by ~6e-3 at skip5 vs fp64), so those checks are: |hip - fp64 truth| <= max(1e-4, 4 x floor)."""
    return float((o32.double() - o64).abs().max())


@pytest.mark.parametrize("name,train", [("enc_eval_64", False), ("enc_eval_96x80", False), ("enc_train_64", True)])
def test_encoder_golden(name, train):
    from oracle import filler
    from oracle import rsis_oracle as O
    from rsis_amd.modules import FeatureExtractor
    g = gold(name)
    oenc = filler.fill_module(O.FeatureExtractor(mk_args()), seed=33)
    enc = FeatureExtractor(mk_args()).cuda()
    enc.load_state_dict(oenc.state_dict())
    enc.train(train)
    shape = tuple(int(v) for v in g["shape"])
    x = filler.tensor(33, name + ".x", shape)
    with torch.no_grad():
        fs = enc(x.cuda())
    if not train:
        for i, f in enumerate(fs):
            # eval-mode BN of these (filler-initialised, un-normalised) nets scales activations to O(100): the absolute part of
            # the tolerance is 1e-4 at unit scale plus 2 ulp-ish of the tensor's largest magnitude (an element that cancels
            # from +-200 down to 0.007 carries fp32 summation-order noise of ~2e-4)
            ref = g["skip%d" % (5 - i)]
            assert_close("%s.skip%d" % (name, 5 - i), f, ref, 1e-4 + 2e-6 * float(np.abs(ref).max()), 1e-4)
        return
    o64 = filler.fill_module(O.FeatureExtractor(mk_args()), seed=33).double().train()
    with torch.no_grad():
        f64 = o64(x.double())
    for i, f in enumerate(fs):
        ref32 = torch.from_numpy(g["skip%d" % (5 - i)])
        floor = _noise_floor(ref32, f64[i])
        assert_close("%s.skip%d (floor %.1e)" % (name, 5 - i, floor), f, f64[i], max(1e-4, 4 * floor), 1e-4)
    sd, sd64 = enc.state_dict(), o64.state_dict()
    for k in g.files:
        if k.startswith("sd."):
            floor = _noise_floor(torch.from_numpy(g[k]), sd64[k[3:]])
            assert_close(name + "." + k, sd[k[3:]], sd64[k[3:]], max(1e-5, 4 * floor), 1e-4)


def test_encoder_backward_vs_oracle():
    """train-mode encoder fwd + bwd on (4,3,128,128) against the oracle's autograd (fp64 truth, fp32 noise floor)."""
    from oracle import filler
    from oracle import rsis_oracle as O
    from rsis_amd.modules import FeatureExtractor
    shape = (4, 3, 128, 128)
    oenc = filler.fill_module(O.FeatureExtractor(mk_args()), seed=33).train()
    o64 = filler.fill_module(O.FeatureExtractor(mk_args()), seed=33).double().train()
    enc = FeatureExtractor(mk_args()).cuda()
    enc.load_state_dict(oenc.state_dict())
    enc.train()
    x = filler.tensor(33, "encbwd.x", shape)
    fs_o, fs_64, fs = oenc(x), o64(x.double()), enc(x.cuda())
    lo, l64, lg = 0.0, 0.0, 0.0
    for i, (a, a64, b) in enumerate(zip(fs_o, fs_64, fs)):
        gy = filler.tensor(33, "encbwd.g%d" % i, a.shape)
        lo = lo + (a * gy).sum()
        l64 = l64 + (a64 * gy.double()).sum()
        lg = lg + (b * gy.cuda()).sum()
        assert_close("skip%d" % (5 - i), b, a64, max(1e-4, 4 * _noise_floor(a, a64)), 1e-4)
    lo.backward()
    l64.backward()
    lg.backward()
    po, p64, pg = dict(oenc.named_parameters()), dict(o64.named_parameters()), dict(enc.named_parameters())
    # BPTT through ~100 train-mode BN layers is chaotic in fp32 (the oracle's own fp32 gradient differs from its fp64
    # gradient by O(10%) in layer4 on this fixture), so the wiring check is a relative-L2 one, scaled by that noise
    # floor; every op's backward is checked tightly on its own in test_gpu_ops.py.
    worst = 0.0
    for k in po:
        if k.startswith("base.fc"):
            assert pg[k].grad is None
            continue
        ref = p64[k].grad
        nrm = float(ref.norm()) + 1e-30
        floor = float((po[k].grad.double() - ref).norm()) / nrm
        err = float((pg[k].grad.cpu().double() - ref).norm()) / nrm
        assert err <= max(1e-3, 4 * floor), "grad.%s: rel-L2 err %.3e vs fp64, fp32 noise floor %.3e" % (k, err, floor)
        worst = max(worst, err / max(floor, 2.5e-4))
    print("encoder backward: worst rel-L2 err / max(noise floor, 2.5e-4) = %.2f" % worst)


def test_e2e_256_north_star():
    """BASELINE north star: per-timestep mask logits / class probs / stop logits within 1e-4 of the reference
    CPU path on identical inputs (256x256, B=2, T=10, ResNet-101, hidden 128)."""
    from oracle import filler
    from oracle import rsis_oracle as O
    from rsis_amd.modules import FeatureExtractor, RSIS
    from rsis_amd.test import test as hip_test
    g = gold("e2e_256")
    a = mk_args(maxseqlen=int(g["T"]))
    oenc = filler.fill_module(O.FeatureExtractor(a), seed=44)
    odec = filler.fill_module(O.RSIS(a), seed=45)
    enc, dec = FeatureExtractor(a).cuda(), RSIS(a).cuda()
    enc.load_state_dict(oenc.state_dict())
    dec.load_state_dict(odec.state_dict())
    x = filler.tensor(44, "e2e_256.x", tuple(int(v) for v in g["shape"])).cuda()
    sub = int(g["sub"])
    masks, classes, stops = hip_test(a, enc, dec, x)
    logits, _, stop_logits = hip_test(a, enc, dec, x, return_logits=True)
    assert_close("e2e.mask_logits", logits[:, :, ::sub, ::sub], g["mask_logits_sub"], 1e-4)
    assert_close("e2e.mask_probs", masks[:, :, ::sub, ::sub], g["mask_probs_sub"], 1e-4)
    assert_close("e2e.classes", classes, g["classes"], 1e-4)
    assert_close("e2e.stops", stops, g["stops"], 1e-4)
    # raw stop logit, the output with the largest fp32 noise of the reference itself on this fixture (its own |fp32 - fp64| is 4.1e-5,
    # tools/exp/e2e_fp64_floor.py): the north-star 1e-4 like every other output.  It used to sit at 1.1e-4: the error entered through
    # the side feature of the coarsest level, whose 18432-deep skip conv this library summed as ONE chain of MFMA accumulations
    # (tools/exp/stop_logit_diag.py); deep reductions and every inference call are now summed in segments (conv3x3_direct.hip).
    assert_close("e2e.stop_logits", stop_logits, g["stop_logits"], 1e-4)
    _loaded_native()


def test_reference_era_checkpoint_runs_the_north_star_case(tmp_path):
    """N4 end to end (reference utils/utils.py:12-32,89-111 + README.md:90-99): a checkpoint directory in the byte format of
    the reference's environment (python 2.7 / torch 0.2: oracle/legacy_ckpt.py -- `module.`-prefixed DataParallel keys,
    torch.cuda.FloatTensor objects, no num_batches_tracked, protocol-0 args.pkl) holding the e2e_256 weights is loaded through
    load_checkpoint -> check_parallel -> load_state_dict exactly as reference eval.py:43-66 does, the modules are built from the
    LOADED args namespace, and the HIP test() reproduces the reference's own outputs for those weights at the north-star 1e-4."""
    from oracle import filler, legacy_ckpt
    from oracle import rsis_oracle as O
    from rsis_amd.modules import FeatureExtractor, RSIS
    from rsis_amd.test import test as hip_test
    from rsis_amd.utils.utils import check_parallel, load_checkpoint
    g = gold("e2e_256")
    a = mk_args(maxseqlen=int(g["T"]))
    oenc = filler.fill_module(O.FeatureExtractor(a), seed=44)
    odec = filler.fill_module(O.RSIS(a), seed=45)
    ns = {k: v for k, v in vars(a).items() if isinstance(v, (bool, int, float, str, type(None)))}
    ns.update(model_name="published", epoch_resume=12, best_val_loss=np.float64(0.731))
    legacy_ckpt.write_reference_checkpoint(str(tmp_path), "published", oenc.state_dict(), odec.state_dict(), ns, parallel=True, cuda=True)
    del oenc, odec
    e_sd, d_sd, _eo, _do, largs = load_checkpoint("published", use_gpu=True, root=str(tmp_path))
    assert next(iter(e_sd)).startswith("module.") and largs.hidden_size == 128 and largs.maxseqlen == int(g["T"])
    e_sd, d_sd = check_parallel(e_sd, d_sd)
    enc, dec = FeatureExtractor(largs).cuda(), RSIS(largs).cuda()
    enc.load_state_dict(e_sd)
    dec.load_state_dict(d_sd)
    x = filler.tensor(44, "e2e_256.x", tuple(int(v) for v in g["shape"])).cuda()
    sub = int(g["sub"])
    masks, classes, stops = hip_test(largs, enc, dec, x)
    logits, _, stop_logits = hip_test(largs, enc, dec, x, return_logits=True)
    assert_close("ckpt.mask_logits", logits[:, :, ::sub, ::sub], g["mask_logits_sub"], 1e-4)
    assert_close("ckpt.mask_probs", masks[:, :, ::sub, ::sub], g["mask_probs_sub"], 1e-4)
    assert_close("ckpt.classes", classes, g["classes"], 1e-4)
    assert_close("ckpt.stops", stops, g["stops"], 1e-4)
    assert_close("ckpt.stop_logits", stop_logits, g["stop_logits"], 1e-4)
    _loaded_native()


def test_e2e_odd_size():
    from oracle import filler
    from oracle import rsis_oracle as O
    from rsis_amd.modules import FeatureExtractor, RSIS
    from rsis_amd.test import test as hip_test
    g = gold("e2e_200x264")
    a = mk_args(maxseqlen=int(g["T"]))
    oenc = filler.fill_module(O.FeatureExtractor(a), seed=44)
    odec = filler.fill_module(O.RSIS(a), seed=45)
    enc, dec = FeatureExtractor(a).cuda(), RSIS(a).cuda()
    enc.load_state_dict(oenc.state_dict())
    dec.load_state_dict(odec.state_dict())
    x = filler.tensor(44, "e2e_200x264.x", tuple(int(v) for v in g["shape"])).cuda()
    masks, classes, stops = hip_test(a, enc, dec, x)
    sub = int(g["sub"])
    assert_close("odd.mask_probs", masks[:, :, ::sub, ::sub], g["mask_probs_sub"], 1e-4)
    assert_close("odd.classes", classes, g["classes"], 1e-4)
    assert_close("odd.stops", stops, g["stops"], 1e-4)


def test_runiter_golden():
    """rsis_amd.train.runIter (eval-mode BN would not match train.py: both are train mode) vs the golden restated
    runIter on the reference modules: losses, scores-derived permutation, outputs (2,3,64,64), T=3."""
    from oracle import filler
    from oracle import rsis_oracle as O
    from rsis_amd.modules import FeatureExtractor, RSIS
    from rsis_amd.train import build_optimizers, runIter
    from rsis_amd.utils.objectives import MaskedBCELoss, MaskedNLLLoss, softIoULoss
    g = gold("runiter_64")
    B, H, W, T = 2, 64, 64, 3
    a = mk_args(maxseqlen=T, optim="adam", optim_cnn="adam", lr=0.0, lr_cnn=0.0, weight_decay=0.0, weight_decay_cnn=0.0)
    oenc = filler.fill_module(O.FeatureExtractor(a), seed=66)
    odec = filler.fill_module(O.RSIS(a), seed=67)
    enc, dec = FeatureExtractor(a).cuda(), RSIS(a).cuda()
    enc.load_state_dict(oenc.state_dict())
    dec.load_state_dict(odec.state_dict())
    x = filler.tensor(66, "runiter_64.x", (B, 3, H, W)).cuda()
    y_mask, y_class, sw_mask, sw_class = [t.cuda() for t in filler.synthetic_targets(66, B, H, W, gt_maxseqlen=20, n_inst=5)]
    enc_opt, dec_opt = build_optimizers(a, enc, dec)
    crits = [softIoULoss(), MaskedNLLLoss(None), MaskedBCELoss(a.stop_balance_weight)]
    losses, outs, perms = runIter(a, enc, dec, x, y_mask, y_class.clone(), sw_mask.double(), sw_class.double(), crits,
                                  [enc_opt, dec_opt], mode="train")
    # (2, 3, 64, 64) in train mode leaves 8 samples per channel at the deepest level: the class / stop heads, fed by those
    # features, are ill-conditioned here, so this small fixture pins only what is well-conditioned -- the matching permutation and
    # the soft-IoU loss of the full-resolution masks.  The complete training step (all four losses at 1e-4, every gradient by the
    # fp64-truth rule, one Adam step) is pinned on the larger fixture: tests/test_gpu_round2.py::test_trainstep_fp32_*.
    assert (perms[1].cpu().numpy() == g["y_class_perm"]).all()
    assert_close("loss_iou", losses[1], g["loss_mask_iou"], 1e-4)


def test_training_reduces_loss():
    """a few fused-Adam steps on one synthetic batch must reduce the loss (end-to-end sanity of bwd + optimizer)."""
    from rsis_amd.modules import FeatureExtractor, RSIS
    from rsis_amd.synthetic import synthetic_batch
    from rsis_amd.train import build_optimizers, runIter
    from rsis_amd.utils.objectives import MaskedBCELoss, MaskedNLLLoss, softIoULoss
    torch.manual_seed(0)
    a = mk_args(hidden_size=32, maxseqlen=3, optim="adam", optim_cnn="adam", lr=1e-3, lr_cnn=1e-6, weight_decay=1e-6,
                weight_decay_cnn=1e-6)
    enc, dec = FeatureExtractor(a).cuda(), RSIS(a).cuda()
    batch = synthetic_batch(5, 4, 64, 64, 20, 3, 21, "cuda")
    opts = list(build_optimizers(a, enc, dec))
    crits = [softIoULoss(), MaskedNLLLoss(None), MaskedBCELoss(0.5)]
    ls = [runIter(a, enc, dec, *batch, crits, opts, mode="train")[0][0] for _ in range(8)]
    assert all(v == v for v in ls), ls
    assert ls[-1] < ls[0], ls


@pytest.mark.parametrize("tcap", [10, 2])
def test_fused_decoder_equals_unfused(tcap):
    """hoisted + time-batched decoder path == plain per-cell path (same kernels, different schedule), incl. all grads;
    tcap=2 < T exercises the beyond-capacity fallback steps."""
    from oracle import filler
    from rsis_amd.modules import RSIS
    hs, B, T = 32, 2, 4
    sizes = [(3, 4), (5, 7), (10, 13), (19, 25), (37, 50)]
    chans = [hs, hs, hs // 2, hs // 4, hs // 8]
    a = mk_args(hidden_size=hs, maxseqlen=tcap)
    torch.manual_seed(1)
    ref = RSIS(a).cuda()
    ref.fused = False
    fus = RSIS(a).cuda()
    fus.load_state_dict(ref.state_dict())
    assert fus.fused
    res = []
    for dec in (ref, fus):
        feats = [filler.tensor(7, "fz.f%d" % i, (B, chans[i]) + sizes[i]).cuda().requires_grad_() for i in range(5)]
        hidden, loss, outs = None, 0.0, []
        for t in range(T):
            m, c, s, hidden = dec(feats, hidden)
            outs += [m, c, s]
            loss = loss + (m * filler.tensor(7, "fz.gm%d" % t, m.shape).cuda()).sum() + (c * c).sum() + s.sum()
        loss = loss + sum((h * h).mean() + c.mean() for h, c in hidden)
        loss.backward()
        res.append((outs, [f.grad for f in feats], {k: p.grad for k, p in dec.named_parameters()}))
    for i, (p, q) in enumerate(zip(res[0][0], res[1][0])):
        assert_close("out%d" % i, q, p, 2e-5, 1e-5)
    for i, (p, q) in enumerate(zip(res[0][1], res[1][1])):
        assert_close("dfeat%d" % i, q, p, 1e-4 * max(1.0, float(p.abs().max())), 1e-4)
    for k in res[0][2]:
        p, q = res[0][2][k], res[1][2][k]
        assert_close("grad." + k, q, p, 1e-4 * max(1.0, float(p.abs().max())), 1e-4)


@pytest.mark.parametrize("shape", ["odd", "pow2"])
def test_wavefront_sequence_equals_per_step_loop(shape):
    """RSIS.forward_sequence (the (level, timestep) wavefront with the gate kernels of a diagonal in ONE rsis_convlstm_fwd_batch call)
    against T calls of RSIS.forward (reference train.py:85-94): the same autograd nodes and the same kernel per cell, only the
    launch order and the grouping differ -- outputs, final states and every gradient must agree to fp32 summation-order noise
    (the grouped kernel may pick another pixel-tile shape than the single launch; the K order per output element is the same).
    `odd`: ragged maps (partial tiles at every level); `pow2`: 8..128-pixel maps at hidden 128, batch 4 (every level of the product's
    variant table, incl. the K-split 8x8 variant and the 32-wide tiles)."""
    from oracle import filler
    from rsis_amd.modules import RSIS
    if shape == "odd":
        hs, B, T, sizes = 32, 2, 4, [(3, 4), (5, 7), (10, 13), (19, 25), (37, 50)]
    else:
        hs, B, T, sizes = 128, 4, 3, [(8, 8), (16, 16), (32, 32), (64, 64), (128, 128)]
    chans = [hs, hs, hs // 2, hs // 4, hs // 8]
    a = mk_args(hidden_size=hs, maxseqlen=T)
    torch.manual_seed(1)
    dec = RSIS(a).cuda()
    res = []
    for seq in (False, True):
        dec.zero_grad()
        feats = [filler.tensor(7, "wf.f%d" % i, (B, chans[i]) + sizes[i]).cuda().requires_grad_() for i in range(5)]
        if seq:
            steps, hidden = dec.forward_sequence(feats, T)
        else:
            hidden, steps = None, []
            for t in range(T):
                m, c, s, hidden = dec(feats, hidden)
                steps.append((m, c, s))
        loss, outs = 0.0, []
        for t, (m, c, s) in enumerate(steps):
            outs += [m, c, s]
            loss = loss + (m * filler.tensor(7, "wf.gm%d" % t, m.shape).cuda()).sum() + (c * c).sum() + s.sum()
        loss = loss + sum((h * h).mean() + c.mean() for h, c in hidden)
        outs += [t_ for st in hidden for t_ in st]
        loss.backward()
        res.append(([o.detach().clone() for o in outs], [f.grad.clone() for f in feats], {k: p.grad.clone() for k, p in dec.named_parameters()}))
    for i, (p, q) in enumerate(zip(res[0][0], res[1][0])):
        assert_close("out%d" % i, q, p, 1e-6 * max(1.0, float(p.abs().max())), 1e-6)
    for i, (p, q) in enumerate(zip(res[0][1], res[1][1])):
        assert_close("dfeat%d" % i, q, p, 1e-4 * max(1.0, float(p.abs().max())), 1e-4)
    for k in res[0][2]:
        p, q = res[0][2][k], res[1][2][k]
        assert_close("grad." + k, q, p, 1e-4 * max(1.0, float(p.abs().max())), 1e-4)
    # inference (no tape, no saved gates): bit-for-bit the per-step loop's outputs
    with torch.no_grad():
        feats = [filler.tensor(7, "wf.f%d" % i, (B, chans[i]) + sizes[i]).cuda() for i in range(5)]
        steps, _hid = dec.forward_sequence(feats, T)
        hidden = None
        for t in range(T):
            m, c, s, hidden = dec(feats, hidden)
            assert_close("inf.mask%d" % t, steps[t][0], m, 1e-6 * max(1.0, float(m.abs().max())), 1e-6)
            assert_close("inf.class%d" % t, steps[t][1], c, 1e-6)


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
@pytest.mark.parametrize("shape", ["odd", "pow2"])
def test_pool_fused_into_gate_kernel_is_bit_identical(shape, dtype):
    """The side features of reference model.py:143 (global max-pool of every hidden state) taken inside the ConvLSTM gate kernels as
    packed (value, first pixel) keys merged by a 64-bit atomic max and decoded by the heads launch (rsis_lstm_job.side_key +
    rsis_heads_fwd_keys, decoder_fused.FUSED_POOL) against the separate rsis_global_maxpool_fwd launches: the maximum of a set does
    not depend on the order it is taken in and ties keep the first pixel in both, so outputs AND gradients (the arg-max routes the
    pooled gradient) must be bit-identical in the deterministic mode."""
    from oracle import filler
    from rsis_amd import decoder_fused, ops
    from rsis_amd.modules import RSIS
    if shape == "odd":
        hs, B, T, sizes = 32, 2, 4, [(3, 4), (5, 7), (10, 13), (19, 25), (37, 50)]
    else:
        hs, B, T, sizes = 128, 4, 3, [(8, 8), (16, 16), (32, 32), (64, 64), (128, 128)]
    chans = [hs, hs, hs // 2, hs // 4, hs // 8]
    a = mk_args(hidden_size=hs, maxseqlen=T, dtype=dtype)
    torch.manual_seed(1)
    dec = RSIS(a).cuda()
    res = []
    from rsis_amd import decoder_seq
    was, det, was_seq = decoder_fused.FUSED_POOL[0], ops.is_deterministic(), decoder_seq.ENABLED[0]
    ops.set_deterministic(True)
    decoder_seq.ENABLED[0] = False      # (both runs on decoder_fused.decoder_sequence: the one-node sequence of decoder_seq always pools in the gate kernels)
    try:
        for fused in (False, True):
            decoder_fused.FUSED_POOL[0] = fused
            dec.zero_grad()
            feats = [filler.tensor(7, "wf.f%d" % i, (B, chans[i]) + sizes[i]).cuda().requires_grad_() for i in range(5)]
            steps, hidden = dec.forward_sequence(feats, T)
            loss, outs = 0.0, []
            for t, (m, c, s) in enumerate(steps):
                outs += [m, c, s]
                loss = loss + (m * filler.tensor(7, "wf.gm%d" % t, m.shape).cuda()).sum() + (c * c).sum() * 50 + s.sum()
            loss = loss + sum((h * h).mean() + c.mean() for h, c in hidden)
            loss.backward()
            res.append(([o.detach().clone() for o in outs], [f.grad.clone() for f in feats], {k: p.grad.clone() for k, p in dec.named_parameters()}))
    finally:
        decoder_fused.FUSED_POOL[0] = was
        decoder_seq.ENABLED[0] = was_seq
        ops.set_deterministic(det)
    for i, (p, q) in enumerate(zip(res[0][0], res[1][0])):
        assert torch.equal(p, q), "out%d differs by %g" % (i, float((p - q).abs().max()))
    for i, (p, q) in enumerate(zip(res[0][1], res[1][1])):
        assert torch.equal(p, q), "dfeat%d differs by %g" % (i, float((p - q).abs().max()))
    for k in res[0][2]:
        p, q = res[0][2][k], res[1][2][k]
        assert torch.equal(p, q), "grad.%s differs by %g" % (k, float((p - q).abs().max()))
    assert float(res[0][2]["fc_class.weight"].abs().max()) > 0


@pytest.mark.parametrize("train_bn", [True, False])
def test_direct_grad_accumulation_equals_autograd(train_bn):
    """ops.DIRECT_GRAD (wgrad kernels accumulate straight into the zeroed flat .grad views) == plain autograd grads.
    train_bn=True: decoder + skip-conv parameters with train-mode BatchNorm (direct accumulation of d(gamma), d(beta)); the trunk
    is not compared there (its gradients pass through ~100 train-mode BN layers on a tiny fixture and are chaotic w.r.t. the fp32
    atomic order of the split-K convs -- two identical runs differ).  train_bn=False (running statistics: well conditioned):
    EVERY parameter gradient, trunk included, to 2e-3 of the gradient scale (it was a 20 % relative-L2 bound)."""
    from rsis_amd import ops
    from rsis_amd.modules import FeatureExtractor, RSIS
    from rsis_amd.optim import FlatGroup
    torch.manual_seed(3)
    a = mk_args(hidden_size=32, maxseqlen=2)
    enc, dec = FeatureExtractor(a).cuda().train(train_bn), RSIS(a).cuda().train()
    tight = list(dec.parameters()) + [p for k, p in enc.named_parameters() if not k.startswith("base.")]
    trunk = [p for k, p in enc.named_parameters() if k.startswith("base.") and not k.startswith("base.fc")]
    g_tight, g_trunk = FlatGroup(tight, lr=0.0), FlatGroup(trunk, lr=0.0)
    x = torch.randn(2, 3, 64, 64, device="cuda")
    flats = []
    for direct in (False, True):
        g_tight.zero_grad()
        g_trunk.zero_grad()
        prev, ops.DIRECT_GRAD[0] = ops.DIRECT_GRAD[0], direct
        try:
            feats = enc(x)
            hidden, loss = None, 0.0
            for _t in range(2):
                m, c, s, hidden = dec(feats, hidden)
                loss = loss + m.square().mean() + c.square().sum() + s.mean()
            loss.backward()
        finally:
            ops.DIRECT_GRAD[0] = prev
        flats.append((g_tight.flat_g.clone(), g_trunk.flat_g.clone()))
    scale = float(flats[0][0].abs().max())
    assert scale > 0
    assert_close("decoder + skip grads", flats[1][0], flats[0][0], 2e-4 * scale, 1e-3)
    if not train_bn:
        tscale = float(flats[0][1].abs().max())
        assert tscale > 0
        assert_close("trunk grads", flats[1][1], flats[0][1], 2e-3 * tscale, 1e-3)     # (fp32 atomics of the split-K sums: ~1e-3 of the scale)


def test_training_step_leaves_no_cyclic_garbage():
    """with Python's cyclic GC disabled, device memory must not grow from step to step (the decoder tape / autograd
    nodes used to form reference cycles that kept a whole iteration alive until a gen-2 collection)."""
    import gc
    from rsis_amd.modules import FeatureExtractor, RSIS
    from rsis_amd.synthetic import synthetic_batch
    from rsis_amd.train import build_optimizers, runIter
    from rsis_amd.utils.objectives import MaskedBCELoss, MaskedNLLLoss, softIoULoss
    a = mk_args(hidden_size=32, maxseqlen=3, optim="adam", optim_cnn="adam", lr=1e-3, lr_cnn=1e-6, weight_decay=1e-6,
                weight_decay_cnn=1e-6)
    enc, dec = FeatureExtractor(a).cuda(), RSIS(a).cuda()
    batch = synthetic_batch(5, 2, 64, 64, 20, 3, 21, "cuda")
    opts = list(build_optimizers(a, enc, dec))
    crits = [softIoULoss(), MaskedNLLLoss(None), MaskedBCELoss(0.5)]
    gc.collect()
    gc.disable()
    try:
        for _ in range(3):
            runIter(a, enc, dec, *batch, crits, opts, mode="train", sync_losses=False, t_run=3)
        torch.cuda.synchronize()
        base = torch.cuda.memory_allocated()
        for _ in range(5):
            runIter(a, enc, dec, *batch, crits, opts, mode="train", sync_losses=False, t_run=3)
        torch.cuda.synchronize()
        grown = torch.cuda.memory_allocated() - base
    finally:
        gc.enable()
    assert grown <= 1 << 20, "device memory grew by %.1f MB over 5 steps with the cyclic GC off" % (grown / 2 ** 20)


@pytest.mark.parametrize("skip_mode", ["sum", "mul", "none"])
def test_decoder_other_skip_modes_match_oracle(skip_mode):
    """model.py:151-158: the non-default skip connections (`-skip_mode sum|mul|none`) run the per-cell (unfused) decoder path;
    2 timesteps forward + backward against the oracle on the same weights"""
    from oracle import filler
    from oracle import rsis_oracle as O
    from rsis_amd.modules import RSIS
    hs, B = 32, 2
    sizes = [(2, 3), (4, 6), (8, 12), (16, 24), (32, 48)]
    a = mk_args(hidden_size=hs, skip_mode=skip_mode)
    odec = filler.fill_module(O.RSIS(a), seed=41).train()
    dec = RSIS(a).cuda().train()
    dec.load_state_dict(odec.state_dict())
    chans = [hs, hs, hs // 2, hs // 4, hs // 8]
    feats = [filler.tensor(41, "skipmode.f%d" % i, (B, chans[i]) + sizes[i]).requires_grad_() for i in range(5)]
    dfeats = [f.detach().cuda().requires_grad_() for f in feats]
    res = []
    for d, fs in ((odec, feats), (dec, dfeats)):
        hidden, loss, outs = None, 0.0, []
        for _t in range(2):
            m, c, s, hidden = d(fs, hidden)
            outs += [m, c, s]
            loss = loss + m.square().mean() + c.square().sum() + s.mean()
        loss.backward()
        res.append(outs)
    for i, (p, q) in enumerate(zip(*res)):
        assert_close("%s.out%d" % (skip_mode, i), q, p.detach(), 1e-4, 1e-4)
    for i, (f, g) in enumerate(zip(feats, dfeats)):
        if f.grad is None:                                   # skip_mode none: the skip features of levels 1-4 are unused
            assert g.grad is None or float(g.grad.abs().max()) == 0.0
            continue
        assert_close("%s.dfeat%d" % (skip_mode, i), g.grad, f.grad, 2e-4 * max(1.0, float(f.grad.abs().max())), 1e-4)
    for (k, p), (_k2, q) in zip(odec.named_parameters(), dec.named_parameters()):
        assert_close("%s.grad.%s" % (skip_mode, k), q.grad, p.grad, 2e-4 * max(1.0, float(p.grad.abs().max())), 1e-4)


@pytest.mark.gpu
@pytest.mark.parametrize("stride", [1, 2])
def test_gradslot_downsample_block_equals_autograd_sum(stride):
    """A downsample Bottleneck whose input also feeds an outside consumer: the gradient hand-overs (ops.GradSlot: outside
    gradient -> strided downsample dgrad accumulating in place -> conv1's dgrad epilogue) must give the same input and
    parameter gradients as letting autograd add the three contributions."""
    from rsis_amd import ops
    from rsis_amd.modules.vision import Bottleneck, HipBatchNorm2d, HipConv2d
    torch.manual_seed(5)
    ds = torch.nn.Sequential(HipConv2d(64, 128, 1, stride=stride, bias=False), HipBatchNorm2d(128))
    blk = Bottleneck(64, 32, stride, ds).cuda().train()
    x = torch.randn(4, 64, 16, 16, device="cuda", requires_grad=True)
    g1 = torch.randn(4, 128, 16 // stride, 16 // stride, device="cuda")
    g2 = torch.randn(4, 64, 16, 16, device="cuda")

    def run():
        x.grad = None
        blk.zero_grad()
        xa = x * 1.0                                   # non-leaf, like a trunk feature
        out = blk(xa)
        side = ops.grad_tap(xa, blk._slot_in)          # the skip connection leaving the trunk
        ((out * g1).sum() + (side * g2).sum()).backward()
        return [x.grad.clone()] + [p.grad.clone() for p in blk.parameters()]

    got = run()
    assert blk._slot.done and blk._slot.grad is None and blk._slot_in.grad is None, "every parked gradient must be consumed"
    park = ops.GradSlot.park
    try:
        ops.GradSlot.park = lambda self, g: False      # no hand-over: autograd adds
        want = run()
    finally:
        ops.GradSlot.park = park
    for i, (a, b) in enumerate(zip(got, want)):
        assert_close("grad %d" % i, a, b, 2e-5 * float(b.abs().max()) + 1e-7)


