"""GPU parity tests, op level: every C-ABI entry point (through rsis_amd.ops) against the CPU oracle's ops
(plain torch fp32 on the host) on seeded inputs.  fp32 tolerance: 1e-4 absolute on O(1) values unless stated."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from helpers import assert_close

pytestmark = pytest.mark.gpu


def _rng_t(seed, shape, scale=1.0):
    return torch.from_numpy(np.random.default_rng(seed).normal(0, scale, shape).astype(np.float32))


def _dev(t):
    return t.detach().cuda().requires_grad_(t.requires_grad) if t is not None else None


@pytest.fixture(autouse=True)
def _reset_tile():
    from rsis_amd import ops
    ops.FORCE_TILE[0] = 0
    yield
    ops.FORCE_TILE[0] = 0


CONV_CASES = [
    # (B, [Cin segs], H, W, Cout, ks, stride, pad, bias)
    (2, [8], 9, 11, 16, 3, 1, 1, True),
    (2, [3], 32, 40, 64, 7, 2, 3, False),       # stem
    (3, [64], 16, 16, 256, 1, 1, 0, False),     # bottleneck 1x1
    (2, [64], 17, 15, 64, 3, 2, 1, False),      # strided 3x3 (odd size)
    (2, [256], 8, 8, 512, 1, 2, 0, False),      # downsample 1x1 s2
    (2, [16, 16], 12, 20, 32, 3, 1, 1, True),   # concat by pointer
    (2, [8], 20, 24, 1, 3, 1, 1, True),         # conv_out (Cout = 1: bandwidth kernel)
    (3, [16], 9, 13, 1, 3, 1, 1, True),         # conv_out, 16 channels, odd size
    (2, [8], 40, 136, 1, 3, 1, 1, True),        # conv_out over several LDS tiles, partial tiles in both directions
    (1, [16], 20, 72, 1, 3, 1, 1, False),       # conv_out, 16 channels (two channel passes per tile)
    (3, [4], 16, 64, 1, 3, 1, 1, True),         # conv_out, 4 channels, exactly one tile per image
    (1, [8], 12, 256, 1, 3, 1, 1, True),        # conv_out, full-width 8 x 256 tiles (the config-2 shape), partial tile rows
    (2, [8], 9, 160, 1, 3, 1, 1, False),        # conv_out, 256-wide tile on a 160-wide image
    (3, [16], 16, 16, 32, 3, 1, 1, False),      # tiled wgrad, 64-wide N tile + in-block K split (N = 144, Cout <= 32)
    (2, [16], 16, 32, 64, 3, 1, 1, True),       # tiled wgrad <64, 64> (N = 144)
    (2, [32], 16, 16, 128, 3, 1, 1, False),     # tiled wgrad <128, 64> (N = 288)
    (2, [64], 8, 32, 96, 1, 1, 0, False),       # tiled 1x1 wgrad <128, 64> (N = 64)
    (2, [16, 8], 16, 16, 32, 3, 1, 1, True),    # two sources: N = 144 (narrow) and N = 72 (wide) in the same conv
    (2, [6], 8, 8, 1, 3, 1, 1, False),          # Cout = 1 with an unsupported Cin -> MFMA path
    (1, [130], 7, 7, 129, 3, 1, 1, True),       # ragged channels
    (2, [2048], 4, 4, 128, 3, 1, 1, True),      # sk5-like deep K
    (2, [20, 12], 16, 24, 40, 3, 1, 1, True),   # tile-aligned map, 2 sources, ragged channels (LDS-DMA tiled wgrad, 64-row tile)
    (2, [72], 8, 16, 200, 3, 1, 1, False),      # tiled wgrad, 128-row tile, several co / n tiles
    (3, [40], 12, 16, 24, 1, 1, 0, False),      # tiled 1x1 wgrad (8x4 tiles), ragged channels
    (2, [24], 32, 48, 40, 3, 2, 1, True),       # strided 3x3, even size, several tiles (parity-class dgrad)
    (1, [16], 9, 64, 72, 3, 2, 1, False),       # strided 3x3, odd height, wide map
    (2, [128], 16, 16, 96, 3, 2, 1, False),     # strided 3x3 forward on the direct kernel, 8x8 output (64-row variant)
    (3, [20, 12], 20, 40, 48, 3, 2, 1, True),   # strided 3x3 forward, two sources with channel tails, partial tiles
]


@pytest.mark.parametrize("case", CONV_CASES)
@pytest.mark.parametrize("tile", [0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 13, 14])
def test_conv2d_fwd_bwd(case, tile):
    from rsis_amd import ops
    B, segs, H, W, Cout, ks, stride, pad, has_bias = case
    if tile in (7, 8, 9) and not (ks == 3 and stride == 1 and pad == 1):
        pytest.skip("variants 7-9 are 512-thread blocks of the direct 3x3 / stride 1 kernel")
    ops.FORCE_TILE[0] = tile
    Ctot = sum(segs)
    xs = [_rng_t(10 + i, (B, c, H, W)).requires_grad_() for i, c in enumerate(segs)]
    w = _rng_t(20, (Cout, Ctot, ks, ks), 1.0 / np.sqrt(Ctot * ks * ks)).requires_grad_()
    b = _rng_t(21, (Cout,)).requires_grad_() if has_bias else None
    ref = F.conv2d(torch.cat(xs, 1), w, b, stride=stride, padding=pad)
    gy = _rng_t(22, tuple(ref.shape))
    ref.backward(gy)
    xd = [_dev(x.detach().clone().requires_grad_()) for x in xs]
    wd = _dev(w.detach().clone().requires_grad_())
    bd = _dev(b.detach().clone().requires_grad_()) if has_bias else None
    pack = ops.PackedConv(ks, segs, stride=stride, pad=pad)
    out = ops.conv2d(xd, wd, bd, stride, pad, pack)
    out.backward(gy.cuda())
    torch.cuda.synchronize()
    tol = 2e-5 * np.sqrt(Ctot * ks * ks) + 1e-5
    assert_close("fwd", out, ref, tol, 1e-5)
    for i, x in enumerate(xs):
        assert_close("dx%d" % i, xd[i].grad, x.grad, 2e-5 * np.sqrt(Cout * ks * ks) + 1e-5, 1e-5)
    assert_close("dW", wd.grad, w.grad, 1e-4 * max(1.0, float(w.grad.abs().max())), 1e-5)
    if has_bias:
        assert_close("db", bd.grad, b.grad, 1e-4 * max(1.0, float(b.grad.abs().max())), 1e-5)


LSTM_CASES = [
    # (B, [x segs], hid, H, W)
    (2, [8], 4, 5, 7),
    (2, [16, 16], 8, 16, 16),       # L4-like
    (3, [24], 16, 9, 12),
    (2, [64, 64], 32, 8, 8),        # L2-like
    (1, [128], 128, 4, 4),          # L0-like
    (2, [6, 5], 3, 6, 5),           # ragged
]


@pytest.mark.parametrize("case", LSTM_CASES)
@pytest.mark.parametrize("tile", [0, 1, 2, 3, 4, 5, 6, 7, 8, 9])
def test_convlstm_fwd_bwd(case, tile):
    from oracle import rsis_oracle as O
    from rsis_amd import ops
    from rsis_amd.modules.clstm import ConvLSTMCell
    from helpers import mk_args
    B, segs, hid, H, W = case
    ops.FORCE_TILE[0] = tile
    Cin = sum(segs)
    ocell = O.ConvLSTMCell(mk_args(), Cin, hid, 3, 1)
    with torch.no_grad():
        ocell.Gates.weight.copy_(_rng_t(1, tuple(ocell.Gates.weight.shape), 2.0 / np.sqrt(9 * (Cin + hid))))
        ocell.Gates.bias.copy_(_rng_t(2, (4 * hid,), 0.2))
    cell = ConvLSTMCell(mk_args(), Cin, hid, 3, 1).cuda()
    cell.load_state_dict(ocell.state_dict())
    x0 = [_rng_t(30 + i, (B, c, H, W)).requires_grad_() for i, c in enumerate(segs)]
    x1 = [_rng_t(40 + i, (B, c, H, W)).requires_grad_() for i, c in enumerate(segs)]
    gh, gc = _rng_t(50, (B, hid, H, W)), _rng_t(51, (B, hid, H, W))
    h0, c0 = ocell(torch.cat(x0, 1), None)
    h1, c1 = ocell(torch.cat(x1, 1), (h0, c0))
    ((h1 * gh).sum() + (c1 * gc).sum() + (h0 * gc).sum()).backward()
    x0d = [_dev(t.detach().clone().requires_grad_()) for t in x0]
    x1d = [_dev(t.detach().clone().requires_grad_()) for t in x1]
    h0d, c0d = cell.forward_multi(x0d, None)
    h1d, c1d = cell.forward_multi(x1d, (h0d, c0d))
    ((h1d * gh.cuda()).sum() + (c1d * gc.cuda()).sum() + (h0d * gc.cuda()).sum()).backward()
    torch.cuda.synchronize()
    for n, a, b in (("h0", h0d, h0), ("c0", c0d, c0), ("h1", h1d, h1), ("c1", c1d, c1)):
        assert_close(n, a, b, 2e-5, 1e-5)
    for i in range(len(segs)):
        assert_close("dx0_%d" % i, x0d[i].grad, x0[i].grad, 5e-5, 1e-4)
        assert_close("dx1_%d" % i, x1d[i].grad, x1[i].grad, 5e-5, 1e-4)
    gw = ocell.Gates.weight.grad
    assert_close("dW", cell.Gates.weight.grad, gw, 1e-4 * max(1.0, float(gw.abs().max())), 1e-4)
    gb = ocell.Gates.bias.grad
    assert_close("db", cell.Gates.bias.grad, gb, 1e-4 * max(1.0, float(gb.abs().max())), 1e-4)


@pytest.mark.parametrize("shape,size", [((2, 3, 4, 5), (7, 9)), ((2, 8, 8, 8), (16, 16)), ((1, 2, 13, 25), (25, 50)),
                                        ((2, 1, 5, 7), (5, 7)), ((2, 4, 1, 1), (3, 3)), ((2, 1, 64, 64), (100, 132)),
                                        ((2, 3, 64, 64), (128, 128)), ((1, 2, 40, 72), (80, 144)), ((1, 2, 128, 128), (256, 256)),
                                        ((2, 2, 16, 16), (32, 32)), ((1, 2, 33, 20), (66, 40))])
def test_upsample(shape, size):
    from rsis_amd import ops
    x = _rng_t(3, shape).requires_grad_()
    ref = F.interpolate(x, size=size, mode="bilinear", align_corners=True)
    gy = _rng_t(4, tuple(ref.shape))
    ref.backward(gy)
    xd = _dev(x.detach().clone().requires_grad_())
    y = ops.upsample_bilinear_ac(xd, size)
    y.backward(gy.cuda())
    # 1 ulp of the fp32 source coordinate (|coord| up to ~130 here -> 1.5e-5) times the local slope of O(1) data
    assert_close("fwd", y, ref, 3e-5)
    assert_close("bwd", xd.grad, x.grad, 4e-5, 1e-5)


@pytest.mark.parametrize("shape", [(2, 5, 7, 9), (3, 8, 16, 16), (1, 3, 1, 1), (2, 2, 40, 33)])
def test_global_maxpool(shape):
    from rsis_amd import ops
    x = _rng_t(5, shape).requires_grad_()
    ref = F.max_pool2d(x, kernel_size=shape[2:])
    gy = _rng_t(6, tuple(ref.shape))
    ref.backward(gy)
    xd = _dev(x.detach().clone().requires_grad_())
    y = ops.global_maxpool(xd)
    y.backward(gy.cuda())
    assert_close("fwd", y, ref, 0)
    assert_close("bwd", xd.grad, x.grad, 0)


@pytest.mark.parametrize("shape", [(2, 4, 8, 8), (3, 5, 9, 11), (1, 2, 16, 6), (2, 3, 7, 7), (2, 8, 56, 56), (1, 3, 13, 30), (2, 16, 64, 64)])
@pytest.mark.parametrize("stride", [2, 3])
def test_subsample2d(shape, stride):
    """rsis_subsample2d: x[:, :, ::s, ::s] as a dense map (the input of the strided 1x1 downsample convs), bit-exact copy; even /
    odd / ragged widths (vector and scalar paths)"""
    from rsis_amd._lib import check, lib, ptr, stream
    x = _dev(_rng_t(70, shape))
    B, C, H, W = shape
    want = x[:, :, ::stride, ::stride].contiguous()
    y = torch.full_like(want, float("nan"))
    check(lib().rsis_subsample2d(ptr(x), ptr(y), B * C, H, W, stride, stream()), "rsis_subsample2d")
    assert torch.equal(y, want)


@pytest.mark.parametrize("shape", [(2, 4, 8, 8), (3, 5, 9, 11), (1, 2, 16, 6), (2, 3, 7, 7), (2, 3, 12, 20), (1, 2, 10, 4), (2, 8, 32, 64)])
@pytest.mark.parametrize("ties", [False, True])
def test_maxpool3x3s2(shape, ties):
    from rsis_amd import ops
    x = _rng_t(7, shape)
    if ties:                         # post-ReLU maps: many equal zeros, the FIRST maximum of a window takes the gradient
        x = torch.clamp(x, min=0.3)
    x = x.requires_grad_()
    ref = F.max_pool2d(x, 3, 2, 1)
    gy = _rng_t(8, tuple(ref.shape))
    ref.backward(gy)
    xd = _dev(x.detach().clone().requires_grad_())
    y = ops.maxpool3x3s2(xd)
    y.backward(gy.cuda())
    assert_close("fwd", y, ref, 0)
    assert_close("bwd", xd.grad, x.grad, 1e-6)


@pytest.mark.parametrize("shape", [(4, 6, 5, 7), (2, 64, 16, 16), (3, 3, 33, 17), (8, 130, 4, 4),
                                   (32, 96, 16, 16), (16, 64, 32, 32), (32, 64, 32, 32), (2, 64, 64, 68)])   # register-resident / split paths
@pytest.mark.parametrize("train,relu,res", [(True, False, False), (True, True, False), (True, True, True), (True, False, True),
                                            (False, True, True), (False, False, False)])
def test_batchnorm(shape, train, relu, res):
    from rsis_amd import ops
    B, C, H, W = shape
    x = (_rng_t(9, shape, 2.0) + 0.5).requires_grad_()
    r = _rng_t(10, shape).requires_grad_() if res else None
    gamma = (_rng_t(11, (C,), 0.3) + 1.0).requires_grad_()
    beta = _rng_t(12, (C,), 0.3).requires_grad_()
    rm, rv = _rng_t(13, (C,), 0.1), _rng_t(14, (C,), 0.1).abs() + 0.5
    rm_ref, rv_ref = rm.clone(), rv.clone()
    ref = F.batch_norm(x, rm_ref, rv_ref, gamma, beta, training=train, momentum=0.1, eps=1e-5)
    if res:
        ref = ref + r
    if relu:
        ref = F.relu(ref)
    gy = _rng_t(15, shape)
    ref.backward(gy)
    xd = _dev(x.detach().clone().requires_grad_())
    rd = _dev(r.detach().clone().requires_grad_()) if res else None
    gd, bd = _dev(gamma.detach().clone().requires_grad_()), _dev(beta.detach().clone().requires_grad_())
    rmd, rvd = rm.cuda(), rv.cuda()
    y = ops.batchnorm(xd, gd, bd, rmd, rvd, train, relu=relu, res=rd)
    y.backward(gy.cuda())
    assert_close("fwd", y, ref, 2e-5, 1e-5)
    assert_close("running_mean", rmd, rm_ref, 1e-6, 1e-5)
    assert_close("running_var", rvd, rv_ref, 1e-6, 1e-5)
    assert_close("dx", xd.grad, x.grad, 5e-5, 1e-4)
    assert_close("dgamma", gd.grad, gamma.grad, 1e-4, 1e-4)
    assert_close("dbeta", bd.grad, beta.grad, 1e-4, 1e-4)
    if res:
        assert_close("dres", rd.grad, r.grad, 1e-6)


def test_adam_matches_torch():
    from rsis_amd import ops
    n = 10007
    p = _rng_t(16, (n,))
    ref_p = p.clone().requires_grad_()
    opt = torch.optim.Adam([ref_p], lr=1e-3, weight_decay=1e-6)
    pd, m, v = p.cuda(), torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda")
    for step in range(1, 4):
        g = _rng_t(17 + step, (n,))
        ref_p.grad = g.clone()
        opt.step()
        ops.adam_step_flat(pd, g.cuda(), m, v, 1e-3, 0.9, 0.999, 1e-8, 1e-6, step)
    assert_close("adam", pd, ref_p, 1e-6, 1e-5)


@pytest.mark.parametrize("B,G,T", [(3, 20, 10), (32, 20, 10), (5, 7, 7), (4, 64, 20), (2, 3, 1)])
def test_assign_min_cost_matches_scipy(B, G, T):
    """device Hungarian == scipy.optimize.linear_sum_assignment (what the oracle uses for munkres) on generic costs"""
    from scipy.optimize import linear_sum_assignment
    from rsis_amd import ops
    rng = np.random.default_rng(B * 1000 + G * 10 + T)
    scores = rng.uniform(0, 1, (B, G, T)).astype(np.float32)
    perm = ops.assign_min_cost(torch.from_numpy(scores).cuda()).cpu().numpy()
    for b in range(B):
        r, c = linear_sum_assignment(scores[b].astype(np.float64))
        want = np.zeros(G, dtype=np.int64)
        want[c] = r
        assert (perm[b] == want).all(), (b, perm[b], want)


def test_assign_min_cost_with_masked_ties_and_golden():
    """the reference's score structure (invalid pairs = 10 -> ties among unused slots): the assignment must have the
    optimal total cost and agree with the golden permutation wherever the loss looks (valid predictions)."""
    from scipy.optimize import linear_sum_assignment
    from helpers import gold
    from rsis_amd import ops
    g = gold("losses")
    scores = np.random.default_rng(56).uniform(0, 1, (3, 20, 10)).astype(np.float32)
    perm = ops.assign_min_cost(torch.from_numpy(scores).cuda()).cpu().numpy()
    assert (perm == g["match_perm"]).all()
    B, G, T, n_inst = 4, 20, 10, 6
    sc = np.random.default_rng(1).uniform(0, 1, (B, G, T)).astype(np.float32)
    sw = np.zeros(G, np.float32)
    sw[:n_inst] = 1
    valid = sw[None, :, None] * sw[None, None, :T]
    sc = sc * valid + (1 - valid) * 10
    perm = ops.assign_min_cost(torch.from_numpy(sc).cuda()).cpu().numpy()
    for b in range(B):
        r, c = linear_sum_assignment(sc[b].astype(np.float64))
        assert abs(sc[b][perm[b, :T], np.arange(T)].sum() - sc[b][r, c].sum()) < 1e-4
        assert len(set(perm[b, :T].tolist())) == T
        want = np.zeros(G, dtype=np.int64)
        want[c] = r
        assert (perm[b, :n_inst] == want[:n_inst]).all()


@pytest.mark.gpu
@pytest.mark.parametrize("B,T,G,N", [(2, 3, 5, 64), (3, 10, 20, 4096), (1, 31, 31, 520), (2, 1, 1, 8), (2, 7, 12, 65536)])
def test_softiou_sums_and_matched_loss(B, T, G, N):
    """fused soft-IoU (rsis_softiou_sums / rsis_softiou_bwd) against the oracle's softIoU (hungarian.py:62-89) on every
    pair, and against autograd of the oracle's matched loss"""
    from rsis_amd import ops
    from oracle import rsis_oracle as O
    rng = np.random.default_rng(5)
    logits = torch.from_numpy(rng.normal(0, 2.0, (B, T, N)).astype(np.float32))
    y = torch.from_numpy((rng.random((B, G, N)) < 0.3).astype(np.float32))
    y[:, -1] = 0                                                     # an empty ground-truth slot
    S = ops.softiou_sums(logits.cuda(), y.cuda())
    cost = ops.softiou_cost_matrix(S).cpu()                          # (B, G, T)
    ref = torch.stack([torch.stack([O.softIoU(y[:, g], logits[:, t]).reshape(B) for t in range(T)], 1) for g in range(G)], 1)
    assert_close("cost", cost, ref, 2e-6, 1e-5)
    perm = torch.stack([torch.from_numpy(rng.permutation(G)) for _ in range(B)]).long()
    lg = logits.clone().requires_grad_()
    ref_cost = O.softIoU(torch.gather(y, 1, perm[:, :T].unsqueeze(-1).expand(-1, -1, N)).reshape(-1, N), lg.reshape(-1, N)).reshape(B, T)
    w = torch.from_numpy(rng.normal(0, 1, (B, T)).astype(np.float32))
    (ref_cost * w).sum().backward()
    ld = logits.cuda().requires_grad_()
    got = ops.softiou_matched(ld, y.cuda(), perm.cuda(), S)
    (got * w.cuda()).sum().backward()
    assert_close("matched", got, ref_cost, 2e-6, 1e-5)
    assert_close("dlogits", ld.grad, lg.grad, 1e-6 * max(1.0, float(lg.grad.abs().max()) * 1e3), 1e-4)


@pytest.mark.gpu
def test_repack_all_equals_lazy_packs():
    """rsis_conv_pack_batch (one launch for every packed copy) must write exactly what the per-conv pack entry points write"""
    from rsis_amd import ops
    cfgs = [  # (Cout, Ctot, ks, stride, pad, segs, offs, lstm_hid)
        (64, 48, 3, 1, 1, [16, 32], None, 0), (256, 64, 1, 1, 0, [64], None, 0), (64, 3, 7, 2, 3, [3], None, 0),
        (40, 24, 3, 2, 1, [24], None, 0), (128, 56, 3, 1, 1, [24, 8], [0, 48], 32), (130, 129, 3, 1, 1, [129], None, 0),
        (512, 256, 1, 2, 0, [256], None, 0)]
    rng = np.random.default_rng(17)
    packs, weights, biases = [], [], []
    for Cout, Ctot, ks, stride, pad, segs, offs, hid in cfgs:
        w = torch.from_numpy(rng.normal(0, 1, (Cout, Ctot, ks, ks)).astype(np.float32)).cuda()
        b = torch.from_numpy(rng.normal(0, 1, (Cout,)).astype(np.float32)).cuda() if hid else None
        p = ops.PackedConv(ks, segs, lstm_hid=hid, stride=stride, pad=pad, offs=offs)
        p.fwd(w, b)
        p.dgrad(w)
        packs.append(p); weights.append(w); biases.append(b)
    for w in weights:
        w.mul_(-1.5).add_(0.25)                      # "optimizer step"
    ops.bump_weight_epoch()
    ops.repack_all()
    torch.cuda.synchronize()
    for (Cout, Ctot, ks, stride, pad, segs, offs, hid), p, w, b in zip(cfgs, packs, weights, biases):
        key_f, key_d = p._key_f, p._key_d
        got_f, got_d = p.fwd(w, b).clone(), p.dgrad(w).clone()
        assert p._key_f == key_f and p._key_d == key_d, "repack_all must leave the caches valid (no lazy repack afterwards)"
        q = ops.PackedConv(ks, segs, lstm_hid=hid, stride=stride, pad=pad, offs=offs)
        assert torch.equal(got_f, q.fwd(w, b)) and torch.equal(got_d, q.dgrad(w)), (Cout, Ctot, ks, stride)
        if hid:
            assert torch.equal(p.bias_p, q.bias_p)


@pytest.mark.gpu
@pytest.mark.parametrize("B,Cs,ncls", [(2, [32, 16, 8, 4, 2], 7), (32, [128, 64, 32, 16, 8], 21), (3, [5], 2), (4, [100, 3, 9], 64),
                                       (320, [128, 64, 32, 16, 8], 21), (97, [16, 8], 33)])      # (320: the T * B rows of the sequence node)
def test_heads_fwd_bwd(B, Cs, ncls):
    """fused class / stop heads (rsis_heads_fwd / _bwd) against model.py:169-182 in plain torch"""
    from rsis_amd import ops
    rng = np.random.default_rng(13)
    K = sum(Cs)
    sides = [torch.from_numpy(rng.normal(0, 1, (B, c, 1, 1)).astype(np.float32)).requires_grad_() for c in Cs]
    fc_c, fc_s = torch.nn.Linear(K, ncls), torch.nn.Linear(K, 1)
    side = torch.cat(sides, 1).squeeze(-1).squeeze(-1)
    ref_p, ref_s = torch.softmax(fc_c(side), dim=1), fc_s(side)
    gp = torch.from_numpy(rng.normal(0, 1, (B, ncls)).astype(np.float32))
    gs = torch.from_numpy(rng.normal(0, 1, (B, 1)).astype(np.float32))
    ((ref_p * gp).sum() + (ref_s * gs).sum()).backward()
    import copy
    dc, ds = copy.deepcopy(fc_c).cuda(), copy.deepcopy(fc_s).cuda()
    for m in (dc, ds):
        for p in m.parameters():
            p.grad = None
    sd = [_dev(s.detach().clone().requires_grad_()) for s in sides]
    p, s = ops.heads(sd, dc, ds)
    ((p * gp.cuda()).sum() + (s * gs.cuda()).sum()).backward()
    assert_close("probs", p, ref_p, 1e-6, 1e-5)
    assert_close("stop", s, ref_s, 2e-6, 1e-5)
    for a, b in zip(sd, sides):
        assert_close("dside", a.grad, b.grad, 2e-6, 1e-4)
    for a, b in ((dc.weight, fc_c.weight), (dc.bias, fc_c.bias), (ds.weight, fc_s.weight), (ds.bias, fc_s.bias)):
        assert_close("dparam", a.grad, b.grad, 5e-6, 1e-4)


@pytest.mark.gpu
@pytest.mark.parametrize("bw,use_w", [(0.5, False), (None, False), (0.3, True)])
def test_loss_tail_matches_oracle(bw, use_w):
    """fused loss tail (rsis_loss_tail) against the oracle's MaskedNLL / StableBalancedMaskedBCE + masked means (train.py:159-176)"""
    from rsis_amd import ops
    from oracle import rsis_oracle as O
    rng = np.random.default_rng(23)
    B, T, C = 6, 5, 9
    probs = torch.softmax(torch.from_numpy(rng.normal(0, 2, (B, T, C)).astype(np.float32)), -1)
    y = torch.from_numpy(rng.integers(0, C, (B, T)))
    stop = torch.from_numpy(rng.normal(0, 3, (B, T)).astype(np.float32))
    siou = torch.from_numpy(rng.random((B, T)).astype(np.float32))
    swm = torch.from_numpy((rng.random((B, T)) < 0.6).astype(np.float32)); swm[0, 0] = 1
    swc = torch.from_numpy((rng.random((B, T)) < 0.7).astype(np.float32)); swc[0, 0] = 1
    cw = torch.from_numpy(rng.random(C).astype(np.float32) + 0.5) if use_w else None
    w_iou, w_cls, w_stop = 1.0, 0.1, 0.5
    leaves = [t.clone().requires_grad_() for t in (probs, stop, siou)]
    nll = O.MaskedNLL(y.reshape(-1, 1), leaves[0].reshape(-1, C), cw)
    bce = O.StableBalancedMaskedBCE(swm, leaves[1], bw)
    mm = lambda c, w: torch.masked_select(c.reshape(-1), w.reshape(-1).bool()).mean()
    l_cls, l_iou, l_stop = mm(nll, swm), mm(leaves[2], swm), mm(bce, swc)
    ref = w_iou * l_iou + w_cls * l_cls + w_stop * l_stop
    (ref * 1.7).backward()
    dev = [t.detach().cuda().requires_grad_() for t in (probs, stop, siou)]
    total, parts = ops.loss_tail(dev[0], y.cuda(), dev[1], dev[2], swm.cuda(), swc.cuda(), cw.cuda() if use_w else None, bw, w_iou, w_cls, w_stop)
    (total * 1.7).backward()
    assert_close("total", total, ref, 2e-6, 1e-5)
    assert_close("parts", parts, torch.stack([l_iou, l_stop, l_cls]), 2e-6, 1e-5)
    for a, b, nm in zip(dev, leaves, ("dprobs", "dstop", "dsiou")):
        assert_close(nm, a.grad, b.grad, 2e-6, 1e-4)


@pytest.mark.gpu
def test_convlstm_kernel_size_1():
    """`-kernel_size 1` (model.py:83-84: padding 0): the gate conv is a 1x1 conv -> implicit-GEMM kernel with the LSTM epilogue"""
    from oracle import rsis_oracle as O
    from rsis_amd.modules.clstm import ConvLSTMCell
    from helpers import mk_args
    B, Cin, hid, H, W = 2, 24, 16, 9, 12
    ocell = O.ConvLSTMCell(mk_args(), Cin, hid, 1, 0)
    with torch.no_grad():
        ocell.Gates.weight.copy_(_rng_t(1, tuple(ocell.Gates.weight.shape), 2.0 / np.sqrt(Cin + hid)))
        ocell.Gates.bias.copy_(_rng_t(2, (4 * hid,), 0.2))
    cell = ConvLSTMCell(mk_args(), Cin, hid, 1, 0).cuda()
    cell.load_state_dict(ocell.state_dict())
    x0, x1 = _rng_t(30, (B, Cin, H, W)).requires_grad_(), _rng_t(31, (B, Cin, H, W)).requires_grad_()
    gh = _rng_t(50, (B, hid, H, W))
    h0, c0 = ocell(x0, None)
    h1, c1 = ocell(x1, (h0, c0))
    ((h1 * gh).sum() + (c1 * gh).sum()).backward()
    x0d, x1d = _dev(x0.detach().clone().requires_grad_()), _dev(x1.detach().clone().requires_grad_())
    h0d, c0d = cell(x0d, None)
    h1d, c1d = cell(x1d, (h0d, c0d))
    ((h1d * gh.cuda()).sum() + (c1d * gh.cuda()).sum()).backward()
    for n, a, b in (("h1", h1d, h1), ("c1", c1d, c1)):
        assert_close(n, a, b, 2e-5, 1e-5)
    assert_close("dx0", x0d.grad, x0.grad, 5e-5, 1e-4)
    assert_close("dx1", x1d.grad, x1.grad, 5e-5, 1e-4)
    gw = ocell.Gates.weight.grad
    assert_close("dW", cell.Gates.weight.grad, gw, 1e-4 * max(1.0, float(gw.abs().max())), 1e-4)
    assert_close("db", cell.Gates.bias.grad, ocell.Gates.bias.grad, 1e-4 * max(1.0, float(ocell.Gates.bias.grad.abs().max())), 1e-4)


@pytest.mark.gpu
@pytest.mark.parametrize("shape,size", [((2, 3, 8, 8), (16, 16)), ((2, 5, 13, 25), (25, 50)), ((1, 8, 64, 64), (128, 128)), ((3, 2, 4, 5), (7, 9))])
def test_upsample_maxpool_bwd_one_launch(shape, size):
    """rsis_upsample_maxpool_bwd (the side max-pool's gradient folded into the upsample backward) == the two separate kernels"""
    from rsis_amd._lib import check, lib, ptr, stream
    B, C, Hi, Wi = shape
    torch.manual_seed(9)
    x = torch.randn(shape, device="cuda")
    dy = torch.randn(B, C, *size, device="cuda")
    dside = torch.randn(B, C, device="cuda")
    L = lib()
    side = torch.empty(B, C, device="cuda")
    arg = torch.empty(B, C, dtype=torch.int32, device="cuda")
    check(L.rsis_global_maxpool_fwd(ptr(x), ptr(side), ptr(arg), B * C, Hi * Wi, stream()), "gmax")
    want = torch.empty_like(x)
    check(L.rsis_upsample_bilinear_ac_bwd(ptr(dy), ptr(want), B * C, Hi, Wi, size[0], size[1], stream()), "up bwd")
    check(L.rsis_global_maxpool_bwd_add(ptr(dside), ptr(arg), ptr(want), B * C, Hi * Wi, stream()), "gmax bwd add")
    got = torch.empty_like(x)
    check(L.rsis_upsample_maxpool_bwd(ptr(dy), ptr(dside), ptr(arg), ptr(got), B * C, Hi, Wi, size[0], size[1], stream()), "fused")
    assert torch.equal(got, want)


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(2, 4, 8, 8), (2, 3, 9, 11), (1, 8, 32, 64)])
def test_maxpool3x3s2_accumulates_into_parked_gradient(shape):
    """a second consumer's gradient of the pooled tensor's input (ops.grad_tap -> GradSlot) is accumulated into by the max-pool
    backward in place: same result as autograd adding the two"""
    from rsis_amd import ops
    torch.manual_seed(4)
    x = torch.randn(shape, device="cuda", requires_grad=True)
    gy = torch.randn(shape[0], shape[1], (shape[2] + 1) // 2, (shape[3] + 1) // 2, device="cuda")
    gs = torch.randn(shape, device="cuda")

    def run(slot):
        x.grad = None
        xa = x * 1.0
        y = ops.maxpool3x3s2(xa, grad_slot=slot)
        side = ops.grad_tap(xa, slot) if slot is not None else xa
        ((y * gy).sum() + (side * gs).sum()).backward()
        return x.grad.clone()

    slot = ops.GradSlot()
    got = run(slot)
    assert slot.done and slot.grad is None
    want = run(None)
    assert_close("dx", got, want, 1e-6)


# fp32 weight gradients on maps no tile shape divides (the RAG instantiations of conv_wgrad_tiled.hip: dword DMA with a per-element
# "beyond the map" test): the 7 / 14 / 28-pixel pyramid of 224 x 224 inputs, odd sizes, every tile configuration (Cout 16..200, narrow
# and wide N), 1x1 and 3x3, two sources into one dW, a gate-interleaved ConvLSTM weight -- single launches and one grouped call
RAGGED_WGRAD_CASES = [(3, [64], 14, 14, 64, 3, 0), (2, [128], 28, 28, 128, 3, 0), (4, [64], 7, 7, 256, 3, 0), (2, [256], 14, 14, 1024, 1, 0),
                      (2, [1024], 14, 14, 256, 1, 0), (3, [96], 7, 7, 48, 1, 0), (2, [8], 9, 11, 16, 3, 0), (2, [20, 12], 17, 23, 40, 3, 0),
                      (2, [72], 5, 13, 200, 3, 0), (3, [40], 12, 18, 24, 1, 0), (2, [24, 8], 14, 14, 32, 3, 8), (1, [16], 30, 27, 96, 1, 0),
                      (2, [16], 28, 28, 32, 3, 0), (2, [200], 7, 7, 72, 3, 0)]


@pytest.mark.parametrize("grouped", [False, True], ids=["single", "grouped"])
def test_conv2d_wgrad_fp32_on_ragged_maps(grouped):
    from rsis_amd import ops
    from rsis_amd._lib import WgradJob, check, lib, ptr, stream
    L = lib()
    jobs, keep, want = [], [], []
    for k, (B, segs, H, W, Cout, ks, hid) in enumerate(RAGGED_WGRAD_CASES):
        Ctot, pad = sum(segs), ks // 2
        xs = [_rng_t(700 + 10 * k + i, (B, c, H, W)) for i, c in enumerate(segs)]
        w = _rng_t(800 + k, (Cout, Ctot, ks, ks)).requires_grad_()
        gy = _rng_t(900 + k, (B, Cout, H, W))
        F.conv2d(torch.cat(xs, 1).double(), w.double(), None, padding=pad).backward(gy.double()) if False else None
        wd = w.detach().double().requires_grad_()
        F.conv2d(torch.cat(xs, 1).double(), wd, None, padding=pad).backward(gy.double())
        ref = wd.grad.clone()
        if hid > 0:
            gy = gy.reshape(B, 4, hid, H, W).transpose(1, 2).reshape(B, Cout, H, W).contiguous()
        prev = _rng_t(1000 + k, (Cout, Ctot, ks, ks))
        dW, dy = _dev(prev.clone()), _dev(gy)
        c_off = 0
        for x in xs:
            xd = _dev(x)
            if grouped:
                j = WgradJob()
                (j.dy, j.x, j.dW, j.B, j.Cs, j.H, j.W, j.Cout, j.Ho, j.Wo, j.ks, j.stride, j.pad, j.Ctot, j.c_off, j.lstm_hid, j.dtype) = (
                    dy.data_ptr(), xd.data_ptr(), dW.data_ptr(), B, x.shape[1], H, W, Cout, H, W, ks, 1, pad, Ctot, c_off, hid, ops.DTYPE_F32)
                jobs.append(j)
            else:
                check(L.rsis_conv2d_wgrad(ptr(dy), ptr(xd), ptr(dW), B, x.shape[1], H, W, Cout, H, W, ks, 1, pad, Ctot, c_off, hid, ops.DTYPE_F32,
                                          stream()), "rsis_conv2d_wgrad")
            keep.append(xd)
            c_off += x.shape[1]
        keep.append(dy)
        want.append((dW, prev.double() + ref))
    if grouped:
        arr = (WgradJob * len(jobs))(*jobs)
        check(L.rsis_conv2d_wgrad_batch(arr, len(jobs), stream()), "rsis_conv2d_wgrad_batch")
    torch.cuda.synchronize()
    for k, (got, ref) in enumerate(want):
        assert_close("dW of case %d %r" % (k, RAGGED_WGRAD_CASES[k]), got, ref, 2e-5 * max(1.0, float(ref.abs().max())), 1e-5)


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_conv2d_wgrad_batch(dtype):
    """rsis_conv2d_wgrad_batch: a mixed bag of weight gradients (tiled 3x3 / 1x1 in several tile configurations, two sources of one
    conv into the same dW, a gate-interleaved ConvLSTM weight, an odd-sized map and a strided conv that fall back to single launches)
    in ONE call, each against torch's autograd; every dW is accumulated on top of its previous contents."""
    from rsis_amd import ops
    from rsis_amd._lib import WgradJob, check, lib, stream
    dt = ops.DTYPES[dtype]
    # (B, [Cin segs], H, W, Cout, ks, stride, pad, lstm_hid)
    cases = [(2, [16], 16, 16, 32, 3, 1, 1, 0), (2, [16], 16, 32, 64, 3, 1, 1, 0), (2, [32], 16, 16, 128, 3, 1, 1, 0),
             (2, [64], 8, 32, 96, 1, 1, 0, 0), (2, [20, 12], 16, 24, 40, 3, 1, 1, 0), (2, [72], 8, 16, 200, 3, 1, 1, 0),
             (3, [40], 12, 16, 24, 1, 1, 0, 0), (2, [24, 8], 16, 16, 32, 3, 1, 1, 8), (2, [8], 9, 11, 16, 3, 1, 1, 0),
             (2, [24], 32, 48, 40, 3, 2, 1, 0), (2, [256], 16, 16, 256, 3, 1, 1, 0), (2, [256], 16, 16, 64, 1, 1, 0, 0),
             (2, [64], 16, 16, 256, 1, 1, 0, 0), (1, [256], 16, 16, 256, 3, 1, 1, 0)]
    jobs, keep, want = [], [], []
    for k, (B, segs, H, W, Cout, ks, stride, pad, hid) in enumerate(cases):
        Ctot = sum(segs)
        Ho, Wo = (H + 2 * pad - ks) // stride + 1, (W + 2 * pad - ks) // stride + 1
        xs = [_rng_t(100 + 10 * k + i, (B, c, H, W)) for i, c in enumerate(segs)]
        w = _rng_t(300 + k, (Cout, Ctot, ks, ks)).requires_grad_()
        gy = _rng_t(400 + k, (B, Cout, Ho, Wo))
        F.conv2d(torch.cat(xs, 1), w, None, stride=stride, padding=pad).backward(gy)
        ref = w.grad.clone()
        if hid > 0:       # the kernel sees gate-interleaved dy rows 4 j + g and writes reference row g * hid + j
            gy = gy.reshape(B, 4, hid, Ho, Wo).transpose(1, 2).reshape(B, Cout, Ho, Wo).contiguous()
        prev = _rng_t(500 + k, (Cout, Ctot, ks, ks))
        dW, dy = _dev(prev.clone()), _dev(gy)
        c_off = 0
        for x in xs:
            xd = _dev(x)
            j = WgradJob()
            (j.dy, j.x, j.dW, j.B, j.Cs, j.H, j.W, j.Cout, j.Ho, j.Wo, j.ks, j.stride, j.pad, j.Ctot, j.c_off, j.lstm_hid, j.dtype) = (
                dy.data_ptr(), xd.data_ptr(), dW.data_ptr(), B, x.shape[1], H, W, Cout, Ho, Wo, ks, stride, pad, Ctot, c_off, hid, dt)
            jobs.append(j)
            keep.append(xd)
            c_off += x.shape[1]
        keep.append(dy)
        want.append((dW, prev + ref, Ctot * ks * ks * B * Ho * Wo))
    arr = (WgradJob * len(jobs))(*jobs)
    check(lib().rsis_conv2d_wgrad_batch(arr, len(jobs), stream()), "rsis_conv2d_wgrad_batch")
    torch.cuda.synchronize()
    for k, (got, ref, _n) in enumerate(want):
        scale = max(1.0, float((ref).abs().max()))
        assert_close("dW of job set %d" % k, got, ref, (1e-4 if dtype == "fp32" else 2e-2) * scale, 1e-5)
