"""GPU parity tests of the recurrent decoder's kernels on channel-blocked bf16 tensors (rsis_amd/csrc/conv_blk_dec.hip, blk_dec.hip;
include/rsis_hip.h rsis_blk_conv3x3_batch / rsis_blk_upsample_*_batch / rsis_blk_lstm_bwd_batch / rsis_blk_conv_out_seq_*): the storage
half of `-dtype bf16` in the decoder (BASELINE.json configs[2..4]; reference ops clstm.py:43-58, model.py:143-167).

Semantics under test, stated before measuring (as tests/test_gpu_blk.py): every kernel computes in fp32 on EXACT bf16 inputs and rounds
ONCE to bf16 at a blk store.  The reference is the same op in float64 on the same bf16-valued inputs (bf16-rounded weights for the
convs); a blk result may differ from it by half a bf16 ulp of the exact value (2^-8 relative with the binade margin) plus fp32
accumulation noise (1e-5 of the output scale); fp32 outputs (cell state, conv_out logits, parameter gradients) by the fp32 noise only."""
import pytest
import torch

from helpers import assert_close
from test_gpu_blk import HALF_ULP, _bf16, from_blk, to_blk

pytestmark = pytest.mark.gpu
F = torch.nn.functional


def _key_decode(key):
    """(value, flat pixel) of rsis_side_key keys (int64 tensor)"""
    u = (key >> 32) & 0xFFFFFFFF
    neg = (u & 0x80000000) == 0
    bits = torch.where(neg, (~u) & 0xFFFFFFFF, u & 0x7FFFFFFF)
    val = bits.to(torch.int32).view(torch.float32) if False else torch.tensor(bits.cpu().numpy().astype("uint32").view("float32")).to(key.device)
    idx = 0x7FFFFFFF - (key & 0xFFFFFFFF)
    return val, idx


CONV_SHAPES = [  # (B, segs, Cout, H, W)
    (2, [128], 512, 7, 7), (2, [128, 64], 256, 14, 14), (2, [64, 32], 128, 28, 28), (2, [32, 16], 64, 56, 56), (2, [16, 8], 32, 112, 112),
    (3, [24, 8, 40], 48, 10, 13), (2, [8], 8, 5, 5),
]


@pytest.mark.parametrize("shape", CONV_SHAPES, ids=lambda s: "%dx%sx%dx%dx%d" % (s[0], "+".join(map(str, s[1])), s[2], s[3], s[4]))
@pytest.mark.parametrize("tile", [0, 1, 4, 5])
def test_blk_conv3x3_plain_multi_source_bias_addend_two_destinations(shape, tile):
    from rsis_amd import ops
    B, segs, Cout, H, W = shape
    torch.manual_seed(sum(segs) + Cout + H)
    Cin = sum(segs)
    xs = [_bf16(torch.randn(B, c, H, W, device="cuda")) for c in segs]
    w = torch.randn(Cout, Cin, 3, 3, device="cuda") / (Cin * 9) ** 0.5
    bias = torch.randn(Cout, device="cuda")
    add = _bf16(torch.randn(B, Cout, H, W, device="cuda"))
    pk = ops.PackedConv(3, segs, stride=1, pad=1, dtype=ops.DTYPE_BF16)
    ref = F.conv2d(torch.cat(xs, 1).double(), _bf16(w).double(), padding=1)
    scale = float(ref.abs().max())
    # one destination, no bias / addend
    y = torch.empty((B, Cout // 8, H, W, 8), dtype=torch.bfloat16, device="cuda")
    ops.blk_conv3x3_batch([ops.blk_conv_job([to_blk(x) for x in xs], pk.fwd(w), Cout, dsts=[y], tile=tile)])
    assert_close("plain", from_blk(y), ref, 1e-5 * scale, HALF_ULP)
    # bias + addend + two destinations
    if Cout >= 16:
        c0 = (Cout // 16) * 8
        d0 = torch.empty((B, c0 // 8, H, W, 8), dtype=torch.bfloat16, device="cuda")
        d1 = torch.empty((B, (Cout - c0) // 8, H, W, 8), dtype=torch.bfloat16, device="cuda")
        ops.blk_conv3x3_batch([ops.blk_conv_job([to_blk(x) for x in xs], pk.fwd(w), Cout, bias=bias, addend=to_blk(add), dsts=[d0, d1], tile=tile)])
        ref2 = ref + bias.double().view(1, -1, 1, 1) + add.double()
        got = torch.cat([from_blk(d0), from_blk(d1)], 1)
        assert_close("bias+addend, two destinations", got, ref2, 1e-5 * float(ref2.abs().max()), HALF_ULP)


@pytest.mark.parametrize("shape", [(2, 128, 0, 128, 7, 7), (2, 64, 128, 64, 14, 14), (2, 32, 64, 32, 28, 28), (2, 16, 32, 16, 56, 56),
                                   (2, 8, 16, 8, 112, 112), (3, 16, 24, 8, 10, 13)], ids=lambda s: "x".join(str(v) for v in s))
@pytest.mark.parametrize("tile", [0, 1, 4, 5])
def test_blk_convlstm_cell_and_its_data_gradient(shape, tile):
    """the fused cell (clstm.py:43-58) with the hoisted skip term as addend, zero state and recurrent state, the side max-pool keys, and
    the data gradient of the dynamic channels through the SAME entry point on the data-gradient pack (two destinations)"""
    from rsis_amd import ops
    B, hid, c_up, c_skip, H, W = shape
    torch.manual_seed(hid + H)
    Ctot = c_up + c_skip + hid
    w = torch.randn(4 * hid, Ctot, 3, 3, device="cuda") / (Ctot * 9) ** 0.5
    segs, offs = ([c_up], [0]) if c_up else ([], [])
    dyn = ops.PackedConv(3, segs + [hid], lstm_hid=hid, stride=1, pad=1, offs=offs + [c_up + c_skip], dtype=ops.DTYPE_BF16)
    up = _bf16(torch.randn(B, c_up, H, W, device="cuda")) if c_up else None
    h_prev = _bf16(torch.tanh(torch.randn(B, hid, H, W, device="cuda")))
    c_prev = torch.randn(B, hid, H, W, device="cuda")
    G = _bf16(torch.randn(B, 4 * hid, H, W, device="cuda"))          # reference rows [i | f | o | g] x hid
    perm = torch.arange(4 * hid, device="cuda").view(4, hid).t().reshape(-1)       # packed row 4 j + gate <- reference row gate * hid + j
    G_blk = to_blk(G[:, perm].contiguous())
    wd = _bf16(w).double()
    for state in (False, True):
        if not state and c_up == 0:
            srcs, ref_in, wsel = [], None, None
        else:
            parts = ([up] if c_up else []) + ([h_prev] if state else [])
            cols = (list(range(c_up)) if c_up else []) + (list(range(c_up + c_skip, Ctot)) if state else [])
            srcs, ref_in, wsel = [to_blk(p) for p in parts], torch.cat(parts, 1).double(), wd[:, cols]
        a = G.double() + (F.conv2d(ref_in, wsel, padding=1) if ref_in is not None else 0.0)
        ai, af, ao, ag = a.chunk(4, 1)
        gi, gf, go, gg = torch.sigmoid(ai), torch.sigmoid(af), torch.sigmoid(ao), torch.tanh(ag)
        c_ref = gf * (c_prev.double() if state else 0.0) + gi * gg
        h_ref = go * torch.tanh(c_ref)
        h = torch.empty((B, hid // 8, H, W, 8), dtype=torch.bfloat16, device="cuda")
        act = torch.empty((B, 4 * hid // 8, H, W, 8), dtype=torch.bfloat16, device="cuda")
        c = torch.empty(B, hid, H, W, device="cuda")
        key = torch.zeros(B, hid, dtype=torch.int64, device="cuda")
        ops.blk_conv3x3_batch([ops.blk_conv_job(srcs, dyn.fwd(w), 4 * hid, addend=G_blk, hid=hid, c_prev=c_prev if state else None, c_out=c,
                                               h_out=h, act_out=act, side_key=key, tile=tile, shape=(B, H, W))])
        # gates saturate: their error is absolute (the bf16 rounding of the conv operands moves the pre-activation by ~1e-2 at most);
        # the inputs here are exact bf16, so only the accumulation order and the ONE rounding of each output remain
        assert_close("c (state=%s)" % state, c, c_ref, 2e-5 * max(1.0, float(c_ref.abs().max())), 2e-5)
        assert_close("h (state=%s)" % state, from_blk(h), h_ref, 2e-5, HALF_ULP)
        act_ref = torch.cat([gi, gf, go, gg], 1)[:, perm]
        assert_close("act (state=%s)" % state, from_blk(act), act_ref, 2e-5, HALF_ULP)
        # side feature keys: the maximum of h BEFORE its rounding for storage (fp32) and the pixel attaining it -- so the stored h at that
        # pixel is the rounded pooled value, and the pooled value is the float64 maximum up to fp32 noise
        hs = from_blk(h).view(B, hid, -1)
        val, idx = _key_decode(key)
        assert torch.equal(_bf16(val), hs.gather(2, idx.unsqueeze(-1)).squeeze(-1)), "stored h at the pooled pixel != rounded pooled value"
        assert_close("pooled value (state=%s)" % state, val, h_ref.view(B, hid, -1).max(dim=2).values, 2e-5, 2e-5)
    # data gradient of [up | h_prev] from d(gates) in packed row order
    da = _bf16(torch.randn(B, 4 * hid, H, W, device="cuda"))          # packed rows
    inv = torch.empty_like(perm)
    inv[perm] = torch.arange(4 * hid, device="cuda")
    da_ref_rows = da[:, inv].double()                                  # reference row order
    cols = (list(range(c_up)) if c_up else []) + list(range(c_up + c_skip, Ctot))
    dref = F.conv_transpose2d(da_ref_rows, wd[:, cols], padding=1)
    dh = torch.empty((B, hid // 8, H, W, 8), dtype=torch.bfloat16, device="cuda")
    dsts = [dh]
    if c_up:
        dup = torch.empty((B, c_up // 8, H, W, 8), dtype=torch.bfloat16, device="cuda")
        dsts = [dup, dh]
    ops.blk_conv3x3_batch([ops.blk_conv_job([to_blk(da)], dyn.dgrad(w), c_up + hid, cpack=dyn.cin, dsts=dsts, tile=tile)])
    got = torch.cat([from_blk(d) for d in dsts], 1)
    assert_close("dgrad", got, dref, 1e-5 * float(dref.abs().max()), HALF_ULP)
    if c_up:        # leading destination only (the t = 0 cells have no recurrent input)
        dup2 = torch.empty_like(dup)
        ops.blk_conv3x3_batch([ops.blk_conv_job([to_blk(da)], dyn.dgrad(w), c_up, cpack=dyn.cin, dsts=[dup2], tile=tile)])
        assert torch.equal(dup2, dup)


def test_grouped_launch_equals_single_launches():
    """the cells of a wavefront diagonal in ONE call: bit-identical to one call per cell"""
    from rsis_amd import ops
    torch.manual_seed(3)
    B = 2
    cfg = [(128, 0, 7), (64, 128, 14), (32, 64, 28), (16, 32, 56), (8, 16, 112)]
    jobs, outs = [], []
    for hid, c_up, hw in cfg:
        w = torch.randn(4 * hid, c_up + hid, 3, 3, device="cuda") / ((c_up + hid) * 9) ** 0.5
        pk = ops.PackedConv(3, ([c_up] if c_up else []) + [hid], lstm_hid=hid, stride=1, pad=1, dtype=ops.DTYPE_BF16)
        srcs = ([to_blk(torch.randn(B, c_up, hw, hw, device="cuda"))] if c_up else []) + [to_blk(torch.randn(B, hid, hw, hw, device="cuda"))]
        G = to_blk(torch.randn(B, 4 * hid, hw, hw, device="cuda"))
        c_prev = torch.randn(B, hid, hw, hw, device="cuda")
        res = []
        for _ in range(2):
            res.append((torch.empty((B, hid // 8, hw, hw, 8), dtype=torch.bfloat16, device="cuda"), torch.empty(B, hid, hw, hw, device="cuda"),
                        torch.empty((B, 4 * hid // 8, hw, hw, 8), dtype=torch.bfloat16, device="cuda"), torch.zeros(B, hid, dtype=torch.int64, device="cuda")))
        outs.append(res)
        jobs.append([ops.blk_conv_job(srcs, pk.fwd(w), 4 * hid, addend=G, hid=hid, c_prev=c_prev, c_out=r[1], h_out=r[0], act_out=r[2], side_key=r[3])
                     for r in res])
    ops.blk_conv3x3_batch([j[0] for j in jobs])
    for j in jobs:
        ops.blk_conv3x3_batch([j[1]])
    for (a, b) in outs:
        for p, q in zip(a, b):
            assert torch.equal(p, q)


@pytest.mark.parametrize("shape", [(2, 16, 7, 7, 14, 14), (2, 8, 56, 56, 112, 112), (3, 24, 5, 7, 9, 13), (2, 8, 112, 112, 224, 224), (2, 8, 1, 1, 3, 3)],
                         ids=lambda s: "x".join(str(v) for v in s))
def test_blk_upsample_forward_and_transpose_with_pooled_gradient(shape):
    from rsis_amd import ops
    B, C, Hi, Wi, Ho, Wo = shape
    torch.manual_seed(Hi + Wo)
    x = _bf16(torch.randn(B, C, Hi, Wi, device="cuda"))
    y = torch.empty((B, C // 8, Ho, Wo, 8), dtype=torch.bfloat16, device="cuda")
    ops.blk_upsample_fwd_batch([ops.blk_resize_job(to_blk(x), y)])
    xd = x.double().requires_grad_()
    ref = F.interpolate(xd, size=(Ho, Wo), mode="bilinear", align_corners=True)
    # (the source coordinate scale * o is computed in fp32, as nn.UpsamplingBilinear2d itself does: its rounding moves the lerp weights
    #  by ~4e-6 on a 224-pixel row, i.e. the result by ~1e-5 of the input scale -- absolute, on top of the one rounding at the store)
    assert_close("fwd", from_blk(y), ref.detach(), 2e-5 * float(ref.abs().max()), HALF_ULP)
    dy = _bf16(torch.randn(B, C, Ho, Wo, device="cuda"))
    ref.backward(dy.double())
    dpool = torch.randn(B, C, device="cuda")
    arg = torch.randint(0, Hi * Wi, (B, C), device="cuda", dtype=torch.int32)
    dx = torch.empty((B, C // 8, Hi, Wi, 8), dtype=torch.bfloat16, device="cuda")
    dx2 = torch.empty_like(dx)
    ops.blk_upsample_bwd_batch([ops.blk_resize_job(to_blk(dy), dx, backward=True), ops.blk_resize_job(to_blk(dy), dx2, dpool, arg, backward=True)])
    dref = xd.grad
    assert_close("bwd", from_blk(dx), dref, 2e-5 * float(dref.abs().max()), HALF_ULP)
    dref2 = dref.clone().view(B, C, -1)
    dref2.scatter_add_(2, arg.long().unsqueeze(-1), dpool.double().unsqueeze(-1))
    assert_close("bwd + pooled gradient", from_blk(dx2), dref2.view_as(dref), 2e-5 * float(dref2.abs().max()), HALF_ULP)


@pytest.mark.parametrize("shape", [(2, 128, 7, 7), (2, 16, 56, 56), (3, 8, 10, 13)], ids=lambda s: "x".join(str(v) for v in s))
@pytest.mark.parametrize("full", [True, False])
def test_blk_lstm_pointwise_backward(shape, full):
    from rsis_amd import ops
    B, hid, H, W = shape
    torch.manual_seed(hid + H)
    perm = torch.arange(4 * hid, device="cuda").view(4, hid).t().reshape(-1)
    act = _bf16(torch.cat([torch.sigmoid(torch.randn(B, 3 * hid, H, W, device="cuda")), torch.tanh(torch.randn(B, hid, H, W, device="cuda"))], 1))
    gi, gf, go, gg = act.double().chunk(4, 1)
    c_prev = torch.randn(B, hid, H, W, device="cuda")
    cp = c_prev.double() if full else torch.zeros_like(gi)
    c = (gf * cp + gi * gg).float()            # the stored fp32 cell state
    dh, dh2 = _bf16(torch.randn(B, hid, H, W, device="cuda")), _bf16(torch.randn(B, hid, H, W, device="cuda"))
    dcn = torch.randn(B, hid, H, W, device="cuda")
    dhv = dh.double() + (dh2.double() if full else 0.0)
    tc = torch.tanh(c.double())
    dcv = dhv * go * (1 - tc * tc) + (dcn.double() if full else 0.0)
    da_ref = torch.cat([dcv * gg * gi * (1 - gi), dcv * cp * gf * (1 - gf), dhv * tc * go * (1 - go), dcv * gi * (1 - gg * gg)], 1)[:, perm]
    da = torch.empty((B, 4 * hid // 8, H, W, 8), dtype=torch.bfloat16, device="cuda")
    dcp = torch.empty(B, hid, H, W, device="cuda")
    ops.blk_lstm_bwd_batch([ops.blk_lstm_bwd_job(to_blk(dh), to_blk(dh2) if full else None, dcn if full else None, to_blk(act[:, perm].contiguous()),
                                                 c_prev if full else None, c, da, dcp if full else None)])
    assert_close("da", from_blk(da), da_ref, 1e-5 * float(da_ref.abs().max()), HALF_ULP)
    if full:
        ref = dcv * gf
        assert_close("dc_prev", dcp, ref, 1e-5 * float(ref.abs().max()), 1e-5)


@pytest.mark.parametrize("shape", [(3, 2, 16, 24), (10, 4, 64, 64), (2, 2, 7, 12)], ids=lambda s: "x".join(str(v) for v in s))
def test_blk_conv_out_over_all_timesteps(shape):
    from rsis_amd._lib import check, lib, ptr, stream
    T, B, H, W = shape
    torch.manual_seed(T + H)
    x = _bf16(torch.randn(T, B, 8, H, W, device="cuda"))
    w = torch.randn(1, 8, 3, 3, device="cuda") / 72 ** 0.5
    bias = torch.randn(1, device="cuda")
    xb = to_blk(x.view(T * B, 8, H, W))
    y = torch.empty(B, T, H * W, device="cuda")
    L = lib()
    check(L.rsis_blk_conv_out_seq_fwd(ptr(xb), ptr(w), ptr(bias), ptr(y), T, B, H, W, stream()), "fwd")
    ref = F.conv2d(x.view(T * B, 8, H, W).double(), w.double(), bias.double(), padding=1).view(T, B, H * W).transpose(0, 1)
    assert_close("fwd", y, ref, 1e-5 * float(ref.abs().max()), 1e-5)
    dy = torch.randn(B, T, H * W, device="cuda")
    dyt = dy.transpose(0, 1).reshape(T * B, 1, H, W).double()
    dx = torch.empty((T * B, 1, H, W, 8), dtype=torch.bfloat16, device="cuda")
    check(L.rsis_blk_conv_out_seq_dgrad(ptr(dy), ptr(w), ptr(dx), T, B, H, W, stream()), "dgrad")
    dref = F.conv_transpose2d(dyt, w.double(), padding=1)
    assert_close("dgrad", from_blk(dx), dref, 1e-5 * float(dref.abs().max()), HALF_ULP)
    dW, db = torch.zeros(72, device="cuda"), torch.zeros(1, device="cuda")
    check(L.rsis_blk_conv_out_seq_wgrad(ptr(dy), ptr(xb), ptr(dW), ptr(db), T, B, H, W, stream()), "wgrad")
    xd = x.view(T * B, 8, H, W).double()
    wref = torch.autograd.grad(F.conv2d(xd, w.double().requires_grad_(), padding=1), [], allow_unused=True) if False else None
    wq = w.double().clone().requires_grad_()
    F.conv2d(xd, wq, padding=1).backward(dyt)
    assert_close("dW", dW.view(1, 8, 3, 3), wq.grad, 2e-5 * float(wq.grad.abs().max()), 2e-5)
    assert_close("db", db, dyt.sum().view(1), 2e-5 * float(dyt.abs().sum()), 0)


@pytest.mark.parametrize("geom", ["224", "odd"])
def test_blk_decoder_sequence_against_the_fp32_storage_decoder_and_float64(geom):
    """The whole T-step decoder on blk storage (decoder_seq._DecoderSeqBlkFn) against (a) the same bf16-operand kernels on fp32 NCHW
    storage (RSIS_DECODER_BLK=0, decoder_seq._DecoderSeqFn) and (b) the float64 oracle decoder on the same weights.  Bars stated before
    measuring: blk storage adds ONE bf16 rounding to tensors the fp32-storage path already rounds when it stages them (h, up(h)) and
    rounds three it keeps in fp32 (the hoisted gate term, the saved gates, the gate gradients): outputs within 2 % of max |reference|
    of the fp32-storage path, every gradient within 6 % relative L2 of it -- and no farther from float64 than 1.5 x the fp32-storage
    path's own distance + 1 %."""
    from oracle import filler
    from oracle import rsis_oracle as O
    from helpers import mk_args
    from rsis_amd import decoder_seq
    from rsis_amd.modules import RSIS
    hs, B, T = 128, 2, 3
    sizes = [(7, 7), (14, 14), (28, 28), (56, 56), (112, 112)] if geom == "224" else [(3, 5), (6, 10), (12, 20), (23, 40), (46, 80)]
    chans = [hs, hs, hs // 2, hs // 4, hs // 8]
    a = mk_args(hidden_size=hs, maxseqlen=T, dtype="bf16")
    odec = filler.fill_module(O.RSIS(mk_args(hidden_size=hs, maxseqlen=T)), seed=5).double()
    dec = RSIS(a).cuda()
    dec.load_state_dict({k: v.float() for k, v in odec.state_dict().items()})
    feats_cpu = [filler.tensor(9, "bd.f%d" % i, (B, chans[i]) + sizes[i]) for i in range(5)]
    gm = [filler.tensor(9, "bd.gm%d" % t, (B, 1, 2 * sizes[-1][0], 2 * sizes[-1][1])) for t in range(T)]

    def loss_of(steps):
        return sum((m * g.to(m.device, m.dtype)).sum() + (c * c).sum() * 20 + s.sum() for (m, c, s), g in zip(steps, gm))

    # float64 truth
    f64 = [f.double().requires_grad_() for f in feats_cpu]
    hidden, steps = None, []
    for _t in range(T):
        m, c, s, hidden = odec(f64, hidden)
        steps.append((m, c, s))
    loss_of(steps).backward()
    truth = ([torch.cat([x.reshape(-1) for x in st]) for st in steps], [f.grad for f in f64], {k: p.grad for k, p in odec.named_parameters()})
    res = []
    was = decoder_seq.BLK_ENABLED[0]
    try:
        for blk in (False, True):
            decoder_seq.BLK_ENABLED[0] = blk
            dec.zero_grad()
            feats = [f.cuda().requires_grad_() for f in feats_cpu]
            assert decoder_seq.supported(dec, feats, T) and decoder_seq.blk_supported(dec, feats) == blk
            steps, _hid = dec.forward_sequence(feats, T)
            loss_of(steps).backward()
            res.append(([torch.cat([x.reshape(-1) for x in st]).detach().cpu().double() for st in steps], [f.grad.cpu().double() for f in feats],
                        {k: p.grad.cpu().double() for k, p in dec.named_parameters()}))
    finally:
        decoder_seq.BLK_ENABLED[0] = was

    def rel(x, y):
        return float((x - y).norm() / y.norm().clamp_min(1e-30))

    worst = []
    for t in range(T):
        ref, got = res[0][0][t], res[1][0][t]
        assert float((got - ref).abs().max()) <= 0.02 * float(ref.abs().max()), "step %d outputs: %g of max" % (t, float((got - ref).abs().max() / ref.abs().max()))
    for name, r0, r1, tr in ([("dfeat%d" % i, res[0][1][i], res[1][1][i], truth[1][i]) for i in range(5)] +
                             [("grad." + k, res[0][2][k], res[1][2][k], truth[2][k]) for k in truth[2]]):
        e_store, e0, e1 = rel(r1, r0), rel(r0, tr), rel(r1, tr)
        worst.append((e_store, name, e0, e1))
        assert e_store <= 0.06, "%s: blk storage is %.3f relative L2 from the fp32-storage path" % (name, e_store)
        assert e1 <= 1.5 * e0 + 0.01, "%s: %.4f from float64 (fp32 storage: %.4f)" % (name, e1, e0)
    print("largest storage distances:", sorted(worst, reverse=True)[:4])


def test_blk_skip_branches_against_the_fp32_storage_branches():
    """FeatureExtractor.forward(blk_skips=True) -- the skip convs + BatchNorms of model.py:59-63 on the trunk's blk features, blk out
    (blk_trunk._SkipsBlkFn) -- against the default branches (the same bf16-operand convs on fp32 NCHW copies) on the SAME trunk
    features: the four blk skip features within half a bf16 ulp of the fp32-storage ones (+ 1 % of the feature scale for the second
    rounding of the conv output ahead of the BatchNorm), the gradients of the skip convs / BatchNorms within 3 % relative L2, and the
    gradient handed to the trunk (summed into x_k) within 3 %."""
    from helpers import mk_args
    from rsis_amd.modules import FeatureExtractor
    torch.manual_seed(0)
    a = mk_args(hidden_size=128, dtype="bf16")
    enc = FeatureExtractor(a).cuda().train()
    x = torch.randn(4, 3, 96, 96, device="cuda")
    res = []
    gws = None
    for blk in (False, True):
        enc.zero_grad()
        torch.manual_seed(1)
        feats = enc(x, blk_skips=blk)
        if blk:
            assert all(f.dtype == torch.bfloat16 and f.dim() == 5 for f in feats[:4]) and feats[4].dtype == torch.float32
        dense = [from_blk(f) if f.dim() == 5 else f for f in feats]
        if gws is None:
            gws = [torch.randn_like(d) for d in dense]
        loss = sum((d.float() * g).sum() for d, g in zip(dense, gws))
        loss.backward()
        res.append(([d.detach().float() for d in dense],
                    {k: p.grad.detach().clone() for k, p in enc.named_parameters() if p.grad is not None and not k.startswith("base.fc")}))
    for i in range(5):
        ref, got = res[0][0][i], res[1][0][i]
        assert_close("skip%d" % (5 - i), got, ref, 1e-2 * float(ref.abs().max()), HALF_ULP)
    bad = []
    for k, g0 in res[0][1].items():
        if k.startswith("sk") and k.endswith("bias"):
            continue                 # (a conv bias in front of a BatchNorm: its gradient is mathematically zero, both sides are noise)
        g1 = res[1][1][k]
        e = float((g1 - g0).norm() / g0.norm().clamp_min(1e-30))
        lim = 0.03 if not k.startswith("base.") else 0.08       # (trunk: the difference is carried back through ~100 train-mode BatchNorms)
        if e > lim:
            bad.append((k, e))
    assert not bad, bad[:8]


def test_blk_skip_branches_train_behind_a_frozen_trunk():
    """with `base` frozen the blk features carry no gradient; the skip node must stay in the graph (anchor parameter) so that
    sk5..sk2 / bn5..bn2 receive the same gradients as with a trainable trunk."""
    from helpers import mk_args
    from rsis_amd.modules import FeatureExtractor
    torch.manual_seed(0)
    a = mk_args(hidden_size=128, dtype="bf16")
    enc = FeatureExtractor(a).cuda().train()
    x = torch.randn(4, 3, 96, 96, device="cuda")
    grads = []
    gws = None
    for frozen in (False, True):
        for p in enc.base.parameters():
            p.requires_grad_(not frozen)
        enc.zero_grad()
        feats = enc(x, blk_skips=True)
        dense = [from_blk(f) if f.dim() == 5 else f for f in feats]
        if gws is None:
            gws = [torch.randn_like(d) for d in dense]
        sum((d.float() * g).sum() for d, g in zip(dense, gws)).backward()
        grads.append({k: p.grad.detach().clone() for k, p in enc.named_parameters() if p.grad is not None and not k.startswith("base.")})
    want = sorted(k for k in grads[0])
    assert sorted(grads[1]) == want and any(k.startswith("sk5") for k in want) and any(k.startswith("bn2") for k in want), sorted(grads[1])
    for k in want:
        if k.startswith("sk") and k.endswith("bias"):
            continue
        e = float((grads[1][k] - grads[0][k]).norm() / grads[0][k].norm().clamp_min(1e-30))
        assert e <= 0.02, (k, e)        # (not bit-equal: train-mode BatchNorm running statistics moved between the two passes; atomics order)
