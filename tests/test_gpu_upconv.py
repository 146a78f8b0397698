"""GPU parity tests of the decoder's fused tail (rsis_amd/csrc/upconv_out.hip): nn.UpsamplingBilinear2d x2 (align_corners, reference
src/modules/model.py:163-164) + conv_out (model.py:109,167) over the images of all timesteps as one op per direction, against the two
torch ops in float64 on the same inputs.  fp32 hidden states: 1e-5 of the result's scale (fp32 accumulation only); blk (bf16) hidden
states: exact bf16 inputs, so the same bar forward, and half a bf16 ulp more on the hidden-state gradient (rounded once at the store)."""
import pytest
import torch
import torch.nn.functional as F

from helpers import assert_close

pytestmark = pytest.mark.gpu

SHAPES = [(2, 3, 16, 16), (1, 2, 12, 20), (2, 1, 7, 9), (1, 1, 40, 70), (2, 2, 112, 112), (1, 2, 128, 128), (1, 1, 33, 65)]      # T, B, Hs, Ws


def _ref(h, w, b, dout, arg, dside, T, B):
    """float64: out (B, T, Ho*Wo), and the gradients of sum(out * dout) + the side term"""
    Hs, Ws = h.shape[-2:]
    hd = h.double().view(T * B, 8, Hs, Ws).clone().requires_grad_(True)
    wd, bd = w.double().clone().requires_grad_(True), b.double().clone().requires_grad_(True)
    up = F.interpolate(hd, size=(2 * Hs, 2 * Ws), mode="bilinear", align_corners=True)
    out = F.conv2d(up, wd, bd, padding=1).view(T, B, -1).transpose(0, 1)           # (B, T, N)
    side = hd.view(T * B * 8, Hs * Ws).gather(1, arg.view(-1, 1).long()).view(-1)    # the arg-max pixel of every (image, channel)
    ((out * dout.double()).sum() + (side * dside.double().view(-1)).sum()).backward()
    return out.detach(), hd.grad.view(T, B, 8, Hs, Ws), wd.grad, bd.grad


@pytest.mark.parametrize("blk", [False, True], ids=["fp32", "blk"])
@pytest.mark.parametrize("shape", SHAPES, ids=lambda s: "x".join(str(v) for v in s))
def test_upconv_out_forward_backward(shape, blk):
    from rsis_amd import ops
    from rsis_amd._lib import check, lib, ptr, stream
    L = lib()
    T, B, Hs, Ws = shape
    Ho, Wo = 2 * Hs, 2 * Ws
    assert L.rsis_upconv_out_supported(8, Hs, Ws, Ho, Wo) == 1
    torch.manual_seed(sum(shape) + int(blk))
    h = torch.randn(T, B, 8, Hs, Ws, device="cuda")
    if blk:
        h = h.to(torch.bfloat16).float()
    w = torch.randn(1, 8, 3, 3, device="cuda") / 8.0
    b = torch.randn(1, device="cuda")
    dout = torch.randn(B, T, Ho * Wo, device="cuda")
    arg = torch.randint(0, Hs * Ws, (T, B, 8), device="cuda", dtype=torch.int32)
    dside = torch.randn(T, B, 8, device="cuda")
    out_r, dh_r, dw_r, db_r = _ref(h, w, b, dout, arg, dside, T, B)

    hx = ops.blk_from_nchw(h.view(T * B, 8, Hs, Ws)).view(T, B, 1, Hs, Ws, 8) if blk else h
    out = torch.empty(B, T, Ho * Wo, device="cuda")
    check(L.rsis_upconv_out_fwd(ptr(hx), int(blk), ptr(w), ptr(b), ptr(out), T, B, 8, Hs, Ws, Ho, Wo, stream()), "fwd")
    assert_close("out", out, out_r, 1e-5 * float(out_r.abs().max()), 1e-5)

    dh = torch.full_like(hx, float("nan"))
    dW = torch.ones(1, 8, 3, 3, device="cuda")          # accumulated into
    db = torch.ones(1, device="cuda")
    nb = L.rsis_upconv_out_bwd_blocks(T, B, Hs, Ws)
    partial = torch.empty(nb * 80, device="cuda")
    args = (ptr(dout), ptr(hx), int(blk), ptr(w), ptr(dh), ptr(dW), ptr(db), ptr(dside), ptr(arg), ptr(partial), T, B, 8, Hs, Ws, Ho, Wo, stream())
    check(L.rsis_upconv_out_bwd(*args), "bwd")
    dhf = ops.blk_to_nchw(dh.view(T * B, 1, Hs, Ws, 8)).view(T, B, 8, Hs, Ws) if blk else dh
    sc = float(dh_r.abs().max())
    if blk:
        assert_close("dh", dhf, dh_r, 1e-5 * sc, 2.0 ** -8)
    else:
        assert_close("dh", dhf, dh_r, 1e-5 * sc, 1e-5)
    assert_close("dW", dW - 1.0, dw_r, 2e-5 * float(dw_r.abs().max()), 1e-5)
    assert_close("db", db - 1.0, db_r, 2e-5 * max(float(db_r.abs().max()), float(dout.abs().sum()) ** 0.5), 1e-5)
    # reproducible run to run (no atomics), and the side term is optional
    dW2, db2, dh2 = torch.ones_like(dW), torch.ones_like(db), torch.empty_like(dh)
    args2 = (ptr(dout), ptr(hx), int(blk), ptr(w), ptr(dh2), ptr(dW2), ptr(db2), ptr(dside), ptr(arg), ptr(partial), T, B, 8, Hs, Ws, Ho, Wo, stream())
    check(L.rsis_upconv_out_bwd(*args2), "bwd")
    assert torch.equal(dW2, dW) and torch.equal(db2, db) and torch.equal(dh2.view(torch.int16 if blk else torch.int32), dh.view(torch.int16 if blk else torch.int32))
    args3 = (ptr(dout), ptr(hx), int(blk), ptr(w), ptr(dh2), None, None, None, None, ptr(partial), T, B, 8, Hs, Ws, Ho, Wo, stream())
    check(L.rsis_upconv_out_bwd(*args3), "bwd without the side term / parameter gradients")
    dh3 = ops.blk_to_nchw(dh2.view(T * B, 1, Hs, Ws, 8)).view(T, B, 8, Hs, Ws) if blk else dh2
    side = torch.zeros(T * B * 8, Hs * Ws, device="cuda", dtype=torch.float64)
    side.scatter_(1, arg.view(-1, 1).long(), dside.double().view(-1, 1))
    assert_close("dh without side", dh3, dh_r - side.view(T, B, 8, Hs, Ws), 1e-5 * sc, 2.0 ** -8 if blk else 1e-5)


def test_upconv_out_rejects_other_geometries():
    from rsis_amd._lib import lib
    L = lib()
    assert L.rsis_upconv_out_supported(4, 16, 16, 32, 32) == 0          # hidden_size / 16 != 8
    assert L.rsis_upconv_out_supported(8, 16, 16, 48, 48) == 0          # x3
    assert L.rsis_upconv_out_supported(8, 16, 16, 16, 16) == 0          # same size
    assert L.rsis_upconv_out_supported(8, 256, 512, 512, 1024) == 1     # configs[4]


@pytest.mark.parametrize("with_masks", [True, False], ids=["mask-loss", "heads-only"])
@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_decoder_node_with_and_without_the_fused_tail(dtype, with_masks):
    """the decoder's sequence node through the fused tail against the same node through upsample + conv_out (decoder_seq.UPCONV off).
    fp32: the same sums in another order -- 1e-5 of each tensor's scale.  bf16 (blk storage): the unfused path rounds the upsampled
    tensor and its gradient to bf16, the fused one rounds neither: outputs within a bf16 ulp (2^-7) of the reference's scale, every
    gradient within 2 % relative L2."""
    from oracle import filler
    from helpers import mk_args
    from rsis_amd import decoder_seq
    from rsis_amd.modules import RSIS
    hs, B, T = 128, 2, 3
    sizes = [(7, 7), (14, 14), (28, 28), (56, 56), (112, 112)]
    chans = [hs, hs, hs // 2, hs // 4, hs // 8]
    dec = RSIS(mk_args(hidden_size=hs, maxseqlen=T, dtype=dtype)).cuda()
    feats_cpu = [filler.tensor(11, "uc.f%d" % i, (B, chans[i]) + sizes[i]) for i in range(5)]
    gm = [filler.tensor(11, "uc.gm%d" % t, (B, 1, 224, 224)) for t in range(T)]
    res = []
    try:
        for fused in (False, True):
            decoder_seq.UPCONV[0] = fused
            dec.zero_grad()
            feats = [f.cuda().requires_grad_() for f in feats_cpu]
            assert decoder_seq.supported(dec, feats, T) and decoder_seq.blk_supported(dec, feats) == (dtype == "bf16")
            steps, _hid = dec.forward_sequence(feats, T)
            # (heads-only: no gradient reaches the mask logits -- the node's backward sees d_masks = None and the tail's backward still
            #  has to deliver the side max-pool gradient of the last level)
            sum(((m * g.to(m.device, m.dtype)).sum() if with_masks else 0.0) + (c * c).sum() * 20 + s.sum() for (m, c, s), g in zip(steps, gm)).backward()
            res.append((torch.cat([st[0].reshape(-1) for st in steps]).detach().double(), [f.grad.double() for f in feats],
                        {k: (p.grad.double() if p.grad is not None else None) for k, p in dec.named_parameters()}))
    finally:
        decoder_seq.UPCONV[0] = True
    (m0, f0, p0), (m1, f1, p1) = res

    def rel(x, y):
        return float((x - y).norm() / y.norm().clamp_min(1e-30))

    if dtype == "fp32":
        assert_close("masks", m1, m0, 1e-5 * float(m0.abs().max()), 1e-5)
        for i in range(5):
            assert_close("d feat %d" % i, f1[i], f0[i], 2e-5 * float(f0[i].abs().max()), 1e-4)
        for k in p0:
            if p0[k] is None or p1[k] is None:
                assert p0[k] is None and p1[k] is None, k
                continue
            assert_close("d " + k, p1[k], p0[k], 2e-5 * float(p0[k].abs().max()) + 1e-12, 1e-4)
    else:
        assert_close("masks", m1, m0, 2.0 ** -7 * float(m0.abs().max()), 2.0 ** -7)
        for i in range(5):
            assert rel(f1[i], f0[i]) < 0.02, (i, rel(f1[i], f0[i]))
        for k in p0:
            if p0[k] is None or p1[k] is None:
                assert p0[k] is None and p1[k] is None, k
                continue
            if float(p0[k].abs().max()) == 0.0:
                assert float(p1[k].abs().max()) == 0.0, k
                continue
            assert rel(p1[k], p0[k]) < 0.02, (k, rel(p1[k], p0[k]))
