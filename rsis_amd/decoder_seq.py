"""The T-step recurrent decoder as ONE autograd node with an explicit back-propagation through time.

What it computes is the loop of reference train.py:85-94 / test.py:37-38 around RSIS.forward (model.py:122-184) from the zero
state: per timestep five ConvLSTM levels (clstm.py:19-62), the global max-pool side features (model.py:143), the align-corners
upsamples between levels (:149-150, :163-164), conv_out (:167) and the class / stop heads (:169-182).

`decoder_fused.decoder_sequence` already walks the (level, timestep) grid by diagonals in the FORWARD pass (cell (i, t) needs
up(h[i-1][t]) and (h, c)[i][t-1], so the cells of a diagonal d = i + t are independent and their gate kernels go out as one
grouped launch), but leaves the backward to autograd: one node per cell, run one by one.  Here the backward is written out:

  * the reverse wavefront -- cell (i, t) needs d(up[i+1][t]) from cell (i+1, t) and (dh, dc) through the recurrence from cell
    (i, t+1), both on diagonal d + 1 -- so per diagonal there are three grouped steps: the gradient of every hidden state
    through its upsample + side max-pool, the LSTM pointwise backward, the data gradient of the gate conv;
  * everything that is independent of the recurrence runs ONCE over all T timesteps: conv_out forward / data gradient / weight
    gradient on the stacked [T][B] images (rsis_conv_out_seq_*: 3 launches instead of 3 T), the mask logits written straight into
    the (B, T, N) layout train.py:118 stacks them into, the sum over t of d(gates) for the time-invariant skip term
    (rsis_sum_leading), the time-batched weight gradients of the recurrent channels (as decoder_fused does).

Arithmetic and kernels per cell are those of the per-step path (tests/test_gpu_modules.py compares them); only the launch
schedule differs.  The per-timestep `RSIS.forward` stays for callers that thread the state themselves.
"""
import ctypes
import os

import torch

from . import _lib, decoder_fused, ops
from ._lib import check, int_array, lib, ptr, ptr_array, stream

ENABLED = [os.environ.get("RSIS_DECODER_SEQ", "1") != "0"]
# test hook: RECORD[0] = True keeps a copy of the arg-max pixels of the side max-pools of the last sequence in LAST["arg"] (per level,
# [T][B][hid] int32) -- what the per-step path's callers can read off the returned hidden states
RECORD, LAST = [False], {}


def _is_blk(t):
    return t.dtype == torch.bfloat16 and t.dim() == 5 and t.shape[-1] == 8


def _chan(t):
    """channels of a skip feature given as fp32 NCHW or as a blk tensor [B][C/8][H][W][8]"""
    return t.shape[1] * 8 if _is_blk(t) else t.shape[1]


def supported(decoder, skip_feats, T):
    """the explicit sequence covers the product configuration: concat skips, 3x3 gates, no dropout, the fused heads kernel"""
    n = len(decoder.clstm_list)
    if not (ENABLED[0] and decoder_fused.WAVEFRONT[0] and decoder_fused.FUSED_POOL[0] and decoder.fused and decoder.skip_mode == "concat" and decoder.dropout == 0 and decoder.dropout_cls == 0 and
            decoder.dropout_stop == 0 and len(skip_feats) == n and T >= 1 and "forward" not in decoder.__dict__):
        return False
    if not all(f.is_cuda and ((f.dtype == torch.float32 and f.dim() == 4) or _is_blk(f)) for f in skip_feats):
        return False
    if any(_is_blk(f) for f in skip_feats) and not blk_supported(decoder, skip_feats):
        return False
    if not all(c.kernel_size == 3 and c.padding == 1 for c in decoder.clstm_list):
        return False
    if decoder.conv_out.kernel_size != 3 or decoder.conv_out.padding != 1 or decoder.conv_out.bias is None:
        return False
    hs = [c.hidden_size for c in decoder.clstm_list]
    if sum(hs) != decoder.fc_class.weight.shape[1] or not ops.heads_supported([skip_feats[0]] * n, decoder.fc_class, decoder.fc_stop):
        return False
    for i, c in enumerate(decoder.clstm_list):      # the channel pyramid of model.py:98-106 (level i consumes up(h[i-1]) | skip[i])
        c_up = 0 if i == 0 else hs[i - 1]
        if c.input_size != c_up + _chan(skip_feats[i]):
            return False
    return True


class ShapeProbe(object):
    """stands in for a skip feature (blk: [B][C/8][H][W][8] bf16, else fp32 NCHW) where only its shape / dtype are consulted: lets the
    callers ask supported() BEFORE the encoder has run (train._blk_skips_ok)"""

    def __init__(self, B, C, H, W, blk):
        self.is_cuda = True
        self.dtype = torch.bfloat16 if blk else torch.float32
        self.shape = (B, C // 8, H, W, 8) if blk else (B, C, H, W)

    def dim(self):
        return len(self.shape)


def supported_for_input(decoder, skip_channels, B, H, W, T):
    """supported() for the skip features a (B, 3, H, W) input will produce when the encoder hands sk5..sk2 over as blk tensors and sk1
    as fp32 NCHW: THE decision whether runIter / test() ask the encoder for blk skip features -- the same predicate the sequence node
    applies afterwards, so the two cannot disagree."""
    sizes, h, w = [], H, W
    for _ in range(5):                      # stem / max-pool / strided 3x3 convs: every halving is ceil(n / 2)
        h, w = (h + 1) // 2, (w + 1) // 2
        sizes.append((h, w))
    sizes = sizes[::-1]                     # x5 .. x1
    if any(c % 8 for c in skip_channels[:4]):
        return False
    probes = [ShapeProbe(B, c, hh, ww, i < 4) for i, (c, (hh, ww)) in enumerate(zip(skip_channels, sizes))]
    return supported(decoder, probes, T) and blk_supported(decoder, probes)


class _Level(object):
    __slots__ = ("cell", "hid", "c_up", "c_skip", "H", "W", "hoist", "dyn", "G", "Hs", "Cs", "ACT", "UP", "KEY", "SIDE", "ARG", "skip")


def _conv_out_seq_ok(Cin, W):
    return Cin in (4, 8, 16) and W % 4 == 0


UPCONV = [os.environ.get("RSIS_UPCONV", "1") != "0"]       # the last level's upsample + conv_out as one op per direction (upconv_out.hip)


def _upconv_ok(L, last, H5, W5):
    return UPCONV[0] and L.rsis_upconv_out_supported(last.hid, last.H, last.W, H5, W5) == 1


def _upconv_bwd(L, d_masks, last, blk, co_w, dW, db, dside, T, B, H5, W5, DH_last):
    """the backward of the fused tail: DH_last <- the gradient of the last level's hidden states through conv_out and the upsample (plus
    the side max-pool gradient at its arg-max pixel); dW / db accumulate (fixed-order sums of per-block partials)"""
    nb = L.rsis_upconv_out_bwd_blocks(T, B, last.H, last.W)
    partial = torch.empty(nb * 80, dtype=torch.float32, device=d_masks.device)
    check(L.rsis_upconv_out_bwd(ptr(d_masks), ptr(last.Hs), 1 if blk else 0, ptr(co_w.detach()), ptr(DH_last), ptr(dW) if dW is not None else None,
                                ptr(db) if db is not None else None, ptr(dside), ptr(last.ARG), ptr(partial), T, B, last.hid, last.H, last.W, H5, W5,
                                stream()), "rsis_upconv_out_bwd")


def _heads_bwd_all_steps(L, levels, hs, n, T, B, Wc, Ws, probs_tb, dp_tb, ds_tb, dsides, hb):
    """backward of the class / stop heads: the T * B rows of all timesteps in ONE launch (every row is independent; the parameter
    gradients are sums over rows, which the kernel reduces itself) -- per timestep when the rows do not fit the kernel's LDS"""
    def call(rows, sl):
        return L.rsis_heads_bwd(ptr_array([lv.SIDE[sl] for lv in levels]), int_array(hs), n, rows, ptr(Wc.detach()), Wc.shape[0], ptr(Ws.detach()),
                                ptr(probs_tb[sl]), ptr(dp_tb[sl]) if dp_tb is not None else None, ptr(ds_tb[sl]) if ds_tb is not None else None,
                                ptr_array([ds[sl] for ds in dsides]), ptr(hb[0]), ptr(hb[1]), ptr(hb[2]), ptr(hb[3]), stream())
    rc = call(T * B, slice(None))
    if rc == 3:          # RSIS_ERR_UNSUPPORTED: more rows than the kernel's LDS holds
        for t in range(T):
            check(call(B, t), "rsis_heads_bwd")
    else:
        check(rc, "rsis_heads_bwd(all steps)")


def _heads_all_steps(L, levels, hs, n, rows, Wc, bc, Ws, bs, ncls, probs_tb, stop_tb):
    """the class / stop heads (model.py:169-182) of ALL timesteps in one launch: every (t, b) row is independent, the per-level key /
    feature / arg-max arrays are [T][B][hid] contiguous, i.e. T * B rows; the launch decodes the pooled keys (writes SIDE / ARG)"""
    check(L.rsis_heads_fwd_keys(ptr_array([v.KEY for v in levels]), ptr_array([v.SIDE for v in levels]), ptr_array([v.ARG for v in levels]),
                                int_array(hs), n, rows, ptr(Wc), ptr(bc), ncls, ptr(Ws), ptr(bs), ptr(probs_tb), ptr(stop_tb), stream()),
          "rsis_heads_fwd_keys(all steps)")


class _DecoderSeqFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, decoder, T, keep, *tensors):
        L = lib()
        n = len(decoder.clstm_list)
        feats = [t if t.is_contiguous() else t.contiguous() for t in tensors[:n]]
        params = tensors[n:]
        gates_w = [params[2 * i] for i in range(n)]
        gates_b = [params[2 * i + 1] for i in range(n)]
        co_w, co_b, Wc, bc, Ws, bs = params[2 * n:2 * n + 6]
        _lib.require_cuda_f32(*feats, *params)
        need_grad = bool(keep) and any(ctx.needs_input_grad)     # (grad mode is off inside forward: the caller says whether a backward may follow)
        dev = feats[0].device
        B = feats[0].shape[0]
        hs = [c.hidden_size for c in decoder.clstm_list]
        tot = sum(hs)
        f32 = dict(dtype=torch.float32, device=dev)
        # one zeroed key buffer for the side max-pools of the whole sequence (rsis_lstm_job.side_key), decoded by the heads launches
        KEY = torch.zeros(T * B * tot, dtype=torch.int64, device=dev)
        SIDE = torch.empty(T * B * tot, **f32)
        ARG = torch.empty(T * B * tot, dtype=torch.int32, device=dev)
        levels, off = [], 0
        for i, cell in enumerate(decoder.clstm_list):
            lv = _Level()
            lv.cell, lv.hid, lv.c_up, lv.c_skip = cell, hs[i], (0 if i == 0 else hs[i - 1]), feats[i].shape[1]
            lv.H, lv.W = feats[i].shape[2], feats[i].shape[3]
            lv.hoist, lv.dyn = decoder_fused._packs(cell, lv.c_up, lv.c_skip)
            # G_i = W[:, skip channels] * skip_i + b, once per iteration (the skip features do not depend on t)
            wp = lv.hoist.fwd(gates_w[i], gates_b[i])
            lv.G = torch.empty((B, 4 * lv.hid, lv.H, lv.W), **f32)
            check(L.rsis_conv2d_fwd(ptr_array([feats[i]]), int_array([lv.c_skip]), 1, B, lv.H, lv.W, ptr(wp), 4 * lv.hid, 3, 1, 1,
                                    ptr(lv.hoist.bias_p), None, ptr(lv.G), lv.H, lv.W, ops.FORCE_TILE[0] + (100 if need_grad else 0), lv.hoist.dtype,
                                    stream()),
                  "rsis_conv2d_fwd(hoist)")
            lv.Hs = torch.empty((T, B, lv.hid, lv.H, lv.W), **f32)
            lv.Cs = torch.empty((T, B, lv.hid, lv.H, lv.W), **f32)
            lv.ACT = torch.empty((T, B, 4 * lv.hid, lv.H, lv.W), **f32) if need_grad else None
            lv.UP = torch.empty((T, B, lv.c_up, lv.H, lv.W), **f32) if lv.c_up > 0 else None
            m = T * B * lv.hid
            lv.KEY, lv.SIDE, lv.ARG = KEY[off:off + m].view(T, B, lv.hid), SIDE[off:off + m].view(T, B, lv.hid), ARG[off:off + m].view(T, B, lv.hid)
            off += m
            levels.append(lv)
        last = levels[-1]
        H5, W5 = 2 * last.H, 2 * last.W
        upc = _upconv_ok(L, last, H5, W5)
        UP5 = None if upc else torch.empty((T, B, last.hid, H5, W5), **f32)            # model.py:163-164, all timesteps
        ncls = Wc.shape[0]
        probs_tb = torch.empty((T, B, ncls), **f32)
        stop_tb = torch.empty((T, B, 1), **f32)
        wps = [lv.dyn.fwd(gates_w[i]) for i, lv in enumerate(levels)]
        Wc_d, bc_d, Ws_d, bs_d = Wc.detach(), bc.detach(), Ws.detach(), bs.detach()

        for d in range(T + n - 1):
            cells = [(i, d - i) for i in range(n) if 0 <= d - i < T]
            jobs = (_lib.LstmJob * len(cells))()
            for j, (i, t) in zip(jobs, cells):
                lv = levels[i]
                srcs = ([lv.UP[t]] if lv.c_up > 0 else []) + ([lv.Hs[t - 1]] if t > 0 else [])
                j.nsrc = len(srcs)
                for k, s in enumerate(srcs):
                    j.src[k], j.Csrc[k] = s.data_ptr(), s.shape[1]
                (j.B, j.H, j.W, j.Wp, j.bias_packed, j.addend, j.c_prev, j.h_out, j.c_out, j.act_out, j.hid, j.ks, j.pad, j.tile, j.dtype,
                 j.side_key) = (B, lv.H, lv.W, wps[i].data_ptr(), None, lv.G.data_ptr(), lv.Cs[t - 1].data_ptr() if t > 0 else None,
                                lv.Hs[t].data_ptr(), lv.Cs[t].data_ptr(), lv.ACT[t].data_ptr() if lv.ACT is not None else None, lv.hid, 3, 1,
                                ops.FORCE_TILE[0], lv.dyn.dtype, lv.KEY[t].data_ptr())
            check(L.rsis_convlstm_fwd_batch(jobs, len(cells), stream()), "rsis_convlstm_fwd_batch")
            for i, t in cells:
                lv = levels[i]
                if i + 1 < n:
                    nx = levels[i + 1]
                    if (nx.H, nx.W) == (lv.H, lv.W):          # align-corners resize to the same size is the identity (ops.upsample_bilinear_ac)
                        nx.UP[t].copy_(lv.Hs[t])
                    else:
                        check(L.rsis_upsample_bilinear_ac_fwd(ptr(lv.Hs[t]), ptr(nx.UP[t]), B * lv.hid, lv.H, lv.W, nx.H, nx.W, stream()),
                              "rsis_upsample_fwd")
                # (the x2 upsample of the last level, model.py:163-164, feeds conv_out only, and the heads only the losses: both run once
                #  over all T steps after the loop)
        if not upc:
            check(L.rsis_upsample_bilinear_ac_fwd(ptr(last.Hs), ptr(UP5), T * B * last.hid, last.H, last.W, H5, W5, stream()), "rsis_upsample_fwd(all steps)")
        _heads_all_steps(L, levels, hs, n, T * B, Wc_d, bc_d, Ws_d, bs_d, ncls, probs_tb, stop_tb)
        # conv_out (model.py:167) on every timestep at once, logits straight into (B, T, N)
        out_masks = torch.empty((B, T, H5 * W5), **f32)
        co_pack = decoder.conv_out._pack
        co_wp = None if upc else co_pack.fwd(co_w)
        seq_ok = _conv_out_seq_ok(last.hid, W5)
        if upc:       # upsample + conv_out in one pass, the upsampled tensor never formed
            check(L.rsis_upconv_out_fwd(ptr(last.Hs), 0, ptr(co_w.detach()), ptr(co_b.detach()), ptr(out_masks), T, B, last.hid, last.H, last.W, H5, W5,
                                        stream()), "rsis_upconv_out_fwd")
        elif seq_ok:
            check(L.rsis_conv_out_seq_fwd(ptr(UP5), ptr(co_wp), ptr(co_b.detach()), ptr(out_masks), T, B, last.hid, H5, W5, stream()),
                  "rsis_conv_out_seq_fwd")
        else:
            tmp = torch.empty((B, 1, H5, W5), **f32)
            for t in range(T):
                check(L.rsis_conv2d_fwd(ptr_array([UP5[t]]), int_array([last.hid]), 1, B, H5, W5, ptr(co_wp), 1, 3, 1, 1, ptr(co_b.detach()), None,
                                        ptr(tmp), H5, W5, ops.FORCE_TILE[0], co_pack.dtype, stream()), "rsis_conv2d_fwd(conv_out)")
                out_masks[:, t].copy_(tmp.view(B, -1))
        out_probs = probs_tb.transpose(0, 1).contiguous()            # (B, T, C)   train.py:119
        out_stops = stop_tb.transpose(0, 1).contiguous()             # (B, T, 1)   train.py:120
        hidden = []
        for lv in levels:
            hidden += [lv.Hs[T - 1], lv.Cs[T - 1]]
        if RECORD[0]:
            LAST["arg"] = [lv.ARG.clone() for lv in levels]
        ctx.set_materialize_grads(False)
        if need_grad:
            ctx.decoder, ctx.T, ctx.levels, ctx.seq_ok, ctx.upc = decoder, T, levels, seq_ok, upc
            ctx.UP5, ctx.probs_tb, ctx.feats = UP5, probs_tb, feats
            ctx.params = params
        else:
            for lv in levels:
                lv.G = lv.ACT = lv.UP = None
        return (out_masks, out_probs, out_stops) + tuple(hidden)

    @staticmethod
    def backward(ctx, d_masks, d_probs, d_stops, *d_hidden):
        L = lib()
        decoder, T, levels = ctx.decoder, ctx.T, ctx.levels
        n = len(levels)
        feats, params = ctx.feats, ctx.params
        gates_w = [params[2 * i] for i in range(n)]
        gates_b = [params[2 * i + 1] for i in range(n)]
        co_w, co_b, Wc, bc, Ws, bs = params[2 * n:2 * n + 6]
        need = ctx.needs_input_grad            # (decoder, T, keep, feats..., params...)
        need_feat = [need[3 + i] for i in range(n)]
        need_par = [need[3 + n + k] for k in range(len(params))]
        dev = feats[0].device
        B = feats[0].shape[0]
        hs = [lv.hid for lv in levels]
        f32 = dict(dtype=torch.float32, device=dev)
        last = levels[-1]
        H5, W5 = 2 * last.H, 2 * last.W
        UP5 = ctx.UP5
        grads_par = [None] * len(params)

        def target(k):
            """(buffer the kernels accumulate parameter k's gradient into, is it the flat gradient buffer)"""
            p = params[k]
            t = ops._direct_target(p)
            if t is not None:
                return t, True
            g = torch.zeros_like(p)
            grads_par[k] = g
            return g, False

        # ---- conv_out: data gradient of all T steps in one launch, weight + bias gradient in another ----
        upc = ctx.upc
        dUP5 = None if upc else torch.empty_like(UP5)
        co_pack = decoder.conv_out._pack
        have_masks = d_masks is not None       # (no gradient on the logits: conv_out's parameters get none either, as on the unfused path)
        if upc:         # (the whole tail runs below, once the side gradients of the heads are known)
            d_masks = torch.zeros((B, T, H5 * W5), **f32) if d_masks is None else (d_masks if d_masks.is_contiguous() else d_masks.contiguous())
        elif d_masks is None:
            dUP5.zero_()
        else:
            d_masks = d_masks if d_masks.is_contiguous() else d_masks.contiguous()
            co_wd = co_pack.dgrad(co_w)
            kw, kb = 2 * n, 2 * n + 1
            dW = db = None
            if need_par[kw]:
                dW, _ = target(kw)
            if need_par[kb]:
                db, _ = target(kb)
            if ctx.seq_ok:
                check(L.rsis_conv_out_seq_dgrad(ptr(d_masks), ptr(co_wd), ptr(dUP5), T, B, last.hid, H5, W5, stream()), "rsis_conv_out_seq_dgrad")
                if dW is not None:
                    check(L.rsis_conv_out_seq_wgrad(ptr(d_masks), ptr(UP5), ptr(dW), ptr(db), T, B, last.hid, H5, W5, stream()),
                          "rsis_conv_out_seq_wgrad")
                elif db is not None:
                    db += d_masks.sum()
            else:
                for t in range(T):
                    dy = d_masks[:, t].contiguous().view(B, 1, H5, W5)
                    check(L.rsis_conv2d_dgrad(ptr(dy), B, 1, H5, W5, ptr(co_wd), co_pack.cin, 3, 1, 1, ptr_array([dUP5[t]]), int_array([last.hid]), 1,
                                              H5, W5, None, ops.FORCE_TILE[0], co_pack.dtype, stream()), "rsis_conv2d_dgrad(conv_out)")
                    if dW is not None:
                        check(L.rsis_conv2d_wgrad(ptr(dy), ptr(UP5[t]), ptr(dW), B, last.hid, H5, W5, 1, H5, W5, 3, 1, 1, last.hid, 0, 0,
                                                  co_pack.dtype, stream()), "rsis_conv2d_wgrad(conv_out)")
                    if db is not None:
                        check(L.rsis_bias_grad(ptr(dy), ptr(db), B, 1, H5 * W5, 0, stream()), "rsis_bias_grad(conv_out)")
        # ---- heads: per timestep (the parameter gradients accumulate over t) ----
        tot = sum(hs)
        DSIDE = torch.empty(T * B * tot, **f32)
        dsides, off = [], 0
        for lv in levels:
            m = T * B * lv.hid
            dsides.append(DSIDE[off:off + m].view(T, B, lv.hid))
            off += m
        if d_probs is None and d_stops is None:
            DSIDE.zero_()
        else:
            dp_tb = d_probs.transpose(0, 1).contiguous() if d_probs is not None else None
            ds_tb = d_stops.transpose(0, 1).contiguous() if d_stops is not None else None
            hb = [target(2 * n + 2 + k)[0] if need_par[2 * n + 2 + k] else None for k in range(4)]
            _heads_bwd_all_steps(L, levels, hs, n, T, B, Wc, Ws, ctx.probs_tb, dp_tb, ds_tb, dsides, hb)
        # ---- reverse wavefront ----
        DA = [torch.empty_like(lv.ACT) for lv in levels]
        DH = [torch.empty((B, lv.hid, lv.H, lv.W), **f32) for lv in levels]                  # gradient of h[i][t] from above (upsample + pool)
        DHP = [torch.empty((B, lv.hid, lv.H, lv.W), **f32) for lv in levels]                 # ... through the recurrence, from step t + 1
        DC = [[torch.empty((B, lv.hid, lv.H, lv.W), **f32) for _ in range(2)] for lv in levels]
        DUP = [torch.empty((B, lv.c_up, lv.H, lv.W), **f32) if lv.c_up > 0 else None for lv in levels]
        dpk = [decoder_fused.dyn_dgrad_pack(lv.cell, lv.c_up, lv.c_skip, lv.H, lv.W) for lv in levels]
        wds = [dpk[i].dgrad(gates_w[i]) for i, lv in enumerate(levels)]
        # the last level's hidden states receive their gradient from conv_out only (no level above): all T steps in one launch
        DH_last = torch.empty((T, B, last.hid, last.H, last.W), **f32)
        if upc:
            kw, kb = 2 * n, 2 * n + 1
            _upconv_bwd(L, d_masks, last, False, co_w, target(kw)[0] if need_par[kw] and have_masks else None,
                        target(kb)[0] if need_par[kb] and have_masks else None, dsides[n - 1],
                        T, B, H5, W5, DH_last)
        else:
            check(L.rsis_upsample_maxpool_bwd(ptr(dUP5), ptr(dsides[n - 1]), ptr(last.ARG), ptr(DH_last), T * B * last.hid, last.H, last.W, H5, W5,
                                              stream()), "rsis_upsample_maxpool_bwd(all steps)")
        for d in range(T + n - 2, -1, -1):
            cells = [(i, d - i) for i in range(n) if 0 <= d - i < T]
            for i, t in cells:          # gradient reaching h[i][t] through its upsample into the next level and its side max-pool
                lv = levels[i]
                if i + 1 == n:
                    continue
                nx = levels[i + 1]
                dy, Ho, Wo = DUP[i + 1], nx.H, nx.W
                if (Ho, Wo) == (lv.H, lv.W):
                    DH[i].copy_(dy)
                    check(L.rsis_global_maxpool_bwd_add(ptr(dsides[i][t]), ptr(lv.ARG[t]), ptr(DH[i]), B * lv.hid, lv.H * lv.W, stream()),
                          "rsis_global_maxpool_bwd_add")
                else:
                    check(L.rsis_upsample_maxpool_bwd(ptr(dy), ptr(dsides[i][t]), ptr(lv.ARG[t]), ptr(DH[i]), B * lv.hid, lv.H, lv.W, Ho, Wo,
                                                      stream()), "rsis_upsample_maxpool_bwd")
            # clstm.py:47-58 backwards: d(gates), dc_{t-1} of the diagonal's cells -- ONE grouped launch
            lb = (_lib.LstmBwdJob * len(cells))()
            keep = []
            for j, (i, t) in zip(lb, cells):
                lv = levels[i]
                if t == T - 1:      # gradients a caller put on the returned final state (none in runIter: train.py never reads it)
                    dh2 = d_hidden[2 * i].contiguous() if d_hidden[2 * i] is not None else None
                    dcn = d_hidden[2 * i + 1].contiguous() if d_hidden[2 * i + 1] is not None else None
                    keep += [dh2, dcn]
                else:
                    dh2, dcn = DHP[i], DC[i][(t + 1) & 1]
                dh = DH_last[t] if i + 1 == n else DH[i]
                (j.dh, j.dh2, j.dc_next, j.act, j.c_prev, j.c, j.da, j.dc_prev, j.B, j.hid, j.HW) = (
                    ptr(dh), ptr(dh2), ptr(dcn), ptr(lv.ACT[t]), ptr(lv.Cs[t - 1]) if t > 0 else None, ptr(lv.Cs[t]), ptr(DA[i][t]),
                    ptr(DC[i][t & 1]) if t > 0 else None, B, lv.hid, lv.H * lv.W)
            check(L.rsis_convlstm_bwd_gates_batch(lb, len(cells), stream()), "rsis_convlstm_bwd_gates_batch")
            # data gradients of the gate convs: d(up[i][t]) for the level below, dh[i][t-1] through the recurrence -- ONE grouped launch
            dgc = [(i, t) for i, t in cells if levels[i].c_up > 0 or t > 0]
            if dgc:
                dg = (_lib.DgradJob * len(dgc))()
                for j, (i, t) in zip(dg, dgc):
                    lv = levels[i]
                    dxs = ([DUP[i]] if lv.c_up > 0 else []) + ([DHP[i]] if t > 0 else [])
                    (j.dy, j.B, j.Cout, j.Hy, j.Wy, j.Wd, j.Cin_packed, j.ks, j.stride, j.pad, j.ndst, j.Hx, j.Wx, j.addend, j.tile, j.dtype) = (
                        ptr(DA[i][t]), B, 4 * lv.hid, lv.H, lv.W, ptr(wds[i]), lv.dyn.cin, 3, 1, 1, len(dxs), lv.H, lv.W, None, ops.FORCE_TILE[0],
                        dpk[i].dtype)
                    for k, x in enumerate(dxs):
                        j.dx[k], j.Cdx[k] = x.data_ptr(), x.shape[1]
                check(L.rsis_conv2d_dgrad_batch(dg, len(dgc), stream()), "rsis_conv2d_dgrad_batch")
        # ---- per level, once: the time-invariant skip term and the time-batched weight gradients ----
        dfeats = [None] * n
        for i, lv in enumerate(levels):
            kw, kb = 2 * i, 2 * i + 1
            hid, H, W = lv.hid, lv.H, lv.W
            if T == 1:
                dG = DA[i][0]
            else:
                dG = torch.empty_like(lv.G)
                check(L.rsis_sum_leading(ptr(DA[i]), ptr(dG), T, dG.numel(), stream()), "rsis_sum_leading")
            if need_feat[i]:
                dfeats[i] = torch.empty_like(feats[i])
                check(L.rsis_conv2d_dgrad(ptr(dG), B, 4 * hid, H, W, ptr(lv.hoist.dgrad(gates_w[i])), lv.hoist.cin, 3, 1, 1, ptr_array([dfeats[i]]),
                                          int_array([lv.c_skip]), 1, H, W, None, ops.FORCE_TILE[0], lv.hoist.dtype, stream()), "rsis_conv2d_dgrad(hoist)")
            if need_par[kb]:
                db, _ = target(kb)
                check(L.rsis_bias_grad(ptr(dG), ptr(db), B, 4 * hid, H * W, hid, stream()), "rsis_bias_grad(hoist)")
            if need_par[kw]:
                dW, direct = target(kw)
                Ctot = gates_w[i].shape[1]
                h_off = lv.c_up + lv.c_skip
                ops.wgrad_launch(L, dG, feats[i], dW, B, lv.c_skip, H, W, 4 * hid, H, W, 3, 1, 1, Ctot, lv.c_up, hid, lv.hoist.dtype,
                                 "rsis_conv2d_wgrad(hoist)", direct)
                dt = lv.dyn.dtype
                # bf16, rows that are not a multiple of 4 pixels (the 7 / 14-pixel levels of a 224 x 224 input): channel-blocked bf16 copies
                # of the (small) stacks and the whole-cell loader, see decoder_fused._StepFn.backward
                as_blk = (dt == ops.DTYPE_BF16 and decoder_fused._BLK_WGRAD[0] and W % 4 != 0 and hid % 8 == 0 and lv.c_up % 8 == 0)
                if as_blk:
                    da_b = ops.blk_from_nchw(DA[i].view(T * B, 4 * hid, H, W))
                    if lv.c_up > 0:
                        ops.wgrad_launch(L, da_b, ops.blk_from_nchw(lv.UP.view(T * B, lv.c_up, H, W)), dW, T * B, lv.c_up, H, W, 4 * hid, H, W, 3, 1, 1,
                                         Ctot, 0, hid, ops.DTYPE_BF16_BLK, "rsis_conv2d_wgrad(batched up, blk)", direct)
                    if T > 1:
                        ops.wgrad_launch(L, da_b[B:], ops.blk_from_nchw(lv.Hs[:T - 1].reshape((T - 1) * B, hid, H, W)), dW, (T - 1) * B, hid, H, W,
                                         4 * hid, H, W, 3, 1, 1, Ctot, h_off, hid, ops.DTYPE_BF16_BLK, "rsis_conv2d_wgrad(batched h, blk)", direct)
                else:
                    if lv.c_up > 0:
                        ops.wgrad_launch(L, DA[i], lv.UP, dW, T * B, lv.c_up, H, W, 4 * hid, H, W, 3, 1, 1, Ctot, 0, hid, dt,
                                         "rsis_conv2d_wgrad(batched up)", direct)
                    if T > 1:
                        ops.wgrad_launch(L, DA[i][1], lv.Hs, dW, (T - 1) * B, hid, H, W, 4 * hid, H, W, 3, 1, 1, Ctot, h_off, hid, dt,
                                         "rsis_conv2d_wgrad(batched h)", direct)
        for lv in levels:
            lv.G = lv.Hs = lv.Cs = lv.ACT = lv.UP = None
        ctx.levels = ctx.UP5 = ctx.feats = ctx.params = ctx.decoder = None
        return (None, None, None) + tuple(dfeats) + tuple(grads_par)


# RSIS_DECODER_BLK=0: keep fp32 NCHW storage in the decoder under -dtype bf16 (bf16 operands only, the round-3 path)
BLK_ENABLED = [os.environ.get("RSIS_DECODER_BLK", "1") != "0"]


def blk_supported(decoder, skip_feats):
    """the decoder's storage half of `-dtype bf16`: every tensor of the recurrence as channel-blocked bf16 (csrc/conv_blk_dec.hip,
    blk_dec.hip) -- needs bf16 cells, whole 8-channel cells at every level and the 8-channel conv_out (hidden_size % 128 == 0)"""
    if not BLK_ENABLED[0]:
        return False
    hs = [c.hidden_size for c in decoder.clstm_list]
    if not all(getattr(c, "dtype", ops.DTYPE_F32) == ops.DTYPE_BF16 for c in decoder.clstm_list):
        return False
    if any(h % 8 for h in hs) or any(_chan(f) % 8 for f in skip_feats) or hs[-1] != 8:
        return False
    return (2 * skip_feats[-1].shape[3]) % 4 == 0


class _DecoderSeqBlkFn(torch.autograd.Function):
    """_DecoderSeqFn with every tensor of the recurrence stored channel-blocked bf16: hidden states, saved gates, up-sampled inputs,
    the hoisted gate terms and every gradient of them; the cell state and its gradient stay fp32.  Per diagonal ONE grouped launch
    per kind of work (gate convs, upsamples; backward: upsample transposes, pointwise LSTM backward, data gradients)."""

    @staticmethod
    def forward(ctx, decoder, T, keep, want_hidden, *tensors):
        L = lib()
        n = len(decoder.clstm_list)
        feats = [t if t.is_contiguous() else t.contiguous() for t in tensors[:n]]      # fp32 NCHW, or already blk (the encoder's blk skip path)
        params = tensors[n:]
        gates_w = [params[2 * i] for i in range(n)]
        gates_b = [params[2 * i + 1] for i in range(n)]
        co_w, co_b, Wc, bc, Ws, bs = params[2 * n:2 * n + 6]
        _lib.require_cuda_f32(*[f for f in feats if not _is_blk(f)], *params)
        need_grad = bool(keep) and any(ctx.needs_input_grad)
        dev = feats[0].device
        B = feats[0].shape[0]
        hs = [c.hidden_size for c in decoder.clstm_list]
        tot = sum(hs)
        f32 = dict(dtype=torch.float32, device=dev)
        b16 = dict(dtype=torch.bfloat16, device=dev)
        KEY = torch.zeros(T * B * tot, dtype=torch.int64, device=dev)
        SIDE = torch.empty(T * B * tot, **f32)
        ARG = torch.empty(T * B * tot, dtype=torch.int32, device=dev)
        levels, off, hoist_jobs = [], 0, []
        for i, cell in enumerate(decoder.clstm_list):
            lv = _Level()
            lv.cell, lv.hid, lv.c_up, lv.c_skip = cell, hs[i], (0 if i == 0 else hs[i - 1]), _chan(feats[i])
            lv.H, lv.W = feats[i].shape[2], feats[i].shape[3]
            lv.hoist, lv.dyn = decoder_fused._packs(cell, lv.c_up, lv.c_skip)
            lv.skip = feats[i] if _is_blk(feats[i]) else ops.blk_from_nchw(feats[i])
            lv.G = torch.empty((B, 4 * lv.hid // 8, lv.H, lv.W, 8), **b16)
            hoist_jobs.append(ops.blk_conv_job([lv.skip], lv.hoist.fwd(gates_w[i], gates_b[i]), 4 * lv.hid, bias=lv.hoist.bias_p, dsts=[lv.G]))
            lv.Hs = torch.empty((T, B, lv.hid // 8, lv.H, lv.W, 8), **b16)
            lv.Cs = torch.empty((T, B, lv.hid, lv.H, lv.W), **f32)
            lv.ACT = torch.empty((T, B, 4 * lv.hid // 8, lv.H, lv.W, 8), **b16) if need_grad else None
            lv.UP = torch.empty((T, B, lv.c_up // 8, lv.H, lv.W, 8), **b16) if lv.c_up > 0 else None
            m = T * B * lv.hid
            lv.KEY, lv.SIDE, lv.ARG = KEY[off:off + m].view(T, B, lv.hid), SIDE[off:off + m].view(T, B, lv.hid), ARG[off:off + m].view(T, B, lv.hid)
            off += m
            levels.append(lv)
        ops.blk_conv3x3_batch(hoist_jobs)          # G_i = W[:, skip channels] * skip_i + b of all five levels: one grouped launch
        last = levels[-1]
        H5, W5 = 2 * last.H, 2 * last.W
        upc = _upconv_ok(L, last, H5, W5)
        UP5 = None if upc else torch.empty((T, B, 1, H5, W5, 8), **b16)
        ncls = Wc.shape[0]
        probs_tb = torch.empty((T, B, ncls), **f32)
        stop_tb = torch.empty((T, B, 1), **f32)
        wps = [lv.dyn.fwd(gates_w[i]) for i, lv in enumerate(levels)]
        Wc_d, bc_d, Ws_d, bs_d = Wc.detach(), bc.detach(), Ws.detach(), bs.detach()
        for d in range(T + n - 1):
            cells = [(i, d - i) for i in range(n) if 0 <= d - i < T]
            jobs, ups = [], []
            for i, t in cells:
                lv = levels[i]
                srcs = ([lv.UP[t]] if lv.c_up > 0 else []) + ([lv.Hs[t - 1]] if t > 0 else [])
                jobs.append(ops.blk_conv_job(srcs, wps[i], 4 * lv.hid, addend=lv.G, hid=lv.hid, c_prev=lv.Cs[t - 1] if t > 0 else None, c_out=lv.Cs[t],
                                             h_out=lv.Hs[t], act_out=lv.ACT[t] if lv.ACT is not None else None, side_key=lv.KEY[t],
                                             shape=(B, lv.H, lv.W)))
                if i + 1 < n:        # (the last level's x2 upsample feeds conv_out only: all T steps in one launch after the loop)
                    ups.append(ops.blk_resize_job(lv.Hs[t], levels[i + 1].UP[t]))
            ops.blk_conv3x3_batch(jobs)
            if ups:
                ops.blk_upsample_fwd_batch(ups)
        _heads_all_steps(L, levels, hs, n, T * B, Wc_d, bc_d, Ws_d, bs_d, ncls, probs_tb, stop_tb)
        out_masks = torch.empty((B, T, H5 * W5), **f32)
        if upc:       # upsample + conv_out in one pass, the upsampled tensor never formed (nor rounded to bf16)
            check(L.rsis_upconv_out_fwd(ptr(last.Hs), 1, ptr(co_w.detach()), ptr(co_b.detach()), ptr(out_masks), T, B, last.hid, last.H, last.W, H5, W5,
                                        stream()), "rsis_upconv_out_fwd")
        else:
            ops.blk_upsample_fwd_batch([ops.blk_resize_job(last.Hs.view(T * B, last.hid // 8, last.H, last.W, 8), UP5.view(T * B, 1, H5, W5, 8))])
            check(L.rsis_blk_conv_out_seq_fwd(ptr(UP5), ptr(co_w.detach()), ptr(co_b.detach()), ptr(out_masks), T, B, H5, W5, stream()),
                  "rsis_blk_conv_out_seq_fwd")
        out_probs = probs_tb.transpose(0, 1).contiguous()
        out_stops = stop_tb.transpose(0, 1).contiguous()
        hidden = []
        if want_hidden:
            for lv in levels:
                hidden += [ops.blk_to_nchw(lv.Hs[T - 1]), lv.Cs[T - 1]]
        if RECORD[0]:
            LAST["arg"] = [lv.ARG.clone() for lv in levels]
        ctx.set_materialize_grads(False)
        if need_grad:
            ctx.decoder, ctx.T, ctx.levels = decoder, T, levels
            ctx.UP5, ctx.probs_tb, ctx.feats, ctx.params, ctx.upc = UP5, probs_tb, feats, params, upc
        else:
            for lv in levels:
                lv.G = lv.ACT = lv.UP = lv.skip = None
        return (out_masks, out_probs, out_stops) + tuple(hidden)

    @staticmethod
    def backward(ctx, d_masks, d_probs, d_stops, *d_hidden):
        L = lib()
        decoder, T, levels = ctx.decoder, ctx.T, ctx.levels
        n = len(levels)
        feats, params = ctx.feats, ctx.params
        gates_w = [params[2 * i] for i in range(n)]
        co_w, co_b, Wc, bc, Ws, bs = params[2 * n:2 * n + 6]
        need = ctx.needs_input_grad            # (decoder, T, keep, want_hidden, feats..., params...)
        need_feat = [need[4 + i] for i in range(n)]
        need_par = [need[4 + n + k] for k in range(len(params))]
        dev = feats[0].device
        B = feats[0].shape[0]
        hs = [lv.hid for lv in levels]
        f32 = dict(dtype=torch.float32, device=dev)
        b16 = dict(dtype=torch.bfloat16, device=dev)
        last = levels[-1]
        H5, W5 = 2 * last.H, 2 * last.W
        UP5 = ctx.UP5
        grads_par = [None] * len(params)

        def target(k):
            p = params[k]
            t = ops._direct_target(p)
            if t is not None:
                return t, True
            g = torch.zeros_like(p)
            grads_par[k] = g
            return g, False

        upc = ctx.upc
        dUP5 = None if upc else torch.empty_like(UP5)
        kw, kb = 2 * n, 2 * n + 1
        have_masks = d_masks is not None       # (no gradient on the logits: conv_out's parameters get none either, as on the unfused path)
        if upc:         # (the whole tail runs below, once the side gradients of the heads are known)
            d_masks = torch.zeros((B, T, H5 * W5), **f32) if d_masks is None else (d_masks if d_masks.is_contiguous() else d_masks.contiguous())
        elif d_masks is None:
            dUP5.zero_()
        else:
            d_masks = d_masks if d_masks.is_contiguous() else d_masks.contiguous()
            check(L.rsis_blk_conv_out_seq_dgrad(ptr(d_masks), ptr(co_w.detach()), ptr(dUP5), T, B, H5, W5, stream()), "rsis_blk_conv_out_seq_dgrad")
            if need_par[kw] or need_par[kb]:
                dW = target(kw)[0] if need_par[kw] else torch.zeros_like(co_w)
                db = target(kb)[0] if need_par[kb] else None
                check(L.rsis_blk_conv_out_seq_wgrad(ptr(d_masks), ptr(UP5), ptr(dW), ptr(db), T, B, H5, W5, stream()), "rsis_blk_conv_out_seq_wgrad")
        tot = sum(hs)
        DSIDE = torch.empty(T * B * tot, **f32)
        dsides, off = [], 0
        for lv in levels:
            m = T * B * lv.hid
            dsides.append(DSIDE[off:off + m].view(T, B, lv.hid))
            off += m
        if d_probs is None and d_stops is None:
            DSIDE.zero_()
        else:
            dp_tb = d_probs.transpose(0, 1).contiguous() if d_probs is not None else None
            ds_tb = d_stops.transpose(0, 1).contiguous() if d_stops is not None else None
            hb = [target(2 * n + 2 + k)[0] if need_par[2 * n + 2 + k] else None for k in range(4)]
            _heads_bwd_all_steps(L, levels, hs, n, T, B, Wc, Ws, ctx.probs_tb, dp_tb, ds_tb, dsides, hb)
        DA = [torch.empty_like(lv.ACT) for lv in levels]
        DH = [torch.empty((B, lv.hid // 8, lv.H, lv.W, 8), **b16) for lv in levels]
        DHP = [torch.empty((B, lv.hid // 8, lv.H, lv.W, 8), **b16) for lv in levels]
        DC = [[torch.empty((B, lv.hid, lv.H, lv.W), **f32) for _ in range(2)] for lv in levels]
        DUP = [torch.empty((B, lv.c_up // 8, lv.H, lv.W, 8), **b16) if lv.c_up > 0 else None for lv in levels]
        wds = [lv.dyn.dgrad(gates_w[i]) for i, lv in enumerate(levels)]
        dhf = [ops.blk_from_nchw(d_hidden[2 * i].contiguous()) if (len(d_hidden) > 2 * i and d_hidden[2 * i] is not None) else None for i in range(n)]
        dcf = [d_hidden[2 * i + 1].contiguous() if (len(d_hidden) > 2 * i + 1 and d_hidden[2 * i + 1] is not None) else None for i in range(n)]
        # the last level's hidden states receive their gradient from conv_out only: all T steps in one launch
        DH_last = torch.empty((T, B, last.hid // 8, last.H, last.W, 8), **b16)
        if upc:
            _upconv_bwd(L, d_masks, last, True, co_w, target(kw)[0] if need_par[kw] and have_masks else None,
                        target(kb)[0] if need_par[kb] and have_masks else None, dsides[n - 1],
                        T, B, H5, W5, DH_last)
        else:
            ops.blk_upsample_bwd_batch([ops.blk_resize_job(dUP5.view(T * B, 1, H5, W5, 8), DH_last.view(T * B, last.hid // 8, last.H, last.W, 8),
                                                            dsides[n - 1], last.ARG, backward=True)])
        for d in range(T + n - 2, -1, -1):
            cells = [(i, d - i) for i in range(n) if 0 <= d - i < T]
            ups, lbs, dgs = [], [], []
            for i, t in cells:
                lv = levels[i]
                if i + 1 < n:
                    ups.append(ops.blk_resize_job(DUP[i + 1], DH[i], dsides[i][t], lv.ARG[t], backward=True))
                dh = DH[i] if i + 1 < n else DH_last[t]
                dh2, dcn = (dhf[i], dcf[i]) if t == T - 1 else (DHP[i], DC[i][(t + 1) & 1])
                lbs.append(ops.blk_lstm_bwd_job(dh, dh2, dcn, lv.ACT[t], lv.Cs[t - 1] if t > 0 else None, lv.Cs[t], DA[i][t],
                                                DC[i][t & 1] if t > 0 else None))
                dxs = ([DUP[i]] if lv.c_up > 0 else []) + ([DHP[i]] if t > 0 else [])
                if dxs:
                    dgs.append(ops.blk_conv_job([DA[i][t]], wds[i], sum(x.shape[1] for x in dxs) * 8, cpack=lv.dyn.cin, dsts=dxs))
            if ups:
                ops.blk_upsample_bwd_batch(ups)
            ops.blk_lstm_bwd_batch(lbs)
            if dgs:
                ops.blk_conv3x3_batch(dgs)
        dfeats = [None] * n
        dG, dskip_jobs = [], []
        for i, lv in enumerate(levels):
            if T == 1:
                dG.append(DA[i][0])
            else:
                g = torch.empty_like(lv.G)
                check(L.rsis_blk_sum_leading(ptr(DA[i]), ptr(g), T, g.numel() // 8, stream()), "rsis_blk_sum_leading")
                dG.append(g)
        dskips = [None] * n
        for i, lv in enumerate(levels):
            if need_feat[i]:
                dskips[i] = torch.empty_like(lv.skip)
                dskip_jobs.append(ops.blk_conv_job([dG[i]], lv.hoist.dgrad(gates_w[i]), lv.c_skip, cpack=lv.hoist.cin, dsts=[dskips[i]]))
        if dskip_jobs:
            ops.blk_conv3x3_batch(dskip_jobs)       # the data gradients of the five skip terms: one grouped launch
        for i, lv in enumerate(levels):
            kw, kb = 2 * i, 2 * i + 1
            hid, H, W = lv.hid, lv.H, lv.W
            if dskips[i] is not None:
                dfeats[i] = dskips[i] if _is_blk(feats[i]) else ops.blk_to_nchw(dskips[i])
            if need_par[kb]:
                db, _ = target(kb)
                check(L.rsis_blk_bias_grad(ptr(dG[i]), ptr(db), B, 4 * hid, H * W, hid, stream()), "rsis_blk_bias_grad")
            if need_par[kw]:
                dW, direct = target(kw)
                Ctot = gates_w[i].shape[1]
                h_off = lv.c_up + lv.c_skip
                blk = ops.DTYPE_BF16_BLK
                ops.wgrad_launch(L, dG[i], lv.skip, dW, B, lv.c_skip, H, W, 4 * hid, H, W, 3, 1, 1, Ctot, lv.c_up, hid, blk, "rsis_conv2d_wgrad(hoist, blk)",
                                 direct)
                if lv.c_up > 0:
                    ops.wgrad_launch(L, DA[i], lv.UP, dW, T * B, lv.c_up, H, W, 4 * hid, H, W, 3, 1, 1, Ctot, 0, hid, blk,
                                     "rsis_conv2d_wgrad(batched up, blk)", direct)
                if T > 1:
                    ops.wgrad_launch(L, DA[i][1], lv.Hs, dW, (T - 1) * B, hid, H, W, 4 * hid, H, W, 3, 1, 1, Ctot, h_off, hid, blk,
                                     "rsis_conv2d_wgrad(batched h, blk)", direct)
        for lv in levels:
            lv.G = lv.Hs = lv.Cs = lv.ACT = lv.UP = lv.skip = None
        ctx.levels = ctx.UP5 = ctx.feats = ctx.params = ctx.decoder = None
        return (None, None, None, None) + tuple(dfeats) + tuple(grads_par)


def decoder_sequence_stacked(decoder, skip_feats, T, want_hidden=True):
    """(out_masks (B, T, H*W) logits, class_probs (B, T, C), stop logits (B, T, 1), hidden_list, (H, W) of the masks)"""
    n = len(decoder.clstm_list)
    params = []
    for c in decoder.clstm_list:
        params += [c.Gates.weight, c.Gates.bias]
    params += [decoder.conv_out.weight, decoder.conv_out.bias, decoder.fc_class.weight, decoder.fc_class.bias, decoder.fc_stop.weight,
               decoder.fc_stop.bias]
    keep = torch.is_grad_enabled() and (any(f.requires_grad for f in skip_feats) or any(p.requires_grad for p in params))
    if blk_supported(decoder, skip_feats):
        res = _DecoderSeqBlkFn.apply(decoder, int(T), keep, bool(want_hidden), *skip_feats, *params)
    else:
        res = _DecoderSeqFn.apply(decoder, int(T), keep, *skip_feats, *params)
    out_masks, out_probs, out_stops = res[:3]
    hidden = [[res[3 + 2 * i], res[4 + 2 * i]] for i in range(n)] if len(res) > 3 else None
    size = (2 * skip_feats[-1].shape[2], 2 * skip_feats[-1].shape[3])
    return out_masks, out_probs, out_stops, hidden, size
