"""Time-invariant hoisting + time-batched weight gradients for the RSIS recurrent decoder (skip_mode == 'concat').

The encoder features are identical at every timestep (reference train.py:77,94), so in every `Gates` conv the channels
fed by the skip features are time-invariant.  Per ConvLSTM level i this module therefore splits the conv of
clstm.py:43-44   gates = W * [x | h_prev] + b   into

    G_i      = W[:, skip channels] * skip_i + b             -- ONE conv per iteration      (_HoistFn)
    gates_t  = G_i + W[:, up(h_{i-1,t}) | h_{i,t-1}] * [..] -- per timestep, fused kernel  (_StepFn, G_i is the kernel's addend)

(identical to the reference up to fp32 summation order).  The backward mirrors it:
  * sum_t d(gates_t) is one reduction over the stacked d(gates) at the end (in-kernel accumulation, rsis_convlstm_bwd_gates
    da_sum, only for steps beyond the tape capacity) and the skip-channel data/weight/bias gradients are computed ONCE from
    that sum (linearity), instead of T times;
  * h, c, saved gates, up-sampled inputs and d(gates) of all timesteps live in stacked [T][B][C][H][W] buffers written
    in place by the kernels, so the weight gradient of the recurrent channels is ONE split-K launch over T*B images per
    source instead of T launches (run by the t = 0 backward, which autograd necessarily executes last).
Module surface is unchanged: RSIS.forward(skip_feats, prev_hidden_list) is still called once per timestep; the tape is a
private per-iteration cache on the module, keyed on the identity of the state tensors it handed out.
"""
import os

import torch

from . import _lib, ops
from ._lib import check, int_array, lib, ptr, ptr_array, require_cuda_f32, stream

# Gate launches parked by _StepFn.forward while decoder_sequence() walks one diagonal of the (level, timestep) wavefront; flushed as
# ONE rsis_convlstm_fwd_batch call (a grouped launch for the exact-f32 kernels).  None: launch where the node is created.
_GATE_QUEUE = [None]
WAVEFRONT = [os.environ.get("RSIS_DECODER_WAVEFRONT", "1") != "0"]
# The global max-pool of every hidden state (the side features of model.py:143) taken in the gate kernel's epilogue as packed
# (value, pixel) keys (rsis_lstm_job.side_key) and decoded by the heads kernel, instead of one rsis_global_maxpool_fwd launch per cell
# that re-reads h.  Wavefront path only.
FUSED_POOL = [os.environ.get("RSIS_FUSED_POOL", "1") != "0"]
# bf16: time-batched weight gradients of the levels with ragged rows (W % 4 != 0) from channel-blocked bf16 copies of their operands
_BLK_WGRAD = [os.environ.get("RSIS_DECODER_BLK_WGRAD", "1") != "0"]


def _flush_gates():
    q, _GATE_QUEUE[0] = _GATE_QUEUE[0], None
    if not q:
        return
    jobs = (_lib.LstmJob * len(q))()
    for j, (f, _keep) in zip(jobs, q):
        srcs, segs, j.B, j.H, j.W, j.Wp, j.bias_packed, j.addend, j.c_prev, j.h_out, j.c_out, j.act_out, j.hid, j.ks, j.pad, j.tile, j.dtype, j.side_key = f
        j.nsrc = len(srcs)
        for k, (s, c) in enumerate(zip(srcs, segs)):
            j.src[k], j.Csrc[k] = s, c
    check(lib().rsis_convlstm_fwd_batch(jobs, len(q), stream()), "rsis_convlstm_fwd_batch")


class LevelTape(object):
    """Per-iteration state of one ConvLSTM level."""

    def __init__(self, cell, level, c_up, c_skip, cap):
        self.cell, self.level, self.c_up, self.c_skip, self.cap = cell, level, c_up, c_skip, cap
        self.hid = cell.hidden_size
        self.ks, self.pad = cell.kernel_size, cell.padding
        self.H = self.C = self.ACT = self.UP = self.DA = self.da_sum = None
        self.DHP, self.dhp_t = None, -1      # gradient of h[dhp_t] through the recurrence, handed from step dhp_t+1 to step dhp_t
        self.G = None
        self.n_fwd = 0
        self.n_bwd = 0
        self.last_h = self.last_c = None
        self.need_grad = False
        self.KEY = self.SIDE = self.ARG = None       # decoder_sequence with FUSED_POOL: per-step [T][B][hid] keys / features / arg-max

    def pooled(self, t):
        """(side feature, arg-max pixel) buffers of step t that the heads kernel fills from the keys, or None"""
        return None if self.KEY is None else (self.SIDE[t], self.ARG[t])

    def release(self):
        """Drop every tensor the level holds.  tl -> G / last_h -> grad_fn -> ctx -> tl is a reference CYCLE: without this
        the whole iteration (stacked buffers + the encoder graph hanging off G) is only reclaimed by Python's cyclic GC,
        i.e. tens of GB per step pile up until a gen-2 collection stalls the host for seconds."""
        self.H = self.C = self.ACT = self.UP = self.DA = self.da_sum = None
        self.DHP, self.dhp_t = None, -1
        self.G = self.last_h = self.last_c = None
        self.KEY = self.SIDE = self.ARG = None

    def alloc_forward(self, B, Hh, Ww, device, need_grad):
        hid, cap = self.hid, self.cap
        self.need_grad = need_grad
        if need_grad:
            self.H = torch.empty((cap, B, hid, Hh, Ww), dtype=torch.float32, device=device)
            self.C = torch.empty_like(self.H)
            self.ACT = torch.empty((cap, B, 4 * hid, Hh, Ww), dtype=torch.float32, device=device)
            if self.c_up > 0:
                self.UP = torch.empty((cap, B, self.c_up, Hh, Ww), dtype=torch.float32, device=device)


def _packs(cell, c_up, c_skip):
    """PackedConv triples of one cell: hoisted (skip channels), dynamic (up + h_prev channels)."""
    key = ("fused", c_up, c_skip, getattr(cell, "dtype", ops.DTYPE_F32))
    if key not in cell._packs:
        hid, ks, pad = cell.hidden_size, cell.kernel_size, cell.padding
        skip_off = c_up
        h_off = c_up + c_skip
        dt = getattr(cell, "dtype", ops.DTYPE_F32)
        hoist = ops.PackedConv(ks, [c_skip], lstm_hid=hid, stride=1, pad=pad, offs=[skip_off], dtype=dt)
        segs, offs = ([c_up], [0]) if c_up > 0 else ([], [])
        dyn = ops.PackedConv(ks, segs + [hid], lstm_hid=hid, stride=1, pad=pad, offs=offs + [h_off], dtype=dt)
        cell._packs[key] = (hoist, dyn)
    return cell._packs[key]


def dyn_dgrad_pack(cell, c_up, c_skip, H, W):
    """The copy set the gate conv's DATA GRADIENT (d(up) | dh_prev, explicit BPTT of decoder_seq) reads: under fp32, on levels whose
    sources are 32-channel multiples and whose map holds at least 12 x 12 pixels, a Winograd F(2x2, 3x3) copy (ops.DTYPE_F32_WINO,
    RSIS_WINOGRAD_GATES, default on: the 16 x 16 and 32 x 32 levels at 256 x 256 input, hidden 128) -- otherwise the dynamic pack itself.
    The forward keeps the direct kernel with the fused LSTM epilogue on every level."""
    _hoist, dyn = _packs(cell, c_up, c_skip)
    hid = cell.hidden_size
    if (dyn.dtype != ops.DTYPE_F32 or not ops.WINOGRAD_GATES[0] or cell.kernel_size != 3 or cell.padding != 1 or min(H, W) < 12
            or (c_up + hid) % 32 or c_up % 32 or c_up == 0 or (4 * hid) % 32):
        return dyn
    key = ("fused-dgrad-wino", c_up, c_skip)
    if key not in cell._packs:
        cell._packs[key] = ops.PackedConv(3, list(dyn.segs), lstm_hid=hid, stride=1, pad=1, offs=list(dyn.offs), dtype=ops.DTYPE_F32_WINO)
    return cell._packs[key]


class _HoistFn(torch.autograd.Function):
    """G = conv(skip, W[:, skip channels]) + b on gate-interleaved rows (once per iteration)."""

    @staticmethod
    def forward(ctx, tl, skip, weight, bias):
        skip = skip if skip.is_contiguous() else skip.contiguous()
        require_cuda_f32(skip, weight, bias)
        L = lib()
        hoist, _dyn = _packs(tl.cell, tl.c_up, tl.c_skip)
        B, Cs, H, W = skip.shape
        wp = hoist.fwd(weight, bias)
        G = torch.empty((B, 4 * tl.hid, H, W), dtype=torch.float32, device=skip.device)
        check(L.rsis_conv2d_fwd(ptr_array([skip]), int_array([Cs]), 1, B, H, W, ptr(wp), 4 * tl.hid, tl.ks, 1, tl.pad,
                                ptr(hoist.bias_p), None, ptr(G), H, W, ops.FORCE_TILE[0], hoist.dtype, stream()), "rsis_conv2d_fwd(hoist)")
        ctx.tl = tl
        ctx.wparam, ctx.bparam = weight, bias
        ctx.save_for_backward(skip, weight)
        return G

    @staticmethod
    def backward(ctx, dG):
        tl = ctx.tl
        skip, weight = ctx.saved_tensors
        L = lib()
        dG = dG if dG.is_contiguous() else dG.contiguous()
        hoist, _dyn = _packs(tl.cell, tl.c_up, tl.c_skip)
        B, Cs, H, W = skip.shape
        dskip = dW = db = None
        if ctx.needs_input_grad[1]:
            wd = hoist.dgrad(weight)
            dskip = torch.empty_like(skip)
            check(L.rsis_conv2d_dgrad(ptr(dG), B, 4 * tl.hid, H, W, ptr(wd), hoist.cin, tl.ks, 1, tl.pad, ptr_array([dskip]),
                                      int_array([Cs]), 1, H, W, None, ops.FORCE_TILE[0], hoist.dtype, stream()), "rsis_conv2d_dgrad(hoist)")
        if ctx.needs_input_grad[2]:
            tgt = ops._direct_target(ctx.wparam)
            dW = tgt if tgt is not None else torch.zeros_like(weight)
            ops.wgrad_launch(L, dG, skip, dW, B, Cs, H, W, 4 * tl.hid, H, W, tl.ks, 1, tl.pad, weight.shape[1], tl.c_up, tl.hid,
                             hoist.dtype, "rsis_conv2d_wgrad(hoist)", tgt is not None)
            if tgt is not None:
                dW = None
        if ctx.needs_input_grad[3]:
            tgt = ops._direct_target(ctx.bparam)
            db = tgt if tgt is not None else torch.zeros(4 * tl.hid, dtype=torch.float32, device=dG.device)
            check(L.rsis_bias_grad(ptr(dG), ptr(db), B, 4 * tl.hid, H * W, tl.hid, stream()), "rsis_bias_grad(hoist)")
            if tgt is not None:
                db = None
        return None, dskip, dW, db


class _StepFn(torch.autograd.Function):
    """One ConvLSTM level at one timestep: gates = G + conv([up | h_prev], W[:, dynamic channels]) -> (h, c)."""

    @staticmethod
    def forward(ctx, tl, t, up, h_prev, c_prev, G, weight):
        require_cuda_f32(up, h_prev, c_prev, G, weight)
        L = lib()
        _hoist, dyn = _packs(tl.cell, tl.c_up, tl.c_skip)
        B, _, H, W = G.shape
        hid = tl.hid
        need_grad = any(ctx.needs_input_grad) and tl.need_grad   # (needs_input_grad is True for parameters even under no_grad)
        stacked = need_grad and t < tl.cap
        srcs = []
        if up is not None:
            srcs.append(up if up.is_contiguous() else up.contiguous())
        if h_prev is not None:
            h_prev = h_prev if h_prev.is_contiguous() else h_prev.contiguous()
            c_prev = c_prev if c_prev.is_contiguous() else c_prev.contiguous()
            srcs.append(h_prev)
        if stacked:
            h, c, act = tl.H[t], tl.C[t], tl.ACT[t]
        else:
            h = torch.empty((B, hid, H, W), dtype=torch.float32, device=G.device)
            c = torch.empty_like(h)
            act = torch.empty((B, 4 * hid, H, W), dtype=torch.float32, device=G.device) if need_grad else None
        wp = dyn.fwd(weight)
        pa = ptr_array(srcs) if srcs else None
        ia = int_array([s.shape[1] for s in srcs]) if srcs else None
        if _GATE_QUEUE[0] is not None:      # decoder_sequence: park the launch, the diagonal's cells go out together
            _GATE_QUEUE[0].append((([s.data_ptr() for s in srcs], [s.shape[1] for s in srcs], B, H, W, wp.data_ptr(), None, G.data_ptr(),
                                    c_prev.data_ptr() if h_prev is not None else None, h.data_ptr(), c.data_ptr(),
                                    act.data_ptr() if act is not None else None, hid, tl.ks, tl.pad, ops.FORCE_TILE[0], dyn.dtype,
                                    tl.KEY[t].data_ptr() if tl.KEY is not None else None),
                                   (srcs, wp, G, c_prev, h, c, act)))
        else:
            check(L.rsis_convlstm_fwd(pa, ia, len(srcs), B, H, W, ptr(wp), None, ptr(G), ptr(c_prev) if h_prev is not None else None,
                                      ptr(h), ptr(c), ptr(act), hid, tl.ks, tl.pad, ops.FORCE_TILE[0], dyn.dtype, stream()), "rsis_convlstm_fwd(step)")
        ctx.tl, ctx.t, ctx.stacked = tl, t, stacked
        ctx.wparam = weight
        ctx.has_up, ctx.has_state = up is not None, h_prev is not None
        # the recurrent input is the tape's own previous output: its gradient is handed over through tl.DHP and summed inside
        # the next (earlier) step's gate-backward kernel instead of by an autograd add
        ctx.handover = bool(stacked and h_prev is not None and t > 0 and h_prev.data_ptr() == tl.H[t - 1].data_ptr())
        if need_grad:
            if stacked:
                tl.n_fwd = max(tl.n_fwd, t + 1)
                ctx.save_for_backward(weight)
            else:   # beyond the tape capacity / no tape: keep what the per-step backward needs
                ctx.save_for_backward(weight, act, c, c_prev if h_prev is not None else None, *srcs)
        return h, c

    @staticmethod
    def backward(ctx, dh, dc):
        tl, t = ctx.tl, ctx.t
        L = lib()
        _hoist, dyn = _packs(tl.cell, tl.c_up, tl.c_skip)
        hid, ks, pad = tl.hid, tl.ks, tl.pad
        weight = ctx.saved_tensors[0]
        dh = (dh if dh.is_contiguous() else dh.contiguous()) if dh is not None else None
        dc = (dc if dc.is_contiguous() else dc.contiguous()) if dc is not None else None
        if ctx.stacked:
            act, c = tl.ACT[t], tl.C[t]
            c_prev = tl.C[t - 1] if ctx.has_state else None
            if tl.DA is None:
                tl.DA = torch.empty_like(tl.ACT)
            da = tl.DA[t]                     # (sum_t d(gates_t) is one reduction over the stacked DA at t == 0: no per-step RMW)
            srcs = ([tl.UP[t]] if ctx.has_up else []) + ([tl.H[t - 1]] if ctx.has_state else [])
        else:
            act, c, c_prev = ctx.saved_tensors[1:4]
            srcs = list(ctx.saved_tensors[4:])
            da = torch.empty_like(act)
            if tl.da_sum is None:
                tl.da_sum = torch.zeros_like(act)
        B, _, H, W = c.shape
        dc_prev = torch.empty_like(c) if ctx.has_state else None
        dh2 = tl.DHP if (ctx.stacked and tl.dhp_t == t) else None
        tl.dhp_t = -1
        check(L.rsis_convlstm_bwd_gates(ptr(dh), ptr(dh2), ptr(dc), ptr(act), ptr(c_prev), ptr(c), ptr(da), ptr(dc_prev),
                                        None if ctx.stacked else ptr(tl.da_sum), B, hid, H * W, stream()), "rsis_convlstm_bwd_gates(step)")
        tl.n_bwd += 1
        d_up = dh_prev = None
        if srcs:
            wd = dyn.dgrad(weight)
            dxs = [torch.empty_like(s) for s in srcs]
            if ctx.handover:
                if tl.DHP is None:
                    tl.DHP = torch.empty_like(srcs[-1])
                dxs[-1] = tl.DHP
            check(L.rsis_conv2d_dgrad(ptr(da), B, 4 * hid, H, W, ptr(wd), dyn.cin, ks, 1, pad, ptr_array(dxs),
                                      int_array([s.shape[1] for s in srcs]), len(srcs), H, W, None, ops.FORCE_TILE[0], dyn.dtype, stream()),
                  "rsis_conv2d_dgrad(step)")
            k = 0
            if ctx.has_up:
                d_up = dxs[k]
                k += 1
            if ctx.has_state:
                if ctx.handover:
                    tl.dhp_t = t - 1          # consumed (and summed in-kernel) by step t-1's backward, which autograd runs next for this level
                else:
                    dh_prev = dxs[k]
        dW = dG = None
        Ctot = weight.shape[1]
        h_off = tl.c_up + tl.c_skip
        tgt = ops._direct_target(ctx.wparam) if ctx.needs_input_grad[6] else None
        if not ctx.stacked:
            # un-batched fallback: this step's own weight gradient
            dW = tgt if tgt is not None else torch.zeros_like(weight)
            off = [0] if ctx.has_up else []
            off += [h_off] if ctx.has_state else []
            for s, o in zip(srcs, off):
                ops.wgrad_launch(L, da, s, dW, B, s.shape[1], H, W, 4 * hid, H, W, ks, 1, pad, Ctot, o, hid, dyn.dtype,
                                 "rsis_conv2d_wgrad(step)", tgt is not None)
        if t == 0:
            # autograd runs the t = 0 backward last (every later step depends on it): flush the time-batched work
            if ctx.stacked:
                if tl.n_bwd < tl.n_fwd:       # (a step without a gradient would leave its DA slot unwritten)
                    raise RuntimeError("fused RSIS decoder: %d of %d timesteps were back-propagated" % (tl.n_bwd, tl.n_fwd))
                if tl.n_fwd == 1:
                    dG = tl.DA[0]
                else:       # sum over the timesteps, t ascending (fixed order; no torch reduction kernel on the path)
                    dG = torch.empty_like(tl.DA[0])
                    check(L.rsis_sum_leading(ptr(tl.DA), ptr(dG), tl.n_fwd, dG.numel(), stream()), "rsis_sum_leading")
                if tl.da_sum is not None:     # steps beyond the tape capacity accumulated theirs in the kernel
                    dG = dG + tl.da_sum
            else:
                dG = tl.da_sum
            if ctx.stacked and ctx.needs_input_grad[6]:
                n = tl.n_fwd
                if tl.n_bwd < n:   # steps that never received a gradient contribute zero
                    raise RuntimeError("fused RSIS decoder: %d of %d timesteps were back-propagated" % (tl.n_bwd, n))
                if dW is None:
                    dW = tgt if tgt is not None else torch.zeros_like(weight)
                # bf16, maps whose rows are not a multiple of 4 pixels (the 7 / 14-pixel levels of a 224 x 224 input): the fp32 loader of
                # the bf16 weight-gradient kernel goes dword by dword there (4x the load instructions: 0.82 ms per step for the two
                # smallest levels).  Their stacks are small: convert them to channel-blocked bf16 once (the same rounding the kernel
                # applies while staging) and use the whole-cell loader (conv_wgrad_bf16.hip, IN = 2).
                as_blk = (dyn.dtype == ops.DTYPE_BF16 and _BLK_WGRAD[0] and ks == 3 and W % 4 != 0 and hid % 8 == 0 and tl.c_up % 8 == 0)
                if as_blk:
                    da_b = ops.blk_from_nchw(tl.DA[:n].reshape(n * B, 4 * hid, H, W))
                    if tl.c_up > 0:
                        ops.wgrad_launch(L, da_b, ops.blk_from_nchw(tl.UP[:n].reshape(n * B, tl.c_up, H, W)), dW, n * B, tl.c_up, H, W, 4 * hid,
                                         H, W, ks, 1, pad, Ctot, 0, hid, ops.DTYPE_BF16_BLK, "rsis_conv2d_wgrad(batched up, blk)", tgt is not None)
                    if n > 1:
                        ops.wgrad_launch(L, da_b[B:], ops.blk_from_nchw(tl.H[:n - 1].reshape((n - 1) * B, hid, H, W)), dW, (n - 1) * B, hid, H, W,
                                         4 * hid, H, W, ks, 1, pad, Ctot, h_off, hid, ops.DTYPE_BF16_BLK, "rsis_conv2d_wgrad(batched h, blk)",
                                         tgt is not None)
                else:
                    if tl.c_up > 0:
                        ops.wgrad_launch(L, tl.DA, tl.UP, dW, n * B, tl.c_up, H, W, 4 * hid, H, W, ks, 1, pad, Ctot, 0, hid, dyn.dtype,
                                         "rsis_conv2d_wgrad(batched up)", tgt is not None)
                    if n > 1:
                        ops.wgrad_launch(L, tl.DA[1], tl.H, dW, (n - 1) * B, hid, H, W, 4 * hid, H, W, ks, 1, pad, Ctot, h_off, hid, dyn.dtype,
                                         "rsis_conv2d_wgrad(batched h)", tgt is not None)
        if tgt is not None:
            dW = None   # accumulated straight into weight.grad
        if t == 0:
            tl.release()   # last use of this level's tape: break the tape <-> autograd-node reference cycle now
        return None, None, d_up, dh_prev, dc_prev, dG, dW


class DecoderTape(object):
    """Per-iteration cache of the fused decoder: the hoisted gate terms and the stacked per-level buffers."""

    def __init__(self, decoder, skip_feats, need_grad):
        self.skip_ids = tuple(f.data_ptr() for f in skip_feats)
        self.t = 0
        self.levels = []
        hs = [cell.hidden_size for cell in decoder.clstm_list]
        for i, cell in enumerate(decoder.clstm_list):
            c_up = 0 if i == 0 else hs[i - 1]
            c_skip = skip_feats[i].shape[1]
            tl = LevelTape(cell, i, c_up, c_skip, decoder._tcap)
            B, _, H, W = skip_feats[i].shape
            tl.alloc_forward(B, H, W, skip_feats[i].device, need_grad)
            tl.G = _HoistFn.apply(tl, skip_feats[i], cell.Gates.weight, cell.Gates.bias)
            self.levels.append(tl)

    def matches(self, skip_feats, prev_hidden_list):
        """the caller is continuing the sequence this tape belongs to"""
        if tuple(f.data_ptr() for f in skip_feats) != self.skip_ids or self.t == 0:
            return False
        for tl, st in zip(self.levels, prev_hidden_list):
            if tl.last_h is None or st[0].data_ptr() != tl.last_h.data_ptr() or st[1].data_ptr() != tl.last_c.data_ptr():
                return False
        return True


class _UpsampleIntoFn(torch.autograd.Function):
    """align-corners bilinear upsample written straight into the level's stacked UP[t] buffer."""

    @staticmethod
    def forward(ctx, tl, t, x):
        x = x if x.is_contiguous() else x.contiguous()
        B, C, Hi, Wi = x.shape
        y = tl.UP[t]
        Ho, Wo = y.shape[-2], y.shape[-1]
        check(lib().rsis_upsample_bilinear_ac_fwd(ptr(x), ptr(y), B * C, Hi, Wi, Ho, Wo, stream()), "rsis_upsample_fwd")
        ctx.dims = (B, C, Hi, Wi, Ho, Wo)
        return y

    @staticmethod
    def backward(ctx, dy):
        B, C, Hi, Wi, Ho, Wo = ctx.dims
        dy = dy if dy.is_contiguous() else dy.contiguous()
        dx = torch.empty((B, C, Hi, Wi), dtype=torch.float32, device=dy.device)
        check(lib().rsis_upsample_bilinear_ac_bwd(ptr(dy), ptr(dx), B * C, Hi, Wi, Ho, Wo, stream()), "rsis_upsample_bwd")
        return None, None, dx


class _SideUpFn(torch.autograd.Function):
    """The two consumers of a level's hidden state besides the recurrence (model.py:143,149-150,163-164): the global max-pool
    side feature and the align-corners upsample into the next level, as ONE autograd node -- the backward writes the upsample
    gradient and adds the pooled gradient at the arg-max pixel in place, instead of materialising a mostly-zero map and
    letting autograd add the two."""

    @staticmethod
    def forward(ctx, tl, t, x, size, pooled=None):
        x = x if x.is_contiguous() else x.contiguous()
        B, C, Hi, Wi = x.shape
        L = lib()
        if pooled is not None:      # the gate kernel left (value, pixel) keys: the heads launch of this timestep fills both (ops._HeadsFn)
            side, arg = pooled
        else:
            side = torch.empty((B, C, 1, 1), dtype=torch.float32, device=x.device)
            arg = torch.empty((B, C), dtype=torch.int32, device=x.device)
            check(L.rsis_global_maxpool_fwd(ptr(x), ptr(side), ptr(arg), B * C, Hi * Wi, stream()), "rsis_global_maxpool_fwd")
        if tl is not None:
            y = tl.UP[t]                      # straight into the next level's stacked buffer
        else:
            y = torch.empty((B, C, size[0], size[1]), dtype=torch.float32, device=x.device)
        Ho, Wo = y.shape[-2], y.shape[-1]
        check(L.rsis_upsample_bilinear_ac_fwd(ptr(x), ptr(y), B * C, Hi, Wi, Ho, Wo, stream()), "rsis_upsample_fwd")
        ctx.dims = (B, C, Hi, Wi, Ho, Wo)
        ctx.save_for_backward(arg)
        return side, y

    @staticmethod
    def backward(ctx, dside, dy):
        (arg,) = ctx.saved_tensors
        B, C, Hi, Wi, Ho, Wo = ctx.dims
        L = lib()
        dx = None
        if dy is not None and dside is not None:      # the usual case: both gradients in one launch
            dy = dy if dy.is_contiguous() else dy.contiguous()
            dside = dside if dside.is_contiguous() else dside.contiguous()
            dx = torch.empty((B, C, Hi, Wi), dtype=torch.float32, device=dy.device)
            check(L.rsis_upsample_maxpool_bwd(ptr(dy), ptr(dside), ptr(arg), ptr(dx), B * C, Hi, Wi, Ho, Wo, stream()),
                  "rsis_upsample_maxpool_bwd")
            return None, None, dx, None, None
        if dy is not None:
            dy = dy if dy.is_contiguous() else dy.contiguous()
            dx = torch.empty((B, C, Hi, Wi), dtype=torch.float32, device=dy.device)
            check(L.rsis_upsample_bilinear_ac_bwd(ptr(dy), ptr(dx), B * C, Hi, Wi, Ho, Wo, stream()), "rsis_upsample_bwd")
        if dside is not None:
            dside = dside if dside.is_contiguous() else dside.contiguous()
            if dx is None:
                dx = torch.empty((B, C, Hi, Wi), dtype=torch.float32, device=dside.device)
                check(L.rsis_global_maxpool_bwd(ptr(dside), ptr(arg), ptr(dx), B * C, Hi * Wi, stream()), "rsis_global_maxpool_bwd")
            else:
                check(L.rsis_global_maxpool_bwd_add(ptr(dside), ptr(arg), ptr(dx), B * C, Hi * Wi, stream()), "rsis_global_maxpool_bwd_add")
        return None, None, dx, None, None


def decoder_levels(decoder, skip_feats, prev_hidden_list):
    """The 5-level ConvLSTM pyramid of RSIS.forward (model.py:129-165) with hoisting; returns (hidden_list, side_feats,
    last up-sampled hidden) -- or
    None when the fused path does not apply to this call."""
    need_grad = torch.is_grad_enabled() and (any(f.requires_grad for f in skip_feats) or
                                            any(p.requires_grad for p in decoder.clstm_list.parameters()))
    tape = decoder._tape
    if prev_hidden_list is None:
        if tape is not None:           # a previous sequence that was never back-propagated: drop its cycles explicitly
            for old in tape.levels:
                old.release()
        tape = decoder._tape = DecoderTape(decoder, skip_feats, need_grad)
    elif tape is None or not tape.matches(skip_feats, prev_hidden_list):
        return None
    t = tape.t
    hidden_list, side_feats = [], []
    up = None
    n_levels = len(tape.levels)
    for i, tl in enumerate(tape.levels):
        cell = tl.cell
        h_prev, c_prev = (None, None) if prev_hidden_list is None else (prev_hidden_list[i][0], prev_hidden_list[i][1])
        h, c = _StepFn.apply(tl, t, up, h_prev, c_prev, tl.G, cell.Gates.weight)
        tl.last_h, tl.last_c = h, c
        hidden_list.append([h, c])                                       # model.py:137
        if i + 1 < n_levels:
            nxt = tape.levels[i + 1]
            into = nxt if (nxt.UP is not None and t < nxt.cap) else None
            side, up = _SideUpFn.apply(into, t, h, tuple(skip_feats[i + 1].shape[-2:]))     # model.py:143,149-150
        else:
            size = (h.shape[-2] * 2, h.shape[-1] * 2)
            side, up = _SideUpFn.apply(None, t, h, size)                                    # model.py:143,163-164
        side_feats.append(side)
    tape.t += 1
    return hidden_list, side_feats, up


def sequence_supported(decoder, skip_feats, T):
    """the wavefront schedule covers what the fused per-step path covers, for sequences that fit the tape"""
    need_grad = torch.is_grad_enabled() and (any(f.requires_grad for f in skip_feats) or
                                            any(p.requires_grad for p in decoder.clstm_list.parameters()))
    return (WAVEFRONT[0] and decoder.fused and decoder.skip_mode == "concat" and decoder.dropout == 0 and
            len(skip_feats) == len(decoder.clstm_list) and T >= 1 and (T <= decoder._tcap or not need_grad))


def decoder_sequence(decoder, skip_feats, T):
    """T decoder timesteps from the zero state (the loop of reference train.py:85-94 / test.py:37-38 around RSIS.forward,
    model.py:122-184) in WAVEFRONT order.  Cell (level i, step t) needs up(h[i-1][t]) and (h, c)[i][t-1] only, so the cells
    (i, d - i) of diagonal d are independent: their gate kernels are issued as ONE rsis_convlstm_fwd_batch call per diagonal
    (T + 4 grouped launches instead of 5 T; include/rsis_hip.h).  Same autograd nodes, same kernels per cell and therefore the same
    results as T calls of RSIS.forward; only the launch order and the grouping differ.  Returns the per-step outputs
    [(out_mask, class_probs, stop_probs)] and the final hidden_list."""
    need_grad = torch.is_grad_enabled() and (any(f.requires_grad for f in skip_feats) or
                                            any(p.requires_grad for p in decoder.clstm_list.parameters()))
    if decoder._tape is not None:          # a previous sequence that was never back-propagated: drop its cycles explicitly
        for old in decoder._tape.levels:
            old.release()
    tape = decoder._tape = DecoderTape(decoder, skip_feats, need_grad)
    n = len(tape.levels)
    B = skip_feats[0].shape[0]
    fused_pool = (FUSED_POOL[0] and all(tl.ks == 3 and tl.pad == 1 for tl in tape.levels) and decoder.dropout_cls == 0 and
                  decoder.dropout_stop == 0 and ops.heads_supported([skip_feats[0]] * n, decoder.fc_class, decoder.fc_stop) and
                  sum(tl.hid for tl in tape.levels) == decoder.fc_class.weight.shape[1])
    if fused_pool:                     # one zeroed key buffer (and its decoded twins) for the whole sequence: a single fill launch
        tot = sum(tl.hid for tl in tape.levels)
        dev = skip_feats[0].device
        KEY = torch.zeros(T * B * tot, dtype=torch.int64, device=dev)
        SIDE = torch.empty(T * B * tot, dtype=torch.float32, device=dev)
        ARG = torch.empty(T * B * tot, dtype=torch.int32, device=dev)
        off = 0
        for tl in tape.levels:
            m = T * B * tl.hid
            tl.KEY = KEY[off:off + m].view(T, B, tl.hid)
            tl.SIDE = SIDE[off:off + m].view(T, B, tl.hid, 1, 1)
            tl.ARG = ARG[off:off + m].view(T, B, tl.hid)
            off += m
    Hs = [[None] * T for _ in range(n)]
    Cs = [[None] * T for _ in range(n)]
    UP = [[None] * T for _ in range(n + 1)]          # UP[i][t]: the up-sampled hidden state level i consumes at step t (i >= 1)
    SIDE = [[None] * T for _ in range(n)]
    outs = [None] * T
    for d in range(T + n - 1):
        cells = [(i, d - i) for i in range(n) if 0 <= d - i < T]
        _GATE_QUEUE[0] = []
        try:
            for i, t in cells:
                tl = tape.levels[i]
                h_prev, c_prev = (Hs[i][t - 1], Cs[i][t - 1]) if t > 0 else (None, None)
                Hs[i][t], Cs[i][t] = _StepFn.apply(tl, t, UP[i][t] if i > 0 else None, h_prev, c_prev, tl.G, tl.cell.Gates.weight)
            _flush_gates()
        finally:
            _GATE_QUEUE[0] = None
        for i, t in cells:
            h = Hs[i][t]
            if i + 1 < n:
                nxt = tape.levels[i + 1]
                into = nxt if (nxt.UP is not None and t < nxt.cap) else None
                SIDE[i][t], UP[i + 1][t] = _SideUpFn.apply(into, t, h, tuple(skip_feats[i + 1].shape[-2:]), tape.levels[i].pooled(t))     # model.py:143,149-150
            else:
                SIDE[i][t], UP[n][t] = _SideUpFn.apply(None, t, h, (h.shape[-2] * 2, h.shape[-1] * 2), tape.levels[i].pooled(t))          # model.py:143,163-164
                hidden_list = [[Hs[k][t], Cs[k][t]] for k in range(n)]
                keys = ([tl.KEY[t] for tl in tape.levels], [tl.ARG[t] for tl in tape.levels]) if fused_pool else None
                outs[t] = decoder._heads(UP[n][t], [SIDE[k][t] for k in range(n)], hidden_list, keys)[:3]
                UP[n][t] = None
    for i, tl in enumerate(tape.levels):
        tl.last_h, tl.last_c = Hs[i][T - 1], Cs[i][T - 1]
    tape.t = T
    return outs, [[Hs[k][T - 1], Cs[k][T - 1]] for k in range(n)]
