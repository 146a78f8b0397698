#!/usr/bin/env python
"""bench.py -- training images/sec of the RSIS hot path on MI355X (BASELINE.json metric).

One "step" = one full training iteration of the encoder -> T=10-step ConvLSTM decoder path on one resident synthetic
batch (B=32 per GPU, 256x256x3, fp32, ResNet-101, hidden 128): encoder fwd, 10 decoder steps, score matrix + Hungarian
matching, the three losses, backward (BPTT + encoder), gradient all-reduce (N>1), two Adam steps.  Nothing is skipped.

Prints ONE JSON line on rank 0 with the contract fields plus
  "roofline":     ConvLSTM gate kernel (rsis_convlstm_fwd_batch: the 5 pyramid levels of one timestep as the product launches
                  them, B=32): achieved = EXECUTED FLOPs of that launch (hoisted skip term not credited: 31.41 GFLOP at 256x256) /
                  its HIP-event time, against the 157.3 TFLOP/s exact-f32 MFMA peak; `achieved_algorithmic` (the reference's full-K
                  53.15 GFLOP / the product's time) and `full_k` (the un-hoisted launch) are reported next to it, labelled;
  "cpu_baseline": the CPU oracle (port of the reference op graph, oracle/rsis_oracle.py) timed on this box's host cores
                  on a bounded sample of the same workload (rank 0, N=1 only).

After the W warm-up steps an untimed "settle" phase keeps stepping until the step time is stable (cold-box power
management; see main()); the timed region is still EXACTLY K steps between barrier + synchronize.

  python bench.py [--gpus N] [--steps K] [--warmup W]        (N > 1 without WORLD_SIZE: launches its own N ranks)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...
"""
import argparse
import json
import os
import re
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

PEAK_F32_MFMA_TFLOPS = 157.3   # /opt/skills/guides/MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
# what a register-only v_mfma_f32_32x32x2_f32 loop SUSTAINS on these boxes (tools/exp/clock_probe.hip, profiles/r05_r: 0.875 of nominal;
# power management holds the block clock near 2.1 GHz): reported beside `peak` as `peak_sustained` / `frac_of_sustained`, never instead of it
PEAK_F32_MFMA_SUSTAINED_TFLOPS = 137.6
# ConvLSTM gate GEMMs at config 2: (x segments, hid, H=W)
GATE_LAYERS = [([128], 128, 8), ([128, 128], 64, 16), ([64, 64], 32, 32), ([32, 32], 16, 64), ([16, 16], 8, 128)]


def bench_args(batch, imsize, T, dtype="fp32"):
    from rsis_amd.args import get_parser
    a = get_parser().parse_args([])
    a.batch_size, a.imsize, a.maxseqlen, a.gt_maxseqlen, a.num_classes = batch, imsize, T, 20, 21
    a.dtype = dtype
    a.use_class_loss = a.use_stop_loss = a.update_encoder = True
    a.synthetic = True
    return a


PEAK_HBM_GBS = 8000.0           # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s (6.3 TB/s achievable)
PEAK_BF16_MFMA_TFLOPS = 2500.0  # same guide: dense bf16 MFMA (v_mfma_f32_32x32x16_bf16), no sparsity


def _time_launch(launch, iters):
    for _ in range(3):
        launch()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        launch()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def _hw(imsize):
    """(H, W) of the input image: `imsize` is an int (square) or a pair"""
    return (imsize, imsize) if isinstance(imsize, int) else (int(imsize[0]), int(imsize[1]))


def gate_kernel_roofline(B, iters, imsize, dtype="fp32", T=10, product_only=False):
    """The fused ConvLSTM gate kernel (rsis_convlstm_fwd) of the 5 pyramid scales of one decoder timestep, timed with HIP events
    on the launch stream, in the TWO forms SURVEY.md 8(d) asks for:
      product : the launch rsis_amd.decoder_fused issues every timestep -- the time-invariant skip-channel term G enters as the
                kernel's addend (computed once per iteration by one plain conv per level), only [up(h) | h_prev] is convolved;
      full_k  : the reference's un-hoisted conv over [x | skip | h_prev] (clstm.py:43-44), as round 1 reported.
    `achieved` is the EXECUTED TFLOP/s of the product launches (pure kernel efficiency; the hoisting is not credited),
    `achieved_algorithmic` divides the reference's full-K FLOPs of a timestep (53.15 G at B=32, 256x256) by the product's time per
    timestep including 1/T of the hoisted convs."""
    from rsis_amd import ops
    from rsis_amd._lib import check, int_array, lib, ptr, ptr_array, stream
    L = lib()
    dt = ops.DTYPES[dtype]
    rows, tot = [], {"full_flops": 0.0, "full_ms": 0.0, "dyn_flops": 0.0, "dyn_ms": 0.0, "hoist_ms": 0.0, "bytes": 0.0}
    diag = []        # the five product launches as jobs of ONE rsis_convlstm_fwd_batch call (a steady-state wavefront diagonal)
    for li, (segs, hid, hw) in enumerate(GATE_LAYERS):
        H, W = hw * _hw(imsize)[0] // 256, hw * _hw(imsize)[1] // 256
        c_skip = segs[-1]
        c_up = segs[0] if len(segs) > 1 else 0
        cin = sum(segs) + hid
        w = torch.randn(4 * hid, cin, 3, 3, device="cuda") * (1.0 / (3.0 * cin ** 0.5))
        bias = torch.randn(4 * hid, device="cuda") * 0.1
        xs = [torch.randn(B, c, H, W, device="cuda") for c in segs]
        h_prev = torch.tanh(torch.randn(B, hid, H, W, device="cuda"))
        c_prev = torch.randn(B, hid, H, W, device="cuda")
        h, c = torch.empty_like(c_prev), torch.empty_like(c_prev)
        act = torch.empty(B, 4 * hid, H, W, device="cuda")
        # ---- full K (reference form) ----
        pack = ops.PackedConv(3, segs + [hid], lstm_hid=hid, dtype=dt)
        wp = pack.fwd(w, bias)
        srcs = xs + [h_prev]
        pa, ia = ptr_array(srcs), int_array(segs + [hid])
        ms_full = 1e-9 if product_only else _time_launch(
            lambda: check(L.rsis_convlstm_fwd(pa, ia, len(srcs), B, H, W, ptr(wp), ptr(pack.bias_p), None, ptr(c_prev), ptr(h), ptr(c), ptr(act),
                                              hid, 3, 1, 0, dt, stream()), "rsis_convlstm_fwd"), iters)
        # ---- product form: hoisted skip term + dynamic channels ----
        hoist = ops.PackedConv(3, [c_skip], lstm_hid=hid, offs=[c_up], dtype=dt)
        dyn = ops.PackedConv(3, ([c_up] if c_up else []) + [hid], lstm_hid=hid, offs=([0] if c_up else []) + [c_up + c_skip], dtype=dt)
        wh, wd = hoist.fwd(w, bias), dyn.fwd(w)
        G = torch.empty(B, 4 * hid, H, W, device="cuda")
        skip = xs[-1]
        ps, is_ = ptr_array([skip]), int_array([c_skip])
        ms_hoist = _time_launch(lambda: check(L.rsis_conv2d_fwd(ps, is_, 1, B, H, W, ptr(wh), 4 * hid, 3, 1, 1, ptr(hoist.bias_p), None, ptr(G),
                                                                 H, W, 0, dt, stream()), "rsis_conv2d_fwd(hoist)"), 1 if product_only else iters)
        dsrc = ([xs[0]] if c_up else []) + [h_prev]
        pd, idd = ptr_array(dsrc), int_array(([c_up] if c_up else []) + [hid])
        ms_dyn = _time_launch(lambda: check(L.rsis_convlstm_fwd(pd, idd, len(dsrc), B, H, W, ptr(wd), None, ptr(G), ptr(c_prev), ptr(h), ptr(c),
                                                                 ptr(act), hid, 3, 1, 0, dt, stream()), "rsis_convlstm_fwd(step)"), iters)
        diag.append((dsrc, wd, G, c_prev, h, c, act, hid, H, W))
        M = B * H * W
        f_full = 2.0 * M * (cin * 9) * (4 * hid)
        f_dyn = 2.0 * M * ((c_up + hid) * 9) * (4 * hid)
        # minimum HBM bytes of the product launch: inputs [up | h_prev], G, c_prev in; h, c, saved gates out (fp32)
        byts = 4.0 * M * ((c_up + hid) + 4 * hid + hid + hid + hid + 4 * hid)
        # SURVEY 8(d)'s minimal bytes of the FULLY fused reference cell (no saved gates, no hoisted term): x, h_prev, c_prev in; h, c out; weights
        tot["min_bytes"] = tot.get("min_bytes", 0.0) + 4.0 * (M * (cin + 3 * hid) + 4 * hid * cin * 9)
        rows.append({"HxW": "%dx%d" % (H, W), "gemm_MKN_full": [M, cin * 9, 4 * hid], "gemm_MKN_product": [M, (c_up + hid) * 9, 4 * hid],
                     "ms_full": round(ms_full, 4), "tflops_full": round(f_full / ms_full / 1e9, 2),
                     "ms_product": round(ms_dyn, 4), "tflops_product": round(f_dyn / ms_dyn / 1e9, 2),
                     "gbs_product": round(byts / ms_dyn / 1e6, 1), "ms_hoist_per_iteration": round(ms_hoist, 4)})
        tot["full_flops"] += f_full
        tot["full_ms"] += ms_full
        tot["dyn_flops"] += f_dyn
        tot["dyn_ms"] += ms_dyn
        tot["hoist_ms"] += ms_hoist
        tot["bytes"] += byts
    # ---- the launch the product issues in steady state: the gate kernels of one (level, timestep) diagonal of the decoder's
    # wavefront -- levels 0..4 at steps t+4..t, one timestep's worth of work -- as ONE rsis_convlstm_fwd_batch call ----
    from rsis_amd._lib import LstmJob
    jobs = (LstmJob * len(diag))()
    keys = []
    for j, (dsrc, wd, G, c_prev, h, c, act, hid, H, W) in zip(jobs, diag):
        j.nsrc = len(dsrc)
        for k, s in enumerate(dsrc):
            j.src[k], j.Csrc[k] = s.data_ptr(), s.shape[1]
        (j.B, j.H, j.W, j.Wp, j.bias_packed, j.addend, j.c_prev, j.h_out, j.c_out, j.act_out, j.hid, j.ks, j.pad, j.tile, j.dtype) = (
            B, H, W, wd.data_ptr(), None, G.data_ptr(), c_prev.data_ptr(), h.data_ptr(), c.data_ptr(), act.data_ptr(), hid, 3, 1, 0, dt)
        # (as the product launches it: the global max-pool of the side feature, model.py:143, folded into the epilogue as packed keys)
        keys.append(torch.zeros(B, hid, dtype=torch.int64, device="cuda"))
        j.side_key = keys[-1].data_ptr()
    ms_diag = _time_launch(lambda: check(L.rsis_convlstm_fwd_batch(jobs, len(diag), stream()), "rsis_convlstm_fwd_batch"), iters)
    singles_ms = tot["dyn_ms"]
    tot["dyn_ms"] = ms_diag
    executed = tot["dyn_flops"] / tot["dyn_ms"] / 1e9
    algorithmic = tot["full_flops"] / (tot["dyn_ms"] + tot["hoist_ms"] / T) / 1e9
    full = tot["full_flops"] / tot["full_ms"] / 1e9
    out = {"kernel": ("conv3x3_direct_group_kernel<EPI_LSTM>" if dtype == "fp32" else "conv_bf16_kernel<3, ..., EPI_LSTM>") +
                     " (rsis_convlstm_fwd_batch): the ConvLSTM gate kernels of the 5 pyramid levels -- one decoder timestep's worth of "
                     "work -- as rsis_amd.decoder_fused.decoder_sequence launches them: the cells of one (level, timestep) wavefront "
                     "diagonal in ONE call (fp32: one grid; hoisted skip term as addend, dynamic channels only)",
           "per_scale": rows, "ms_per_timestep": round(tot["dyn_ms"], 4),
           "ms_per_timestep_as_five_single_launches": round(singles_ms, 4),
           "tflops_as_five_single_launches": round(tot["dyn_flops"] / singles_ms / 1e9, 2),
           "executed_gflop_per_timestep": round(tot["dyn_flops"] / 1e9, 3),
           "algorithmic_gflop_per_timestep": round(tot["full_flops"] / 1e9, 3),
           "hoisted_convs_ms_per_iteration": round(tot["hoist_ms"], 4),
           "achieved_executed": round(executed, 2), "achieved_algorithmic": round(algorithmic, 2),
           "full_k": {"ms_per_timestep": round(tot["full_ms"], 4), "achieved": round(full, 2),
                      "note": "the reference's un-hoisted launch (what round 1 reported as roofline.achieved)"},
           "traffic": None}
    out["algorithmic_mbytes_per_timestep"] = round(tot["bytes"] / 1e6, 1)
    out["algorithmic_mbytes_note"] = ("bytes of the launch AS BUILT (inputs [up | h_prev], hoisted term G, c_prev in; h, c, saved gates out) -- the "
                                      "denominator of traffic_vs_algorithmic; SURVEY 8(d)'s minimal bytes of a fully fused inference cell (no saved "
                                      "gates, no G) are in survey_minimal_mbytes_per_timestep, the ratio to them in traffic_vs_survey_minimal")
    out["survey_minimal_mbytes_per_timestep"] = round(tot["min_bytes"] / 1e6, 1)
    if dtype == "fp32":
        out.update({"bound": "mfma", "achieved": round(executed, 2), "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                    "frac": round(executed / PEAK_F32_MFMA_TFLOPS, 4), "frac_algorithmic": round(algorithmic / PEAK_F32_MFMA_TFLOPS, 4),
                    "frac_full_k": round(full / PEAK_F32_MFMA_TFLOPS, 4),
                    "peak_sustained": PEAK_F32_MFMA_SUSTAINED_TFLOPS, "frac_of_sustained": round(executed / PEAK_F32_MFMA_SUSTAINED_TFLOPS, 4),
                    "peak_sustained_note": "what a register-only v_mfma_f32_32x32x2_f32 loop holds on these boxes (profiles/r05_r); "
                                           "`frac` is against the guide's nominal `peak`"})
    else:
        gbs = tot["bytes"] / tot["dyn_ms"] / 1e6
        out.update({"bound": "hbm", "achieved": round(gbs, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": round(gbs / PEAK_HBM_GBS, 4),
                    "note": "bf16 operands at 16x the f32 MFMA rate: the launch is bound by its fp32 activation traffic"})
    return out


def gate_kernel_roofline_blk(B, iters, imsize, T=10):
    """bf16 (`-dtype bf16`): the gate kernels as rsis_amd.decoder_seq._DecoderSeqBlkFn launches them -- every tensor of the recurrence
    channel-blocked bf16 (cell state fp32), the five levels of a wavefront diagonal as ONE rsis_blk_conv3x3_batch call
    (conv_blk_dec_group_kernel<1>).  Bound: HBM.  Algorithmic bytes per pixel of a level: [up | h_prev] in (2 B per channel), the hoisted
    gate term G in (4 hid x 2 B), c_prev in / c out (hid x 4 B each), h out (hid x 2 B), saved gates out (4 hid x 2 B)."""
    from rsis_amd import ops
    L_rows, jobs, tot_bytes, tot_flops, keep = [], [], 0.0, 0.0, []
    min_bytes = 0.0
    dt = ops.DTYPE_BF16
    b16 = dict(dtype=torch.bfloat16, device="cuda")
    singles_ms = 0.0
    hoist_ms = 0.0
    for li, (segs, hid, hw) in enumerate(GATE_LAYERS):
        H, W = hw * _hw(imsize)[0] // 256, hw * _hw(imsize)[1] // 256
        c_skip = segs[-1]
        c_up = segs[0] if len(segs) > 1 else 0
        cin = sum(segs) + hid
        w = torch.randn(4 * hid, cin, 3, 3, device="cuda") * (1.0 / (3.0 * cin ** 0.5))
        bias = torch.randn(4 * hid, device="cuda") * 0.1
        hoist = ops.PackedConv(3, [c_skip], lstm_hid=hid, offs=[c_up], dtype=dt)
        dyn = ops.PackedConv(3, ([c_up] if c_up else []) + [hid], lstm_hid=hid, offs=([0] if c_up else []) + [c_up + c_skip], dtype=dt)
        wh, wd = hoist.fwd(w, bias), dyn.fwd(w)
        skip = torch.randn((B, c_skip // 8, H, W, 8), device="cuda").to(torch.bfloat16)
        G = torch.empty((B, 4 * hid // 8, H, W, 8), **b16)
        ms_h = _time_launch(lambda: ops.blk_conv3x3_batch([ops.blk_conv_job([skip], wh, 4 * hid, bias=hoist.bias_p, dsts=[G])]), max(2, iters // 4))
        srcs = ([torch.randn((B, c_up // 8, H, W, 8), device="cuda").to(torch.bfloat16)] if c_up else []) + \
               [torch.tanh(torch.randn((B, hid // 8, H, W, 8), device="cuda")).to(torch.bfloat16)]
        c_prev = torch.randn(B, hid, H, W, device="cuda")
        h, c = torch.empty((B, hid // 8, H, W, 8), **b16), torch.empty_like(c_prev)
        act = torch.empty((B, 4 * hid // 8, H, W, 8), **b16)
        key = torch.zeros(B, hid, dtype=torch.int64, device="cuda")          # (the side feature's max-pool keys, as the product launches it)
        job = ops.blk_conv_job(srcs, wd, 4 * hid, addend=G, hid=hid, c_prev=c_prev, c_out=c, h_out=h, act_out=act, side_key=key)
        # (RSIS_BENCH_DIAG_ONLY=1, set for the PMC child passes: only the timestep's call is launched, so that every gate-kernel launch in
        #  the counter file belongs to it -- the call may be one grid or, for very large jobs, several)
        ms = 1e-9 if os.environ.get("RSIS_BENCH_DIAG_ONLY") == "1" else _time_launch(lambda: ops.blk_conv3x3_batch([job]), iters)
        M = B * H * W
        fl = 2.0 * M * ((c_up + hid) * 9) * (4 * hid)
        byts = 1.0 * M * (2 * (c_up + hid) + 2 * 4 * hid + 4 * hid + 2 * hid + 4 * hid + 2 * 4 * hid)
        min_bytes = min_bytes + 1.0 * M * (2 * cin + 4 * hid + 2 * hid + 4 * hid) + 2.0 * 4 * hid * cin * 9      # SURVEY 8(d) form, bf16 x / h, fp32 c
        L_rows.append({"HxW": "%dx%d" % (H, W), "gemm_MKN_product": [M, (c_up + hid) * 9, 4 * hid], "ms_product": round(ms, 4),
                       "tflops_product": round(fl / ms / 1e9, 2), "gbs_product": round(byts / ms / 1e6, 1), "mbytes": round(byts / 1e6, 1),
                       "ms_hoist_per_iteration": round(ms_h, 4)})
        jobs.append(job)
        keep.append((w, wh, wd, skip, G, srcs, c_prev, h, c, act, key))
        tot_bytes += byts
        tot_flops += fl
        singles_ms += ms
        hoist_ms += ms_h
    ms_diag = _time_launch(lambda: ops.blk_conv3x3_batch(jobs), iters)
    gbs = tot_bytes / ms_diag / 1e6
    return {"kernel": "conv_blk_dec_group_kernel<1> (rsis_blk_conv3x3_batch): the ConvLSTM gate kernels of the 5 pyramid levels -- one decoder "
                      "timestep's worth of work -- as rsis_amd.decoder_seq launches them under -dtype bf16: channel-blocked bf16 operands / "
                      "hidden state / saved gates, fp32 cell state, the cells of one (level, timestep) wavefront diagonal in ONE grid",
            "per_scale": L_rows, "ms_per_timestep": round(ms_diag, 4), "ms_per_timestep_as_five_single_launches": round(singles_ms, 4),
            "executed_gflop_per_timestep": round(tot_flops / 1e9, 3), "hoisted_convs_ms_per_iteration": round(hoist_ms, 4),
            "achieved_executed": round(tot_flops / ms_diag / 1e9, 2), "achieved_algorithmic": None, "full_k": {"achieved": None, "note": "fp32 only"},
            "algorithmic_mbytes_per_timestep": round(tot_bytes / 1e6, 1), "traffic": None,
            "algorithmic_mbytes_note": "bytes of the launch AS BUILT (see the docstring); SURVEY 8(d)'s minimal bytes of a fully fused inference cell "
                                       "(no saved gates, no hoisted term) are in survey_minimal_mbytes_per_timestep",
            "survey_minimal_mbytes_per_timestep": round(min_bytes / 1e6, 1),
            "bound": "hbm", "achieved": round(gbs, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": round(gbs / PEAK_HBM_GBS, 4),
            "note": "bf16 MFMA at 16x the f32 rate: the launch is bound by its activation traffic (2 B per blk element, 4 B per cell-state element)"}


# (Cin, Cout, ks, out HxW at 256^2, layers of that shape in ResNet-101 + skip convs) -- SURVEY.md Appendix A
N_TRUNK_SHAPES = 11      # the first 11 are trunk layers (blocked bf16 activations under -dtype bf16), the last three the skip convs
TRUNK_SHAPES = [(256, 256, 3, 16, 22), (256, 1024, 1, 16, 23), (1024, 256, 1, 16, 22), (64, 64, 3, 64, 3), (128, 128, 3, 32, 3),
                (512, 512, 3, 8, 2), (64, 256, 1, 64, 4), (128, 512, 1, 32, 4), (512, 128, 1, 32, 3), (512, 2048, 1, 8, 3),
                (2048, 512, 1, 8, 2), (2048, 128, 3, 8, 1), (1024, 128, 3, 16, 1), (64, 16, 3, 128, 1)]


def trunk_kernel_rooflines(B, iters, imsize, dtype="fp32", inference=False):
    """inference=True: the FORWARD launches as test() issues them -- fp32: rsis_conv2d_fwd on its inference path (segmented
    accumulation, no split-K); bf16: rsis_blk_conv2d_bn_eval (the eval-mode BatchNorm (+ ReLU) folded into the conv's epilogue) --
    two families, conv1x1 / conv3x3 forward.  Otherwise --
    roofline_kernels: the four kernel families that hold most of the step next to the gate kernel -- the 1x1 GEMM (forward and
    data gradient), the direct 3x3 conv with the plain epilogue (forward and data gradient), and the tiled weight gradients (3x3,
    1x1) -- on the stride-1 layer shapes of the ResNet-101 trunk and the skip convs, weighted by how many layers have each shape."""
    from rsis_amd import ops
    from rsis_amd._lib import check, int_array, lib, ptr, ptr_array, stream
    from rsis_amd import blk_trunk
    L = lib()
    dt = ops.DTYPES[dtype]
    blk_on = dtype != "fp32" and blk_trunk.ENABLED[0]      # the trunk's layers then run on channel-blocked bf16 tensors (conv_blk.hip)
    fam = {}
    for si, (cin, cout, ks, hw, count) in enumerate(TRUNK_SHAPES):
        hh, ww = hw * _hw(imsize)[0] // 256, hw * _hw(imsize)[1] // 256
        pad = ks // 2
        if blk_on and si < N_TRUNK_SHAPES:
            xb = ops.blk_from_nchw(torch.randn(B, cin, hh, ww, device="cuda"))
            yb = ops.blk_from_nchw(torch.randn(B, cout, hh, ww, device="cuda"))
            w = torch.randn(cout, cin, ks, ks, device="cuda") / (ks * cin ** 0.5)
            pack = ops.PackedConv(ks, [cin], stride=1, pad=pad, dtype=dt)
            wp, wd = pack.fwd(w), pack.dgrad(w)
            ob, dxb, dW = torch.empty_like(yb), torch.empty_like(xb), torch.zeros_like(w)
            fl = 2.0 * B * hh * ww * cin * ks * ks * cout
            if inference:
                g_, b_, m_, v_ = (torch.rand(cout, device="cuda") + 0.5, torch.randn(cout, device="cuda"), torch.randn(cout, device="cuda"),
                                  torch.rand(cout, device="cuda") + 0.5)
                ms_f = _time_launch(lambda: check(L.rsis_blk_conv2d_bn_eval(ptr(xb), B, cin, hh, ww, ptr(wp), cout, ks, None, ptr(g_), ptr(b_), ptr(m_),
                                                                            ptr(v_), 1e-5, 1, 0, ptr(ob), 0, stream()), "blk fwd + bn eval"), iters)
                f = fam.setdefault("conv%dx%d fwd (eval)" % (ks, ks), {"flops": 0.0, "ms": 0.0, "bytes": 0.0, "launches": 0})
                f["flops"] += fl * count
                f["ms"] += ms_f * count
                f["bytes"] += 2.0 * B * hh * ww * (cin + cout) * count
                f["launches"] += count
                continue
            ms_f = _time_launch(lambda: check(L.rsis_blk_conv2d(ptr(xb), B, cin, hh, ww, ptr(wp), cout, ks, None, ptr(ob), 0, stream()), "blk fwd"), iters)
            ms_d = _time_launch(lambda: check(L.rsis_blk_conv2d(ptr(yb), B, cout, hh, ww, ptr(wd), cin, ks, None, ptr(dxb), 0, stream()), "blk dgrad"), iters)
            ms_w = _time_launch(lambda: check(L.rsis_conv2d_wgrad(ptr(yb), ptr(xb), ptr(dW), B, cin, hh, ww, cout, hh, ww, ks, 1, pad, cin, 0, 0,
                                                                  ops.DTYPE_BF16_BLK, stream()), "blk wgrad"), iters)
            act_bytes = 2.0 * B * hh * ww * (cin + cout)
            for name, ms, n in (("conv%dx%d fwd+dgrad" % (ks, ks), ms_f + ms_d, 2), ("conv%dx%d wgrad" % (ks, ks), ms_w, 1)):
                f = fam.setdefault(name, {"flops": 0.0, "ms": 0.0, "bytes": 0.0, "launches": 0})
                f["flops"] += n * fl * count
                f["ms"] += ms * count
                f["bytes"] += n * act_bytes * count
                f["launches"] += n * count
            continue
        x = torch.randn(B, cin, hh, ww, device="cuda")
        w = torch.randn(cout, cin, ks, ks, device="cuda") / (ks * cin ** 0.5)
        # the copy flavour the product's module builds for this conv: training calls of the layer 1-3 bottleneck convs run the Winograd
        # kernel (ops.conv_dtype, RSIS_WINOGRAD); the skip convs have a bias and their own module, inference calls keep the direct kernel
        cdt = dt
        if not inference and si < N_TRUNK_SHAPES:
            cdt = ops.conv_dtype(dt, ks, 1, pad, cin, cout)
        wino = cdt == ops.DTYPE_F32_WINO
        pack = ops.PackedConv(ks, [cin], stride=1, pad=pad, dtype=cdt)
        wp, wd = pack.fwd(w), pack.dgrad(w)
        y = torch.empty(B, cout, hh, ww, device="cuda")
        dx, dW = torch.empty_like(x), torch.zeros_like(w)
        pa, ia, pd = ptr_array([x]), int_array([cin]), ptr_array([dx])
        fl = 2.0 * B * hh * ww * cin * ks * ks * cout
        # tile 100 = a TRAINING call (split-K allowed, one accumulation chain); tile 0 = the inference / parity path (segmented sums)
        ms_f = _time_launch(lambda: check(L.rsis_conv2d_fwd(pa, ia, 1, B, hh, ww, ptr(wp), cout, ks, 1, pad, None, None, ptr(y), hh, ww,
                                                            0 if inference else 100, cdt, stream()), "fwd"), iters)
        if inference:
            f = fam.setdefault("conv%dx%d fwd (eval)" % (ks, ks), {"flops": 0.0, "ms": 0.0, "bytes": 0.0, "launches": 0})
            f["flops"] += fl * count
            f["ms"] += ms_f * count
            f["bytes"] += 4.0 * B * hh * ww * (cin + cout) * count
            f["launches"] += count
            continue
        ms_d = _time_launch(lambda: check(L.rsis_conv2d_dgrad(ptr(y), B, cout, hh, ww, ptr(wd), cin, ks, 1, pad, pd, ia, 1, hh, ww, None, 0, cdt,
                                                              stream()), "dgrad"), iters)
        ms_w = _time_launch(lambda: check(L.rsis_conv2d_wgrad(ptr(y), ptr(x), ptr(dW), B, cin, hh, ww, cout, hh, ww, ks, 1, pad, cin, 0, 0, dt,
                                                              stream()), "wgrad"), iters)
        act_bytes = 4.0 * B * hh * ww * (cin + cout)
        for name, ms, n in (("conv%dx%d fwd+dgrad%s" % (ks, ks, " (winograd)" if wino else ""), ms_f + ms_d, 2), ("conv%dx%d wgrad" % (ks, ks), ms_w, 1)):
            f = fam.setdefault(name, {"flops": 0.0, "ms": 0.0, "bytes": 0.0, "launches": 0})
            f["flops"] += n * fl * count
            f["ms"] += ms * count
            f["bytes"] += n * act_bytes * count
            f["launches"] += n * count
    # the weight gradients as the training step launches them: parked during backward, flushed in ONE rsis_conv2d_wgrad_batch call
    # (grouped launches over all layers of a tile configuration); per family = all layers of that kernel size in one call
    from rsis_amd._lib import WgradJob
    for ksz in (() if inference else (3, 1)):
        jobs, keep, fl, by = [], [], 0.0, 0.0
        for si, (cin, cout, ks, hw, count) in enumerate(TRUNK_SHAPES):
            if ks != ksz:
                continue
            hh, ww = hw * _hw(imsize)[0] // 256, hw * _hw(imsize)[1] // 256
            x = torch.randn(B, cin, hh, ww, device="cuda")
            y = torch.randn(B, cout, hh, ww, device="cuda")
            jdt, ebytes = dt, 4.0
            if blk_on and si < N_TRUNK_SHAPES:
                x, y, jdt, ebytes = ops.blk_from_nchw(x), ops.blk_from_nchw(y), ops.DTYPE_BF16_BLK, 2.0
            keep += [x, y]
            for _ in range(count):
                dW = torch.zeros(cout, cin, ks, ks, device="cuda")
                keep.append(dW)
                j = WgradJob()
                (j.dy, j.x, j.dW, j.B, j.Cs, j.H, j.W, j.Cout, j.Ho, j.Wo, j.ks, j.stride, j.pad, j.Ctot, j.c_off, j.lstm_hid, j.dtype) = (
                    y.data_ptr(), x.data_ptr(), dW.data_ptr(), B, cin, hh, ww, cout, hh, ww, ks, 1, ks // 2, cin, 0, 0, jdt)
                jobs.append(j)
                fl += 2.0 * B * hh * ww * cin * ks * ks * cout
                by += ebytes * B * hh * ww * (cin + cout)
        arr = (WgradJob * len(jobs))(*jobs)
        ms = _time_launch(lambda: check(L.rsis_conv2d_wgrad_batch(arr, len(jobs), stream()), "wgrad_batch"), max(2, iters // 2))
        f = fam["conv%dx%d wgrad" % (ksz, ksz)]
        f.update({"flops": fl, "ms": ms, "bytes": by, "launches": len(jobs), "grouped": True})
        del keep
    kern = {"conv1x1 fwd+dgrad": ("conv_igemm_kernel<..., V4>", "conv_blk_kernel<1, ...> on blocked bf16 activations" if blk_on else "conv_bf16_kernel<1, ...>"),
            "conv3x3 fwd+dgrad": ("conv3x3_direct_kernel<..., EPI_PLAIN>",
                                  "conv_blk_kernel<3, ...> (trunk, blocked bf16) + conv_bf16_kernel<3, ..., EPI_PLAIN> (skip convs, fp32 activations)"
                                  if blk_on else "conv_bf16_kernel<3, ..., EPI_PLAIN>"),
            "conv1x1 wgrad": ("conv_wgrad_tiled_group_kernel<..., 1, ...> (rsis_conv2d_wgrad_batch)",
                              "wgrad1_tr_group_kernel (rsis_conv2d_wgrad_batch: blk operands by LDS-DMA, MFMA operands by ds_read_b64_tr_b16)" if blk_on
                              else "wgrad1_bf16_group_kernel (rsis_conv2d_wgrad_batch)"),
            "conv3x3 wgrad": ("conv_wgrad_tiled_group_kernel<..., 3, ...> (rsis_conv2d_wgrad_batch)",
                              "wgrad3_tr_group_kernel (trunk, blk operands: LDS-DMA + ds_read_b64_tr_b16) + wgrad3_bf16_group_kernel (skip convs, fp32 operands)"
                              if blk_on else "wgrad3_bf16_group_kernel (rsis_conv2d_wgrad_batch)"),
            "conv3x3 fwd+dgrad (winograd)": ("conv_wino_f32_kernel (Winograd F(2x2,3x3), training calls of the layer 1-3 bottleneck convs)",) * 2,
            "conv1x1 fwd (eval)": ("conv_igemm_kernel<..., V4> (inference path)",
                                   "conv_blk_kernel<1, ...> + eval BatchNorm (+ ReLU) epilogue (rsis_blk_conv2d_bn_eval)" if blk_on else "conv_bf16_kernel<1, ...>"),
            "conv3x3 fwd (eval)": ("conv3x3_direct_kernel<..., EPI_PLAIN, FLUSH> (segmented accumulation)",
                                   "conv_blk_kernel<3, ...> + eval BatchNorm (+ ReLU) epilogue (trunk) + conv_bf16_kernel<3, ...> (skip convs)"
                                   if blk_on else "conv_bf16_kernel<3, ...>")}
    out = []
    for name, f in sorted(fam.items(), key=lambda kv: -kv[1]["ms"]):
        tf = f["flops"] / f["ms"] / 1e9
        r = {"family": name, "kernel": kern[name][0 if dtype == "fp32" else 1],
             ("layers_per_step" if f.get("grouped") else "launches_per_step"): f["launches"],
             "ms_per_step": round(f["ms"], 3), "avg_us": round(1e3 * f["ms"] / f["launches"] * (2 if "fwd+dgrad" in name else 1), 1),
             "tflops": round(tf, 1)}
        if dtype == "fp32" and "winograd" in name:
            # `tflops` is the DIRECT-EQUIVALENT rate (2 M K N of the convolution / time: what the throughput metric sees); the kernel executes
            # 16/36 of those flops on the matrix cores, and `frac` is EXECUTED / peak -- a roofline fraction, never above 1
            ex = tf * 16.0 / 36.0
            r.update({"tflops_executed": round(ex, 1), "bound": "mfma", "peak": PEAK_F32_MFMA_TFLOPS, "frac": round(ex / PEAK_F32_MFMA_TFLOPS, 4),
                      "frac_algorithmic": round(tf / PEAK_F32_MFMA_TFLOPS, 4), "peak_sustained": PEAK_F32_MFMA_SUSTAINED_TFLOPS,
                      "frac_of_sustained": round(ex / PEAK_F32_MFMA_SUSTAINED_TFLOPS, 4),
                      "note": "Winograd F(2x2,3x3): 2.25x fewer matrix flops than the direct form; bound in practice by the LDS (profiles/r06 wino counters)"})
        elif dtype == "fp32":
            r.update({"bound": "mfma", "peak": PEAK_F32_MFMA_TFLOPS, "frac": round(tf / PEAK_F32_MFMA_TFLOPS, 4),
                      "peak_sustained": PEAK_F32_MFMA_SUSTAINED_TFLOPS, "frac_of_sustained": round(tf / PEAK_F32_MFMA_SUSTAINED_TFLOPS, 4)})
        else:
            # bf16: the family is priced against the roof its arithmetic intensity (algorithmic FLOP per activation byte) puts it under --
            # the 3x3 convs of the trunk sit right of the ridge (2500 TFLOP/s / 8 TB/s = 312 FLOP/B), the 1x1 ones left of it
            gbs = f["bytes"] / f["ms"] / 1e6
            ai = f["flops"] / f["bytes"]
            r.update({"gbs": round(gbs, 1), "flop_per_byte": round(ai, 1)})
            if ai > PEAK_BF16_MFMA_TFLOPS * 1e3 / PEAK_HBM_GBS:
                r.update({"bound": "mfma", "peak": PEAK_BF16_MFMA_TFLOPS, "frac": round(tf / PEAK_BF16_MFMA_TFLOPS, 4)})
            else:
                r.update({"bound": "hbm", "peak": PEAK_HBM_GBS, "frac": round(gbs / PEAK_HBM_GBS, 4)})
        out.append(r)
    return out


# forward GFLOP per image of the whole model (trunk + skip convs + T x (gates + conv_out)), for the inference record
MODEL_FWD_GFLOP = {(256, 256): (20.374 + 2.416, 1.6704), (224, 224): (15.599 + 1.850, 1.2789), (512, 1024): (162.991 + 19.327, 13.363)}


def inference_record(o):
    """`bench.py --inference`: test() (reference src/test.py:16-50 -- eval-mode encoder once, T decoder steps, masks resized to the input,
    sigmoid; the caller of reference src/eval.py:262) on a resident synthetic batch, as a replayed hipGraph (rsis_amd.test.GraphedTest), timed
    like the training line: W untimed calls (the capture among them), then EXACTLY K replays between synchronisations, HIP events per
    replay.  One JSON line; `value` = images / s of whole batches through test()."""
    from rsis_amd.modules import FeatureExtractor, RSIS
    from rsis_amd.test import GraphedTest
    assert torch.cuda.is_available(), "bench.py needs the GPU (there is no CPU path)"
    imw = o.imsize_w or o.imsize
    a = bench_args(o.batch, o.imsize, o.T, o.dtype)
    torch.manual_seed(a.seed)
    enc, dec = FeatureExtractor(a).cuda().eval(), RSIS(a).cuda().eval()
    x = torch.randn(o.batch, 3, o.imsize, imw, device="cuda")
    run = GraphedTest(a, enc, dec)
    for _ in range(max(o.warmup, 4)):
        run(x)
    assert run.graph is not None, "test() was not captured"
    t_end = time.time() + 1.5                      # settle (clock ramp), untimed
    while time.time() < t_end:
        run(x)
    torch.cuda.synchronize()
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(o.steps + 1)]
    t0 = time.time()
    evs[0].record()
    for i in range(o.steps):
        run(x)
        evs[i + 1].record()
    torch.cuda.synchronize()
    dt = time.time() - t0
    per = sorted(evs[i].elapsed_time(evs[i + 1]) for i in range(o.steps))
    masks = run.outs[0]
    assert tuple(masks.shape[:2]) == (o.batch, o.T) and bool(torch.isfinite(masks).all())
    ms = 1000.0 * dt / o.steps
    out = {"metric": "inference images/sec at %dx%d, T=%d, batch=%d" % (o.imsize, imw, o.T, o.batch), "value": round(o.batch * o.steps / dt, 1),
           "unit": "images/s", "n_gpus": 1, "steps": o.steps, "warmup": o.warmup, "ms_per_step": round(ms, 3), "higher_is_better": True,
           "scaling": "weak", "vs_baseline": None, "dtype": "f32" if o.dtype == "fp32" else "bf16", "data": "synthetic",
           "config": {"workload": "test() of reference src/test.py:16-50 (eval-mode ResNet-101 encoder once, %d decoder timesteps over the 5-scale "
                                  "ConvLSTM pyramid, masks resized to %dx%d, sigmoid), batch %d, %s; a 'step' = one batch through test()"
                                  % (o.T, o.imsize, imw, o.batch, o.dtype),
                      "launch": "hipGraph replay of the captured test() (rsis_amd.test.GraphedTest)",
                      "event_ms_per_batch": {"median": round(per[len(per) // 2], 3), "min": round(per[0], 3), "max": round(per[-1], 3)}}}
    g = MODEL_FWD_GFLOP.get((o.imsize, imw))
    if g is not None:
        gf = (g[0] + o.T * g[1]) * o.batch
        tf = gf / ms
        peak = PEAK_F32_MFMA_TFLOPS if o.dtype == "fp32" else PEAK_BF16_MFMA_TFLOPS
        out["model_flops"] = {"algorithmic_gflop_per_batch": round(gf, 1), "tflops": round(tf, 1), "peak": peak, "frac": round(tf / peak, 4),
                              "note": "whole-model forward FLOPs (SURVEY Appendix A, full-K gates) / time of a batch: a throughput figure, "
                                      "not a kernel roofline (the per-family ones are in roofline_kernels)"}
    if not o.skip_roofline:
        out["roofline_kernels"] = trunk_kernel_rooflines(o.batch, max(3, o.kernel_iters // 4), (o.imsize, imw), o.dtype, inference=True)
    print(json.dumps(out))


def cpu_baseline(imsize, T, budget_s=25.0):
    """The oracle's restated train step (oracle.run_iter_forward + backward) on the host cores, bounded sample."""
    from oracle import rsis_oracle as O
    from rsis_amd.synthetic import synthetic_batch
    cores = min(os.cpu_count() or 1, 32)     # a 256-core host oversubscribes a B=2 step; 32 threads are what is timed
    torch.set_num_threads(cores)
    B = 4
    a = bench_args(B, imsize, T)
    a.use_gpu = False
    torch.manual_seed(0)
    enc, dec = O.FeatureExtractor(a), O.RSIS(a)
    x, y_mask, y_class, sw_mask, sw_class = synthetic_batch(123, B, imsize, imsize, 20, 12, 21, device="cpu")

    # the reference's two optimizers (train.py:236-240: decoder + skip convs at lr, trunk at lr_cnn), stepped like runIter does
    skip = [p for k, p in enc.named_parameters() if not k.startswith("base.")]
    base = [p for k, p in enc.named_parameters() if k.startswith("base.") and not k.startswith("base.fc")]
    dec_opt = torch.optim.Adam(list(dec.parameters()) + skip, lr=a.lr, weight_decay=a.weight_decay)
    enc_opt = torch.optim.Adam(base, lr=a.lr_cnn, weight_decay=a.weight_decay_cnn)

    def step():
        enc.zero_grad()
        dec.zero_grad()
        r = O.run_iter_forward(a, enc, dec, x, y_mask, y_class, sw_mask, sw_class, mode="train")
        r["loss"].backward()
        dec_opt.step()
        enc_opt.step()
    t0 = time.time()
    step()                       # warm-up
    warm = time.time() - t0
    n, t0 = 0, time.time()
    while n < 1 or (time.time() - t0 + warm) < budget_s and n < 12:
        step()
        n += 1
    dt = (time.time() - t0) / n
    return {"value": round(B / dt, 3), "unit": "images/s", "cores": cores, "kind": "port",
            "sample": "oracle train step (fwd+match+losses+bwd+2 Adam steps), B=%d, %dx%d, T=%d, fp32, %d timed steps after 1 warm-up"
                      % (B, imsize, imsize, T, n)}


# the fused ConvLSTM gate kernel in a profiler's kernel-name column: template arguments <BM, TW, TH, NI, EPI, KSP, NWV>, EPI == 1 is
# the fused LSTM epilogue (tests/test_abi.py checks the pattern against the symbols of the built library)
REAL_STDOUT = 1
GATE_KERNEL_RE = re.compile(r"conv3x3_direct_kernel<\d+, \d+, \d+, \d+, 1, \d+, \d+(, (true|false))*>")
# ... and the grouped launch of one wavefront diagonal (rsis_convlstm_fwd_batch): template arguments <EPI, FLUSH> (FLUSH: the segmented-
# accumulation instantiation of inference / deep-K calls, not what a training step launches)
GATE_GROUP_RE = re.compile(r"conv3x3_direct_group_kernel<1(, false)?>")
# ... and its bf16 twin, conv_bf16_kernel<KS, BM, TW, TH, EPI, CKB, V4> with KS == 3 and EPI == 1 (five single launches per diagonal)
GATE_BF16_RE = re.compile(r"conv_bf16_kernel<3, \d+, \d+, \d+, 1, \d+, \w+>")
# ... and the grouped launch on channel-blocked bf16 tensors (rsis_blk_conv3x3_batch, LSTM epilogue): template argument <EPI>
GATE_BLK_RE = re.compile(r"conv_blk_dec_group_kernel<1, \d>")


def gate_kernel_traffic(batch, imsize, dtype="fp32", timeout=150):
    """HBM bytes of the gate-kernel launch of one timestep (the grouped launch of one wavefront diagonal: five levels) from the memory-side PMC counters: two rocprofv3 passes
    (FETCH_SIZE, then WRITE_SIZE -- they do not fit one pass) over `bench.py --roofline-only` in a child process.
    FETCH_SIZE is doubled as MI355X_MICROARCH.md prescribes for gfx950 (re-measured for this library's load flavours in
    profiles/r02_fetch_calibration.txt).  fp32: ONE launch shape (the grouped kernel); bf16: the five single launches of the
    diagonal, summed.  Returns (bytes, detail) or (None, why)."""
    name_re, n_shapes = (GATE_GROUP_RE, 1) if dtype == "fp32" else (GATE_BF16_RE, 5)
    blk = False
    if dtype != "fp32":
        from rsis_amd import decoder_seq
        blk = decoder_seq.BLK_ENABLED[0]
        if blk:        # the child launches only the timestep's call (RSIS_BENCH_DIAG_ONLY): one grid, or several when jobs go out alone
            name_re, n_shapes = GATE_BLK_RE, 0
    import csv
    import shutil
    import statistics
    import tempfile
    if shutil.which("rocprofv3") is None:
        return None, "rocprofv3 not found"
    tmp = tempfile.mkdtemp(prefix="rsis_pmc_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp", RSIS_BENCH_DIAG_ONLY="1")
    per_counter = {}
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            cmd = ["rocprofv3", "--kernel-trace", "--pmc", counter, "--output-format", "csv", "-d", tmp, "-o", counter.lower(), "--",
                   sys.executable, os.path.abspath(__file__), "--roofline-only", "--product-only", "--kernel-iters", "4", "--batch", str(batch),
                   "--imsize", str(_hw(imsize)[0]), "--imsize-w", str(_hw(imsize)[1]), "--dtype", dtype]
            subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=timeout, check=True)
            path = None
            for root, _d, files in os.walk(tmp):
                for f in files:
                    if f.startswith(counter.lower()) and f.endswith("counter_collection.csv"):
                        path = os.path.join(root, f)
            if path is None:
                return None, "no counter_collection.csv from rocprofv3"
            vals = {}
            with open(path) as f:
                for r in csv.DictReader(f):
                    k = r["Kernel_Name"]
                    if r["Counter_Name"] == counter and name_re.search(k):
                        vals.setdefault((k, r["Grid_Size"]), []).append(float(r["Counter_Value"]))
            if (n_shapes and len(vals) != n_shapes) or not vals:
                return None, "expected %d gate-kernel launch shape(s) in the counter file, found %d" % (n_shapes, len(vals))
            per_counter[counter] = sum(statistics.median(v) for v in vals.values()) * 1024.0      # counters are in KiB
    except Exception as e:  # noqa: BLE001  (profiler missing / refused / timed out: the figure stays null)
        return None, "rocprofv3 pass failed: %r" % (e,)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    fetch, write = 2.0 * per_counter["FETCH_SIZE"], per_counter["WRITE_SIZE"]
    return fetch + write, {"fetch_bytes_x2": fetch, "write_bytes": write}


def _gate_roofline(batch, iters, imsize, dtype, T, product_only=False):
    """the gate-kernel roofline leg in the form the product launches for this dtype"""
    if dtype != "fp32":
        from rsis_amd import decoder_seq
        if decoder_seq.BLK_ENABLED[0]:
            return gate_kernel_roofline_blk(batch, iters, imsize, T)
    return gate_kernel_roofline(batch, iters, imsize, dtype, T, product_only=product_only)


def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def visible_devices():
    return torch.cuda.device_count() if torch.cuda.is_available() else 0


def check_world(gpus):
    """`--gpus N` means N ranks, one per GPU (replaces nn.DataParallel, /root/reference src/train.py:269-274).  A launch whose world size
    is not N, or that sees fewer than N devices (unless the test hook RSIS_SHARE_GPU=1 puts several gloo ranks on one GPU), exits
    non-zero instead of printing a line labelled with another GPU count."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != gpus:
        sys.exit("bench.py: --gpus %d but WORLD_SIZE=%d: launch with `python bench.py --gpus %d` (self-launching) or "
                 "`python -m torch.distributed.run --nproc-per-node %d bench.py --gpus %d`" % (gpus, world, gpus, gpus, gpus))
    ndev = visible_devices()
    if ndev < gpus and os.environ.get("RSIS_SHARE_GPU", "") != "1":
        sys.exit("bench.py: --gpus %d but only %d device(s) visible (one rank per GPU; no N-rank run is faked)" % (gpus, ndev))


def self_launch(gpus):
    """plain `python bench.py --gpus N` (the shape of the driver's N=1 command): become `torch.distributed.run --standalone` with N ranks
    of this very command line.  Rank 0 of the child job prints the JSON line; the launcher adds nothing to stdout."""
    ndev = visible_devices()
    if ndev < gpus and os.environ.get("RSIS_SHARE_GPU", "") != "1":
        sys.exit("bench.py: --gpus %d but only %d device(s) visible (one rank per GPU; no N-rank run is faked)" % (gpus, ndev))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"), OMP_NUM_THREADS=os.environ.get("OMP_NUM_THREADS", "8"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    print("[bench] self-launch: %s" % " ".join(cmd), file=sys.stderr, flush=True)
    os.execve(sys.executable, cmd, env)


def exchange_report(a, encoder, decoder, crits, optims, reducer, gstep, seg, batch, t_run, rank_ms, o, fence, note):
    """What the gradient exchange of this run was and what it cost -- in the JSON line, so that a scaling curve explains itself
    (VERDICT r4 item 7).  Runs on EVERY rank after the timed region (it issues collectives): (1) which schedule ran (direct RCCL inside
    the iteration's graph / cut graphs over torch.distributed / eager staged) and why, (2) the bytes of the three gradient ranges, (3)
    the collectives ALONE -- the three ranges all-reduced back to back with nothing to overlap: the un-hidden cost of the exchange,
    (4) the same iteration WITHOUT any exchange (a second captured graph, reducer off; after the timed region the replicas may
    drift): step - this = the EXPOSED part of the exchange, (5) per-rank step times."""
    if reducer is None or not getattr(reducer, "active", False):
        return {"mode": "none", "world": 1, "note": "single process, no gradient exchange (RSIS_FORCE_DIST=1 runs the collective path at world 1)"}
    from rsis_amd import comm as _comm
    from rsis_amd.train import EXCHANGE_CUTS, GraphedStep, exchange_plan
    world = dist.get_world_size()
    captured = gstep is not None and gstep.graph is not None
    direct = captured and gstep.split and gstep.direct is not None
    cuts = gstep.cuts if gstep is not None else EXCHANGE_CUTS
    plan = exchange_plan(encoder, optims, cuts, bool(a.update_encoder))
    rep = {"mode": "direct-in-graph" if direct else ("cut-graphs" if captured else "eager-staged"),
           "backend": dist.get_backend(), "world": world, "cuts": 0 if direct else cuts,
           "rccl_direct": dict(_comm.LAST_STATUS),
           "range_bytes": {k: int(sum(b.numel() * b.element_size() for b in v)) for k, v in plan.items()},
           "rank_ms_per_step": {"min": round(min(rank_ms), 3), "max": round(max(rank_ms), 3), "all": [round(v, 3) for v in rank_ms]}}
    try:
        red = gstep._reduce if gstep is not None else (lambda b, _a=False: dist.all_reduce(b, op=dist.ReduceOp.SUM))
        bufs = [b for k in ("dec", "trunk_hi", "rest") for b in plan[k]]
        for b in bufs:
            b.zero_()
        fence()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 5
        for b in bufs:
            red(b)
        e0.record()
        for _ in range(reps):
            for b in bufs:
                red(b)
        e1.record()
        fence()
        rep["allreduce_alone_ms"] = round(e0.elapsed_time(e1) / reps, 3)
        nbytes = sum(rep["range_bytes"].values())
        rep["allreduce_alone_busbw_gbs"] = round(2.0 * (world - 1) / world * nbytes / (e0.elapsed_time(e1) / reps) / 1e6, 1) if world > 1 else None
    except Exception as ex:  # noqa: BLE001
        rep["allreduce_alone_ms"] = "failed: %r" % (ex,)
    if captured and not o.no_graph:
        try:
            if seg is not None and "exposed_allreduce" in seg and not direct:
                rep["exposed_allreduce_ms_events"] = round(seg["exposed_allreduce"], 3)
            gstep.release()
            g2 = GraphedStep(a, encoder, decoder, crits, optims, None, warm=1)
            while g2.graph is None and g2.failed is None:
                g2(batch, t_run)
            if g2.graph is None:
                raise RuntimeError("capture failed: %s" % g2.failed)
            for _ in range(3):
                g2(batch, t_run)
            n = max(5, o.steps // 3)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(n):
                g2(batch, t_run)
            e1.record()
            torch.cuda.synchronize()
            ms0 = e0.elapsed_time(e1) / n
            rep["step_ms_without_exchange"] = round(ms0, 3)
            rep["exposed_allreduce_ms"] = round(max(rank_ms) - ms0, 3)
            rep["exposed_note"] = ("max-over-ranks step time minus this rank's step time of the same captured iteration with the exchange off "
                                   "(measured after the timed region); includes schedule overheads of the exchange (fork / join, cuts), not only wire time")
            g2.release()
        except Exception as ex:  # noqa: BLE001
            rep["step_ms_without_exchange"] = "failed: %r" % (ex,)
    note("exchange: %s" % json.dumps(rep))
    return rep


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)      # (2 s of timed region at the headline configuration)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=32, help="per-GPU batch (weak scaling)")
    ap.add_argument("--imsize", type=int, default=256)
    ap.add_argument("--imsize-w", type=int, default=0, help="image width when it differs from --imsize (the height): 512 x 1024 = --imsize 512 --imsize-w 1024")
    ap.add_argument("--T", type=int, default=10)
    ap.add_argument("--dtype", default="fp32", choices=["fp32", "bf16"], help="arithmetic of the conv / gate MFMA kernels")
    ap.add_argument("--kernel-iters", type=int, default=20)
    ap.add_argument("--skip-cpu", action="store_true")
    ap.add_argument("--skip-roofline", action="store_true")
    ap.add_argument("--skip-traffic", action="store_true", help="do not run the two rocprofv3 --pmc passes behind roofline.traffic")
    ap.add_argument("--no-graph", action="store_true", help="launch every kernel of a step from Python instead of replaying the "
                    "captured hipGraph of the iteration (rsis_amd.train.GraphedStep)")
    ap.add_argument("--no-settle", action="store_true", help="skip the untimed settle phase after the warm-up steps")
    ap.add_argument("--settle-min", type=float, default=2.0, help="minimum seconds of the untimed settle phase")
    ap.add_argument("--settle-cap", type=float, default=10.0, help="maximum seconds of the untimed settle phase")
    ap.add_argument("--cpu-baseline-only", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--product-only", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--skip-secondary", action="store_true", help="do not run the bf16 224x224 leg (BASELINE configs[2]) after the headline run")
    ap.add_argument("--encoder-frozen", action="store_true", help="update_encoder off -- the reference's state until -finetune_after epochs have passed "
                    "(args.py:39-45): rsis_amd then does not compute the trunk's backward at all (FeatureExtractor.trunk_grad); a SECONDARY record, "
                    "never the headline line")
    ap.add_argument("--inference", action="store_true", help="time test() (reference src/test.py:16-50) instead of the training iteration")
    ap.add_argument("--roofline-only", action="store_true",
                    help="run only the gate-kernel roofline leg and print its object (for `rocprofv3 --kernel-trace --stats`: the "
                         "profile then holds exactly the launches the `roofline` figure is computed from)")
    o = ap.parse_args()
    if o.cpu_baseline_only:
        print(json.dumps(cpu_baseline(o.imsize, o.T)))
        return
    if o.roofline_only:
        assert torch.cuda.is_available(), "bench.py needs the GPU (there is no CPU path)"
        print(json.dumps(_gate_roofline(o.batch, o.kernel_iters, (o.imsize, o.imsize_w or o.imsize), o.dtype, o.T, product_only=o.product_only)))
        return

    if o.inference:
        inference_record(o)
        return

    if o.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(o.gpus)              # (does not return: the process becomes the launcher of N ranks)

    from rsis_amd.train import GraphedStep, build_optimizers, init_distributed, runIter
    from rsis_amd.modules import FeatureExtractor, RSIS
    from rsis_amd.optim import BucketedAllReduce
    from rsis_amd.synthetic import synthetic_batch
    from rsis_amd.utils.objectives import MaskedBCELoss, MaskedNLLLoss, softIoULoss

    # ONE line on stdout, whatever the libraries print: RCCL writes its version banner to stdout when the first communicator is
    # created.  From here on file descriptor 1 is stderr; the JSON line goes to the saved descriptor.
    global REAL_STDOUT
    sys.stdout.flush()
    REAL_STDOUT = os.dup(1)
    os.dup2(2, 1)
    assert torch.cuda.is_available(), "bench.py needs the GPU (there is no CPU path)"
    check_world(o.gpus)                  # never a mislabelled line: --gpus N runs N ranks on N devices or exits non-zero
    rank, local_rank, world = init_distributed()
    a = bench_args(o.batch, o.imsize, o.T, o.dtype)
    if o.encoder_frozen:
        a.update_encoder = False
    torch.manual_seed(a.seed)
    encoder, decoder = FeatureExtractor(a).cuda(), RSIS(a).cuda()
    if world > 1 or os.environ.get("RSIS_FORCE_DIST", "") == "1":
        for p in list(encoder.parameters()) + list(decoder.parameters()) + list(encoder.buffers()):
            dist.broadcast(p.data, 0)
    enc_opt, dec_opt = build_optimizers(a, encoder, decoder)
    force = os.environ.get("RSIS_FORCE_DIST", "") == "1"
    reducer = BucketedAllReduce([dec_opt.group, enc_opt.group], force=force) if (world > 1 or force) else None
    if reducer is not None and reducer.active:
        from rsis_amd.comm import make_direct_reducer
        reducer.direct = make_direct_reducer(lambda m: print("[bench] rank %d: %s" % (rank, m), file=sys.stderr, flush=True) if rank == 0 else None)
    crits = [softIoULoss(), MaskedNLLLoss(None), MaskedBCELoss(a.stop_balance_weight)]
    imw = o.imsize_w or o.imsize
    geom = (o.imsize, imw)
    # 12 instances per image; with T > 12 every slot is an instance, so that all T steps run (train.py:87-92 stops after the
    # first step whose slot is empty in every image)
    batch = synthetic_batch(a.seed + 1000 * rank, o.batch, o.imsize, imw, a.gt_maxseqlen, max(12, min(o.T, a.gt_maxseqlen)), a.num_classes, "cuda")

    from rsis_amd.train import steps_to_run
    t_run = steps_to_run(a, batch[3])      # early-stop rule evaluated once for the resident batch (it is all T steps here)

    # the whole iteration (fwd, matching, losses, bwd, all-reduce, Adam, repack) is captured once as a hipGraph and replayed:
    # every step still executes all of its kernels, the host just stops paying ~35 us of Python per launch
    gstep = None if o.no_graph else GraphedStep(a, encoder, decoder, crits, [enc_opt, dec_opt], reducer, warm=min(2, max(1, o.warmup - 1)))

    def step():
        if gstep is not None:
            return gstep(batch, t_run)
        return runIter(a, encoder, decoder, *batch, crits, [enc_opt, dec_opt], mode="train", reducer=reducer, sync_losses=False,
                       t_run=t_run, want_outs=False)

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def note(msg):
        if rank == 0:
            print("[bench] %s" % msg, file=sys.stderr, flush=True)

    def health(tag, losses):
        """RSIS_BENCH_DEBUG=1: loss values and non-finite parameters / optimizer moments after each phase"""
        if os.environ.get("RSIS_BENCH_DEBUG", "") != "1":
            return
        torch.cuda.synchronize()
        bad = [k for k, p in list(encoder.named_parameters()) + list(decoder.named_parameters()) if not torch.isfinite(p).all()]
        mom = [(o.group.name, int((~torch.isfinite(o.group.exp_avg)).sum()), int((~torch.isfinite(o.group.exp_avg_sq)).sum()),
                int((~torch.isfinite(o.group.flat_g)).sum())) for o in (enc_opt, dec_opt)]
        note("health[%s]: losses %s, non-finite params %d %s, (group, m, v, g non-finite) %s"
             % (tag, [round(float(v), 5) for v in losses], len(bad), bad[:4], mom))

    roof = roof_kernels = None
    if rank == 0 and not o.skip_roofline:
        roof = _gate_roofline(o.batch, o.kernel_iters, geom, o.dtype, o.T)
        note("gate kernel roofline: executed %s, algorithmic %s, full-K %s TFLOP/s; %s %s of %s %s" % (
            roof["achieved_executed"], roof["achieved_algorithmic"], roof["full_k"]["achieved"], roof["bound"], roof["achieved"], roof["peak"], roof["unit"]))
        roof_kernels = trunk_kernel_rooflines(o.batch, max(3, o.kernel_iters // 4), geom, o.dtype)
        note("roofline_kernels: %s" % "; ".join("%s %.1f TF/s" % (r["family"], r["tflops"]) for r in roof_kernels))
        if world == 1 and not o.skip_traffic:
            t0 = time.time()
            traffic, detail = gate_kernel_traffic(o.batch, geom, o.dtype)
            roof["traffic"] = traffic
            if traffic is not None:
                roof["traffic_unit"] = ("bytes per timestep (the gate launches of the 5 levels): 2 x FETCH_SIZE + WRITE_SIZE, rocprofv3 --pmc, separate "
                                        "passes; the x2 is the guide's gfx950 correction, re-measured for this kernel's dword LDS-DMA reads in "
                                        "profiles/r02_fetch_calibration.txt")
                roof["traffic_vs_algorithmic"] = round(traffic / (roof["algorithmic_mbytes_per_timestep"] * 1e6), 3)
                roof["traffic_vs_survey_minimal"] = round(traffic / (roof["survey_minimal_mbytes_per_timestep"] * 1e6), 3)
                roof["traffic_detail"] = detail
            note("gate kernel HBM traffic: %s (%.0f s)" % (traffic if traffic is not None else detail, time.time() - t0))
    tw = time.time()
    for i in range(o.warmup):
        losses = step()[0]
        if i == 0:
            torch.cuda.synchronize()
            note("first step %.2f s" % (time.time() - tw))
        health("warmup %d" % i, losses)
    while gstep is not None and gstep.graph is None and gstep.failed is None:
        losses = step()[0]              # (still warm-up: the call that captures the graph must not fall into the timed region)
    # Untimed settle phase (still warm-up): keep stepping until the step time has been stable for a while (cold-box clock
    # ramp, allocator growth, first-use code loading), so that the K timed steps measure steady state.
    if not o.no_settle:
        hist, t_settle = [], time.time()
        while time.time() - t_settle < o.settle_cap:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            losses = step()[0]
            e1.record()
            e1.synchronize()
            hist.append(e0.elapsed_time(e1))
            done = len(hist) >= 10 and time.time() - t_settle > o.settle_min and max(hist[-10:]) < 1.05 * min(hist)
            if world > 1:     # every rank must leave the loop in the same iteration
                flag = torch.tensor([1.0 if done else 0.0], device="cuda")
                dist.all_reduce(flag, op=dist.ReduceOp.MIN)
                done = bool(flag.item() > 0.5)
            if done:
                break
        note("settle: %d extra untimed steps, last %s ms" % (len(hist), " ".join("%.1f" % h for h in hist[-4:])))
    health("after settle", losses)
    fence()
    if gstep is not None:
        note("hipGraph: %s" % ("captured, replaying" if gstep.graph is not None else "NOT captured (%s): eager launches" % gstep.failed))
    note("warmup done %.2f s" % (time.time() - tw))
    if dist.is_initialized():
        direct = gstep is not None and gstep.graph is not None and gstep.split and gstep.direct is not None
        note("gradient exchange: backend %s, communicator size %d (%s)" % (dist.get_backend(), dist.get_world_size(),
             "RCCL bound directly, ONE graph: the all-reduces of the three gradient ranges are nodes on a forked branch" if direct else
             "%d cuts: graph A | all-reduce(dec) || graph B1 | all-reduce(layers 3-4) || graph B2 | all-reduce(rest) | graph C" % gstep.cuts
             if (gstep is not None and gstep.graph is not None and gstep.split)
             else "staged all-reduce at the cuts of the split backward (eager launches)"))
    if gstep is not None and gstep.graph is not None and gstep.split:
        gstep.timing = True            # HIP events around graph A / graph B (+ overlapped collective) / exposed collective / graph C
    t0 = time.time()
    marks, evs = [], [torch.cuda.Event(enable_timing=True) for _ in range(o.steps + 1)]
    evs[0].record()
    for i in range(o.steps):
        losses = step()[0]
        evs[i + 1].record()
        marks.append(time.time() - t0)      # host enqueue progress (no sync): shows a host-bound step at a glance
    fence()
    dt = time.time() - t0
    health("after timed", losses)
    note("host enqueue marks (s): %s | end %.3f" % (" ".join("%.3f" % m for m in marks), dt))
    note("GPU ms per step (events): %s" % " ".join("%.1f" % evs[i].elapsed_time(evs[i + 1]) for i in range(o.steps)))
    seg = gstep.segment_ms() if (gstep is not None and gstep.timing) else None
    if seg is not None and "mode" in seg:
        note("direct in-graph exchange, ms per step over %d replays: %.3f (one graph)" % (seg["replays"], seg["one_graph_fwd_bwd_allreduce_adam_repack"]))
    elif seg is not None:
        note("split-graph schedule (%d cuts), ms per step over %d replays: %s | EXPOSED all-reduce (wait for the ranges in flight + the "
             "remaining range) %.2f | graph C (Adam + repack) %.2f"
             % (seg["cuts"], seg["replays"], " | ".join("%s %.2f" % (k, v) for k, v in seg.items() if k.startswith(("graph_A", "graph_B"))),
                seg["exposed_allreduce"], seg["graph_C_adam_repack"]))
    rank_ms = [1000.0 * dt / o.steps]
    if world > 1:
        allt = [torch.zeros(1, device="cuda", dtype=torch.float64) for _ in range(world)]
        dist.all_gather(allt, torch.tensor([dt], device="cuda", dtype=torch.float64))
        rank_ms = [1000.0 * float(t.item()) / o.steps for t in allt]
        dt = max(float(t.item()) for t in allt)                                   # MAX over ranks
    loss_val = float(losses[0])
    assert loss_val == loss_val, "loss is NaN"
    captured_launch = gstep is not None and gstep.graph is not None      # (exchange_report releases the captured graphs)
    # The reference hands runIter HOST tensors (utils.py:batch_to_var: x, y_mask, y_class, sw_mask, sw_class of one DataLoader batch, `.cuda()`
    # per step).  `value` above is measured with the batch resident in HBM; this is the same step with the five tensors crossing PCIe from
    # pinned host memory in the timed region -- (a) in stream order in front of every step, (b) double-buffered on a copy stream under the
    # previous step.  A reported figure (measurement), never `value`.
    pcie = None
    if rank == 0 and world == 1 and not o.skip_secondary and os.environ.get("RSIS_BENCH_PCIE", "1") != "0":
        try:
            host = [t.detach().cpu().pin_memory() for t in batch]
            nbytes = sum(t.numel() * t.element_size() for t in host)
            n_h = max(5, min(20, o.steps))

            def timed(fn):
                for _ in range(2):
                    fn()
                torch.cuda.synchronize()
                t_ = time.time()
                for _ in range(n_h):
                    fn()
                torch.cuda.synchronize()
                return 1000.0 * (time.time() - t_) / n_h

            def serial():
                for d_, h_ in zip(batch, host):
                    d_.copy_(h_, non_blocking=True)
                step()

            ms_serial = timed(serial)
            sets = [batch, tuple(torch.empty_like(t) for t in batch)]
            side, ready, free = torch.cuda.Stream(), [torch.cuda.Event(), torch.cuda.Event()], [torch.cuda.Event(), torch.cuda.Event()]
            for e_ in ready + free:
                e_.record()
            state = {"k": 0}

            def overlapped():
                k = state["k"]
                torch.cuda.current_stream().wait_event(ready[k])           # this step's batch has arrived
                if gstep is not None:
                    gstep(sets[k], t_run)
                else:
                    runIter(a, encoder, decoder, *sets[k], crits, [enc_opt, dec_opt], mode="train", reducer=reducer, sync_losses=False, t_run=t_run, want_outs=False)
                free[k].record()                                            # (conservative: the set is free once the whole step has run)
                side.wait_event(free[1 - k])
                with torch.cuda.stream(side):                               # the NEXT step's batch crosses PCIe under this step
                    for d_, h_ in zip(sets[1 - k], host):
                        d_.copy_(h_, non_blocking=True)
                    ready[1 - k].record()
                state["k"] = 1 - k

            ms_over = timed(overlapped)
            torch.cuda.current_stream().wait_stream(side)
            pcie = {"host_bytes_per_step": nbytes, "ms_per_step_h2d_in_stream_order": round(ms_serial, 3),
                    "images_per_s_h2d_in_stream_order": round(o.batch / ms_serial * 1e3, 1),
                    "ms_per_step_h2d_double_buffered": round(ms_over, 3), "images_per_s_h2d_double_buffered": round(o.batch / ms_over * 1e3, 1),
                    "steps": n_h, "note": "the five runIter tensors of one batch (reference utils.py:batch_to_var) from pinned host memory every step; "
                                          "`value` is measured with the batch resident in HBM"}
            note("PCIe-inclusive: %s" % (pcie,))
        except Exception as ex:  # noqa: BLE001
            pcie = {"error": repr(ex)}
    exchange = exchange_report(a, encoder, decoder, crits, [enc_opt, dec_opt], reducer, gstep, seg, batch, t_run, rank_ms, o, fence, note)

    cpu, out, secondary = None, None, None
    if rank == 0:
        note("timed region %.3f s for %d steps" % (dt, o.steps))
        if world == 1 and not o.skip_cpu:
            # own process + hard timeout: the CPU leg can never stall the GPU bench
            import subprocess
            try:
                r = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-baseline-only", "--imsize", str(o.imsize),
                                    "--T", str(o.T)], capture_output=True, text=True, timeout=240)
                cpu = json.loads(r.stdout.strip().splitlines()[-1])
            except Exception as ex:  # noqa: BLE001
                cpu = {"value": None, "unit": "images/s", "cores": None, "kind": "port", "sample": "failed: %r" % (ex,)}
            note("cpu baseline: %s" % (cpu,))
        secondary = None
        if world == 1 and o.dtype == "fp32" and not o.skip_secondary:
            # BASELINE configs[2] (224x224, T=10, batch 32, bf16) as a secondary record, and the SAME geometry under fp32 next to it
            # (so that the speed-up of the bf16 kernels is stated on equal geometry): own processes, same harness
            import subprocess
            secondary = []
            # (third record: bf16 operands on fp32 NCHW activations, RSIS_BF16_STORAGE=0 -- what the blocked bf16 trunk is measured against)
            for sd, extra, env in (("bf16", [], {}), ("fp32", ["--skip-traffic"], {}),
                                   ("bf16", ["--skip-traffic", "--skip-roofline"], {"RSIS_BF16_STORAGE": "0", "RSIS_DECODER_BLK": "0"})):
                try:
                    r = subprocess.run([sys.executable, os.path.abspath(__file__), "--dtype", sd, "--imsize", "224", "--batch", str(o.batch), "--T",
                                        str(o.T), "--steps", str(o.steps), "--warmup", str(o.warmup), "--skip-cpu", "--skip-secondary",
                                        "--kernel-iters", str(o.kernel_iters)] + extra, capture_output=True, text=True, timeout=600,
                                       env=dict(os.environ, **env))
                    sj = json.loads(r.stdout.strip().splitlines()[-1])
                    sj["config"]["workload"] = sj["config"]["workload"].replace("configs[1]", "configs[2] geometry" if sd == "fp32" else "configs[2]")
                    secondary.append({k: sj.get(k) for k in ("metric", "value", "unit", "ms_per_step", "dtype", "config", "roofline", "roofline_kernels")})
                except Exception as ex:  # noqa: BLE001
                    secondary.append({"dtype": sd, "error": repr(ex)})
            if all(s_.get("value") for s_ in secondary):
                secondary[0]["speedup_over_fp32_same_geometry"] = round(secondary[0]["value"] / secondary[1]["value"], 3)
                secondary[0]["speedup_over_bf16_operands_on_fp32_activations"] = round(secondary[0]["value"] / secondary[2]["value"], 3)
            # BASELINE configs[4]'s per-GPU workload (Cityscapes geometry 512x1024, T=20, batch 8 per GPU, bf16), same harness, graph replay
            try:
                r = subprocess.run([sys.executable, os.path.abspath(__file__), "--dtype", "bf16", "--imsize", "512", "--imsize-w", "1024", "--batch", "8", "--T", "20",
                                    "--steps", str(max(5, o.steps // 2)), "--warmup", str(o.warmup), "--skip-cpu", "--skip-secondary",
                                    "--kernel-iters", str(max(4, o.kernel_iters // 2))],
                                   capture_output=True, text=True, timeout=900, env=dict(os.environ))
                sj = json.loads(r.stdout.strip().splitlines()[-1])
                sj["config"]["workload"] = sj["config"]["workload"].replace("configs[1]", "configs[4] per-GPU workload")
                secondary.append({k: sj.get(k) for k in ("metric", "value", "unit", "ms_per_step", "dtype", "config", "roofline", "roofline_kernels")})
            except Exception as ex:  # noqa: BLE001
                secondary.append({"dtype": "bf16", "config": "configs[4]", "error": repr(ex)})
            # the headline configuration once more with the collective path forced on at world size 1 (RSIS_FORCE_DIST=1: communicator,
            # captured all-reduces, 1 / world scaling all run; the wire is a loop-back): what the exchange machinery costs per step
            if os.environ.get("RSIS_FORCE_DIST", "") != "1":
                try:
                    r = subprocess.run([sys.executable, os.path.abspath(__file__), "--batch", str(o.batch), "--imsize", str(o.imsize), "--T", str(o.T),
                                        "--steps", str(max(10, o.steps // 2)), "--warmup", str(o.warmup), "--skip-cpu", "--skip-secondary", "--skip-roofline"],
                                       capture_output=True, text=True, timeout=600,
                                       env=dict(os.environ, RSIS_FORCE_DIST="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port())))
                    sj = json.loads(r.stdout.strip().splitlines()[-1])
                    sj["config"]["workload"] = sj["config"]["workload"].replace("configs[1]", "configs[1] with the gradient exchange forced on at world size 1")
                    secondary.append({k: sj.get(k) for k in ("metric", "value", "unit", "ms_per_step", "dtype", "config", "exchange")})
                except Exception as ex:  # noqa: BLE001
                    secondary.append({"dtype": "fp32", "config": "configs[1] + RSIS_FORCE_DIST=1", "error": repr(ex)})
                # ... and the CUT schedule (RSIS_EXCHANGE=cuts: graphs A | B1 | B2 | C with torch.distributed collectives between them -- the
                # fallback whenever the direct communicator is not available) next to it, so that both schedules are driver-timed every round
                try:
                    r = subprocess.run([sys.executable, os.path.abspath(__file__), "--batch", str(o.batch), "--imsize", str(o.imsize), "--T", str(o.T),
                                        "--steps", str(max(10, o.steps // 2)), "--warmup", str(o.warmup), "--skip-cpu", "--skip-secondary", "--skip-roofline"],
                                       capture_output=True, text=True, timeout=600,
                                       env=dict(os.environ, RSIS_FORCE_DIST="1", RSIS_EXCHANGE="cuts", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port())))
                    sj = json.loads(r.stdout.strip().splitlines()[-1])
                    sj["config"]["workload"] = sj["config"]["workload"].replace(
                        "configs[1]", "configs[1] with the gradient exchange forced on at world size 1, CUT schedule (RSIS_EXCHANGE=cuts)")
                    secondary.append({k: sj.get(k) for k in ("metric", "value", "unit", "ms_per_step", "dtype", "config", "exchange")})
                except Exception as ex:  # noqa: BLE001
                    secondary.append({"dtype": "fp32", "config": "configs[1] + RSIS_FORCE_DIST=1 + RSIS_EXCHANGE=cuts", "error": repr(ex)})
            # update_encoder off (the reference's training state until -finetune_after epochs: args.py:39-45), headline geometry
            try:
                r = subprocess.run([sys.executable, os.path.abspath(__file__), "--encoder-frozen", "--batch", str(o.batch), "--imsize", str(o.imsize), "--T", str(o.T),
                                    "--steps", str(o.steps), "--warmup", str(o.warmup), "--skip-cpu", "--skip-secondary", "--skip-roofline"],
                                   capture_output=True, text=True, timeout=600, env=dict(os.environ))
                sj = json.loads(r.stdout.strip().splitlines()[-1])
                secondary.append({k: sj.get(k) for k in ("metric", "value", "unit", "ms_per_step", "dtype", "config")})
            except Exception as ex:  # noqa: BLE001
                secondary.append({"metric": "training images/sec", "config": "--encoder-frozen", "error": repr(ex)})
            # inference: test() (reference src/test.py:16-50, the caller of src/eval.py:262) at the headline geometry in fp32 and at configs[2]'s in bf16
            for extra in (["--imsize", str(o.imsize), "--dtype", "fp32"], ["--imsize", "224", "--dtype", "bf16"]):
                try:
                    r = subprocess.run([sys.executable, os.path.abspath(__file__), "--inference", "--batch", str(o.batch), "--T", str(o.T), "--steps",
                                        str(o.steps), "--warmup", str(o.warmup), "--kernel-iters", str(o.kernel_iters)] + extra,
                                       capture_output=True, text=True, timeout=600, env=dict(os.environ))
                    secondary.append(json.loads(r.stdout.strip().splitlines()[-1]))
                except Exception as ex:  # noqa: BLE001
                    secondary.append({"metric": "inference images/sec", "config": " ".join(extra), "error": repr(ex)})
            note("secondary (224x224): bf16 %s images/s, fp32 %s images/s" % (secondary[0].get("value"), secondary[1].get("value")))
        value = world * o.batch * o.steps / dt
        out = {"metric": "training images/sec at %dx%d, T=%d, batch=%d per GPU" % (o.imsize, imw, o.T, o.batch),
               "value": round(value, 2), "unit": "images/s", "n_gpus": world, "steps": o.steps, "warmup": o.warmup,
               "ms_per_step": round(1000.0 * dt / o.steps, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
               "dtype": "f32" if o.dtype == "fp32" else "bf16", "data": "synthetic",
               "config": {"workload": "configs[1]: synthetic %dx%dx3, T=%d, batch=%d/GPU, ResNet-101 encoder + 5-scale ConvLSTM "
                                      "decoder, %s, fwd+match+3 losses+bwd+Adam (%s)" % (o.imsize, imw, o.T, o.batch, o.dtype,
                                      "update_encoder OFF: the reference's state before -finetune_after; trunk forward in train mode, no trunk backward, "
                                      "decoder + skip-branch update" if o.encoder_frozen else "update_encoder on"),
                          "global_batch": world * o.batch, "parallelism": "dp%d" % world, "final_loss": round(loss_val, 5),
                          "launch": "hipGraph replay of the captured iteration" if captured_launch
                                    else "eager (one Python launch per kernel)"},
               "roofline": roof, "roofline_kernels": roof_kernels, "cpu_baseline": cpu, "secondary": secondary}
        if o.dtype != "fp32":
            from rsis_amd import blk_trunk
            from rsis_amd import decoder_seq
            out["config"]["activations"] = ("trunk layers 1-4: channel-blocked bf16 (rsis_amd/blk_trunk.py); decoder: %s; stem, skip convs: fp32 NCHW"
                                            % ("channel-blocked bf16, fp32 cell state (rsis_amd/decoder_seq.py)" if decoder_seq.BLK_ENABLED[0] else "fp32 NCHW")
                                            if blk_trunk.ENABLED[0] else "fp32 NCHW everywhere (RSIS_BF16_STORAGE=0: bf16 operands only)")
        if seg is not None:
            out["config"]["exchange_ms_per_step"] = {k: (round(v, 3) if isinstance(v, float) else v) for k, v in seg.items()}
        out["exchange"] = exchange
        if pcie is not None:
            out["pcie_inclusive"] = pcie
        # never a line whose exchange ran over another number of ranks than it is labelled with
        rd = (exchange or {}).get("rccl_direct") or {}
        if (exchange or {}).get("mode") not in (None, "none"):
            assert exchange.get("world") == o.gpus == world, "exchange over %s ranks in a --gpus %d run" % (exchange.get("world"), o.gpus)
            if rd.get("mode") == "direct":
                assert rd.get("world") == o.gpus, "direct RCCL communicator of %s ranks in a --gpus %d run" % (rd.get("world"), o.gpus)
    # RCCL prints a banner (version, library path) through C stdio, which on a pipe is flushed only at exit, i.e. AFTER anything
    # python printed: every rank flushes its C streams before the final barrier, rank 0 prints after it, so that the JSON line
    # is the last line of the job's (merged) stdout
    def flush_c():
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:  # noqa: BLE001
            pass
        sys.stdout.flush()

    flush_c()
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        flush_c()
        sys.stdout.flush()
        os.write(REAL_STDOUT, (json.dumps(out) + "\n").encode())


if __name__ == "__main__":
    main()
