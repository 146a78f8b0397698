"""Golden-vector generator -- runs ONLY in the build container (needs /root/reference).

Imports the UNMODIFIED reference modules (src/modules/{clstm,model,vision}.py, src/test.py,
src/utils/{hungarian,objectives}.py) through the four shims of SURVEY.md Appendix B
(torchvision stub, python-2 integer `hidden_size`, munkres stub, masked_select bool shim),
fills reference and oracle modules with the same deterministic weights (oracle/filler.py),
and for every fixture case

  1. asserts   oracle == reference   (<= 1e-6 abs, fp32; the two are the same op graph), and
  2. writes the REFERENCE's outputs to tests/golden/<case>.npz (seeds + outputs only).

The fixtures are data (inputs are re-derived from seeds by the tests); the reference source
never leaves this container.  Usage:  python -m oracle.make_golden
"""
from __future__ import annotations

import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/src"
GOLD = os.path.join(ROOT, "tests", "golden")


class Py2Int(int):
    """python-2 `/` on ints (model.py:45-47,52-54,91-93 divide hidden_size with `/`)."""
    def __truediv__(self, o): return Py2Int(int(self) // int(o))
    def __floordiv__(self, o): return Py2Int(int(self) // int(o))
    def __mul__(self, o): return Py2Int(int(self) * int(o))
    __rmul__ = __mul__
    def __add__(self, o): return Py2Int(int(self) + int(o))
    __radd__ = __add__


def import_reference():
    # order matters: `utils` must resolve to the package src/utils/ (model.py:13), while
    # `hungarian`/`objectives` are imported top-level from inside it (objectives.py:2)
    sys.path[:0] = [os.path.join(ROOT, "oracle", "ref_shims"), REF, os.path.join(REF, "modules"),
                    os.path.join(REF, "utils"), ROOT]
    import model as ref_model          # noqa: E402  (reference src/modules/model.py)
    import clstm as ref_clstm          # noqa: E402
    import test as ref_test            # noqa: E402  (reference src/test.py)
    import hungarian as ref_hung       # noqa: E402
    import objectives as ref_obj       # noqa: E402
    _orig = torch.masked_select
    torch.masked_select = lambda t, m: _orig(t, m.bool())   # objectives.py:13,23,32 use .byte() masks
    return ref_model, ref_clstm, ref_test, ref_hung, ref_obj


def mk_args(hidden_size=128, num_classes=21, maxseqlen=10, py2=False, **kw):
    hs = Py2Int(hidden_size) if py2 else hidden_size
    a = argparse.Namespace(use_gpu=False, base_model="resnet101", hidden_size=hs, kernel_size=3,
                           num_classes=num_classes, dropout=0.0, dropout_stop=0.0, dropout_cls=0.0,
                           skip_mode="concat", maxseqlen=maxseqlen, gt_maxseqlen=20, iou_weight=1.0,
                           class_weight=0.1, stop_weight=0.5, stop_balance_weight=0.5,
                           use_class_loss=True, use_stop_loss=True, curriculum_learning=False)
    for k, v in kw.items():
        setattr(a, k, v)
    return a


def close(a, b, tol, what):
    a = a.detach() if torch.is_tensor(a) else torch.as_tensor(a)
    b = b.detach() if torch.is_tensor(b) else torch.as_tensor(b)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    err = (a.double() - b.double()).abs().max().item() if a.numel() else 0.0
    assert err <= tol, "%s: oracle vs reference max abs err %.3e > %.1e" % (what, err, tol)
    return err


def npf(t):
    return t.detach().cpu().numpy()


def sub_idx(n, cap=4096):
    """deterministic sub-sample of a flat gradient / parameter vector: every k-th element, at most ~cap of them"""
    return slice(0, n, max(1, n // cap))


def round2_cases(O, filler, ref_model, ref_test, report):
    """Fixtures added in round 2: the BASELINE geometries that had none (224x224: 7/14/28/56/112-pixel pyramid, partial MFMA
    tiles everywhere; 512x1024: the Cityscapes configuration) and a well-conditioned TRAINING step (forward, matching, the three
    losses, backward, one Adam step of both optimizers) on the reference's modules."""
    import utils.utils as ref_utils     # reference src/utils/utils.py: get_optimizer / get_skip_params (train.py:236-240)
    a_ref, a_ora = mk_args(py2=True), mk_args()
    renc, oenc = ref_model.FeatureExtractor(a_ref), O.FeatureExtractor(a_ora)
    rdec, odec = ref_model.RSIS(a_ref), O.RSIS(a_ora)

    # ---------------- F7: end-to-end test() at 224x224 (T=10) and 512x1024 (T=3) ----------------
    for name, shape, T, sub in (("e2e_224", (2, 3, 224, 224), 10, 4), ("e2e_512x1024", (2, 3, 512, 1024), 3, 8)):
        for m in (renc, oenc):
            filler.fill_module(m, seed=44)
            m.eval()
        for m in (rdec, odec):
            filler.fill_module(m, seed=45)
            m.eval()
        x = filler.tensor(44, name + ".x", shape)
        a_ref.maxseqlen = a_ora.maxseqlen = T
        logits = []
        hook = rdec.register_forward_hook(lambda mod, inp, outp: logits.append(outp[0].detach().clone()))
        with torch.no_grad():
            rm, rc, rs = ref_test.test(a_ref, renc, rdec, x)
        hook.remove()
        om, oc, os_ = O.test(a_ora, oenc, odec, x)
        _ol, _, osl = O.test(a_ora, oenc, odec, x, return_logits=True)
        report.append((name + ".masks", close(om, rm, 1e-5, name + ".masks")))
        report.append((name + ".classes", close(oc, rc, 1e-5, name + ".classes")))
        report.append((name + ".stops", close(os_, rs, 1e-5, name + ".stops")))
        ref_logits = torch.cat(logits, 1)
        with torch.no_grad():
            feats = oenc(x)
            hidden, ora_native = None, []
            for _ in range(T):
                m, _c, _s, hidden = odec(feats, hidden)
                ora_native.append(m)
        ora_native = torch.cat(ora_native, 1)
        report.append((name + ".logits_native", close(ora_native, ref_logits, 5e-5, name + ".logits")))
        # the reference's own fp32 noise floor on this fixture: the same op graph evaluated in float64
        e64, d64 = O.FeatureExtractor(a_ora).double(), O.RSIS(a_ora).double()
        e64.load_state_dict(oenc.state_dict())
        d64.load_state_dict(odec.state_dict())
        e64.eval()
        d64.eval()
        with torch.no_grad():
            f64 = e64(x.double())
            hidden, nat64, stop64 = None, [], []
            for _ in range(T):
                m, _c, s_, hidden = d64(f64, hidden)
                nat64.append(m)
                stop64.append(s_)
        nat64 = torch.cat(nat64, 1)
        floor = (ref_logits.double() - nat64).abs().max().item()
        report.append((name + ".fp32_floor_logits", floor))
        np.savez_compressed(os.path.join(GOLD, name + ".npz"), shape=np.array(shape), T=np.array(T), sub=np.array(sub),
                            mask_logits_sub=npf(ref_logits[:, :, ::sub, ::sub]), mask_probs_sub=npf(rm[:, :, ::sub, ::sub]),
                            classes=npf(rc), stops=npf(rs), stop_logits=npf(osl),
                            mask_logits_sub_f64=npf(nat64[:, :, ::sub, ::sub]), stop_logits_f64=npf(torch.stack(stop64, 1)),
                            logit_absmax=np.array(ref_logits.abs().max().item()), fp32_floor_logits=np.array(floor))

    # ---------------- F8: one full training iteration, 160x160, B=4, T=4 (train-mode BN over >= 100 samples per channel) ----------
    name = "trainstep_160"
    B, H, W, T = 4, 160, 160, 4
    x = filler.tensor(88, name + ".x", (B, 3, H, W))
    y_mask, y_class, sw_mask, sw_class = filler.synthetic_targets(88, B, H, W, gt_maxseqlen=20, n_inst=6)
    LR, LR_CNN, WD = 1e-3, 1e-6, 1e-6                      # src/args.py defaults
    res = {}
    cases = (("ref", renc, rdec, a_ref, torch.float32), ("ora", oenc, odec, a_ora, torch.float32),
             ("f64", O.FeatureExtractor(a_ora), O.RSIS(a_ora), a_ora, torch.float64))
    for tag, enc, dec, a, dt in cases:
        filler.fill_module(enc, seed=88)
        filler.fill_module(dec, seed=89)
        enc.to(dt)
        dec.to(dt)
        a.maxseqlen, a.gt_maxseqlen = T, 20
        a.update_encoder = True
        # train.py:236-240 with the reference's own helpers; the trunk group is de-duplicated (get_base_params yields tensors 1-4
        # times -- SURVEY Appendix C -- which modern torch.optim refuses)
        dec_params = list(dec.parameters()) + list(ref_utils.get_skip_params(enc))
        seen, base = set(), []
        for q in ref_utils.get_base_params(a, enc):
            if id(q) not in seen:
                seen.add(id(q))
                base.append(q)
        dec_opt = ref_utils.get_optimizer("adam", LR, dec_params, WD)
        enc_opt = ref_utils.get_optimizer("adam", LR_CNN, base, WD)
        enc.zero_grad()
        dec.zero_grad()
        # arg-max pixel of every hidden-state plane (the global max-pool side features of model.py:143 route their gradient there)
        # and the relative gap between the two largest values: a plane whose gap is at fp32-noise level may legitimately pick
        # another pixel in another fp32 implementation, which moves its gradient (a discontinuity of the reference function)
        picks, gaps = [], []

        def hook(_mod, _inp, outp, _p=picks, _g=gaps):
            for h, _c in outp[3]:
                flat = h.detach().flatten(2)
                top = flat.topk(2, dim=-1).values
                _p.append(flat.argmax(-1))
                _g.append((top[..., 0] - top[..., 1]) / top[..., 0].abs().clamp_min(1e-30))
        hk = dec.register_forward_hook(hook)
        r = O.run_iter_forward(a, enc, dec, x.to(dt), y_mask.to(dt), y_class, sw_mask, sw_class, mode="train")
        hk.remove()
        r["loss"].backward()
        g = {"loss": r["loss"], "loss_mask_iou": r["loss_mask_iou"], "loss_stop": r["loss_stop"], "loss_class": r["loss_class"],
             "scores": r["scores"], "out_masks_sub": r["out_masks"].view(B, T, H, W)[:, :, ::4, ::4], "out_classes": r["out_classes"],
             "out_stops": r["out_stops"], "y_class_perm": r["y_class_perm"]}
        for j, (pk, gp) in enumerate(zip(picks, gaps)):
            g["argmax.t%d.l%d" % (j // 5, j % 5)] = pk.to(torch.int32)
            g["gap.t%d.l%d" % (j // 5, j % 5)] = gp.float()
        named = [("dec." + k, q) for k, q in dec.named_parameters()] + [("enc." + k, q) for k, q in enc.named_parameters()
                                                                        if not k.startswith("base.fc")]
        # decoder + skip-conv tensors (the dec_opt group): ~2048 evenly spaced elements each; trunk tensors: 64 + the norm
        cap = lambda k: 2048 if (k.startswith("dec.") or not k.startswith("enc.base.")) else 64   # noqa: E731
        for k, q in named:
            flat = q.grad.detach().reshape(-1)
            g["grad." + k] = flat[sub_idx(flat.numel(), cap(k))].clone()
            g["gnorm." + k] = flat.norm()
        dec_opt.step()
        enc_opt.step()
        for k, q in named:
            if cap(k) == 2048:
                flat = q.detach().reshape(-1)
                g["post." + k] = flat[sub_idx(flat.numel(), 2048)].clone()
        res[tag] = g
    out = {"B": np.array(B), "H": np.array(H), "W": np.array(W), "T": np.array(T), "n_inst": np.array(6),
           "lr": np.array(LR), "lr_cnn": np.array(LR_CNN), "weight_decay": np.array(WD)}
    worst = 0.0
    for k in res["ref"]:
        ref = res["ref"][k]
        scale = max(1.0, float(torch.as_tensor(ref).double().abs().max()))
        tol = 1e-5 if not k.startswith(("grad.", "gnorm.", "post.")) else 2e-3 * scale   # (two fp32 evaluations of an ill-conditioned BN stack)
        e = close(res["ora"][k].double(), ref.double(), tol, name + "." + k)
        worst = max(worst, e / scale)
        out[k] = npf(ref) if torch.is_tensor(ref) else np.asarray(ref)
        if k.startswith(("loss", "grad.", "gnorm.", "argmax.", "gap.")) or k in ("out_masks_sub", "out_classes", "out_stops"):
            out["f64." + k] = npf(res["f64"][k])          # exact (float64) value of the same quantity: the fp32 noise floor
    report.append((name + ".worst_rel", worst))
    np.savez_compressed(os.path.join(GOLD, name + ".npz"), **out)
    report.append((name + ".bytes", os.path.getsize(os.path.join(GOLD, name + ".npz"))))


def round6_cases(O, filler, ref_model, ref_clstm, ref_test, report):
    """Fixtures added in round 6 (VERDICT r5 "Missing 2" / SURVEY 8(c) last paragraph: logits O(1-5), saturated gates, T = 20).
    `gates_gain` scales the ConvLSTM gate and conv_out weights of oracle/filler.py (2.0 in every earlier fixture):
      e2e_256_hot : test() at 256x256, B=2, T=10 with gates_gain 6 -> |mask logit| reaches 5.5, half of all gate pre-activations
                    are beyond |a| > 4 (sigmoid / tanh within 2 % of their asymptote); the share is stored per level.
      e2e_256_T20 : the same geometry over T = 20 timesteps (the configs[4] sequence length) at gates_gain 4.
      cell_hot    : ConvLSTMCell forward (t=0, t=1) + all gradients with inputs of scale 3 and gates_gain 6.
    Every fixture also stores the float64 evaluation of the same graph: the reference arithmetic's own fp32 noise floor."""
    a_ref, a_ora = mk_args(py2=True), mk_args()
    renc, oenc = ref_model.FeatureExtractor(a_ref), O.FeatureExtractor(a_ora)
    rdec, odec = ref_model.RSIS(a_ref), O.RSIS(a_ora)
    for name, shape, T, sub, gain in (("e2e_256_hot", (2, 3, 256, 256), 10, 4, 6.0), ("e2e_256_T20", (2, 3, 256, 256), 20, 8, 4.0)):
        for m in (renc, oenc):
            filler.fill_module(m, seed=44)
            m.eval()
        for m in (rdec, odec):
            filler.fill_module(m, seed=45, gates_gain=gain)
            m.eval()
        x = filler.tensor(44, name + ".x", shape)
        a_ref.maxseqlen = a_ora.maxseqlen = T
        logits, pre = [], [[] for _ in range(5)]
        hooks = [rdec.register_forward_hook(lambda mod, inp, outp: logits.append(outp[0].detach().clone()))]
        for i, cell in enumerate(rdec.clstm_list):
            hooks.append(cell.Gates.register_forward_hook(lambda mod, inp, outp, _i=i: pre[_i].append(outp.detach().abs())))
        with torch.no_grad():
            rm, rc, rs = ref_test.test(a_ref, renc, rdec, x)
        for h in hooks:
            h.remove()
        om, oc, os_ = O.test(a_ora, oenc, odec, x)
        _ol, _, osl = O.test(a_ora, oenc, odec, x, return_logits=True)
        report.append((name + ".masks", close(om, rm, 1e-5, name + ".masks")))
        report.append((name + ".classes", close(oc, rc, 1e-5, name + ".classes")))
        report.append((name + ".stops", close(os_, rs, 1e-5, name + ".stops")))
        ref_logits = torch.cat(logits, 1)
        with torch.no_grad():
            feats = oenc(x)
            hidden, ora_native = None, []
            for _ in range(T):
                m, _c, _s, hidden = odec(feats, hidden)
                ora_native.append(m)
        ora_native = torch.cat(ora_native, 1)
        report.append((name + ".logits_native", close(ora_native, ref_logits, 5e-5, name + ".logits")))
        e64, d64 = O.FeatureExtractor(a_ora).double(), O.RSIS(a_ora).double()
        e64.load_state_dict(oenc.state_dict())
        d64.load_state_dict(odec.state_dict())
        e64.eval()
        d64.eval()
        with torch.no_grad():
            f64 = e64(x.double())
            hidden, nat64, stop64 = None, [], []
            for _ in range(T):
                m, _c, s_, hidden = d64(f64, hidden)
                nat64.append(m)
                stop64.append(s_)
        nat64 = torch.cat(nat64, 1)
        stop64 = torch.stack(stop64, 1)
        floor = (ref_logits.double() - nat64).abs().max().item()
        floor_stop = (osl.double().reshape(-1) - stop64.reshape(-1)).abs().max().item()
        sat4 = np.array([float((torch.cat([q.flatten() for q in pre[i]]) > 4).float().mean()) for i in range(5)])
        allp = torch.cat([q.flatten() for lv in pre for q in lv])
        report.append((name + ".|logit|max", ref_logits.abs().max().item()))
        report.append((name + ".logit_std", ref_logits.std().item()))
        report.append((name + ".share_|logit|>3", float((ref_logits.abs() > 3).float().mean())))
        report.append((name + ".share_gate_preact_|a|>4", float((allp > 4).float().mean())))
        report.append((name + ".share_gate_preact_|a|>8", float((allp > 8).float().mean())))
        report.append((name + ".fp32_floor_logits", floor))
        report.append((name + ".fp32_floor_stop_logit", floor_stop))
        np.savez_compressed(os.path.join(GOLD, name + ".npz"), shape=np.array(shape), T=np.array(T), sub=np.array(sub),
                            gates_gain=np.array(gain),
                            mask_logits_sub=npf(ref_logits[:, :, ::sub, ::sub]), mask_probs_sub=npf(rm[:, :, ::sub, ::sub]),
                            classes=npf(rc), stops=npf(rs), stop_logits=npf(osl),
                            mask_logits_sub_f64=npf(nat64[:, :, ::sub, ::sub]), stop_logits_f64=npf(stop64),
                            logit_absmax=np.array(ref_logits.abs().max().item()), logit_std=np.array(ref_logits.std().item()),
                            gate_sat4_per_level=sat4, gate_sat4=np.array(float((allp > 4).float().mean())),
                            fp32_floor_logits=np.array(floor), fp32_floor_stop_logit=np.array(floor_stop))

    # ---------------- cell_hot: saturated ConvLSTM cell, forward (None state, then with state) + every gradient ----------------
    name, (B, Cin, hid, H, W), gain, xs = "cell_hot", (2, 40, 16, 12, 20), 6.0, 3.0
    a = mk_args()
    cells = {"ref": ref_clstm.ConvLSTMCell(a, Cin, hid, 3, 1), "ora": O.ConvLSTMCell(a, Cin, hid, 3, 1),
             "f64": O.ConvLSTMCell(a, Cin, hid, 3, 1)}
    for c in cells.values():
        filler.fill_module(c, seed=12, gates_gain=gain)
    cells["f64"].double()
    out = {"shape": np.array([B, Cin, hid, H, W]), "gates_gain": np.array(gain), "x_scale": np.array(xs)}
    res = {}
    for tag, cell in cells.items():
        dt = torch.float64 if tag == "f64" else torch.float32
        x0 = filler.tensor(12, name + ".x0", (B, Cin, H, W), xs).to(dt).requires_grad_()
        x1 = filler.tensor(12, name + ".x1", (B, Cin, H, W), xs).to(dt).requires_grad_()
        gh = filler.tensor(12, name + ".gh", (B, hid, H, W)).to(dt)
        gc = filler.tensor(12, name + ".gc", (B, hid, H, W)).to(dt)
        pre = []
        hk = cell.Gates.register_forward_hook(lambda mod, inp, outp: pre.append(outp.detach().abs()))
        cell.zero_grad()
        h0, c0 = cell(x0, None)
        h1, c1 = cell(x1, (h0, c0))
        hk.remove()
        ((h1 * gh).sum() + (c1 * gc).sum()).backward()
        res[tag] = dict(h0=h0, c0=c0, h1=h1, c1=c1, dx0=x0.grad.clone(), dx1=x1.grad.clone(),
                        dW=cell.Gates.weight.grad.clone(), db=cell.Gates.bias.grad.clone())
        if tag == "ref":
            allp = torch.cat([q.flatten() for q in pre])
            out["gate_sat4"] = np.array(float((allp > 4).float().mean()))
            report.append((name + ".share_gate_preact_|a|>4", float(out["gate_sat4"])))
            report.append((name + ".gate_preact_max", float(allp.max())))
    for k in res["ref"]:
        scale = max(1.0, float(res["ref"][k].abs().max()))
        report.append((name + "." + k, close(res["ora"][k], res["ref"][k], 1e-5 * scale, name + "." + k)))
        out[k] = npf(res["ref"][k])
        out["f64." + k] = npf(res["f64"][k])
        report.append((name + ".fp32_floor." + k, float((res["ref"][k].double() - res["f64"][k]).abs().max())))
    np.savez_compressed(os.path.join(GOLD, name + ".npz"), **out)


def main():
    from oracle import rsis_oracle as O
    from oracle import filler
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", default="all", choices=["all", "r1", "r2", "r6"],
                    help="r1: the round-1 fixtures (REPORT.txt), r2: the fixtures added in round 2 (REPORT_r2.txt), "
                         "r6: the hot / T=20 fixtures of round 6 (REPORT_r6.txt)")
    opt = ap.parse_args()
    ref_model, ref_clstm, ref_test, ref_hung, ref_obj = import_reference()
    os.makedirs(GOLD, exist_ok=True)
    torch.manual_seed(0)
    if opt.cases in ("all", "r6"):
        report = []
        round6_cases(O, filler, ref_model, ref_clstm, ref_test, report)
        w = max(len(k) for k, _ in report)
        with open(os.path.join(GOLD, "REPORT_r6.txt"), "w") as f:
            f.write("oracle vs imported reference (max abs err), dynamic range and fp32 noise floors of the hot / T=20 fixtures,\n"
                    "generated by oracle/make_golden.py --cases r6\n")
            for k, v in report:
                f.write("%-*s %.3e\n" % (w, k, v))
        print("round-6 fixtures written (%d checks)" % len(report))
        if opt.cases == "r6":
            return
    if opt.cases in ("all", "r2"):
        report = []
        round2_cases(O, filler, ref_model, ref_test, report)
        w = max(len(k) for k, _ in report)
        with open(os.path.join(GOLD, "REPORT_r2.txt"), "w") as f:
            f.write("oracle vs imported reference (max abs err) and fp32 noise floors, generated by oracle/make_golden.py --cases r2\n")
            for k, v in report:
                f.write("%-*s %.3e\n" % (w, k, v))
        print("round-2 fixtures written (%d checks)" % len(report))
        if opt.cases == "r2":
            return
    report = []

    # ---------------- F1: ConvLSTMCell fwd (t=0 None state, t=1 with state) + grads ----------------
    for name, (B, Cin, hid, H, W) in {"cell_small": (2, 8, 4, 5, 7), "cell_l4like": (2, 40 - 8, 8, 16, 16),
                                      "cell_wide": (3, 24, 16, 9, 12)}.items():
        a = mk_args()
        rc = ref_clstm.ConvLSTMCell(a, Cin, hid, 3, 1)
        oc = O.ConvLSTMCell(a, Cin, hid, 3, 1)
        filler.fill_module(rc, seed=11)
        filler.fill_module(oc, seed=11)
        out = {"shape": np.array([B, Cin, hid, H, W])}
        x0 = filler.tensor(11, name + ".x0", (B, Cin, H, W)).requires_grad_()
        x1 = filler.tensor(11, name + ".x1", (B, Cin, H, W)).requires_grad_()
        gh = filler.tensor(11, name + ".gh", (B, hid, H, W))
        gc = filler.tensor(11, name + ".gc", (B, hid, H, W))
        res = {}
        for tag, cell in (("ref", rc), ("ora", oc)):
            cell.zero_grad()
            for t in (x0, x1):
                t.grad = None
            h0, c0 = cell(x0, None)
            h1, c1 = cell(x1, (h0, c0))
            ((h1 * gh).sum() + (c1 * gc).sum()).backward()
            res[tag] = dict(h0=h0, c0=c0, h1=h1, c1=c1, dx0=x0.grad.clone(), dx1=x1.grad.clone(),
                            dW=cell.Gates.weight.grad.clone(), db=cell.Gates.bias.grad.clone())
        for k in res["ref"]:
            report.append((name + "." + k, close(res["ora"][k], res["ref"][k], 1e-5, name + "." + k)))
            out[k] = npf(res["ref"][k])
        np.savez_compressed(os.path.join(GOLD, name + ".npz"), **out)

    # ---------------- F2: RSIS decoder, 3 timesteps, hidden 32, regular + odd pyramid ----------------
    for name, sizes in {"dec_pow2": [(2, 2), (4, 4), (8, 8), (16, 16), (32, 32)],
                        "dec_odd": [(3, 4), (5, 7), (10, 13), (19, 25), (37, 50)]}.items():
        hs, B, T = 32, 2, 3
        rd = ref_model.RSIS(mk_args(hidden_size=hs, py2=True))
        od = O.RSIS(mk_args(hidden_size=hs))
        filler.fill_module(rd, seed=22)
        filler.fill_module(od, seed=22)
        chans = [hs, hs, hs // 2, hs // 4, hs // 8]
        feats = [filler.tensor(22, "%s.f%d" % (name, i), (B, chans[i]) + sizes[i]).requires_grad_() for i in range(5)]
        out = {"sizes": np.array(sizes), "hidden_size": np.array(hs), "B": np.array(B), "T": np.array(T)}
        res = {}
        for tag, dec in (("ref", rd), ("ora", od)):
            dec.zero_grad()
            for f in feats:
                f.grad = None
            hidden, loss, r = None, 0.0, {}
            for t in range(T):
                m, c, s, hidden = dec(feats, hidden)
                r["mask%d" % t], r["class%d" % t], r["stop%d" % t] = m, c, s
                loss = loss + (m * filler.tensor(22, "%s.gm%d" % (name, t), m.shape)).sum() \
                    + (c * filler.tensor(22, "%s.gc%d" % (name, t), c.shape)).sum() \
                    + (s * filler.tensor(22, "%s.gs%d" % (name, t), s.shape)).sum()
            for i, (h, c) in enumerate(hidden):
                r["h%d" % i], r["c%d" % i] = h, c
            loss.backward()
            for i, f in enumerate(feats):
                r["dfeat%d" % i] = f.grad.clone()
            for k, p in dec.named_parameters():
                r["grad." + k] = p.grad.clone()
            res[tag] = r
        for k in res["ref"]:
            tol = 1e-5 if not k.startswith(("grad.", "dfeat")) else 2e-4
            report.append((name + "." + k, close(res["ora"][k], res["ref"][k], tol, name + "." + k)))
            out[k] = npf(res["ref"][k])
        np.savez_compressed(os.path.join(GOLD, name + ".npz"), **out)

    # ---------------- F3: FeatureExtractor eval + train (BN batch stats, running-stat update) ----------------
    a_ref, a_ora = mk_args(py2=True), mk_args()
    renc = ref_model.FeatureExtractor(a_ref)
    oenc = O.FeatureExtractor(a_ora)
    filler.fill_module(renc, seed=33)
    filler.fill_module(oenc, seed=33)
    for name, shape, train in (("enc_eval_64", (2, 3, 64, 64), False), ("enc_eval_96x80", (1, 3, 96, 80), False),
                               ("enc_train_64", (2, 3, 64, 64), True)):
        x = filler.tensor(33, name + ".x", shape)
        out = {"shape": np.array(shape), "train": np.array(train)}
        res = {}
        for tag, enc in (("ref", renc), ("ora", oenc)):
            filler.fill_module(enc, seed=33)
            enc.train(train)
            with torch.no_grad():
                fs = enc(x)
            r = {"skip%d" % (5 - i): f for i, f in enumerate(fs)}
            if train:
                sd = enc.state_dict()
                for k in ("bn5.running_mean", "bn5.running_var", "base.bn1.running_mean", "base.bn1.running_var",
                          "base.layer3.22.bn3.running_mean", "base.layer3.22.bn3.running_var"):
                    r["sd." + k] = sd[k].clone()
            res[tag] = r
        for k in res["ref"]:
            report.append((name + "." + k, close(res["ora"][k], res["ref"][k], 2e-5, name + "." + k)))
            out[k] = npf(res["ref"][k])
        np.savez_compressed(os.path.join(GOLD, name + ".npz"), **out)

    # ---------------- F4: end-to-end test() 256x256, B=2, T=10, hidden 128 (the north-star check) ----------
    rdec = ref_model.RSIS(a_ref)
    odec = O.RSIS(a_ora)
    for name, shape, T, sub in (("e2e_256", (2, 3, 256, 256), 10, 4), ("e2e_200x264", (2, 3, 200, 264), 3, 4)):
        for m in (renc, oenc):
            filler.fill_module(m, seed=44)
            m.eval()
        for m in (rdec, odec):
            filler.fill_module(m, seed=45)
            m.eval()
        x = filler.tensor(44, name + ".x", shape)
        a_ref.maxseqlen = a_ora.maxseqlen = T
        # reference test() returns sigmoid(masks); logits captured by hooking the decoder output
        logits = []
        hook = rdec.register_forward_hook(lambda mod, inp, outp: logits.append(outp[0].detach().clone()))
        with torch.no_grad():
            rm, rc, rs = ref_test.test(a_ref, renc, rdec, x)
        hook.remove()
        om, oc, os_ = O.test(a_ora, oenc, odec, x)
        ol, _, osl = O.test(a_ora, oenc, odec, x, return_logits=True)
        report.append((name + ".masks", close(om, rm, 1e-5, name + ".masks")))
        report.append((name + ".classes", close(oc, rc, 1e-5, name + ".classes")))
        report.append((name + ".stops", close(os_, rs, 1e-5, name + ".stops")))
        # logits at the decoder's native output size (2*h1), before the resize-to-input of test.py:39
        ref_logits = torch.cat(logits, 1)
        with torch.no_grad():
            feats = oenc(x)
            hidden, ora_native = None, []
            for _ in range(T):
                m, _c, _s, hidden = odec(feats, hidden)
                ora_native.append(m)
        ora_native = torch.cat(ora_native, 1)
        report.append((name + ".logits_native", close(ora_native, ref_logits, 5e-5, name + ".logits")))
        np.savez_compressed(os.path.join(GOLD, name + ".npz"), shape=np.array(shape), T=np.array(T), sub=np.array(sub),
                            mask_logits_sub=npf(ref_logits[:, :, ::sub, ::sub]), mask_probs_sub=npf(rm[:, :, ::sub, ::sub]),
                            classes=npf(rc), stops=npf(rs), stop_logits=npf(osl),
                            logit_absmax=np.array(ref_logits.abs().max().item()),
                            logit_std=np.array(ref_logits.std().item()))
        report.append((name + ".|logit|max", ref_logits.abs().max().item()))

    # ---------------- F5: loss / matching functions ----------------
    B, G, T, N, C = 3, 20, 10, 96, 21
    P = filler.tensor(55, "f5.P", (B * G, N), 2.0)
    Y = (filler.tensor(55, "f5.Y", (B * G, N)) > 0.3).float()
    probs = torch.softmax(filler.tensor(55, "f5.logits", (B * T, C)), 1)
    tgt = torch.from_numpy(np.random.default_rng(55).integers(0, C, (B * T, 1)))
    stop_logit = filler.tensor(55, "f5.stop", (B, T), 3.0)
    stop_tgt = (filler.tensor(55, "f5.stopt", (B, T)) > 0).float()
    sw = (filler.tensor(55, "f5.sw", (B * T, 1)) > -0.5).float()
    out = {}
    pairs = {
        "softIoU": (O.softIoU(Y, P), ref_hung.softIoU(Y, P)),
        "MaskedNLL": (O.MaskedNLL(tgt, probs), ref_hung.MaskedNLL(tgt, probs)),
        "BCE_bw05": (O.StableBalancedMaskedBCE(stop_tgt, stop_logit, 0.5), ref_hung.StableBalancedMaskedBCE(stop_tgt, stop_logit, 0.5)),
        "BCE_auto": (O.StableBalancedMaskedBCE(stop_tgt, stop_logit), ref_hung.StableBalancedMaskedBCE(stop_tgt, stop_logit)),
        "softIoULoss": (O.softIoULoss(Y[:B * T], P[:B * T], sw), ref_obj.softIoULoss()(Y[:B * T], P[:B * T], sw)),
        "MaskedNLLLoss": (O.MaskedNLLLoss(tgt, probs, sw), ref_obj.MaskedNLLLoss()(tgt, probs, sw)),
        "MaskedBCELoss": (O.MaskedBCELoss(stop_tgt, stop_logit, sw, 0.5), ref_obj.MaskedBCELoss(0.5)(stop_tgt, stop_logit, sw)),
    }
    for k, (o, r) in pairs.items():
        report.append(("f5." + k, close(o, r, 1e-6, "f5." + k)))
        out[k] = npf(r)
    # match(): rectangular 20x10 cost matrices with distinct entries (unique optimum)
    scores = torch.from_numpy(np.random.default_rng(56).uniform(0, 1, (B, G, T))).float()
    ym = (filler.tensor(56, "f5.ym", (B, G, N)) > 0).float()
    yc = torch.from_numpy(np.random.default_rng(57).integers(0, C, (B, G)))
    pm = filler.tensor(56, "f5.pm", (B, T, N))
    pc = filler.tensor(56, "f5.pc", (B, T, C))
    o_m, o_c, o_p = O.match([ym, pm], [yc, pc], scores)
    r_m, r_c, r_p = ref_hung.match([ym.clone(), pm], [yc.clone(), pc], scores)
    assert (o_p == r_p).all() and (o_m == r_m).all() and (o_c == r_c).all()
    out.update(match_perm=r_p, match_class=r_c, match_mask_sum=r_m.sum(-1))
    np.savez_compressed(os.path.join(GOLD, "losses.npz"), **out)

    # ---------------- F6: restated runIter math on reference modules (2,3,64,64), T=3 ----------------
    name = "runiter_64"
    B, H, W, T = 2, 64, 64, 3
    x = filler.tensor(66, name + ".x", (B, 3, H, W))
    y_mask, y_class, sw_mask, sw_class = filler.synthetic_targets(66, B, H, W, gt_maxseqlen=20, n_inst=5)
    res = {}
    for tag, enc, dec, a in (("ref", renc, rdec, a_ref), ("ora", oenc, odec, a_ora)):
        filler.fill_module(enc, seed=66)
        filler.fill_module(dec, seed=67)
        a.maxseqlen, a.gt_maxseqlen = T, 20
        enc.zero_grad()
        dec.zero_grad()
        r = O.run_iter_forward(a, enc, dec, x, y_mask, y_class, sw_mask, sw_class, mode="train")
        r["loss"].backward()
        g = {"loss": r["loss"], "loss_mask_iou": r["loss_mask_iou"], "loss_stop": r["loss_stop"],
             "loss_class": r["loss_class"], "scores": r["scores"], "out_masks_sub": r["out_masks"].view(B, T, H, W)[:, :, ::4, ::4],
             "out_classes": r["out_classes"], "out_stops": r["out_stops"], "y_class_perm": r["y_class_perm"]}
        for k in ("clstm_list.0.Gates.weight", "clstm_list.4.Gates.weight", "conv_out.weight", "fc_class.weight", "fc_stop.bias"):
            g["gnorm.dec." + k] = dict(dec.named_parameters())[k].grad.norm()
        for k in ("sk5.weight", "bn1.weight", "base.conv1.weight", "base.layer1.0.conv1.weight",
                  "base.layer3.10.conv2.weight", "base.layer4.2.bn3.bias"):
            g["gnorm.enc." + k] = dict(enc.named_parameters())[k].grad.norm()
        res[tag] = g
    out = {}
    for k in res["ref"]:
        tol = 1e-5 if not k.startswith("gnorm") else 1e-3 * max(1.0, float(res["ref"][k]))
        report.append((name + "." + k, close(res["ora"][k], res["ref"][k], tol, name + "." + k)))
        out[k] = npf(res["ref"][k])
    np.savez_compressed(os.path.join(GOLD, name + ".npz"), **out)

    w = max(len(k) for k, _ in report)
    with open(os.path.join(GOLD, "REPORT.txt"), "w") as f:
        f.write("oracle vs imported reference (max abs err), generated by oracle/make_golden.py\n")
        for k, v in report:
            f.write("%-*s %.3e\n" % (w, k, v))
    print("golden fixtures written to", GOLD, "(%d checks)" % len(report))
    tot = sum(os.path.getsize(os.path.join(GOLD, f)) for f in os.listdir(GOLD))
    print("total fixture bytes:", tot)


if __name__ == "__main__":
    main()
