"""Folded vs unfolded eval-mode BatchNorm in the blk trunk on the e2e_256 golden case: distance of the five skip features and of the
stop logits from the fp32 oracle."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import torch
from helpers import mk_args
from oracle import filler
from oracle import rsis_oracle as O
from rsis_amd import blk_trunk
from rsis_amd.modules import FeatureExtractor, RSIS
from rsis_amd.test import test as hip_test

T = 10
a32, a = mk_args(maxseqlen=T), mk_args(maxseqlen=T, dtype="bf16")
oenc = filler.fill_module(O.FeatureExtractor(a32), seed=44).eval()
odec = filler.fill_module(O.RSIS(a32), seed=45).eval()
enc, dec = FeatureExtractor(a).cuda(), RSIS(a).cuda()
enc.load_state_dict(oenc.state_dict()); dec.load_state_dict(odec.state_dict())
x = filler.tensor(44, "e2e_256.x", (2, 3, 256, 256))
with torch.no_grad():
    ref = [f.double() for f in oenc(x)]
    hidden, ostops = None, []
    for _t in range(T):
        m, c, s, hidden = odec(list(ref_f.float() for ref_f in ref), hidden)
        ostops.append(s)
    ostops = torch.cat(ostops, 1).double()


def rel(p, q):
    return float((p.double().cpu() - q).norm() / q.norm())


enc.eval(); dec.eval()
for fold in (False, True):
    blk_trunk.EVAL_FOLD[0] = fold
    with torch.no_grad():
        feats = enc(x.cuda())
        _m, _c, st = hip_test(a, enc, dec, x.cuda(), return_logits=True)
    print("fold %s: features rel L2 vs fp32 oracle: %s | stop logits: max abs err %.4f (|ref| max %.3f)" % (
        fold, " ".join("%.4f" % rel(f, r) for f, r in zip(feats, ref)), float((st.double().cpu().view(2, -1) - ostops).abs().max()), float(ostops.abs().max())))
