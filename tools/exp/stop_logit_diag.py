#!/usr/bin/env python
"""Where does the stop-logit error of the e2e_256 fixture come from?  Per timestep and pyramid level: the side features (global max of
the hidden state) of the HIP path and of the fp32 oracle against the float64 oracle, and each level's contribution
fc_stop.weight[level slice] . (side - side64) to the stop-logit error."""
import os
import sys

import torch

R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [R, os.path.join(R, "tests")]
from helpers import mk_args  # noqa: E402
from oracle import filler, rsis_oracle as O  # noqa: E402
from rsis_amd.modules import FeatureExtractor, RSIS  # noqa: E402

a = mk_args(maxseqlen=10)
oenc = filler.fill_module(O.FeatureExtractor(a), seed=44).eval()
odec = filler.fill_module(O.RSIS(a), seed=45).eval()
enc, dec = FeatureExtractor(a).cuda().eval(), RSIS(a).cuda().eval()
enc.load_state_dict(oenc.state_dict())
dec.load_state_dict(odec.state_dict())
x = filler.tensor(44, "e2e_256.x", (2, 3, 256, 256))
torch.set_num_threads(32)


def run(e, d, xin, cast):
    with torch.no_grad():
        feats, hid, out = e(xin), None, []
        fe = [f.detach().double().cpu() for f in feats]
        for _ in range(10):
            _m, _c, s, hid = d(feats, hid)
            out.append(([h.amax((2, 3)).double().cpu() for h, _c2 in hid], s.reshape(-1).double().cpu()))
    return fe, out


f32, o32 = run(oenc, odec, x, float)
fh, oh = run(enc, dec, x.cuda(), float)
f64, o64 = run(oenc.double(), odec.double(), x.double(), float)
Ws = odec.fc_stop.weight.double().reshape(-1)
hs = [128, 64, 32, 16, 8]
offs = [0, 128, 192, 224, 240, 248]
print("skip features max |err| vs f64 (hip | ref32):", [("%.1e" % float((a_ - c_).abs().max()), "%.1e" % float((b_ - c_).abs().max())) for a_, b_, c_ in zip(fh, f32, f64)])
for t in range(10):
    row = []
    for l in range(5):
        dh, dr = oh[t][0][l] - o64[t][0][l], o32[t][0][l] - o64[t][0][l]
        w = Ws[offs[l]:offs[l + 1]]
        row.append("L%d side %.1e/%.1e contrib %+.1e/%+.1e" % (l, float(dh.abs().max()), float(dr.abs().max()), float((dh @ w).abs().max()), float((dr @ w).abs().max())))
    print("t=%d stop err hip %.2e ref32 %.2e | %s" % (t, float((oh[t][1] - o64[t][1]).abs().max()), float((o32[t][1] - o64[t][1]).abs().max()), " ; ".join(row)))
