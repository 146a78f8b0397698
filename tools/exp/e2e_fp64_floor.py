#!/usr/bin/env python
"""How far is the REFERENCE's own fp32 result from the exact (fp64) one on the e2e_256 fixture?  Runs the oracle in float64 on
the CPU (about a minute) and compares with the fp32 golden outputs of tests/golden/.  Measured here: stop logits 4.1e-05,
class probabilities 2.9e-06 -- the scale of the disagreement to expect between two equally accurate fp32 implementations."""
import os
import sys

import torch

R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [R, os.path.join(R, "tests")]
from helpers import gold, mk_args  # noqa: E402
from oracle import filler, rsis_oracle as O  # noqa: E402

g = gold("e2e_256")
a = mk_args(maxseqlen=10)
oenc = filler.fill_module(O.FeatureExtractor(a), seed=44).double().eval()
odec = filler.fill_module(O.RSIS(a), seed=45).double().eval()
x = filler.tensor(44, "e2e_256.x", (2, 3, 256, 256)).double()
with torch.no_grad():
    feats, hid, stops, classes = oenc(x), None, [], []
    for _ in range(10):
        _mask, cls, stop, hid = odec(feats, hid)
        stops.append(stop.reshape(2, 1))
        classes.append(cls.reshape(2, 1, -1))
stops, classes = torch.cat(stops, 1), torch.cat(classes, 1)
gs = torch.from_numpy(g["stop_logits"]).double().reshape(2, 10)
gc = torch.from_numpy(g["classes"]).double().reshape(classes.shape)
print("reference fp32 vs fp64: stop logits max |err| %.3e ; class probs %.3e" % (float((gs - stops).abs().max()), float((gc - classes).abs().max())))
