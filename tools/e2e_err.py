import sys, torch, numpy as np
import os; R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path[:0] = [R, os.path.join(R, 'tests')]
from oracle import filler, rsis_oracle as O
from helpers import mk_args, gold
from rsis_amd.modules import FeatureExtractor, RSIS
from rsis_amd.test import test as hip_test
g = gold("e2e_256"); a = mk_args(maxseqlen=10)
oenc = filler.fill_module(O.FeatureExtractor(a), seed=44); odec = filler.fill_module(O.RSIS(a), seed=45)
enc, dec = FeatureExtractor(a).cuda(), RSIS(a).cuda()
enc.load_state_dict(oenc.state_dict()); dec.load_state_dict(odec.state_dict())
x = filler.tensor(44, "e2e_256.x", (2, 3, 256, 256)).cuda()
logits, classes, stops = hip_test(a, enc, dec, x, return_logits=True)
print("mask %.3e class %.3e stop %.3e" % (float((logits[:, :, ::4, ::4].cpu() - torch.from_numpy(g["mask_logits_sub"])).abs().max()),
      float((classes.cpu() - torch.from_numpy(g["classes"])).abs().max()), float((stops.cpu() - torch.from_numpy(g["stop_logits"])).abs().max())))
