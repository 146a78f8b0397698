import os, sys, torch
sys.path.insert(0, '' + __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))) + ''); sys.path.insert(0, os.path.join(sys.path[0], 'tests'))
from oracle import filler
from oracle import rsis_oracle as O
from rsis_amd.modules import FeatureExtractor, RSIS
from helpers import mk_args
a = mk_args(hidden_size=32, maxseqlen=2)
oenc = filler.fill_module(O.FeatureExtractor(a), seed=1).eval()
odec = filler.fill_module(O.RSIS(a), seed=2).train()
enc, dec = FeatureExtractor(a).cuda().eval(), RSIS(a).cuda().train()
enc.load_state_dict(oenc.state_dict()); dec.load_state_dict(odec.state_dict())
x = filler.tensor(3, "smoke.x", (2, 3, 64, 64))
res = []
import copy
oenc64, odec64 = copy.deepcopy(oenc).double(), copy.deepcopy(odec).double()
for e, d, xin, dt in ((oenc, odec, x, torch.float32), (oenc64, odec64, x.double(), torch.float64), (enc, dec, x.cuda(), torch.float32)):
    with torch.no_grad():
        feats = [f.detach().requires_grad_() for f in e(xin)]
    hidden, loss, outs = None, 0.0, []
    for _t in range(2):
        m, c, s, hidden = d(feats, hidden)
        outs += [m, c, s]
        loss = loss + m.mean() + c.square().sum() + s.mean()
    loss.backward()
    res.append((outs, [f.grad for f in feats], [p.grad for p in d.parameters()], [k for k, _ in d.named_parameters()]))
names = ["m0","c0","s0","m1","c1","s1"]
for i,n in enumerate(names):
    t = res[1][0][i].detach()
    print("out %s: |ref|max %.3g  oracle32 err %.2e  hip err %.2e" % (n, float(t.abs().max()), float((res[0][0][i].detach().double()-t).abs().max()), float((res[2][0][i].detach().cpu().double()-t).abs().max())))
for i in range(5):
    t = res[1][1][i]
    print("dfeat%d: |ref|max %.3g  oracle32 relerr %.2e  hip relerr %.2e" % (i, float(t.abs().max()), float((res[0][1][i].double()-t).abs().max()/t.abs().max()), float((res[2][1][i].cpu().double()-t).abs().max()/t.abs().max())))
for i,k in enumerate(res[1][3]):
    t = res[1][2][i]
    if t is None: continue
    print("d%s: |ref|max %.3g  oracle32 relerr %.2e  hip relerr %.2e" % (k, float(t.abs().max()), float((res[0][2][i].double()-t).abs().max()/t.abs().max()), float((res[2][2][i].cpu().double()-t).abs().max()/t.abs().max())))
