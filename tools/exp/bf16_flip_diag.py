"""Does the bf16 decoder's gradient error on tests/golden/dec_odd come from arg-max flips of the global max-pool side features?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import numpy as np
import torch
from helpers import gold, mk_args
from oracle import filler
from oracle import rsis_oracle as O
from rsis_amd.modules import RSIS

name = sys.argv[1] if len(sys.argv) > 1 else "dec_odd"
g = gold(name)
hs, B, T = int(g["hidden_size"]), int(g["B"]), int(g["T"])
sizes = [tuple(int(v) for v in s) for s in g["sizes"]]
odec = filler.fill_module(O.RSIS(mk_args(hidden_size=hs)), seed=22)
chans = [hs, hs, hs // 2, hs // 4, hs // 8]
res = {}
for dt in ("fp32", "bf16"):
    dec = RSIS(mk_args(hidden_size=hs, dtype=dt)).cuda()
    dec.load_state_dict(odec.state_dict())
    feats = [filler.tensor(22, "%s.f%d" % (name, i), (B, chans[i]) + sizes[i]).cuda().requires_grad_() for i in range(5)]
    hidden, loss, args_ = None, 0.0, []
    for t in range(T):
        m, c, s, hidden = dec(feats, hidden)
        args_.append([h.detach().flatten(2).argmax(-1).cpu() for h, _ in hidden])
        loss = loss + (m * filler.tensor(22, "%s.gm%d" % (name, t), m.shape).cuda()).sum() \
            + (c * filler.tensor(22, "%s.gc%d" % (name, t), c.shape).cuda()).sum() \
            + (s * filler.tensor(22, "%s.gs%d" % (name, t), s.shape).cuda()).sum()
    loss.backward()
    res[dt] = (args_, [f.grad.cpu() for f in feats])
for t in range(T):
    for i in range(5):
        a, b = res["fp32"][0][t][i], res["bf16"][0][t][i]
        n = int((a != b).sum())
        if n:
            print("t=%d level %d: %d of %d arg-max positions differ" % (t, i, n, a.numel()))
for i in range(5):
    a, b = res["fp32"][1][i], res["bf16"][1][i]
    ref = torch.from_numpy(g["dfeat%d" % i])
    print("dfeat%d: rel L2 bf16 vs golden %.3e, fp32-hip vs golden %.3e; worst element diff %.3e of max %.3e"
          % (i, float((b - ref).norm() / ref.norm()), float((a - ref).norm() / ref.norm()), float((b - ref).abs().max()), float(ref.abs().max())))
