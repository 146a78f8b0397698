"""Which Python call sites launch stock torch kernels inside one training step (the hand-written library is supposed to hold the
hot path): torch.profiler with stacks over one eager runIter at the bench configuration.  python tools/glue_profile.py [--B 32]"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--B", type=int, default=32)
    ap.add_argument("--S", type=int, default=256)
    ap.add_argument("--dtype", default="fp32")
    o = ap.parse_args()
    import bench
    from rsis_amd.modules import FeatureExtractor, RSIS
    from rsis_amd.synthetic import synthetic_batch
    from rsis_amd.train import build_optimizers, runIter, steps_to_run
    from rsis_amd.utils.objectives import MaskedBCELoss, MaskedNLLLoss, softIoULoss
    a = bench.bench_args(o.B, o.S, 10, o.dtype)
    torch.manual_seed(0)
    enc, dec = FeatureExtractor(a).cuda(), RSIS(a).cuda()
    opts = list(build_optimizers(a, enc, dec))
    crits = [softIoULoss(), MaskedNLLLoss(None), MaskedBCELoss(a.stop_balance_weight)]
    batch = synthetic_batch(1, o.B, o.S, o.S, 20, 12, 21, "cuda")
    t_run = steps_to_run(a, batch[3])
    for _ in range(2):
        runIter(a, enc, dec, *batch, crits, opts, mode="train", sync_losses=False, t_run=t_run, want_outs=False)
    torch.cuda.synchronize()
    from torch.profiler import ProfilerActivity, profile
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
        runIter(a, enc, dec, *batch, crits, opts, mode="train", sync_losses=False, t_run=t_run, want_outs=False)
        torch.cuda.synchronize()
    # call sites by a dispatch-mode trace of one more step (the profiler's own stacks come back empty on this stack)
    import collections
    import traceback
    from torch.utils._python_dispatch import TorchDispatchMode
    sites = collections.Counter()

    class Trace(TorchDispatchMode):
        def __torch_dispatch__(self, func, types, args=(), kwargs=None):
            name = str(func).replace("aten.", "")
            if any(k in name for k in ("copy_", "add", "fill_", "zero_", "mul", "sub", "div", "gather", "gt", "neg", "clone", "_to_copy", "contiguous", "ones", "zeros", "sum", "stack", "cat", "where")):
                fr = [f for f in traceback.extract_stack() if "/rsis_amd/" in f.filename or f.filename.endswith("bench.py")]
                tensors = [a_ for a_ in args if torch.is_tensor(a_)]
                shape = tuple(tensors[0].shape) if tensors else ()
                if fr:
                    sites[(name, "%s:%d %s" % (fr[-1].filename.split("/root/repo/")[-1].split("rsis_amd/")[-1], fr[-1].lineno, fr[-1].name), str(shape)[:40])] += 1
            return func(*args, **(kwargs or {}))
    with Trace():
        runIter(a, enc, dec, *batch, crits, opts, mode="train", sync_losses=False, t_run=t_run, want_outs=False)
    torch.cuda.synchronize()
    print("dispatch trace of one eager step (op, innermost rsis_amd frame, first tensor shape) x calls:")
    for (name, site, shape), n in sorted(sites.items(), key=lambda kv: (kv[0][1], kv[0][0])):
        print("  %3d  %-22s %-60s %s" % (n, name, site, shape))
    rows = {}
    for ev in prof.key_averages(group_by_stack_n=12):
        dev = getattr(ev, "self_device_time_total", 0) or getattr(ev, "self_cuda_time_total", 0)
        if dev <= 0 or not ev.key.startswith("aten::"):
            continue
        site = next((s for s in ev.stack if "/rsis_amd/" in s or "bench.py" in s), ev.stack[0] if ev.stack else "?")
        key = (ev.key, site.split("/root/repo/")[-1][:120] if "/root/repo/" in site else site[-120:])
        r = rows.setdefault(key, [0, 0.0])
        r[0] += ev.count
        r[1] += dev
    tot = sum(v[1] for v in rows.values())
    print("stock torch ops with device time in one eager step: %d calls, %.3f ms" % (sum(v[0] for v in rows.values()), tot / 1e3))
    for (name, site), (n, us) in sorted(rows.items(), key=lambda kv: -kv[1][1])[:45]:
        print("%5d  %8.1f us  %-28s %s" % (n, us, name, site))


if __name__ == "__main__":
    main()
