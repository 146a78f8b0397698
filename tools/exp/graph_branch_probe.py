#!/usr/bin/env python
"""Do parallel branches of a captured hipGraph run concurrently on this stack?  Two independent chains of small-grid, deep-K
convolutions (each launch occupies a fraction of the CUs): one stream, two streams eager, and two branches inside a graph."""
import sys
import time

import torch

sys.path.insert(0, ".")
from rsis_amd import ops  # noqa: E402


def main():
    dev = "cuda"
    torch.manual_seed(0)
    C, n = 1024, 24
    w = torch.randn(C, C, 3, 3, device=dev) * 0.01
    pk = [ops.PackedConv(3, [C]) for _ in range(2)]
    xs = [torch.randn(2, C, 16, 16, device=dev) for _ in range(2)]

    def chain(i):
        x = xs[i]
        for _ in range(n):
            x = ops.conv2d([x], w, None, 1, 1, pk[i])
        return x

    def timed(fn, reps=5):
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps * 1e3

    side = torch.cuda.Stream()

    def one_stream():
        chain(0)
        chain(1)

    def two_streams():
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            chain(1)
        chain(0)
        torch.cuda.current_stream().wait_stream(side)

    with torch.no_grad():
        print("eager one stream   %.3f ms" % timed(one_stream))
        print("eager two streams  %.3f ms" % timed(two_streams))
        for name, fn in (("graph one stream ", one_stream), ("graph two branches", two_streams)):
            s = torch.cuda.Stream()
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):
                fn()
            torch.cuda.current_stream().wait_stream(s)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                fn()
            print("%s %.3f ms" % (name, timed(g.replay)))


if __name__ == "__main__":
    main()
