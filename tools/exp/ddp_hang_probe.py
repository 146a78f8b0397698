#!/usr/bin/env python
"""Repeat the world-size-1 RCCL + split-graph worker of tests/test_gpu_ddp.py and dump the Python stack if an iteration stalls."""
import faulthandler
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))


class Q(object):
    def put(self, v):
        print("result", v[0], v[1], [round(x, 4) for x in v[2]], flush=True)


if __name__ == "__main__":
    faulthandler.dump_traceback_later(60, exit=True)
    t0 = time.time()
    import test_gpu_ddp as T
    T._worker_force_dist_graph(Q())
    print("ok %.1f s" % (time.time() - t0), flush=True)
