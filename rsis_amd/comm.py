"""RCCL bound directly (include/rsis_hip.h: rsis_comm_*) -- the gradient exchange of one-process-per-GPU training without
torch.distributed in the data path.  Replaces nn.DataParallel of reference src/train.py:269-274.

torch.distributed still does the rendezvous (it distributes the ncclUniqueId, broadcasts the initial parameters, runs barriers); the
per-iteration all-reduces of the flat gradient buffers go through a communicator this module owns.  What that buys: a collective
issued here is an ordinary stream operation, so it can be CAPTURED into the hipGraph of the training iteration -- on a forked stream,
overlapping the rest of the backward -- whereas ProcessGroupNCCL's watchdog thread aborts a concurrent stream capture, which forced
train.GraphedStep to cut the iteration into several graphs with eager collectives between them (RSIS_EXCHANGE=cuts keeps that
schedule; it is also the fallback when the direct communicator cannot be built, e.g. several gloo ranks sharing one GPU in the tests).
"""
import ctypes
import os

import torch
import torch.distributed as dist

from ._lib import RsisHipError, lib, stream


def _check(rc, what):
    if rc != 0:
        L = lib()
        raise RsisHipError("%s failed: %s (code %d): %s" % (what, L.rsis_error_string(rc).decode(), rc, (L.rsis_comm_last_error() or b"").decode()))


class DirectComm(object):
    """one RCCL communicator over the ranks of the default process group, each rank on its current HIP device"""

    def __init__(self):
        if not dist.is_initialized():
            raise RsisHipError("DirectComm needs torch.distributed for the rendezvous")
        L = lib()
        self.world, self.rank = dist.get_world_size(), dist.get_rank()
        buf = ctypes.create_string_buffer(128)
        if self.rank == 0:
            _check(L.rsis_comm_unique_id(buf), "rsis_comm_unique_id")
        box = [bytes(buf.raw)]
        if self.world > 1:
            dist.broadcast_object_list(box, src=0)
        self._id = ctypes.create_string_buffer(box[0], 128)
        handle = ctypes.c_void_p()
        _check(L.rsis_comm_init(ctypes.byref(handle), self.world, self.rank, self._id), "rsis_comm_init")
        self.handle = handle
        n = L.rsis_comm_size(self.handle)
        if n != self.world:
            raise RsisHipError("RCCL communicator has %d ranks, expected %d" % (n, self.world))

    def all_reduce(self, buf):
        """buf (contiguous fp32 CUDA tensor) <- sum over the ranks, in place, on the current stream"""
        assert buf.is_cuda and buf.dtype == torch.float32 and buf.is_contiguous()
        _check(lib().rsis_comm_allreduce_sum_f32(self.handle, buf.data_ptr(), buf.numel(), stream()), "rsis_comm_allreduce_sum_f32")

    def close(self):
        if self.handle is not None:
            lib().rsis_comm_destroy(self.handle)
            self.handle = None


class _Join(object):
    def __init__(self, side):
        self.side = side

    def wait(self):
        torch.cuda.current_stream().wait_stream(self.side)


class DirectReducer(object):
    """reduce(buf, async_op) in the shape train.StagedExchange / GraphedStep use: the asynchronous form runs the collective on a side
    stream forked from the current one (eagerly: it overlaps the kernels enqueued afterwards; under stream capture: a parallel branch
    of the graph) and returns a handle whose wait() joins the side stream back"""

    def __init__(self, comm):
        self.comm, self.world = comm, comm.world
        self.side = torch.cuda.Stream()

    def __call__(self, buf, async_op=False):
        if not async_op:
            self.comm.all_reduce(buf)
            return None
        ev = torch.cuda.Event()
        ev.record()
        self.side.wait_event(ev)
        with torch.cuda.stream(self.side):
            self.comm.all_reduce(buf)
        return _Join(self.side)

    def self_test(self):
        """a small all-reduce captured on the forked stream and replayed twice: what GraphedStep relies on"""
        x = torch.arange(1024, dtype=torch.float32, device="cuda") + self.comm.rank
        want = self.world * torch.arange(1024, dtype=torch.float32, device="cuda") + sum(range(self.world))
        y = x.clone()
        g = torch.cuda.CUDAGraph()
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            g.capture_begin(capture_error_mode="thread_local")
            try:
                y.copy_(x)
                h = self(y, True)
                z = x * 2.0            # work on the main branch while the collective runs on the other
                h.wait()
                y.add_(z - 2.0 * x)
            finally:
                g.capture_end()
        torch.cuda.current_stream().wait_stream(s)
        for _ in range(2):
            g.replay()
        torch.cuda.synchronize()
        if not torch.equal(y, want):
            raise RsisHipError("captured RCCL all-reduce returned wrong sums")
        return True


def make_direct_reducer(log=None):
    """DirectReducer over the default process group, or None (with the reason logged) when the direct exchange does not apply:
    RSIS_EXCHANGE=cuts / hooks, a backend other than nccl (RCCL needs one GPU per rank: the shared-GPU gloo tests), or any failure
    to build / self-test the communicator."""
    say = log or (lambda _m: None)
    mode = os.environ.get("RSIS_EXCHANGE", "direct")
    if mode in ("cuts", "hooks", "staged"):
        say("direct RCCL exchange off (RSIS_EXCHANGE=%s)" % mode)
        return None
    if not (dist.is_initialized() and torch.cuda.is_available()):
        return None
    if dist.get_backend() != "nccl":
        say("direct RCCL exchange off (backend %s)" % dist.get_backend())
        return None
    red, why = None, None
    try:
        red = DirectReducer(DirectComm())
        red.self_test()
    except Exception as e:  # noqa: BLE001  (the cut-graph schedule over torch.distributed remains)
        red, why = None, repr(e)
    # every rank must take the same schedule: one rank on the direct exchange and another on the cut graphs would wait for each
    # other's collectives forever.  Agree over the (working) torch.distributed group: direct only if it came up everywhere.
    ok = torch.tensor([1 if red is not None else 0], dtype=torch.int32, device="cuda")
    dist.all_reduce(ok, op=dist.ReduceOp.MIN)
    if int(ok.item()) == 0:
        say("direct RCCL exchange not available%s" % (": " + why if why else " on another rank"))
        return None
    say("direct RCCL exchange: communicator of %d rank(s), captured all-reduce self-test passed" % red.world)
    return red
