#!/usr/bin/env python
"""Calibrate rocprofv3's FETCH_SIZE / WRITE_SIZE on this GPU for the access widths the RSIS kernels use, with micro-kernels
that move a KNOWN number of bytes (tools/calib/fetch_calib.hip; every byte of a 1 GiB buffer exactly once -- far beyond the
256 MiB Infinity Cache).  Prints counter bytes / known bytes per kernel; the reciprocal is the correction to apply.

  python tools/fetch_calib.py            # both rocprofv3 --pmc passes + the table (run on the GPU box)
  python tools/fetch_calib.py --run      # just launch the kernels (what rocprofv3 wraps)
"""
import csv
import ctypes
import os
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "tools", "calib", "libfetch_calib.so")
N_FLOATS = 1 << 28          # 1 GiB
READS = ["buffer_load_dword ... lds (4 B/lane LDS-DMA)", "buffer_load_dwordx4 ... lds (16 B/lane LDS-DMA)", "global_load_dword",
         "global_load_dwordx4"]
WRITES = ["global_store_dword (contiguous)", "global_store_dwordx4 (contiguous)", "global_store_dword, 128-byte runs scattered"]


def run():
    import torch
    L = ctypes.CDLL(LIB)
    L.calib_read.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_long, ctypes.c_void_p]
    L.calib_write.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_long, ctypes.c_void_p]
    x = torch.randn(N_FLOATS, device="cuda")
    y = torch.empty(N_FLOATS, device="cuda")
    sink = torch.zeros(4, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    torch.cuda.synchronize()
    for rep in range(2):
        for m in range(4):
            assert L.calib_read(m, x.data_ptr(), sink.data_ptr(), N_FLOATS, st) == 0
        for m in range(3):
            assert L.calib_write(m, y.data_ptr(), N_FLOATS, st) == 0
    torch.cuda.synchronize()


def profile():
    tmp = tempfile.mkdtemp(prefix="rsis_calib_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp")
    res = {}
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            cmd = ["rocprofv3", "--kernel-trace", "--pmc", counter, "--output-format", "csv", "-d", tmp, "-o", counter.lower(), "--",
                   sys.executable, os.path.abspath(__file__), "--run"]
            subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=600, check=True)
            for root, _d, files in os.walk(tmp):
                for f in files:
                    if f.startswith(counter.lower()) and f.endswith("counter_collection.csv"):
                        with open(os.path.join(root, f)) as fh:
                            for r in csv.DictReader(fh):
                                if r["Counter_Name"] == counter and ("read_kernel" in r["Kernel_Name"] or "write_kernel" in r["Kernel_Name"]):
                                    res.setdefault((counter, r["Kernel_Name"]), []).append(float(r["Counter_Value"]) * 1024.0)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    known = N_FLOATS * 4.0
    print("# rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) over tools/fetch_calib.py --run; known bytes per launch = %.0f" % known)
    print("%-62s %-11s %14s %9s" % ("kernel", "counter", "counter bytes", "/ known"))
    for kind, names, counter in (("read_kernel", READS, "FETCH_SIZE"), ("write_kernel", WRITES, "WRITE_SIZE")):
        for m, label in enumerate(names):
            for (c, k), v in sorted(res.items()):
                if c == counter and ("%s<%d>" % (kind, m)) in k:
                    med = sorted(v)[len(v) // 2]
                    print("%-62s %-11s %14.0f %9.3f" % (label, counter, med, med / known))


if __name__ == "__main__":
    if "--run" in sys.argv:
        run()
    else:
        profile()
