"""CPU tests of the drop-in boundary: librsis_hip.so loads without a GPU and exports every symbol that
include/rsis_hip.h declares (no compute calls here)."""
import ctypes
import os
import re

import pytest

from rsis_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    src = open(os.path.join(ROOT, "include", "rsis_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(rsis_[a-z0-9_]+)\s*\(", src)))


def test_header_declares_symbols():
    syms = _declared_symbols()
    assert len(syms) >= 20 and "rsis_convlstm_fwd" in syms and "rsis_conv2d_wgrad" in syms


def test_library_exports_every_declared_symbol():
    assert os.path.exists(_lib.LIB_PATH), "build the library first: python -c 'import __graft_entry__ as g; g.build()'"
    L = ctypes.CDLL(_lib.LIB_PATH)
    for s in _declared_symbols():
        assert hasattr(L, s), "librsis_hip.so does not export %s" % s


def test_binding_covers_header():
    assert sorted(_lib.SIGNATURES) == _declared_symbols()


def test_version_and_error_strings():
    L = _lib.lib()
    assert L.rsis_version() == 2
    assert L.rsis_error_string(0) == b"ok"
    assert b"argument" in L.rsis_error_string(1)


def test_packed_size_queries():
    L = _lib.lib()
    segs = _lib.int_array([16, 16, 8])
    F32, BF16 = 0, 1
    # 3x3/s1/p1 -> direct layout: 8-channel chunks per source (2+2+1) x 72 rows; Cout 32 -> row stride 128; 4 bytes per element
    assert L.rsis_conv_packed_bytes_fwd(F32, 32, 3, 1, 1, 3, segs) == 5 * 72 * 128 * 4
    assert L.rsis_conv_packed_bytes_dgrad(F32, 32, 3, 1, 1, 40) == 4 * 72 * 128 * 4
    # 3x3/s2/p1 forward runs on the direct kernel too (EPI_F2): same direct layout
    assert L.rsis_conv_packed_bytes_fwd(F32, 32, 3, 2, 1, 3, segs) == 5 * 72 * 128 * 4
    # other strided / padded 3x3 -> implicit-GEMM layout: K = 40*9 = 360 -> 384 rows (multiple of 32)
    assert L.rsis_conv_packed_bytes_fwd(F32, 32, 3, 2, 0, 3, segs) == 384 * 128 * 4
    assert L.rsis_conv_packed_bytes_dgrad(F32, 32, 3, 2, 1, 40) == 288 * 128 * 4
    # bf16 cells: 16-channel chunks per source (1+1+1) x 9 taps x 2 cells of 8 channels, 16 bytes per cell and column
    assert L.rsis_conv_packed_bytes_fwd(BF16, 32, 3, 1, 1, 3, segs) == 3 * 9 * 2 * 128 * 16
    assert L.rsis_conv_packed_bytes_dgrad(BF16, 32, 3, 1, 1, 40) == 2 * 9 * 2 * 128 * 16
    # 1x1: 64-channel chunks of 8 cells
    one = _lib.int_array([256])
    assert L.rsis_conv_packed_bytes_fwd(BF16, 64, 1, 1, 0, 1, one) == 4 * 8 * 128 * 16
    # layers without a bf16 kernel keep the f32 layout under either dtype
    assert L.rsis_conv_uses_bf16(3, 1, 1, 64) == 1 and L.rsis_conv_uses_bf16(1, 2, 0, 64) == 1
    assert L.rsis_conv_uses_bf16(3, 2, 1, 64) == 0 and L.rsis_conv_uses_bf16(7, 2, 3, 64) == 0 and L.rsis_conv_uses_bf16(3, 1, 1, 1) == 0
    assert L.rsis_conv_packed_bytes_fwd(BF16, 32, 3, 2, 1, 3, segs) == 5 * 72 * 128 * 4


def test_product_has_no_cpu_path():
    import pytest
    import torch
    from rsis_amd import ops
    with pytest.raises(_lib.RsisHipError):
        ops.upsample_bilinear_ac(torch.zeros(1, 1, 2, 2), (4, 4))


def test_bench_finds_the_gate_kernel_by_name():
    """bench.py picks the gate kernel's rows out of rocprofv3's counter file by its demangled name: the pattern must match the
    EPI_LSTM instantiations the library actually contains (a template parameter added to the kernel once silently turned
    `roofline.traffic` into null)."""
    import shutil
    import subprocess
    import sys
    if shutil.which("nm") is None:
        pytest.skip("nm not available")
    sys.path.insert(0, ROOT)
    import bench
    out = subprocess.run(["nm", "-C", _lib.LIB_PATH], stdout=subprocess.PIPE, universal_newlines=True, check=True).stdout
    names = [l for l in out.splitlines() if "conv3x3_direct_kernel<" in l]
    gates = [l for l in names if bench.GATE_KERNEL_RE.search(l)]
    assert len(names) >= 20 and len(gates) >= 5, (len(names), len(gates))
    assert any(bench.GATE_GROUP_RE.search(l) for l in out.splitlines()), "the grouped gate kernel is not in the library under the name bench.py filters on"
    assert any(bench.GATE_BLK_RE.search(l) for l in out.splitlines()), "the blk grouped gate kernel (bf16) is not in the library under the name bench.py filters on"
