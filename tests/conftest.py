import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


# Run order of the test modules.  The driver runs `pytest -x`: the oracle-parity suites (fp32 first: the north-star 1e-4 checks, every
# C-ABI entry point, the BASELINE geometries and the training step; then the bf16 twin) must not be hidden behind an auxiliary
# test that happens to sort earlier alphabetically.  Modules not listed run after the parity suites and before the last group
# (graph replay / multi-process tests, which exercise launch modes rather than arithmetic).
_ORDER_FIRST = ["test_gpu_modules", "test_gpu_round2", "test_gpu_ops", "test_gpu_bf16", "test_gpu_bf16s", "test_maskpost", "test_targets",
                "test_augment", "test_leaves_loader"]
_ORDER_LAST = ["test_gpu_determinism", "test_gpu_graph", "test_gpu_ddp", "test_gpu_bench"]


def _module_rank(item):
    name = os.path.splitext(os.path.basename(str(item.fspath)))[0]
    if name in _ORDER_FIRST:
        return _ORDER_FIRST.index(name)
    if name in _ORDER_LAST:
        return 1000 + _ORDER_LAST.index(name)
    return 500


def pytest_collection_modifyitems(config, items):
    """`gpu` tests need a HIP device and the built library: skip (not fail) them elsewhere, so that a plain `pytest tests` on a
    CPU box shows only real CPU-side regressions.  Also fixes the module run order (stable within a module), see _ORDER_FIRST."""
    import torch
    items.sort(key=_module_rank)
    lib_path = os.path.join(ROOT, "rsis_amd", "lib", "librsis_hip.so")
    why = None
    if not torch.cuda.is_available():
        why = "no HIP GPU visible"
    elif not os.path.exists(os.environ.get("RSIS_HIP_LIB") or lib_path):
        why = "librsis_hip.so is not built"
    if why is None:
        return
    skip = pytest.mark.skip(reason="gpu test: " + why)
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")
