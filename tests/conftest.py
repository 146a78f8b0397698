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
_ORDER_FIRST = ["test_gpu_modules", "test_gpu_hot", "test_gpu_round2", "test_gpu_ops", "test_gpu_wino", "test_gpu_bf16", "test_gpu_bf16s", "test_maskpost", "test_targets",
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
    """`gpu` tests need a HIP device: on a box without one they are skipped, so that a plain `pytest tests` on a CPU box shows only
    real CPU-side regressions.  On a box WITH a device a missing librsis_hip.so is a broken build, not a reason to skip: every `gpu`
    test then FAILS ("green with skips" must not be possible).  Also fixes the module run order (stable within a
    module), see _ORDER_FIRST."""
    import torch
    items.sort(key=_module_rank)
    lib_path = os.environ.get("RSIS_HIP_LIB") or os.path.join(ROOT, "rsis_amd", "lib", "librsis_hip.so")
    if not torch.cuda.is_available():
        skip = pytest.mark.skip(reason="gpu test: no HIP GPU visible")
        for item in items:
            if "gpu" in item.keywords:
                item.add_marker(skip)
    elif not os.path.exists(lib_path):
        config._rsis_missing_lib = lib_path


def pytest_runtest_setup(item):
    missing = getattr(item.config, "_rsis_missing_lib", None)
    if missing and "gpu" in item.keywords:
        pytest.fail("a HIP device is visible but %s is not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
                    "(gpu tests fail, not skip, on a GPU box without the library)" % missing, pytrace=False)


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")
