import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """`gpu` tests need a HIP device and the built library: skip (not fail) them elsewhere, so that a plain `pytest tests` on a
    CPU box shows only real CPU-side regressions."""
    import torch
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
