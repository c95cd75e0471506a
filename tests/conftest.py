import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """Plain ``pytest`` on a box without a HIP device skips the gpu tests instead of failing inside them."""
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no HIP device visible (gpu tests run on the MI355X box: pytest -m gpu)")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def built_library():
    """libserhip.so, built in-tree by hipcc (cross-compiles without a GPU)."""
    path = os.path.join(ROOT, "interspeech_ser_amd", "lib", "libserhip.so")
    if not os.path.isfile(path):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "interspeech_ser_amd", "csrc"), "-j4"])
    return path


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
