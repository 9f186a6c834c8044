import os
import sys

import numpy as np
import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """`gpu` tests need the HIP library and an MI355X: skip them (instead of failing) anywhere else."""
    try:
        from starfish_amd import _lib

        have_gpu = _lib.load().sf_device_count() > 0
    except Exception:
        have_gpu = False
    if have_gpu:
        return
    skip = pytest.mark.skip(reason="no MI355X / libstarfish_amd.so: the HIP path has no CPU fallback")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


@pytest.fixture(scope="session")
def golden():
    return load_golden


@pytest.fixture(params=["fused", "unfused", "wide", "wide_then_narrow", "dataflow"])
def chol_sequence(request):
    """Runs a test under both launch sequences of the batched Cholesky (by default the library picks by batch size,
    which would leave the fused panel kernel -- the bench path -- untested at the small batches of the tests)."""
    from starfish_amd import _lib

    lib = _lib.require_gpu()
    assert lib.sf_debug_cholesky_sequence({"fused": 0, "unfused": 1, "wide": 2, "wide_then_narrow": 3, "dataflow": 4}[request.param]) == 0
    yield request.param
    lib.sf_debug_cholesky_sequence(-1)
