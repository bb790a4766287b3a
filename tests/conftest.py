import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from tests import oracle_lib
    return oracle_lib.load()


def pytest_collection_modifyitems(config, items):
    """`pytest tests` on a box without a HIP device: gpu-marked tests are skipped instead of failing with RH_E_DEVICE
    (the engine has no CPU fallback).  `-m gpu` on the GPU box runs them all."""
    try:
        from rainier_amd import _capi
        have = _capi.lib().rh_device_count() > 0
    except Exception:
        have = True   # a missing / unloadable library must fail loudly, never skip
    if have:
        return
    skip = pytest.mark.skip(reason="no HIP device: the engine has no CPU fallback")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)
