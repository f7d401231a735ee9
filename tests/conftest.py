import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session')
def c_oracle():
    from oracle import c_oracle as co
    co.build()
    return co


@pytest.fixture(scope='session')
def pm_ctx():
    """One device handle for the GPU tests; fails loudly if the HIP library is missing."""
    from sea_ice_drift_amd import _capi
    ctx = _capi.PMContext(0)
    yield ctx
    ctx.close()


def pytest_collection_modifyitems(config, items):
    """The eviction soaks (tests/test_gpu_soak.py) run FIRST: each is a process of its own that provokes evictions of its GPU
    queues, and on the pool's boxes such an eviction stalls for minutes once another process - this one, after the first GPU
    test - holds queues and gigabytes of allocations on the same device."""
    first = [it for it in items if 'survive_queue_evictions' in it.nodeid]
    if first:
        rest = [it for it in items if 'survive_queue_evictions' not in it.nodeid]
        items[:] = first + rest
