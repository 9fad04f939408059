import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _oracle(prec):
    """the CPU oracle with its OpenMP team bounded to the process's CPU share (the GPU box shows 256 logical CPUs to a process whose
    share is 16; every thread of the oracle's backward owns private gradient buffers)"""
    from oracle.nso import Oracle
    o = Oracle(prec)
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    o.lib.nso_set_num_threads(max(1, min(n, 16)))
    return o


@pytest.fixture(scope="session")
def oracle32():
    return _oracle("f32")


@pytest.fixture(scope="session")
def oracle64():
    return _oracle("f64")
