import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "last: collected after everything else (heavy north-star-size cases, tests whose counters depend on host timing) "
                            "so that under `pytest -x` one of them going red cannot hide the rest of the suite")


def pytest_collection_modifyitems(config, items):
    head = [it for it in items if it.get_closest_marker("last") is None]
    tail = [it for it in items if it.get_closest_marker("last") is not None]
    items[:] = head + tail


@pytest.fixture(scope="session")
def oracle():
    import oracle as o

    o.build()
    return o
